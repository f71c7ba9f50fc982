import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
os.environ.setdefault("PBRT_HIP_DEBUG_KNOBS","1")
from oracle import binding as ob
from pbrt_amd import scenes
from pbrt_amd.api import quad_build_host_ex
import walk_sim

def e4m3_grid():
    v=[0.0]+[m*2.0**-9 for m in range(1,8)]
    for e in range(1,16):
        for m in range(8):
            if e==15 and m==7: continue
            v.append(2.0**(e-7)*(1+m/8))
    return np.array(sorted(set(v)))
def e5m2_grid():
    v=[0.0]+[m*2.0**-16 for m in range(1,4)]
    for e in range(1,31):
        for m in range(4):
            v.append(2.0**(e-15)*(1+m/4))
    return np.array(sorted(set(v)))

def quantise(exact, grid, two_origin):
    """exact: (n,4,6) child boxes; returns boxes rounded outwards onto origin + grid*cell (cell: grid max covers the node extent)"""
    ex=exact.reshape(-1,4,6).astype(np.float64).copy()
    used=ex[:,:,0]<=ex[:,:,3]
    lo=np.where(used[:,:,None], ex[:,:,0:3], np.inf).min(1); hi=np.where(used[:,:,None], ex[:,:,3:6], -np.inf).max(1)
    ext=np.maximum(hi-lo,1e-30)
    g=grid/grid.max()   # normalised 0..1
    out=ex.copy()
    for a in range(3):
        for k in range(4):
            l=(ex[:,k,a]-lo[:,a])/ext[:,a]; h=(ex[:,k,3+a]-lo[:,a])/ext[:,a]
            if not two_origin:
                li=np.clip(np.searchsorted(g,l,side='right')-1,0,len(g)-1); hi_i=np.clip(np.searchsorted(g,h,side='left'),0,len(g)-1)
                ql=g[li]; qh=g[hi_i]
            else:
                # lo planes on the grid measured from the low corner; hi planes on the grid measured DOWN from the high corner
                li=np.clip(np.searchsorted(g,l,side='right')-1,0,len(g)-1); ql=g[li]
                hd=1.0-h  # distance below the high corner
                hi_i=np.clip(np.searchsorted(g,hd,side='right')-1,0,len(g)-1); qh=1.0-g[hi_i]
            out[:,k,a]=np.where(used[:,k], lo[:,a]+ql*ext[:,a], np.inf); out[:,k,3+a]=np.where(used[:,k], lo[:,a]+qh*ext[:,a], -np.inf)
    # outward float rounding
    o32=out.astype(np.float32)
    o32[:,:,0:3]=np.where(o32[:,:,0:3].astype(np.float64)>out[:,:,0:3], np.nextafter(o32[:,:,0:3],-np.inf,dtype=np.float32), o32[:,:,0:3])
    o32[:,:,3:6]=np.where(o32[:,:,3:6].astype(np.float64)<out[:,:,3:6], np.nextafter(o32[:,:,3:6],np.inf,dtype=np.float32), o32[:,:,3:6])
    return o32.reshape(-1,24)

n=int(sys.argv[1]) if len(sys.argv)>1 else 100000
sd=scenes.random_mesh_scene(n,256,256).normalized()
osc=ob.OracleScene(sd)
(co,cd,ct),(so,sdd,stm)=walk_sim.path_rays(sd,osc)
rt,rprim,_,_,_=osc.intersect(co,cd,ct)
q=quad_build_host_ex(sd.P,sd.idx,tree="sah")
def run(name, boxes):
    c=ob.quad_walk(q["quads"],q["root_box"],sd.P,sd.idx,q["order"],co,cd,ct,exact_boxes=boxes)
    s=ob.quad_walk(q["quads"],q["root_box"],sd.P,sd.idx,q["order"],so,sdd,stm,any_hit=True,exact_boxes=boxes)
    ok=np.array_equal(c["prim"],rprim)
    print(f"{name:28s} steps/ray {(c['steps'].sum()+s['steps'].sum())/(len(co)+len(so)):6.2f} tris/ray {(c['tris'].sum()+s['tris'].sum())/(len(co)+len(so)):5.2f} hits {'ok' if ok else 'DIFFER'}")
run("kernel's own 8-bit planes", None)
run("exact boxes", q["exact_boxes"])
u8=np.arange(256.0)
run("emulated uniform 8 bit", quantise(q["exact_boxes"],u8,False))
run("uniform 7 bit", quantise(q["exact_boxes"],np.arange(128.0),False))
run("uniform 6 bit", quantise(q["exact_boxes"],np.arange(64.0),False))
run("e4m3 one origin", quantise(q["exact_boxes"],e4m3_grid(),False))
run("e4m3 two origins", quantise(q["exact_boxes"],e4m3_grid(),True))
run("e5m2 two origins", quantise(q["exact_boxes"],e5m2_grid(),True))


# ---- r03u: signed fp8 planes around a per-node, per-axis ORIGIN chosen by the builder (the decode stays linear:
# t = v * (cell * inv) - (o - origin) * inv, v = the fp8 value in -448 .. 448, cell a power of two) ----
def quantise_signed(exact, grid, origin_mode, pow2=True):
    ex = exact.reshape(-1, 4, 6).astype(np.float64).copy()
    used = ex[:, :, 0] <= ex[:, :, 3]
    lo = np.where(used[:, :, None], ex[:, :, 0:3], np.inf).min(1)
    hi = np.where(used[:, :, None], ex[:, :, 3:6], -np.inf).max(1)
    sg = np.concatenate([-grid[::-1], grid[1:]])  # signed value set, ascending
    gmax = grid.max()
    out = ex.copy()
    for a in range(3):
        L = np.where(used, ex[:, :, a], np.nan)      # (n, 4)
        H = np.where(used, ex[:, :, 3 + a], np.nan)
        if origin_mode == "centre":
            cands = [0.5 * (lo[:, a] + hi[:, a])]
        elif origin_mode == "lo":
            cands = [lo[:, a]]
        else:  # every plane position, both corners and the centre
            cands = [lo[:, a], hi[:, a], 0.5 * (lo[:, a] + hi[:, a])] + [np.where(used[:, k], ex[:, k, a], lo[:, a]) for k in range(4)] + \
                    [np.where(used[:, k], ex[:, k, 3 + a], hi[:, a]) for k in range(4)]
        best_err = np.full(len(ex), np.inf)
        best_ql = np.zeros_like(L)
        best_qh = np.zeros_like(H)
        for org in cands:
            reach = np.maximum(np.maximum(hi[:, a] - org, org - lo[:, a]), 1e-30)
            cell = reach / gmax
            if pow2:
                cell = 2.0 ** np.ceil(np.log2(cell))
            l = (L - org[:, None]) / cell[:, None]
            h = (H - org[:, None]) / cell[:, None]
            li = np.clip(np.searchsorted(sg, np.nan_to_num(l), side='right') - 1, 0, len(sg) - 1)
            hi_i = np.clip(np.searchsorted(sg, np.nan_to_num(h), side='left'), 0, len(sg) - 1)
            ql = org[:, None] + sg[li] * cell[:, None]
            qh = org[:, None] + sg[hi_i] * cell[:, None]
            err = np.nansum(np.where(used, (L - ql) + (qh - H), 0.0), axis=1)
            better = err < best_err
            best_err = np.where(better, err, best_err)
            best_ql = np.where(better[:, None], ql, best_ql)
            best_qh = np.where(better[:, None], qh, best_qh)
        for k in range(4):
            out[:, k, a] = np.where(used[:, k], best_ql[:, k], np.inf)
            out[:, k, 3 + a] = np.where(used[:, k], best_qh[:, k], -np.inf)
    o32 = out.astype(np.float32)
    o32[:, :, 0:3] = np.where(o32[:, :, 0:3].astype(np.float64) > out[:, :, 0:3], np.nextafter(o32[:, :, 0:3], -np.inf, dtype=np.float32), o32[:, :, 0:3])
    o32[:, :, 3:6] = np.where(o32[:, :, 3:6].astype(np.float64) < out[:, :, 3:6], np.nextafter(o32[:, :, 3:6], np.inf, dtype=np.float32), o32[:, :, 3:6])
    return o32.reshape(-1, 24)


run("uniform 8 bit, pow2 cell", quantise_signed(q["exact_boxes"], np.arange(128.0), "lo") if False else quantise(q["exact_boxes"], u8, False))
for mode in ("lo", "centre", "best"):
    for p2 in (False, True):
        run(f"e4m3 signed, origin {mode}{', pow2 cell' if p2 else ''}", quantise_signed(q["exact_boxes"], e4m3_grid(), mode, p2))
