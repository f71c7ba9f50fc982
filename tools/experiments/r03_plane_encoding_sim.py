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
