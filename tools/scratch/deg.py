import sys, os, dataclasses
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, pbrt_amd, util
from pbrt_amd import LIGHT_DISTANT, LIGHT_POINT, INTEGRATOR_PATH, INTEGRATOR_DIRECT
from oracle import binding as ob
def make(seed, drop=()):
    rng = np.random.default_rng(1000 + seed)
    sd = util.SMALL_SCENES["mesh1k"]()
    P, idx = sd.P.copy(), sd.idx.copy()
    lights = [[LIGHT_POINT, *P[int(rng.integers(0, len(P)))], 5, 5, 5],[LIGHT_POINT, *rng.uniform(-1, 1, 3), 0, 0, 0],[LIGHT_POINT, *rng.uniform(-1, 1, 3), 1e30, 1e30, 1e30],[LIGHT_DISTANT, 0, 0, 0, 2, 2, 2]]
    extra = [[LIGHT_POINT, *rng.uniform(-1.5, 1.5, 3), *rng.uniform(0, 0.3, 3)] for _ in range(300 if seed % 2 else 0)]
    mats = sd.materials.copy(); mats[0,1:4]=0.0; mats[1%len(mats),1:4]=1.7
    tri=int(rng.integers(0,len(idx)));
    if "zeroarea" not in drop: idx[tri]=idx[tri][[0,0,1]]
    mats=np.concatenate([mats,[[0,0.5,0.5,0.5,4,4,4]]]).astype(np.float32)
    mat_id=sd.mat_id.copy(); mat_id[tri]=len(mats)-1
    eye=sd.cam_to_world[:3,3]
    spheres=[[0.3,0.2,0.1,0.25,0],[0.3,0.2,0.1,0.25,1%len(mats)],[*rng.uniform(-1,1,3),1e-6,0],[0,0,0,1e6,0],[*eye,0.05,0]]
    L = [l for i,l in enumerate(lights) if f"light{i}" not in drop] + ([] if "extra" in drop else extra)
    S = [] if "spheres" in drop else spheres[:2+seed%4]
    if "kd17" in drop: mats[1%len(mats),1:4]=0.7
    return dataclasses.replace(sd, idx=idx, mat_id=mat_id, materials=mats, mat_tex=np.zeros(len(mats),np.uint32), lights=np.array(L,np.float32).reshape(-1,7), spheres=np.array(S,np.float32).reshape(-1,5)).normalized()
seed=5
for drop in ((), ("extra",), ("light0",), ("light1",), ("light2",), ("light3",), ("spheres",), ("zeroarea",), ("kd17",)):
    sd=make(seed, drop)
    for integ in (0,2):
        for sampler in ("stratified","halton"):
            kw=dict(max_depth=6, spp=(2,2), seed=seed, integrator=integ, sampler=sampler)
            ref,_=ob.OracleScene(sd).render(**kw)
            with pbrt_amd.Scene(sd) as sc: film,_=sc.render(**kw)
            bad=np.argwhere((film.view(np.uint32)!=ref.view(np.uint32)).any(-1))
            print("drop",drop,"integ",integ,sampler,"mismatched pixels",len(bad), bad[:3].tolist(), flush=True)
