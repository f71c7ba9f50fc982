import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tools/scratch")
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("deg", "tools/scratch/deg_make.py"); deg = importlib.util.module_from_spec(spec); spec.loader.exec_module(deg)
import pbrt_amd
from oracle import binding as ob
sd = deg.make(5)
kw = dict(max_depth=6, spp=(2, 2), seed=5, integrator=0, sampler="halton")
if sys.argv[1] == "hip":
    with pbrt_amd.Scene(sd) as sc:
        film, _ = sc.render(**kw)
    print("HIPFILM", film[23, 47].view(np.uint32).tolist())
else:
    os.environ["ORC_DEBUG_LI"] = "1"
    ref = ob.OracleScene(sd).pixel_samples(47, 23, **kw).reshape(-1, 3)
    print("ORACLE samples", ref.tolist())
