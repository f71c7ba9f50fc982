#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02_job7; mkdir -p $O
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*\|SQ_THREAD_CYCLES[A-Z0-9_]*" | sort -u > $O/counters.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_ubench -- $R/tools/ubench/valu_issue pmc > $O/pmc_ubench.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pmc_ubench/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Kernel_Name"][:40])
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
with open(out + "/pmc_ubench_summary.txt", "w") as fo:
    for (d, k), c in rows.items():
        if c.get("SQ_INSTS_VALU", 0) < 1e8: continue
        line = f"{d:>4} {k:40s} INSTS_VALU {c.get('SQ_INSTS_VALU',0):.4g} ACTIVE_INST_VALU {c.get('SQ_ACTIVE_INST_VALU',0):.4g} ratio {c.get('SQ_ACTIVE_INST_VALU',0)/max(c.get('SQ_INSTS_VALU',1),1):.3f} WAVE_CYCLES {c.get('SQ_WAVE_CYCLES',0):.4g} BUSY_CYCLES {c.get('SQ_BUSY_CYCLES',0):.4g} GUI_ACTIVE {c.get('GRBM_GUI_ACTIVE',0):.4g}"
        print(line); fo.write(line + "\n")
PY
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "multi_gpu or limits or sampler or chunk or handout" 2>&1 | tail -8 > $O/pytest_new.txt
cat $O/pytest_new.txt
timeout 900 python3 - > $O/big.txt 2>&1 <<'PY'
import time, sys
sys.path.insert(0, ".")
import pbrt_amd
from pbrt_amd import scenes
t0 = time.time(); sd = scenes.big_mesh_scene(); print("scene", time.time() - t0, flush=True)
t0 = time.time()
with pbrt_amd.Scene(sd) as sc:
    print("build+upload", time.time() - t0, sc.info(), flush=True)
    for spp in ((2, 2), (4, 4)):
        film, st = sc.render(max_depth=8, spp=spp, seed=0)
        print("big", spp, "kernel_ms", st["kernel_ms"], "Msamples/s", st["samples"] / st["kernel_ms"] / 1e3, flush=True)
    _, wk = sc.render(max_depth=8, spp=(2, 2), seed=0, counters="walk")
    _, ex = sc.render(max_depth=8, spp=(2, 2), seed=0, counters=True)
    rays = ex["camera_rays"] + ex["bounce_rays"] + ex["shadow_rays"]
    print("rays/sample", rays / ex["samples"], "exact nodes/ray", ex["nodes_visited"] / rays, "tris/ray", ex["tris_tested"] / rays,
          "| walk fetches/ray", wk["nodes_visited"] / rays, "tris/ray", wk["tris_tested"] / rays)
PY
cat $O/big.txt
