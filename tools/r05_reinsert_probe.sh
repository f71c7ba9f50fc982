#!/bin/bash
# round 5: the device re-insertion pass after the refit of what moved / key tie rule / fixed-point growth term / area guard: build time,
# passes, moves, summed area before / after, and the production walk's fetches per ray on the C3 / C2 / 12 M-triangle probe frames --
# against round 4's refit of every box after every pass (PBRT_HIP_REINSERT_FULL_REFIT=1) --, then the builder's kernels of one C3 build
# under rocprofv3 --kernel-trace --stats.  usage (GPU box): tools/r05_reinsert_probe.sh   (profiles/r05e_reinsert_probe.txt)
cd "$(dirname "$0")/.."
export PBRT_HIP_DEBUG_KNOBS=1 PROBE_COUNTERS=1 PROBE_BUILDER=gpu
mkdir -p gpurun_out/r05e
for wl in c3 c2 big; do
  for full in "" 1; do
    echo "== $wl full_refit=${full:-0}"
    if [ -n "$full" ]; then export PBRT_HIP_REINSERT_FULL_REFIT=1; else unset PBRT_HIP_REINSERT_FULL_REFIT; fi
    timeout 900 python3 tools/pmc_probe.py $wl 4 4 2>&1 | grep -v "^RAYS"
  done
done
unset PBRT_HIP_REINSERT_FULL_REFIT PROBE_COUNTERS
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05e/build_trace -- python3 $R/tools/pmc_probe.py c3 1 1 > $R/gpurun_out/r05e/build_trace.log 2>&1
cd $R; python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05e/build_trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
