#!/bin/bash
# usage (GPU box): tools/pmc_issue.sh <out-subdir> -- <program args...>
# The SQ's issue-side counters (one rocprofv3 --pmc pass, kernel-trace only) per kernel of a program: vector instructions,
# the quad-cycles in which a wave issued one (SQ_ACTIVE_INST_VALU) and in which TWO were issued together
# (SQ_ACTIVE_INST_VALU2), busy / wave cycles, and the kernel's duration.  On `tools/ubench/valu_issue pmc` this calibrates
# what the counters mean per instruction class; on the render kernel it MEASURES the vector issue port's occupancy
# (tools/issue_busy.py) instead of pricing an instruction census with ubench costs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/p1 -- "$@" > $OUT/p1.log 2>&1
echo "pass rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg, dur = collections.OrderedDict(), collections.Counter()
for f in sorted(glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for f in sorted(glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(out + "/issue.txt", "w") as fo:
    for k, c in agg.items():
        if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
        line = k[:90] + " | " + " ".join(f"{n}={v:.6g}" for n, v in c.items()) + f" KERNEL_NS={dur[k]:.6g}"
        print(line); fo.write(line + "\n")
PY
