#!/bin/bash
# usage (GPU box): tools/pmc_classes.sh <out-subdir> -- <program args...>
# The instruction-class counters of the SQ (two rocprofv3 --pmc passes, kernel-trace only) for every kernel of a
# program, summarised per kernel name: how many of its VALU instructions fall into which class.  Used twice: on
# tools/ubench/valu_issue pmc (which instruction increments which counter: the calibration) and on the render kernel.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(out + "/classes.txt", "w") as fo:
    for k, c in agg.items():
        if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
        tot = c["SQ_INSTS_VALU"]
        line = k[:70] + " | VALU %.4g | " % tot + " ".join(f"{n.replace('SQ_INSTS_', '')}={v / tot:.3f}" for n, v in c.items() if n != "SQ_INSTS_VALU")
        print(line); fo.write(line + "\n")
PY
