#!/bin/bash
# usage (on the GPU box): tools/pmc_passes.sh <out-subdir> [probe args...]; one rocprofv3 run per counter group
# (--pmc with --kernel-trace only, as the pool requires; FETCH_SIZE and WRITE_SIZE each in a pass of their own:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  The probe (tools/pmc_probe.py) renders one frame at a low sample count
# and, with PROBE_COUNTERS=1, counts its rays with the counting instantiations of the kernel (other kernel names).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_COUNTERS=1
i=0
# (a group that starts with "ISSUE:" is the vector issue port's occupancy, tools/pmc_issue.sh / profiles/r03c_issue_counter_calibration.txt:
# its counters come from ONE pass -- the others repeat elsewhere -- and are kept apart as ISSUE_<name>)
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  prefix=""
  case "$group" in ISSUE:*) prefix="ISSUE_"; group=${group#ISSUE: };; esac
  echo "$prefix" > $OUT/p$i.prefix
  timeout 600 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/pmc_probe.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $group"
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH
SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT
ISSUE: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
import os
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    pdir = f[len(out) + 1:].split("/")[0]
    prefix = open(os.path.join(out, pdir + ".prefix")).read().strip() if os.path.exists(os.path.join(out, pdir + ".prefix")) else ""
    for r in csv.DictReader(open(f)):
        if re.search(r"render_kernel<(true|false), false, false", r["Kernel_Name"]):  # the production kernel only
            agg[prefix + r["Counter_Name"]] = agg.get(prefix + r["Counter_Name"], 0.0) + float(r["Counter_Value"])
probe = {}
for line in open(out + "/p1.log"):
    m = re.match(r"RAYS (\d+) SAMPLES (\d+) KERNEL_MS ([0-9.]+)", line)
    if m:
        probe = {"PROBE_RAYS": float(m.group(1)), "PROBE_SAMPLES": float(m.group(2))}
    m = re.match(r"TREE FETCHES_PER_RAY ([0-9.]+) TRIS_PER_RAY ([0-9.]+) QUAD_NODES (\d+) STACK_NEED (\d+) LDS_ROWS (\d+) WAVES_PER_CU (\d+) SPP (\d+)", line)
    if m:
        probe.update({"TREE_FETCHES_PER_RAY": float(m.group(1)), "TREE_TRIS_PER_RAY": float(m.group(2)), "TREE_QUAD_NODES": float(m.group(3)),
                      "TREE_STACK_NEED": float(m.group(4)), "LAUNCH_LDS_ROWS": float(m.group(5)), "LAUNCH_WAVES_PER_CU": float(m.group(6)), "PROBE_SPP": float(m.group(7))})
# the kernel's duration inside the PMC runs (kernel trace of pass 1)
for f in sorted(glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if re.search(r"render_kernel<(true|false), false, false", r["Kernel_Name"]):
            probe["PROBE_KERNEL_NS"] = probe.get("PROBE_KERNEL_NS", 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg.update(probe)
with open(out + "/summary.txt", "w") as fo:
    for k, v in agg.items():
        fo.write(f"{k} {v:.9g}\n")
        print(k, f"{v:.6g}")
PY
