#!/bin/bash
# Round 5 (VERDICT r04 task 2b): the counters of the treetop variant (PBRT_HIP_TREETOP=128, 4 waves per workgroup) beside the product
# kernel's on the C3 probe frame, three rocprofv3 --pmc passes each (kernel-trace only): vector L1 accesses, issue quad-cycles, active
# lanes per vector instruction, LDS instructions and bank conflicts.  usage (GPU box): tools/experiments/r05_treetop_lds/pmc.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
OUT=$R/gpurun_out/r05c_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PBRT_HIP_DEBUG_KNOBS=1 PROBE_COUNTERS=1 PBRT_HIP_TREETOP_WAVES=4
for K in 0 128; do
  export PBRT_HIP_TREETOP=$K
  i=0
  while read -r group; do
    [ -z "$group" ] && continue
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/k${K}_p$i -- python3 $R/tools/pmc_probe.py c3 8 8 > $OUT/k${K}_p$i.log 2>&1
    echo "treetop $K pass $i rc=$? : $group"
  done <<'GROUPS'
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum
GROUPS
done
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
for K in (0, 128):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(f"{out}/k{K}_p*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if re.search(r"render_kernel(_top)?<false, (false, false|\d+)", r["Kernel_Name"]):  # the production kernel / its treetop variant (not the counting ones)
                agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                agg["KERNEL"] = r["Kernel_Name"][:80]
    rays = ms = None
    for line in open(f"{out}/k{K}_p1.log"):
        m = re.match(r"RAYS (\d+) SAMPLES (\d+) KERNEL_MS ([0-9.]+)", line)
        if m:
            rays, ms = float(m.group(1)), float(m.group(3))
    qc = agg["SQ_ACTIVE_INST_VALU"] - agg["SQ_ACTIVE_INST_VALU2"]
    print(f"treetop {K:4d}: {agg['KERNEL']}")
    print(f"   kernel {ms:.1f} ms under the profiler | issue quad-cycles/ray {qc / rays:.2f} | vector instructions/ray {agg['SQ_INSTS_VALU'] / rays:.2f} | "
          f"lanes per vector instruction {agg['SQ_THREAD_CYCLES_VALU'] / agg['SQ_ACTIVE_INST_VALU']:.2f} of 64 | "
          f"L1 accesses/ray {agg['TCP_TOTAL_CACHE_ACCESSES_sum'] / rays:.2f} | VMEM read instr/ray {agg['SQ_INSTS_VMEM_RD'] / rays:.3f} | LDS instr/ray {agg['SQ_INSTS_LDS'] / rays:.3f} | "
          f"LDS bank-conflict cycles / LDS active cycles {agg['SQ_LDS_BANK_CONFLICT'] / max(agg['SQ_LDS_IDX_ACTIVE'], 1):.3f} | L2 hit rate {agg['TCC_HIT_sum'] / (agg['TCC_HIT_sum'] + agg['TCC_MISS_sum']):.3f}")
    for k, v in agg.items():
        if k != "KERNEL":
            print(f"   {k} {v:.6g}")
PY
