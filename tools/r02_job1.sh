#!/bin/bash
# round-2 job 1: the VALU-issue ubench + A-B sensitivity of render_kernel (extra VALU / extra loads / occupancy)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02_job1; mkdir -p $O
cd $R
timeout 300 tools/ubench/valu_issue > $O/valu_issue.txt 2>&1
for v in base valu32 valu64 loads; do
  for rep in 1 2; do
    echo -n "$v: " >> $O/ab.txt
    PBRT_HIP_LIB_DIR=$R/pbrt_amd/lib_$v timeout 300 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -1 >> $O/ab.txt
  done
done
for wg in 1024 2048 3072 4096; do
  echo -n "workgroups=$wg: " >> $O/ab.txt
  PBRT_HIP_RENDER_WORKGROUPS=$wg PBRT_HIP_LIB_DIR=$R/pbrt_amd/lib_base timeout 300 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -1 >> $O/ab.txt
done
cat $O/ab.txt
