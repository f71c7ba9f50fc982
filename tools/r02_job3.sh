#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02_job3; mkdir -p $O
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
for v in lib_base lib; do
  echo "== $v" >> $O/probe.txt
  PROBE_COUNTERS=1 PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 600 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -2 >> $O/probe.txt
  PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 600 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -1 >> $O/probe.txt
  PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 600 python3 tools/pmc_probe.py c2 4 4 2>&1 | tail -1 >> $O/probe.txt
done
cat $O/probe.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
