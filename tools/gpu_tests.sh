#!/bin/bash
# usage: tools/r02_gpu_tests.sh <tag> [pytest args]: the -m gpu suite + a C3/C2 probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
timeout 2400 python3 -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -25 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
for w in c3 c2; do PROBE_COUNTERS=1 timeout 600 python3 tools/pmc_probe.py $w 4 4 2>&1 | tail -2 >> $O/probe.txt; done
cat $O/probe.txt
