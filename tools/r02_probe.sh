#!/bin/bash
# usage: tools/r02_probe.sh <tag> [lib dirs...]   quick C3/C2 16-spp probes (+ walk counters) for each library variant
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
for v in "$@"; do
  echo "== $v" >> $O/probe.txt
  PROBE_COUNTERS=1 PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 600 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -2 >> $O/probe.txt
  PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 600 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -1 >> $O/probe.txt
  PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 600 python3 tools/pmc_probe.py c2 4 4 2>&1 | tail -1 >> $O/probe.txt
done
cat $O/probe.txt
