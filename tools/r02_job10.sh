#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02_job10; mkdir -p $O
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
for v in lib lib_sdwa lib lib_sdwa; do
  echo -n "$v: " >> $O/ab.txt
  PBRT_HIP_LIB_DIR=$R/pbrt_amd/$v timeout 300 python3 tools/pmc_probe.py c3 4 4 2>&1 | tail -1 >> $O/ab.txt
done
echo "== min_walkers x min_parked (c3 16 spp)" >> $O/ab.txt
bash tools/tune.sh c3 4 4 "28 32 36 40 44" "12 16 20 24" >> $O/ab.txt 2>&1
echo "== c2" >> $O/ab.txt
bash tools/tune.sh c2 4 4 "32 36 40" "12 16 20" >> $O/ab.txt 2>&1
cat $O/ab.txt
