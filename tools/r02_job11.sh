#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02_job11; mkdir -p $O
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
timeout 900 python3 -m pytest tests -m gpu -x -q -k "gpu_built or random_scenes" 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
for b in host gpu; do for w in c3 c2; do
  echo -n "$b $w: " >> $O/probe.txt
  PROBE_BUILDER=$b timeout 600 python3 tools/pmc_probe.py $w 4 4 2>&1 | grep -E "accelerator|kernel_ms" | tr '\n' ' ' >> $O/probe.txt; echo >> $O/probe.txt
  PROBE_BUILDER=$b timeout 600 python3 tools/pmc_probe.py $w 4 4 2>&1 | grep -E "kernel_ms" >> $O/probe.txt
done; done
echo -n "lbvh c3: " >> $O/probe.txt
PBRT_HIP_GPU_BUILDER=lbvh PROBE_BUILDER=gpu timeout 600 python3 tools/pmc_probe.py c3 4 4 2>&1 | grep -E "accelerator|kernel_ms" | tr '\n' ' ' >> $O/probe.txt; echo >> $O/probe.txt
cat $O/probe.txt
