#!/bin/bash
# round 4: the out-of-cache workload (12 M triangles) with and without the device builder's re-insertion pass
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04p3_big_probe.txt
: > $out
for b in gpu-plain gpu; do
  echo "== big builder=$b" >> $out
  PROBE_BUILDER=$b PROBE_COUNTERS=1 timeout 900 python tools/pmc_probe.py big 4 4 2>&1 | egrep "accelerator|kernel_ms|production" >> $out
  PROBE_BUILDER=$b timeout 900 python tools/pmc_probe.py big 8 8 2>&1 | grep kernel_ms >> $out
done
cat $out
