#!/bin/bash
# round 4: the device builder's re-insertion pass against the plain device tree (walk counters, probe frame times)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04p1_reinsert_probe.txt
: > $out
for wl in c3 c2; do
  for b in gpu-plain gpu; do
    echo "== $wl builder=$b" >> $out
    PROBE_BUILDER=$b PROBE_COUNTERS=1 timeout 600 python tools/pmc_probe.py $wl 4 4 >> $out 2>&1
    PROBE_BUILDER=$b timeout 600 python tools/pmc_probe.py $wl 4 4 2>&1 | grep kernel_ms >> $out
  done
done
cat $out
