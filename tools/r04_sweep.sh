#!/bin/bash
# round 4: sweep of the device re-insertion's knobs on the C3 probe frame (16 spp)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04p2_reinsert_sweep.txt
: > $out
export PBRT_HIP_DEBUG_KNOBS=1
run() {  # label, env...
  echo "== $1" >> $out; shift
  env "$@" PROBE_COUNTERS=1 timeout 900 python tools/pmc_probe.py ${WL:-c3} 4 4 2>&1 | egrep "accelerator|kernel_ms|production" >> $out
  env "$@" timeout 900 python tools/pmc_probe.py ${WL:-c3} 4 4 2>&1 | grep kernel_ms >> $out
}
run plain PROBE_BUILDER=gpu-plain
for p in 2 4 8 12 20 32; do run "passes=$p" PROBE_BUILDER=gpu PBRT_HIP_GPU_REINSERT=$p; done
for qk in 0 0.004 0.016; do run "qk=$qk" PROBE_BUILDER=gpu PBRT_HIP_REINSERT_QK=$qk; done
for qw in 1 4; do run "qw=$qw" PROBE_BUILDER=gpu PBRT_HIP_REINSERT_QW=$qw; done
run host-optimized PROBE_BUILDER=host-optimized
cat $out
