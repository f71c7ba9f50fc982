#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
bash tools/pmc_classes.sh r02_job8/ubench -- $R/tools/ubench/valu_issue pmc
bash tools/pmc_classes.sh r02_job8/render -- python3 $R/tools/pmc_probe.py c3 4 4
grep -h "kernel_ms" gpurun_out/r02_job8/render/p1.log
