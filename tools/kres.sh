#!/bin/bash
# usage: tools/kres.sh [extra hipcc flags]  -> VGPRs / spills / scratch of the production render kernels (device-only compile of
# kernels.hip, ~10 s).  For register-pressure work: tools/kres.sh -DPBRT_RENDER_WAVES_PER_SIMD=5 -DPBRT_QUAD_LDS_STACK=32
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 --offload-arch=gfx950 "$@" \
  --offload-device-only -Rpass-analysis=kernel-resource-usage -I$R/include -x hip -c $R/pbrt_amd/csrc/kernels.hip -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|VGPRs:|VGPR Spill|ScratchSize|error" | sed 's/.*remark: *//;s/ \[-Rpass.*//' |
  awk '/Function Name/{n=$3} /VGPRs:/{v=$2} /Spill/{sp=$NF} /ScratchSize/{print n, "VGPRs", v, "spill", sp, "scratch", $NF; sp=0} /error/{print}' |
  grep -E "render_kernelILb0ELb0ELb0E|error" | sed 's/_ZN8pbrt_hip12_GLOBAL__N_113//;s/EEvNS_8DevSceneENS_12RenderParamsE//'
