#!/bin/bash
# usage (here, on the CPU box): tools/ab_variants.sh name1="-DFLAG ..." name2="..."  -> builds pbrt_amd/lib_<name>/libpbrt_hip.so per variant
# (hipcc cross-compiles; the .so files travel to the GPU box with the snapshot).  On the GPU box select one with
# PBRT_HIP_LIB_DIR=$GRAFT_REPO_ROOT/pbrt_amd/lib_<name>.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( PBRT_HIP_LIB_DIR=$R/pbrt_amd/lib_$name PBRT_HIP_EXTRA_FLAGS="$flags" python3 -c "from pbrt_amd.build import build_hip; build_hip(force=True)" \
      > /tmp/ab_$name.log 2>&1 && echo "built $name" || { echo "FAILED $name"; tail -5 /tmp/ab_$name.log; } ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 2 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
