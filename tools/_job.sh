export PBRT_HIP_DEBUG_KNOBS=1
echo "== c3"; bash tools/tune.sh c3 4 4 "28 32 36 40" "12 16 20" 2>&1 | grep -v "^\["
echo "== c2"; bash tools/tune.sh c2 4 4 "28 32 36 40" "12 16 20" 2>&1 | grep -v "^\["
for v in lib lib_s2 lib_s4; do for wl in c3 c2; do echo -n "$v $wl: "; PBRT_HIP_LIB_DIR=$GRAFT_REPO_ROOT/pbrt_amd/$v timeout 300 python tools/pmc_probe.py $wl 4 4 2>&1 | grep kernel_ms; done; done
