#!/bin/bash
# round 5: the device re-insertion pass after the sparse refit / key tie-break / fixed-point growth term: build time, passes, moves,
# area before / after, and the production walk's fetches per ray on the C3 / C2 probe frames; PBRT_HIP_REINSERT_FULL_REFIT=1 = round 4's refit
cd "$(dirname "$0")/.."
export PBRT_HIP_DEBUG_KNOBS=1 PROBE_COUNTERS=1 PROBE_BUILDER=gpu
for wl in c3 c2 big; do
  for full in "" 1; do
    echo "== $wl full_refit=${full:-0}"
    PBRT_HIP_REINSERT_FULL_REFIT=$full timeout 900 python3 tools/pmc_probe.py $wl 4 4 2>&1 | grep -v "^RAYS"
  done
done
