#!/bin/bash
# usage: tools/tune.sh <workload> <sx> <sy> "<walkers list>" "<parked list>"   (runs tools/pmc_probe.py per setting)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for w in $4; do for p in $5; do
  echo -n "min_walkers=$w min_parked=$p : "
  PBRT_HIP_MIN_WALKERS=$w PBRT_HIP_MIN_PARKED=$p python3 $R/tools/pmc_probe.py $1 $2 $3 2>&1 | tail -1
done; done
