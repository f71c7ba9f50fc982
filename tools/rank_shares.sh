#!/bin/bash
# rank 0's share of the C3 frame at the full 512 spp for 1, 2, 4, 8 ranks (strong scaling projection: one MI355X renders
# the share rank 0 of an N-GPU job would render) -- DESIGN.md section 7
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/rank_shares; mkdir -p $O; : > $O/shares.txt
cd $R
export PROBE_BUILDER=${PROBE_BUILDER:-gpu}
for w in 1 2 4 8; do
  echo -n "world $w: " >> $O/shares.txt
  PROBE_WORLD=$w timeout 600 python3 tools/pmc_probe.py c3 32 16 2>&1 | tail -1 >> $O/shares.txt
done
for r in 3 7; do
  echo -n "world 8 rank $r: " >> $O/shares.txt
  PROBE_RANK=$r PROBE_WORLD=8 timeout 600 python3 tools/pmc_probe.py c3 32 16 2>&1 | tail -1 >> $O/shares.txt
done
cat $O/shares.txt
