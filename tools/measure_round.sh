#!/bin/bash
# usage (on the GPU box): tools/measure_round.sh <round-tag>
# The measurement set behind profiles/<tag>_*: GPU parity tests, the default bench under rocprofv3
# --kernel-trace --stats, the two HBM-traffic PMC passes (each in its own run), and the other workloads' benches.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/${TAG}_pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_trace -- python3 $R/bench.py --steps 2 --warmup 1 > $O/bench_c3_${TAG}.json 2> $O/bench_c3_${TAG}.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-counters > $O/prof_${TAG}_fetch.json 2> $O/prof_${TAG}_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-counters > $O/prof_${TAG}_write.json 2> $O/prof_${TAG}_write.err
cd $R
for w in c2 c1 c4; do
  timeout 900 python3 bench.py --workload $w --steps 2 --warmup 1 > $O/bench_${w}_${TAG}.json 2> $O/bench_${w}_${TAG}.err
done
tail -n 2 $O/${TAG}_pytest_gpu.log; cat $O/bench_c3_${TAG}.json
