#!/bin/bash
# usage (on the GPU box): tools/measure_round.sh <round-tag> [workloads...]   default workloads: c3 big
# The measurement set behind profiles/<tag>_*:
#   1. the -m gpu parity suite;
#   2. per workload: bench.py under rocprofv3 --kernel-trace --stats (no CPU leg under the profiler: ADVICE r01), the
#      two HBM-traffic PMC passes on the same command (FETCH_SIZE / WRITE_SIZE, each in its own run), and the SQ / TCC /
#      instruction-class PMC passes on the workload's own frame (tools/pmc_passes.sh);
#   3. bench.py itself, unprofiled, for every workload (with the CPU baseline leg where it has one).
# tools/summarize_profile.py then condenses gpurun_out/ into profiles/ (run it here, on the CPU box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
WLS=${@:-c3 big}
O=$R/gpurun_out
cd $R
export PBRT_HIP_DEBUG_KNOBS=1
export PROBE_BUILDER=gpu   # the probes use the builder bench.py's default does (the device builder)
timeout 2400 python3 -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -40 > $O/${TAG}_pytest_gpu.log
python3 -c "from oracle import binding as ob; ob.build(native=True)"   # (the CPU leg's oracle is built before any profiler runs)
# FETCH_SIZE against known bytes in the kernel's access shape (round 6): the factor summarize_profile.py stores in pmc_<wl>.json
bash tools/fetch_size_calibration.sh fetch_cal_${TAG} > $O/fetch_cal_${TAG}.log 2>&1
for w in $WLS; do
  cd /tmp && export TMPDIR=/tmp
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_${w}_trace -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_${w}_trace.json 2> $O/prof_${TAG}_${w}_trace.err
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_${TAG}_${w}_fetch -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-counters > $O/prof_${TAG}_${w}_fetch.json 2> $O/prof_${TAG}_${w}_fetch.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_${TAG}_${w}_write -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-counters > $O/prof_${TAG}_${w}_write.json 2> $O/prof_${TAG}_${w}_write.err
  cd $R
  # PMC on the workload's own frame (C3: 512 spp, big: 64): the chunk count and so the hand-out granularity follow the sample count
  if [ $w = c3 ]; then bash tools/pmc_passes.sh pmc_${TAG}_${w} $w 32 16 > $O/pmc_${TAG}_${w}.log 2>&1; else bash tools/pmc_passes.sh pmc_${TAG}_${w} $w 8 8 > $O/pmc_${TAG}_${w}.log 2>&1; fi
done
# The profile of THIS build is in place before the bench lines are taken (on this box's copy of profiles/; run the same
# command on the CPU box afterwards to keep it): their roofline.frac then rests on counters of the library they time
# instead of being withheld as stale.
python3 tools/summarize_profile.py $TAG $WLS > $O/summarize_${TAG}.log 2>&1
# the headline with the DRIVER's arguments, clocks logged beside it; the other workloads with 2 timed steps
( while true; do date +%s.%N; rocm-smi --showclocks 2>/dev/null | grep -i "sclk"; sleep 2; done ) > $O/clocks_${TAG}.txt 2>&1 &
CL=$!
timeout 1800 python3 bench.py --workload c3 --steps 20 --warmup 5 > $O/bench_c3_${TAG}.json 2> $O/bench_c3_${TAG}.err
kill $CL
for w in c2 c1 c4 big; do
  timeout 1200 python3 bench.py --workload $w --steps 2 --warmup 1 > $O/bench_${w}_${TAG}.json 2> $O/bench_${w}_${TAG}.err
done
# the device builder's tree as built (no re-insertion passes) on this same box: what the optimisation is worth here
timeout 900 python3 bench.py --workload c3 --steps 3 --warmup 1 --builder gpu-plain --no-cpu-baseline --no-counters > $O/bench_c3_plaintree_${TAG}.json 2> $O/bench_c3_plaintree_${TAG}.err
timeout 900 python3 bench.py --workload big --steps 2 --warmup 1 --builder gpu-plain --no-cpu-baseline --no-counters > $O/bench_big_plaintree_${TAG}.json 2> $O/bench_big_plaintree_${TAG}.err
bash tools/rank_shares.sh > $O/rank_shares_${TAG}.txt 2>&1
tail -n 2 $O/${TAG}_pytest_gpu.log; cat $O/bench_c3_${TAG}.json
