#!/bin/bash
# Soak run of the randomised parity test (tests/test_gpu_parity.py::test_random_scenes_match_oracle) beyond the 48 seeds of
# the suite: seeds 0..N-1, those from 48 on also drawing 2 500- and 20 000-triangle scenes.  On an MI355X:
#   bash tools/soak.sh 2000 > gpurun_out/soak.txt          (bash tools/soak.sh 60000 90000: seeds 90000 ... 149999)
n=${1:-1000}
first=${2:-0}
cd "$(dirname "$0")/.."
PBRT_SOAK_SEEDS=$n PBRT_SOAK_FIRST=$first python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k test_random_scenes_match_oracle -n 4 -p no:cacheprovider 2>&1 | tail -15
