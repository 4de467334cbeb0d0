#!/bin/bash
# GPU box: what has to be re-taken after the LAST source change of a round -- kernel stats + PMC passes (profiles/traffic.json carries the
# build id), the resident tests, a short soak of both drivers, the whole -m gpu suite with its slowest tests, a longer random-shape sweep.
set -u
TAG=${1:-r06final}
mkdir -p gpurun_out/fin
bash tools/profile_round.sh $TAG > gpurun_out/fin/profile.log 2>&1
find gpurun_out/prof_$TAG -name "*.rocpd" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
timeout 600 python -m pytest tests/test_gpu_resident.py -q -m gpu 2>&1 | tail -3 > gpurun_out/fin/resident_tests.txt
python tools/soak.py 60 30 small > gpurun_out/fin/soak_resident_60s.txt 2>&1
python tools/soak.py 40 20 > gpurun_out/fin/soak_pipelined_40s.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/fin/suite_tail.txt
PBA_RANDOM_CASES=${2:-500} timeout 1500 python -m pytest tests/test_gpu_random_shapes.py -q -m gpu 2>&1 | grep -E "passed|failed|arbit" | tail -20 > gpurun_out/fin/random_sweep.txt
tail -2 gpurun_out/fin/resident_tests.txt; tail -3 gpurun_out/fin/soak_resident_60s.txt; grep -E "passed|failed" gpurun_out/fin/suite_tail.txt; tail -2 gpurun_out/fin/random_sweep.txt; tail -1 gpurun_out/prof_$TAG/bench.json | cut -c1-400
