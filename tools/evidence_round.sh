#!/bin/bash
# GPU box: profile data of the final build + the whole suite + bench matrix (final evidence of a round: tools/evidence_round.sh <tag, e.g. r04final>)
set -u
TAG=${1:-r06final}
mkdir -p gpurun_out/g     # bench matrix + rocprof + whole GPU suite of one round (copy what you keep into profiles/rNN/)
bash tools/profile_round.sh $TAG > gpurun_out/g/profile.log 2>&1
find gpurun_out/prof_$TAG -name "*.rocpd" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
python bench.py --steps 20 --warmup 5 > gpurun_out/g/bench_driver_args.json 2> /dev/null
python bench.py --no-cpu-baseline --steps 20 --repeats 9 --config 3 > gpurun_out/g/bench_config3.json 2> /dev/null
python bench.py --no-cpu-baseline --steps 20 --repeats 9 --config 4 > gpurun_out/g/bench_config4.json 2> /dev/null
python bench.py --no-cpu-baseline --steps 20 --repeats 9 --channels 3 > gpurun_out/g/bench_channels3.json 2> /dev/null
python bench.py --no-cpu-baseline --steps 20 --repeats 9 --channels 8 > gpurun_out/g/bench_channels8.json 2> /dev/null
python bench.py --no-cpu-baseline --steps 20 --repeats 9 --inverse-depth > gpurun_out/g/bench_inverse_depth.json 2> /dev/null
PBA_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 9 --points 25000 > gpurun_out/g/bench_2ranks_1gpu_peer.json 2> /dev/null
PBA_PEER=0 PBA_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 9 --points 25000 > gpurun_out/g/bench_2ranks_1gpu_hoststaged.json 2> /dev/null
python bench.py --config 3 --emulate-rank-of 8 --steps 20 --repeats 9 > gpurun_out/g/config3_rank_of_8.json 2> /dev/null
python bench.py --config 2 --steps 20 --warmup 5 --repeats 9 > gpurun_out/g/bench_config2.json 2> /dev/null
bash tools/ab_resident.sh > gpurun_out/g/ab_resident_vs_pipelined.txt 2>&1
PBA_RES_TRACE=2 python bench.py --no-cpu-baseline --frames 5 --points 5000 --radius 1 --steps 30 --repeats 3 2>&1 > /dev/null | grep -A3 "resident solve" | tail -4 > gpurun_out/g/resident_phase_trace_final.txt
bash tools/dryrun_8ranks.sh > gpurun_out/g/dry8.log 2>&1; cp gpurun_out/dry8/dryrun_8ranks_1gpu.json gpurun_out/dry8/dryrun_8ranks_1gpu.txt gpurun_out/g/ 2> /dev/null
PBA_RANDOM_CASES=240 timeout 2400 python -m pytest tests/test_gpu_random_shapes.py -q -m gpu 2>&1 | grep -E "passed|failed|arbit" | tail -20 > gpurun_out/g/random_sweep_240.txt
bash tools/phase_only.sh g > gpurun_out/g/phase_only.log 2>&1
PBA_TRACE_SOLVE=1 python tools/time_addframe.py 12 4096 5 1 > gpurun_out/g/addframe_timings.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/g/stats3" -o s -- python $OLDPWD/bench.py --no-cpu-baseline --repeats 3 --steps 20 --config 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/g/stats2" -o s -- python $OLDPWD/bench.py --config 2 --repeats 3 --steps 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/g/stats_small" -o s -- python $OLDPWD/bench.py --no-cpu-baseline --frames 5 --points 5000 --radius 1 --repeats 3 --steps 30 > /dev/null 2>&1
cd "$OLDPWD"
find gpurun_out/g/stats2 gpurun_out/g/stats_small -name "*.csv" ! -name "*kernel_stats.csv" -delete
find gpurun_out/g -name "*.rocpd" -delete; find gpurun_out/g -name "*.db" -delete; find gpurun_out/g/stats3 -name "*.csv" ! -name "*kernel_stats.csv" -delete
timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/g/suite.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/g/suite.log | tail -8
