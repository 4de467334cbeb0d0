#!/bin/bash
# GPU box: parity subset around the new reduced solve + kernel stats of both solve variants + engine traces.
set -u
mkdir -p gpurun_out/nf gpurun_out/a
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_random_shapes.py tests/test_gpu_multirank.py tests/test_gpu_inverse_depth.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/a/parity.log
timeout 900 python -m pytest tests/test_gpu_configs0.py -x -q -m gpu -s 2>&1 | tail -60 > gpurun_out/a/configs0.log
OUT=$PWD/gpurun_out/a
cd /tmp && export TMPDIR=/tmp
for K in 0 1; do
  PBA_SOLVE=$K rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_solve$K" -o s -- python $OLDPWD/bench.py --no-cpu-baseline --repeats 3 --steps 50 > "$OUT/bench_solve$K.json" 2> /dev/null
  PBA_SOLVE=$K rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats3_solve$K" -o s -- python $OLDPWD/bench.py --config 3 --no-cpu-baseline --repeats 2 --steps 20 > "$OUT/bench3_solve$K.json" 2> /dev/null
done
cd "$OLDPWD"
for K in 0 1; do PBA_SOLVE=$K python bench.py --no-cpu-baseline --steps 20 > gpurun_out/a/bench20_solve$K.json 2>/dev/null; done
python tools/engine_trace.py poor > gpurun_out/nf/engine_poor.log 2>&1
python tools/engine_trace.py good > gpurun_out/nf/engine_good.log 2>&1
find gpurun_out/a -name "*.csv" ! -name "*kernel_stats.csv" -delete
find gpurun_out/a -name "*.rocpd" -delete -o -name "*.db" -delete
tail -3 gpurun_out/a/parity.log; tail -3 gpurun_out/a/configs0.log
