#!/bin/bash
# Dev loop ON THE GPU BOX (gpurun): targeted parity tests of the reduced solve, A/B bench lines, per-phase cycle stamps.
set -u
OUT=gpurun_out/${1:-dev}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_random_shapes.py tests/test_gpu_multirank.py tests/test_gpu_inverse_depth.py tests/test_gpu_multichannel.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
for a in "--steps 20" "--steps 50" "--config 3 --steps 20 --repeats 5"; do
  python bench.py --no-cpu-baseline $a 2>$OUT/bench.err | python -c '
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith("{")][0])
r = d["roofline"]
print("%s: ms/step %.4f (min %.4f max %.4f)" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"]), {k[:14]: round(1e3 * v, 2) for k, v in r["kernels_ms_per_launch"].items()})
' "$a" | tee -a $OUT/bench_lines.txt
done
# phase stamps: timing build into a second library
make -s -C photobundle_amd/csrc TIMING=1 OUT=../libpba_hip_timing.so > $OUT/make_timing.log 2>&1
PBA_LIB=photobundle_amd/libpba_hip_timing.so PBA_ASYNC=0 PBA_SCHUR_TIMING=3 python tools/solve_phase_timing.py 8 50000 2>&1 | grep -E "solve_blocked|k_reduce_solve" | tail -4 | tee $OUT/solve_phase.txt
PBA_LIB=photobundle_amd/libpba_hip_timing.so PBA_ASYNC=0 PBA_SCHUR_TIMING=3 python tools/solve_phase_timing.py 16 50000 2>&1 | grep -E "solve_blocked|k_reduce_solve" | tail -4 | tee -a $OUT/solve_phase.txt
