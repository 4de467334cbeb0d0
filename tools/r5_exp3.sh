#!/bin/bash
# GPU box: k_schur launch size against the reduction volume, rank-of-8 emulation of configs[3] and the full configs[3] window
OUT=gpurun_out/e4; mkdir -p $OUT
export PBA_WINDOW_CACHE=/tmp/pba_window_cache
(timeout 700 python -m pytest tests/test_gpu_dropin_class.py tests/test_gpu_frontend.py tests/test_gpu_multichannel.py tests/test_gpu_inverse_depth.py tests/test_gpu_multirank.py -q -m gpu -x 2>&1 | tail -4
for g in 1024 768 512 384 256; do
  PBA_SCHUR_GRID=$g python bench.py --config 3 --emulate-rank-of 8 --steps 20 --repeats 7 2>/dev/null | python -c '
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith("{")][0]); s = d["strong_projection"]
print("SCHUR_GRID=%s rank-of-8: %.1f us per iteration per rank (full window %.1f)" % (sys.argv[1], s["us_per_iteration_one_rank"], s["us_per_iteration_full_window_1gpu"]), {k[:12]: round(v, 1) for k, v in s["rank_kernels_us_host_stepped"].items()})' $g
done
for g in 1024 512; do PBA_SCHUR_GRID=$g bash tools/ab_bench.sh "SCHUR_GRID=$g" --config 3 --steps 20 --repeats 5; done
for g in 1024 512 256; do PBA_SCHUR_GRID=$g bash tools/ab_bench.sh "SCHUR_GRID=$g" --steps 20 --warmup 5; done
) 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tee $OUT/log.txt
