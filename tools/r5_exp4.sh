#!/bin/bash
# GPU box: (1) the two failures of the 240-case sweep in detail; (2) k_schur with the 23-double LDS rows at 2 (s2) and 3 (s3) waves per SIMD
OUT=gpurun_out/e9; mkdir -p $OUT
export PBA_WINDOW_CACHE=/tmp/pba_window_cache
PBA_RANDOM_CASES=240 timeout 1500 python -m pytest tests/test_gpu_random_shapes.py -q -rf --tb=short -m gpu > $OUT/sweep240_full.txt 2>&1
grep -E "^FAILED|passed|failed" $OUT/sweep240_full.txt | tail -6
for v in s2 s3; do
  PBA_LIB=photobundle_amd/libpba_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py tests/test_gpu_inverse_depth.py -q -m gpu -x > $OUT/tests_$v.txt 2>&1
  echo "$v: $(grep -E 'passed|failed' $OUT/tests_$v.txt | tail -1)"
done
bash tools/ab_dist.sh 4 "--steps 20 --warmup 5" main s2 s3 2>&1 | tee $OUT/ab1.txt
bash tools/ab_dist.sh 2 "--config 3 --steps 20 --repeats 5" main s3 2>&1 | tee $OUT/ab3.txt
for g in 1042 1280 1536; do PBA_LIB=photobundle_amd/libpba_hip_s3.so PBA_SCHUR_GRID=$g bash tools/ab_bench.sh "s3_GRID=$g" --steps 20 --warmup 5; done 2>&1 | tee $OUT/grid.txt
