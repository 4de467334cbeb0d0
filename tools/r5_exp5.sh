#!/bin/bash
# GPU box: k_schur on 64-observation tiles (one wave per workgroup, eight per CU): correctness on <= 7 free cameras, then timing
OUT=gpurun_out/e11; mkdir -p $OUT
export PBA_WINDOW_CACHE=/tmp/pba_window_cache
PBA_LIB=photobundle_amd/libpba_hip_t64.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "linearisation or trajectory_matches or huber_causal or determinism or clamped" > $OUT/tests_t64.txt 2>&1
echo "t64: $(grep -E 'passed|failed' $OUT/tests_t64.txt | tail -1)"
PBA_LIB=photobundle_amd/libpba_hip_t64.so timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "records" >> $OUT/tests_t64.txt 2>&1
echo "t64 fullsize records: $(grep -E 'passed|failed' $OUT/tests_t64.txt | tail -1)"
bash tools/ab_dist.sh 4 "--steps 20 --warmup 5" main t64 2>&1 | tee $OUT/ab1.txt
for g in 1024 1536 2048; do PBA_LIB=photobundle_amd/libpba_hip_t64.so PBA_SCHUR_GRID=$g bash tools/ab_bench.sh "t64_GRID=$g" --steps 20 --warmup 5; done 2>&1 | tee $OUT/grid.txt
