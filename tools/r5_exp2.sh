#!/bin/bash
# GPU box: round-5 experiment 2 -- fused final pass (decide + flush in the last workgroup), lm init in the first kernel, k_schur grid sweep
OUT=gpurun_out/e2; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_dropin_class.py tests/test_gpu_multirank.py tests/test_gpu_inverse_depth.py tests/test_gpu_multichannel.py -q -m gpu -x 2>&1 | tail -6
bash tools/ab_prebuilt.sh "--steps 20 --warmup 5" base main
PBA_FUSE_FINAL=0 bash tools/ab_bench.sh "V=main_nofusefinal" --steps 20 --warmup 5
for g in 782 896 960 1000; do PBA_SCHUR_GRID=$g bash tools/ab_bench.sh "SCHUR_GRID=$g" --steps 20 --warmup 5; done
PBA_TRACE_SOLVE=1 PBA_LIB=photobundle_amd/libpba_hip_base.so python bench.py --no-cpu-baseline --steps 20 --repeats 3 2>&1 | grep solve_async | tail -2
PBA_TRACE_SOLVE=1 python bench.py --no-cpu-baseline --steps 20 --repeats 3 2>&1 | grep solve_async | tail -2
) 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tee $OUT/log.txt
