#!/bin/bash
# GPU box: the whole -m gpu suite (no -x), bench lines.
set -u
mkdir -p gpurun_out/c
python bench.py --steps 20 --warmup 5 > gpurun_out/c/bench20.json 2> gpurun_out/c/bench20.err
timeout 3500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/c/suite.log
tail -8 gpurun_out/c/suite.log
