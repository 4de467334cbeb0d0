#!/bin/bash
# GPU box: headline window with k_schur launched over fewer / more workgroups (PBA_SCHUR_GRID), ms per LM step + kernel averages
mkdir -p gpurun_out/abgrid
for g in 512 640 768 896 1024; do
  PBA_SCHUR_GRID=$g python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('PBA_SCHUR_GRID=$g  %7.2f us/iter (min %.2f max %.2f)' % (1e3 * d['ms_per_step'], 1e3 * d['ms_per_step_min'], 1e3 * d['ms_per_step_max']))"
done
cd /tmp && export TMPDIR=/tmp
for g in 512 1024; do
  PBA_SCHUR_GRID=$g rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/abgrid/g$g -o g$g -- python /root/repo/bench.py --no-cpu-baseline --steps 50 --warmup 5 --repeats 3 > /dev/null 2>&1
  f=$(find /root/repo/gpurun_out/abgrid/g$g -name '*kernel_stats.csv' | head -1)
  echo "== grid $g"; head -6 "$f" | cut -d, -f1-6
done
