#!/bin/bash
# Runs ON THE GPU BOX: throughput of the other BASELINE.json shapes (one JSON line each, no CPU baseline).
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-70s %8.1f us/iter  %8.1f it/s  whole_iteration_frac %.2f  %s' % (d['config']['workload'][:70], 1e3 * d['ms_per_step'], d['value'], d['roofline']['whole_iteration_frac'], {k.split(' ')[0]: round(1e3 * v, 1) for k, v in d['roofline']['kernels_ms_per_launch'].items()}))"; }
run
run --visibility causal
run --frames 5 --points 2000 --radius 1
run --frames 16 --points 120000 --steps 20
run --radius 5 --huber 0.05 --steps 20
run --radius 3 --steps 20
run --radius 1
