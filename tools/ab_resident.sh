#!/bin/bash
# GPU box: resident vs pipelined driver over window shapes that fit one resident round (one line per shape and driver) + the drop-in class's addFrame timers
mkdir -p gpurun_out/abres
run() { for r in 1 0; do PBA_RESIDENT=$r python bench.py --no-cpu-baseline --steps 30 --repeats 11 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('PBA_RESIDENT=$r %-64s %7.2f us/iter (min %.2f max %.2f) obs %d' % (d['config']['workload'][:64], 1e3 * d['ms_per_step'], 1e3 * d['ms_per_step_min'], 1e3 * d['ms_per_step_max'], d['config']['observations']))"; done; }
run --frames 5 --points 5000 --radius 1
run --frames 5 --points 2500 --radius 1
run --frames 5 --points 10000 --radius 1
run --frames 5 --points 5000 --radius 2
run --frames 8 --points 3000 --radius 2
run --frames 8 --points 8000 --radius 2
run --frames 3 --points 4000 --radius 1
for r in 1 0; do echo "== addFrame, PBA_RESIDENT=$r"; PBA_RESIDENT=$r PBA_TRACE_SOLVE=1 python tools/time_addframe.py 14 4096 5 1 2>&1 | grep -E "^addFrame|^optimize phases|^pba_solve" | tail -12; done
