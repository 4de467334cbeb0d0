#!/bin/bash
# GPU box, with profiles/r06/experiments/k_sample_polled_finalisation.patch applied (the switch does not exist otherwise): polled step finalisation (PBA_FIN_POLL=1, default) against ticket -> gather (=0): configs[1], 50 and 20 steps, and a small window on the pipelined driver
run() { python bench.py --no-cpu-baseline --repeats 15 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s %7.2f us/iter (min %.2f max %.2f)  final cost %.12e' % ('$LABEL', 1e3 * d['ms_per_step'], 1e3 * d['ms_per_step_min'], 1e3 * d['ms_per_step_max'], d['lm']['final_cost']))"; }
for rep in 1 2; do
for v in 0 1; do
  LABEL="FIN_POLL=$v configs[1] 50 steps" PBA_FIN_POLL=$v run --steps 50 --warmup 5
  LABEL="FIN_POLL=$v configs[1] 20 steps" PBA_FIN_POLL=$v run --steps 20 --warmup 5
done
done
for v in 0 1; do
  LABEL="FIN_POLL=$v 5x5000x3x3 pipelined" PBA_RESIDENT=0 PBA_FIN_POLL=$v run --frames 5 --points 5000 --radius 1 --steps 30
done
