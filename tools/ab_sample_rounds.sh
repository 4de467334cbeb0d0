#!/bin/bash
# GPU box, EXPERIMENT lib (PBA_LIB) = a build with profiles/r06/experiments/k_sample_rounds_pair_pad.patch applied: k_sample's 1.53 rounds of workgroups (1563 blocks over 1024 slots) against (a) 3 workgroups per CU via
# unused dynamic LDS at a window of 1532 blocks (= 2 x 766), (b) a paired launch: 784 workgroups running two blocks each
export PBA_LIB=photobundle_amd/libpba_hip_exp.so
run() { python bench.py --no-cpu-baseline --steps 50 --warmup 5 --repeats 11 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('kernel_us') or d.get('kernels') or {}
print('%-44s %7.2f us/iter (min %.2f)  %s' % ('$LABEL', 1e3 * d['ms_per_step'], 1e3 * d['ms_per_step_min'], json.dumps(d.get('kernel_shares_us', d.get('kernel_shares', '')))[:200]))"; }
LABEL="49000 pts, 4 wg/CU" run --points 49000
LABEL="49000 pts, pad 13000 (3 wg/CU)" PBA_SAMPLE_PAD_LDS=13000 run --points 49000
LABEL="49000 pts, pad 41000 (2 wg/CU)" PBA_SAMPLE_PAD_LDS=41000 run --points 49000
LABEL="50000 pts, baseline" run
LABEL="50000 pts, PAIR=2" PBA_SAMPLE_PAIR=2 run
LABEL="50000 pts, PAIR=2 + pad 13000" PBA_SAMPLE_PAIR=2 PBA_SAMPLE_PAD_LDS=13000 run
LABEL="50000 pts, PAIR=3" PBA_SAMPLE_PAIR=3 run
