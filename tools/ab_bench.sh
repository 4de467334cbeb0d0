#!/bin/bash
# usage: tools/ab_bench.sh "<ENV=..>" [bench args]: prints ms/step and per-kernel shares
env $1 python bench.py --no-cpu-baseline "${@:2}" 2>/dev/null | python -c '
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith("{")][0])
r = d["roofline"]
print("%s: ms/step %.4f (min %.4f max %.4f)" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"]), {k[:14]: round(1e3 * v, 2) for k, v in r["kernels_ms_per_launch"].items()})
' "$1"
