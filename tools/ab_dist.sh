#!/bin/bash
# Distribution A/B on the GPU box: tools/ab_dist.sh N "<bench args>" spec1 spec2 ...   with spec = name[:ENV=VAL[,ENV=VAL...]]
# (name = prebuilt photobundle_amd/libpba_hip_<name>.so, "main" = libpba_hip.so).  N alternating processes per spec (the window
# comes from the pickle cache after the first), then min / median / max of the per-process median ms per step.
N=$1; ARGS="$2"; shift; shift
export PBA_WINDOW_CACHE=/tmp/pba_window_cache
TMP=$(mktemp -d)
for rep in $(seq 1 $N); do
  for spec in "$@"; do
    nm=${spec%%:*}; envs=""; [ "$spec" != "$nm" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    lib=photobundle_amd/libpba_hip_$nm.so; [ "$nm" = main ] && lib=photobundle_amd/libpba_hip.so
    env PBA_LIB=$lib $envs python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c '
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith("{")][0])
k = d["roofline"]["kernels_ms_per_launch"]
print("%.4f %.4f %s" % (d["ms_per_step"], d["ms_per_step_min"], " ".join("%.2f" % (1e3 * v) for v in k.values())))' >> "$TMP/$(echo $spec | tr ':,=' '___')"
  done
done
for spec in "$@"; do
  python - "$spec" "$TMP/$(echo $spec | tr ':,=' '___')" <<'PY'
import sys, statistics as st
rows = [l.split() for l in open(sys.argv[2])]
med = sorted(float(r[0]) for r in rows); mn = sorted(float(r[1]) for r in rows)
ks = [[float(x) for x in r[2:]] for r in rows]
kmed = [st.median(c) for c in zip(*ks)] if ks else []
print("%-28s n=%d  ms/step median-of-medians %.4f (min %.4f max %.4f)  best repeat %.4f  kernels(med us): %s" % (sys.argv[1], len(med), st.median(med), med[0], med[-1], mn[0], " ".join("%.2f" % k for k in kmed)))
PY
done
