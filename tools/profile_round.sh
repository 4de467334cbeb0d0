#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: kernel stats, two PMC passes and the bench line of configs[1].
# Output: gpurun_out/prof_<tag>/ ; copy the summaries into profiles/rNN/ afterwards (tools/collect_profiles.py).
set -u
TAG=${1:-final}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (configs[1] runs on the pipelined driver; rocprofv3 crashes in its exit handlers -- after its output is written -- in a process that made a cooperative launch)
BENCH="python $OLDPWD/bench.py --no-cpu-baseline --repeats 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $BENCH > "$OUT/bench_under_rocprof.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -o q -- $BENCH > /dev/null 2>&1
cd "$OLDPWD"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json"
