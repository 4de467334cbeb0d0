#!/bin/bash
# GPU box: round-5 baseline numbers of the build as checked in (bench lines + rank-of-8 record)
set -u
OUT=gpurun_out/${1:-r5base}
mkdir -p $OUT
for a in "--steps 20 --warmup 5" "--steps 50" "--config 3 --steps 20 --repeats 5"; do
  bash tools/ab_bench.sh "X=1" $a | tee -a $OUT/bench_lines.txt
done
python bench.py --config 3 --emulate-rank-of 8 --steps 20 --repeats 9 > $OUT/config3_rank_of_8.json 2> /dev/null
tail -c 1500 $OUT/config3_rank_of_8.json
