#!/bin/bash
# GPU box: candidate geometry in the sampling kernel -- correctness, then same-library A/B (PBA_GEOM_IN_SAMPLE)
OUT=gpurun_out/e13; mkdir -p $OUT
export PBA_WINDOW_CACHE=/tmp/pba_window_cache
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py tests/test_gpu_inverse_depth.py tests/test_gpu_dropin_class.py tests/test_gpu_random_shapes.py tests/test_gpu_configs0.py -q -m gpu -x > $OUT/tests.txt 2>&1
grep -E "passed|failed|^FAILED|^E  " $OUT/tests.txt | tail -6
bash tools/ab_dist.sh 6 "--steps 20 --warmup 5" main:PBA_GEOM_IN_SAMPLE=0 main 2>&1 | tee $OUT/ab1.txt
bash tools/ab_dist.sh 2 "--config 3 --steps 20 --repeats 5" main:PBA_GEOM_IN_SAMPLE=0 main 2>&1 | tee $OUT/ab3.txt
for v in 0 1; do PBA_GEOM_IN_SAMPLE=$v python bench.py --config 3 --emulate-rank-of 8 --steps 20 --repeats 7 2>/dev/null | python -c '
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith("{")][0]); s = d["strong_projection"]
print("GEOM_IN_SAMPLE=%s rank-of-8: %.1f us per iteration per rank" % (sys.argv[1], s["us_per_iteration_one_rank"]), {k[:12]: round(v, 1) for k, v in s["rank_kernels_us_host_stepped"].items()})' $v; done 2>&1 | tee $OUT/r8.txt
