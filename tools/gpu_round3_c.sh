#!/bin/bash
# GPU box: failing tests again, bench lines incl. multi-channel / inverse depth / two ranks on one GPU, multi-rank tests.
set -u
mkdir -p gpurun_out/e
timeout 1500 python -m pytest tests/test_gpu_configs0.py tests/test_gpu_multirank.py tests/test_gpu_bench_contract.py -q -m gpu -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/e/suite.log
grep -E "passed|failed|^FAILED|^ERROR|configs\[" gpurun_out/e/suite.log | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/e/bench20.json 2> gpurun_out/e/bench20.err
python bench.py --no-cpu-baseline --steps 20 --repeats 5 --channels 3 > gpurun_out/e/bench_c3.json 2> gpurun_out/e/bench_c3.err
python bench.py --no-cpu-baseline --steps 20 --repeats 5 --channels 8 > gpurun_out/e/bench_c8.json 2> gpurun_out/e/bench_c8.err
python bench.py --no-cpu-baseline --steps 20 --repeats 5 --inverse-depth > gpurun_out/e/bench_invd.json 2> gpurun_out/e/bench_invd.err
python bench.py --no-cpu-baseline --steps 20 --repeats 5 --config 4 > gpurun_out/e/bench_cfg4.json 2> gpurun_out/e/bench_cfg4.err
python bench.py --no-cpu-baseline --steps 20 --repeats 5 --config 3 > gpurun_out/e/bench_cfg3.json 2> gpurun_out/e/bench_cfg3.err
PBA_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 5 --points 25000 > gpurun_out/e/bench_2rank_peer.json 2> gpurun_out/e/bench_2rank_peer.err
PBA_PEER=0 PBA_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 5 --points 25000 > gpurun_out/e/bench_2rank_host.json 2> gpurun_out/e/bench_2rank_host.err
for f in gpurun_out/e/bench*.json; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print("  ms/step %.4f [%.4f, %.4f] value %.1f  dom %s frac %.3f  exchange %s" % (d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("exchange")))
    print("  per kernel us:", {k: round(1e3 * v, 2) for k, v in d["roofline"]["kernels_ms_per_launch"].items()})
except Exception as ex:
    print("  ERR", ex)
PY
done
