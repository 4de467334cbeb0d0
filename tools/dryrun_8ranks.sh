#!/bin/bash
# GPU box (ONE GPU): the driver's multi-GPU command line -- torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 -- with all eight ranks
# oversubscribing device 0 (PBA_BENCH_BACKEND=gloo: RCCL refuses duplicate devices, so the base transport is the host-staged one; the
# peer mailboxes, the self-test with fall-back, the strong-scaling record and the JSON schema are the ones of a real 8-GPU run).
# Writes gpurun_out/dry8/dryrun_8ranks_1gpu.json + .txt (copy into profiles/rNN/).
set -u
mkdir -p gpurun_out/dry8
export PBA_BENCH_BACKEND=gloo
export PBA_BENCH_STRONG_POINTS=${PBA_BENCH_STRONG_POINTS:-200000}
timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 8 --steps 20 --warmup 5 \
  > gpurun_out/dry8/dryrun_8ranks_1gpu.json 2> gpurun_out/dry8/dryrun_8ranks_1gpu.err
echo "exit code $?" > gpurun_out/dry8/dryrun_8ranks_1gpu.txt
python - <<'PY' >> gpurun_out/dry8/dryrun_8ranks_1gpu.txt 2>&1
import json
lines = [l for l in open("gpurun_out/dry8/dryrun_8ranks_1gpu.json") if l.startswith("{")]
assert len(lines) == 1, "expected ONE JSON line from rank 0, got %d" % len(lines)
d = json.loads(lines[0])
assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["value"] > 0
assert d["exchange"]["ranks_seen_by_transport"] == 8, d["exchange"]
assert d["config"]["observations"] == 8 * 50000 * 8
st = d["strong"]
assert "error" not in st, st
assert {"workload", "us_per_iteration_1gpu", "us_per_iteration", "n_gpus", "speedup_vs_1gpu_same_run", "transport", "steps", "repeats"} <= set(st), st
assert st["n_gpus"] == 8 and st["us_per_iteration"] > 0 and st["us_per_iteration_1gpu"] > 0
print("dry run ok: transport %s, ranks seen %d, %.1f us per LM iteration of the 8 x 50k-point window (all ranks on ONE device), strong record %s"
      % (d["exchange"]["transport"], d["exchange"]["ranks_seen_by_transport"], 1e3 * d["ms_per_step"], json.dumps(st)))
PY
cat gpurun_out/dry8/dryrun_8ranks_1gpu.txt; tail -3 gpurun_out/dry8/dryrun_8ranks_1gpu.err
