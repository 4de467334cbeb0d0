"""bench.py contract on the GPU box: one JSON line with the agreed keys at N = 1, and the N = 2 launch path
(torch.distributed.run, per-rank point shards, max-over-ranks timing, rank 0 prints).  Two ranks share the box's GPU
through the PBA_BENCH_BACKEND=gloo hook: RCCL refuses duplicate devices, which also exercises the host-staged fallback
transport bench.py switches to when RCCL cannot be initialised."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--points", "20000",
                        "--cpu-points", "2000", "--cpu-steps", "3"], capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 10 and d["value"] > 0 and d["scaling"] == "weak"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"


@pytest.mark.timeout(900)
def test_two_rank_launch_path():
    env = dict(os.environ, PBA_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
                        "--warmup", "2", "--points", "10000"], capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["value"] > 0
    assert d["config"]["observations"] == 2 * 10000 * 8           # whole-job aggregate over both shards
    assert abs(d["value"] - 2 * d["iters_per_sec"]) < 1e-6 * d["value"]
