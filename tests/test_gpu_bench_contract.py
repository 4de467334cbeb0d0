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
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_source_id", "traffic_matches_build"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    # core count and CPU model stated; the all-cores leg beside the reference's 4-thread cap wherever the host has more
    assert d["cpu_baseline"]["host_cores"] >= d["cpu_baseline"]["cores"] and d["cpu_baseline"]["cpu_model"]
    # "all cores" = every core the process may use (affinity mask capped by the cgroup CPU quota), not every hardware thread of the host
    assert d["cpu_baseline"]["usable_cores"] <= d["cpu_baseline"]["host_cores"]
    if d["cpu_baseline"]["usable_cores"] > d["cpu_baseline"]["cores"]:
        a = d["cpu_baseline_all_cores"]
        assert a["cores"] == d["cpu_baseline"]["usable_cores"] and a["value"] > 0 and a["kind"] == "port"


@pytest.mark.timeout(900)
def test_two_rank_launch_path():
    env = dict(os.environ, PBA_BENCH_BACKEND="gloo", PBA_BENCH_STRONG_POINTS="6000")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
                        "--warmup", "2", "--points", "10000"], capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["value"] > 0
    assert d["config"]["observations"] == 2 * 10000 * 8           # whole-job aggregate over both shards
    # `value` is the rate of the ONE window both ranks solve together, never multiplied by the rank count ...
    assert d["value"] == d["iters_per_sec"] and abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    # ... what scales with N is the residual rate: all 2 x 80k blocks x 25 residuals per Jacobian pass
    assert d["residuals_per_sec"] > 0.9 * d["value"] * 2 * 10000 * 8 * 25
    # the strong-scaling record of the same launch (configs[3] shape, shrunk here): rank 0 alone, then both ranks
    st = d["strong"]
    assert "error" not in st, st
    assert st["n_gpus"] == 2 and st["us_per_iteration_1gpu"] > 0 and st["us_per_iteration"] > 0
    assert abs(st["speedup_vs_1gpu_same_run"] - st["us_per_iteration_1gpu"] / st["us_per_iteration"]) < 1e-9


@pytest.mark.timeout(900)
def test_rank_of_eight_emulation_and_counter_file_id():
    """--config 3 --emulate-rank-of 8: the multi-rank code path at world = 1 on a shard, projected speed-up printed;
    and the roofline object says which build the committed counter file belongs to."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--emulate-rank-of", "8", "--points", "16000",
                        "--steps", "6", "--warmup", "2", "--repeats", "3"], capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    sp = d["strong_projection"]
    assert sp["ranks"] == 8 and sp["transport"].endswith("+peer"), sp
    assert 0 < sp["us_per_iteration_one_rank"] < sp["us_per_iteration_full_window_1gpu"]
    assert abs(sp["projected_speedup"] - sp["us_per_iteration_full_window_1gpu"] / sp["projected_us_per_iteration"]) < 1e-9
    assert len(d["roofline"]["kernel_source_id"]) == 12


@pytest.mark.timeout(900)
def test_config2_pyramid_line():
    """--config 2: BASELINE configs[2], the 3-level pyramid path, as ONE JSON line -- per level (1241x376, 621x188,
    311x94) the time per LM iteration, the kernel shares and the roofline figures, plus the device-side pyramid build."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--points", "6000", "--steps", "6", "--warmup", "2",
                        "--repeats", "3"], capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["workload"].startswith("configs[2]")
    assert [l["image"] for l in d["levels"]] == ["1241x376 u8", "621x188 u8", "311x94 u8"]
    for l in d["levels"]:
        assert l["iterations"] == 6 and l["us_per_iteration"] > 0 and l["final_cost"] < l["initial_cost"]
        assert l["observations"] == 8 * l["points"] and l["points"] <= 6000
        assert set(l["kernels_us_per_launch"]) == {"k_sample<JAC> (Jacobian pass)", "k_schur (point elimination)", "k_reduce_solve (partials + reduced solve)"}
        assert all(v > 0 for v in l["kernels_us_per_launch"].values()), l["kernels_us_per_launch"]
        assert 0 < l["roofline"]["whole_iteration_frac"] < 1 and l["roofline"]["algorithmic_bytes_per_obs"]["b_jac"] > 700
    assert abs(d["ms_per_step"] - 1e-3 * sum(l["us_per_iteration"] for l in d["levels"])) < 1e-9
    assert d["pyramid_build"]["ms_per_window"] > 0 and {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
