"""Dev tool (GPU box): step arbiter (tests/gpu_util.step_accuracy) of ONE case of the random sweep, per iteration -- python tools/replay_arbiter.py <case index> [iterations]
(PBA_LIB selects the engine build)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PBA_RANDOM_CASES", str(int(sys.argv[1]) + 1))
import test_gpu_random_shapes as t
from gpu_util import step_accuracy
k = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = t.CASES[k]
print("case", k, c)
p = t._make(c)
print("observations", p.n_obs, "points", p.n_points)
for r in step_accuracy(p, n):
    print("it %d cond %.3g  bwd engine %.3e f64 %.3e band %.3e (x%.1f)  fwd engine %.3e f64 %.3e band %.3e (x%.1f)  data_shift %.2e" %
          (r["it"], r["cond"], r["bwd_engine"], r["bwd_f64"], r["bwd_f64_band"], r["bwd_engine"] / max(r["bwd_f64_band"], 1e-300), r["fwd_engine"], r["fwd_f64"], r["fwd_f64_band"],
           r["fwd_engine"] / max(r["fwd_f64_band"], 1e-300), r["data_shift"]))
