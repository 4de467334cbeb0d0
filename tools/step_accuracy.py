"""Dev tool (GPU box): the engine's LM step against the extended-precision / float64 block restatement of the same step
(tests/gpu_util.step_accuracy: backward error in the full normal equations, forward error of the camera step) on windows of the random
sweep (tests/test_gpu_random_shapes.py).   usage: python tools/step_accuracy.py case [case ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("PBA_RANDOM_CASES", "200")
import test_gpu_random_shapes as T
from gpu_util import step_accuracy

for ci in [int(a) for a in sys.argv[1:]]:
    for r in step_accuracy(T._make(T.CASES[ci]), 5):
        print("case %d iteration %d: cond(S) %.1e   backward error engine %.2e float64 %.2e   forward error engine %.2e float64 %.2e   (data shift %.2e)"
              % (ci, r["it"], r["cond"], r["bwd_engine"], r["bwd_f64"], r["fwd_engine"], r["fwd_f64"], r["data_shift"]), flush=True)
