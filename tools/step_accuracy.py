"""Dev tool (GPU box): the engine's LM camera step against the extended-precision / float64 block restatement of the same step
(tests/gpu_util.step_accuracy) on windows of the random sweep (tests/test_gpu_random_shapes.py).  An engine in the float64 band is as
accurate as double arithmetic allows on that window.   usage (from the tree to be judged): python <path>/step_accuracy.py case [case ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("PBA_RANDOM_CASES", "200")
import test_gpu_random_shapes as T
from gpu_util import step_accuracy

for ci in [int(a) for a in sys.argv[1:]]:
    for it, cond, err_e, err_d in step_accuracy(T._make(T.CASES[ci]), 5):
        print("case %d iteration %d: cond(S) %.1e   engine - exact %.2e   float64 - exact %.2e   (relative to the step)" % (ci, it, cond, err_e, err_d), flush=True)
