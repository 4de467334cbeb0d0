"""CPU baseline of configs[1] (the oracle in reference-faithful mode) at 4 threads and at all cores, same host as the GPU.

Runs ON THE GPU BOX: python tools/cpu_baseline.py [points] [iterations]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from photobundle_amd import synthetic

n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prob = synthetic.make_window(n_frames=8, n_points=n_pts, radius=2)
cores = os.cpu_count() or 1
try:
    model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    model = "?"
print("host: %d logical cores, %s" % (cores, model))
for threads in sorted({1, 4, cores}):
    o = oracle.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0,
                               parameter_tolerance=0.0, num_threads=threads, use_autodiff=1)
    t = time.perf_counter()
    res = oracle.solve(prob, o)
    dt = time.perf_counter() - t
    it = len(res["iterations"]) - 1
    print("threads %3d: %d LM iterations in %.1f s -> %.3f iters/s (%d points, %d observations)" % (threads, it, dt, it / dt, n_pts, prob.n_obs))
