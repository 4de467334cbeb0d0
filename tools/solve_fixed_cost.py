"""Per-solve fixed cost of pba_solve: wall time of solves with K = 1, 10, 50, 100 iterations on configs[1].  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options

prob = synthetic.make_window(n_frames=8, n_points=50000, radius=2)
rows, cols = prob.images.shape[1:]
eng = Engine(rows, cols, prob.K, prob.radius, prob.n_frames, huber=prob.huber)
eng.load(prob)
def opts(k):
    return default_solver_options(max_num_iterations=k, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
eng.solve(opts(5))
pts = []
for k in (1, 10, 50, 100):
    best = 1e9
    for _ in range(3):
        eng.set_problem(prob.xyz, prob.desc, prob.obs_point, prob.obs_slot, prob.weights)
        eng.set_cameras(prob.cams, prob.fixed_slot)
        t = time.perf_counter()
        res = eng.solve(opts(k), fetch_state=False)
        best = min(best, time.perf_counter() - t)
    n = len(res["iterations"]) - 1
    pts.append((n, best))
    print("K=%3d: %d iterations, %.1f us total, %.1f us/iter" % (k, n, 1e6 * best, 1e6 * best / max(1, n)))
(n0, t0), (n1, t1) = pts[1], pts[3]
slope = (t1 - t0) / (n1 - n0)
print("marginal %.1f us/iter, fixed %.1f us per solve" % (1e6 * slope, 1e6 * (t0 - slope * n0)))
