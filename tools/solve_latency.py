"""Fixed latency of one pba_solve at configs[1] (scratch tool): host time in the call, C-side total, trailing sync, unpacking."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options
p = synthetic.make_window()
_, _, rows, cols = p.planes.shape
eng = Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber, keep_reduced_system=False)
eng.load(p)
def reset():
    eng.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights); eng.set_cameras(p.cams, p.fixed_slot)
bufs = Engine.solve_buffers()
for K in (0, 1, 2, 5, 20, 20, 20):
    o = default_solver_options(max_num_iterations=K, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    reset(); torch.cuda.synchronize()
    t0 = time.perf_counter(); s, its = eng.solve_raw(o, buffers=bufs); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    res = Engine.unpack_solve(s, its); t3 = time.perf_counter()
    print("K %d: pba_solve %.1f us (C-side total %.1f us), sync %.1f us, unpack %.1f us" % (K, 1e6*(t1-t0), 1e6*s.total_time_in_seconds, 1e6*(t2-t1), 1e6*(t3-t2)))
