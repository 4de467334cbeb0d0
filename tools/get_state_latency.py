import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options
from gpu_util import make_engine
p = synthetic.make_window(n_points=8000, n_frames=5)
with make_engine(p, keep_reduced_system=False) as e:
    t0 = time.perf_counter(); res = e.solve(default_solver_options(max_num_iterations=5), fetch_state=False); t1 = time.perf_counter()
    for k in range(3):
        ta = time.perf_counter(); c, x = e.get_state(); tb = time.perf_counter()
        print("solve %.2f ms" % (1e3 * (t1 - t0)) if k == 0 else "", "get_state call %d: %.3f ms" % (k, 1e3 * (tb - ta)))
