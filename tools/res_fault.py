"""Dev aid: the fault-injection run of tests/test_gpu_resident.py by hand (PBA_RES_STOP=100 PBA_WAIT_TIMEOUT_S=4 python -u tools/res_fault.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, EngineError, default_solver_options
p = synthetic.make_window(n_frames=5, n_points=2000, radius=1, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6))
e = Engine(188, 621, p.K, p.radius, p.n_frames, huber=p.huber)
e.load(p)
print("loaded", flush=True)
for k in range(2):
    t0 = time.time()
    try:
        r = e.solve(default_solver_options(max_num_iterations=6))
        print("solve %d: no error, %s, %d iterations" % (k, e.solve_driver(), len(r["iterations"]) - 1), flush=True)
    except EngineError as exc:
        print("solve %d: error after %.2f s: %s" % (k, time.time() - t0, exc), flush=True)
print("closing", flush=True)
e.close()
print("closed", flush=True)
