"""Dev tool (GPU box): per-phase cycle stamps of the reduced solve.  PBA_LIB=photobundle_amd/libpba_hip_timing.so (make TIMING=1)
PBA_ASYNC=0 PBA_SCHUR_TIMING=3 python tools/solve_phase_timing.py [frames] [points]"""
import os, sys
sys.path.insert(0, os.getcwd())
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
points = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
p = synthetic.make_window(n_frames=frames, n_points=points, dense_births=(0, 8) if frames > 8 else (0,))
_, _, rows, cols = p.planes.shape
with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber, keep_reduced_system=False) as e:
    e.load(p)
    e.solve(default_solver_options(max_num_iterations=6, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
