import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, time
from photobundle_amd import synthetic
from photobundle_amd.engine import default_solver_options
from gpu_util import make_engine
p = synthetic.make_window(n_frames=8, n_points=50000, radius=2)
o = default_solver_options(max_num_iterations=100, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
with make_engine(p, keep_reduced_system=False) as e:
    e.solve(default_solver_options(max_num_iterations=3))
    e.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights); e.set_cameras(p.cams, p.fixed_slot)
    t=time.perf_counter(); res = e.solve(o, fetch_state=False); dt=time.perf_counter()-t
    print(len(res['iterations']), res['num_successful_steps'], res['num_unsuccessful_steps'], res['num_jacobian_passes'], res['num_cost_passes'], res['num_resolve_passes'], dt/100*1e6, res['message'])
    print([ (i['iteration'], i['step_is_successful']) for i in res['iterations'][-8:]])
