import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from oracle import oracle
from photobundle_amd import synthetic
from photobundle_amd.engine import default_solver_options
from gpu_util import make_engine
p = synthetic.make_window(n_frames=4, n_points=300, radius=2, size=(120,160), K=(200.,200.,80.,60.))
ref = oracle.solve(p, oracle.default_options(max_num_iterations=25))
with make_engine(p) as e:
    res = e.solve(default_solver_options(max_num_iterations=25))
for a,b in zip(ref['iterations'], res['iterations']):
    print(a['iteration'], a['step_is_successful'], b['step_is_successful'], '%.6e %.6e'%(a['cost'],b['cost']), '%.4e %.4e'%(a['gradient_max_norm'], b['gradient_max_norm']))
print(ref['message'], '|', res['message'])
