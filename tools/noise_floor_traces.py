"""Dev tool: CPU traces (extended-precision referee + double twins) of configs[1] run to convergence, saved for
offline comparison with an engine trace (tests/test_gpu_fullsize.py does the same thing inside the suite)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle
from photobundle_amd import synthetic

kind = sys.argv[1] if len(sys.argv) > 1 else "poor"
kw = dict(n_frames=8, n_points=50000, radius=2)
if kind == "good":
    kw.update(rot_deg=0.02, trans=0.003, depth_noise=0.002)
p = synthetic.make_window(**kw)
out = {}
for name, o in (("q", dict(use_autodiff=0, extended_precision=1)), ("dual", dict(use_autodiff=1)), ("analytic", dict(use_autodiff=0))):
    t = time.time()
    r = oracle.solve(p, oracle.default_options(num_threads=8, **o))
    print(name, "%.1f s" % (time.time() - t), len(r["iterations"]), r["message"], flush=True)
    out[name] = dict(iterations=r["iterations"], final_cost=r["final_cost"], cams=r["cams"].tolist(), message=r["message"],
                     termination_type=r["termination_type"])
json.dump(out, open("gpurun_out/nf/traces_%s.json" % kind, "w"))
