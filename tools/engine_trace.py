"""Dev tool (GPU box): engine trace of configs[1] run to convergence -> gpurun_out/nf/engine_<kind>.json, plus truncated
solves (poses after k iterations) for the pose-bar table."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options

kind = sys.argv[1] if len(sys.argv) > 1 else "poor"
kw = dict(n_frames=8, n_points=50000, radius=2)
if kind == "good":
    kw.update(rot_deg=0.02, trans=0.003, depth_noise=0.002)
p = synthetic.make_window(**kw)
rows, cols = p.images.shape[1:]
os.makedirs("gpurun_out/nf", exist_ok=True)
out = {}
with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber, device=0) as e:
    e.load(p)
    r = e.solve(default_solver_options())
    out["full"] = dict(iterations=r["iterations"], final_cost=r["final_cost"], cams=r["cams"].tolist(), message=r["message"],
                       termination_type=r["termination_type"])
    for k in (2, 4, 6, 8, 10, 15, 20, 30, 40, 60):
        e.load(p)
        r = e.solve(default_solver_options(max_num_iterations=k))
        out["k%d" % k] = dict(cams=r["cams"].tolist(), final_cost=r["final_cost"])
json.dump(out, open("gpurun_out/nf/engine_%s.json" % kind, "w"))
print("engine", kind, len(out["full"]["iterations"]), out["full"]["message"])
