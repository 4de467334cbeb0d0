"""BASELINE configs[4]: 11x11 patch + Huber loss, tolerance sweep of the sampler / accumulation precision.

The product default is the reference-exact sampler ("exact": float products, double blends, float results, fp64
accumulation - bit-compatible with sample_eigen.h:82-101).  The opt-in modes trade that for speed:
  fp32 : fp32 interpolation, fp32 accumulation of the per-patch structure tensor
  bf16 : as fp32, with the residual and the gradients rounded to bf16 before they are accumulated
The sweep solves the same window to convergence in every mode and compares refined poses with the CPU oracle; the bar
of the north star is a pose RMSE of 1e-5."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options

pytestmark = pytest.mark.gpu


def _solve(p, precision, iters):
    _, _, rows, cols = p.planes.shape
    with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber, precision=precision) as e:
        e.load(p)
        return e.solve(default_solver_options(max_num_iterations=iters))


@pytest.mark.parametrize("radius,huber", [(5, 0.05), (2, 0.0)])
def test_precision_sweep(radius, huber, capsys):
    p = synthetic.make_window(n_frames=8, n_points=1500, radius=radius, huber=huber, rot_deg=0.03, trans=0.005,
                              depth_noise=0.002)
    iters = 40
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=iters))
    out = {}
    for prec in ("exact", "fp32", "bf16"):
        res = _solve(p, prec, iters)
        rmse = float(np.sqrt(np.mean((res["cams"][1:] - ref["cams"][1:]) ** 2)))
        rel_cost = abs(res["final_cost"] - ref["final_cost"]) / ref["final_cost"]
        out[prec] = (rmse, rel_cost, len(res["iterations"]) - 1)
    with capsys.disabled():
        print("\nprecision sweep %dx%d patch, huber %g: pose RMSE vs oracle / relative final-cost difference / iterations" % (2 * radius + 1, 2 * radius + 1, huber))
        for k, v in out.items():
            print("  %-5s  %.3e  %.3e  %d" % ((k,) + v))
    # only the reference-exact sampler is held to the parity bar; the other two modes are measured (DESIGN.md records
    # the numbers: fp32 meets 1e-5 on the 11x11 / Huber shape but not on the 5x5 one, bf16 operands meet it nowhere)
    assert out["exact"][0] <= 1e-5 and out["exact"][1] <= 1e-9
    assert out["fp32"][0] <= 5e-3 and out["bf16"][0] <= 5e-2, out        # sanity: still the same basin
    assert out["bf16"][0] >= out["fp32"][0] >= out["exact"][0]
