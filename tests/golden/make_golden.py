#!/usr/bin/env python3
"""Generates tests/golden/window_small.npz: inputs of one small sliding window + the outputs the CPU oracle produces
for them (the reference itself cannot be built or run in this image -- see oracle/pba_oracle.h -- so these vectors
pin the ORACLE's behaviour: any later change to the restatement or its build flags must reproduce them bit for bit,
and the HIP engine is compared against the same numbers on the GPU box).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle            # noqa: E402
from photobundle_amd import synthetic  # noqa: E402


def main():
    p = synthetic.make_window(n_frames=3, n_points=64, radius=2, size=(96, 128), K=(150.0, 150.0, 64.0, 48.0),
                              huber=0.05, seed_offset=21)
    lin = oracle.linearize(p)
    res = oracle.solve(p, oracle.default_options(max_num_iterations=12))
    its = res["iterations"]
    out = dict(
        images=p.images, K=np.array(p.K), radius=np.array(p.radius), cams=p.cams, xyz=p.xyz, desc=p.desc,
        obs_point=p.obs_point, obs_slot=p.obs_slot, weights=p.weights, huber=np.array(p.huber),
        fixed_slot=np.array(p.fixed_slot),
        # expected (oracle) outputs
        exp_cost=np.array(lin["cost"]), exp_block_sqnorm=lin["block_sqnorm"], exp_grad_cams=lin["grad_cams"],
        exp_grad_pts=lin["grad_pts"], exp_U=lin["U"], exp_V=lin["V"],
        exp_it_cost=np.array([i["cost"] for i in its]), exp_it_ok=np.array([i["step_is_successful"] for i in its]),
        exp_it_radius=np.array([i["trust_region_radius"] for i in its]),
        exp_it_gmax=np.array([i["gradient_max_norm"] for i in its]),
        exp_final_cams=res["cams"], exp_final_xyz=res["xyz"], exp_final_cost=np.array(res["final_cost"]),
    )
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "window_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
