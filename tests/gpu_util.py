"""Helpers shared by the -m gpu parity tests: dense reference algebra built from the oracle."""
import numpy as np

from oracle import oracle
from photobundle_amd.engine import Engine


def make_engine(prob, device=0, keep_reduced_system=True):
    _, _, rows, cols = prob.planes.shape
    e = Engine(rows, cols, prob.K, prob.radius, prob.n_frames, huber=prob.huber, device=device,
               keep_reduced_system=keep_reduced_system)
    e.load(prob)
    return e


def dense_system(p, cams=None, xyz=None):
    """Dense loss-corrected Jacobian / residual from per-block oracle evaluations (free columns only)."""
    P = p.patch_len
    n_c, n_p = p.n_frames, p.n_points
    free = [c for c in range(n_c) if c != p.fixed_slot]
    cols_c = {c: 6 * i for i, c in enumerate(free)}
    n_cam = 6 * len(free)
    J = np.zeros((p.n_obs * P, n_cam + 3 * n_p))
    r = np.zeros(p.n_obs * P)
    for o in range(p.n_obs):
        rb, jc, jp = oracle.eval_block(p, o, cams=cams, xyz=xyz)
        s = rb @ rb
        k = 1.0
        if p.huber > 0 and s > p.huber ** 2:
            k = np.sqrt(p.huber / np.sqrt(s))
        rows = slice(o * P, (o + 1) * P)
        r[rows] = k * rb
        c = p.obs_slot[o]
        if c in cols_c:
            J[rows, cols_c[c]:cols_c[c] + 6] = k * jc
        q = n_cam + 3 * p.obs_point[o]
        J[rows, q:q + 3] = k * jp
    return J, r, n_cam


def reference_step(J, r, n_cam, radius, scale=None):
    """One Ceres LM step with dense algebra.  Returns dict(scale, S, rhs, delta, model_cost_change)."""
    if scale is None:
        scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    Js = J * scale
    diag = np.clip((Js * Js).sum(0), 1e-6, 1e32)
    H = Js.T @ Js + np.diag(diag / radius)
    g = Js.T @ r
    Hcc, Hcp, Hpp = H[:n_cam, :n_cam], H[:n_cam, n_cam:], H[n_cam:, n_cam:]
    Hpp_inv = np.linalg.inv(Hpp)
    S = Hcc - Hcp @ Hpp_inv @ Hcp.T
    rhs = g[:n_cam] - Hcp @ Hpp_inv @ g[n_cam:]
    y = np.linalg.solve(H, g)
    step = -y
    model = Js @ step
    return dict(scale=scale, S=S, rhs=rhs, delta=step * scale, model_cost_change=-model @ (r + model / 2),
                gradient=J.T @ r)
