"""Helpers shared by the -m gpu parity tests: dense reference algebra built from the oracle."""
import numpy as np

from oracle import oracle
from photobundle_amd.engine import Engine


def make_engine(prob, device=0, keep_reduced_system=True):
    _, _, rows, cols = prob.planes.shape
    e = Engine(rows, cols, prob.K, prob.radius, prob.n_frames, huber=prob.huber, device=device,
               keep_reduced_system=keep_reduced_system, channels=getattr(prob, "channels", 1))
    e.load(prob)
    return e


def dense_system(p, cams=None, xyz=None):
    """Dense loss-corrected Jacobian / residual from per-block oracle evaluations (free columns only)."""
    P = p.patch_len * getattr(p, "channels", 1)       # residuals of one block
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


def projection_jacobians_numpy(p, cams=None, xyz=None):
    """Ac [n_obs, 2, 6], Ap [n_obs, 2, 3]: d(u, v)/d(camera [w, t]) and d(u, v)/d(point) for every observation, from
    scipy rotations and the right Jacobian of SO(3): d(R(w) X)/dw = -R [X]x Jr(w),
    Jr = I - (1 - cos t)/t^2 [w]x + (t - sin t)/t^3 [w]x^2 -- a formula neither the device code (Gallego-Yezzi form) nor
    the oracle (dual numbers) uses."""
    from scipy.spatial.transform import Rotation
    cams = p.cams if cams is None else cams
    xyz = p.xyz if xyz is None else xyz
    fx, fy, _, _ = p.K
    w = cams[p.obs_slot, :3]
    t = cams[p.obs_slot, 3:]
    X = xyz[p.obs_point]
    R = Rotation.from_rotvec(w).as_matrix()                       # [n, 3, 3]
    xw = np.einsum("nij,nj->ni", R, X) + t

    def skew(v):
        z = np.zeros(len(v))
        return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1),
                         np.stack([-v[:, 1], v[:, 0], z], 1)], 1)
    th = np.linalg.norm(w, axis=1)
    th2 = th * th
    small = th < 1e-6
    ths = np.where(small, 1.0, th)
    a = np.where(small, 0.5 - th2 / 24.0, (1.0 - np.cos(ths)) / (ths * ths))
    b = np.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - np.sin(ths)) / (ths ** 3))
    Wx = skew(w)
    Jr = np.eye(3)[None] - a[:, None, None] * Wx + b[:, None, None] * (Wx @ Wx)
    dXw_dw = -R @ skew(X) @ Jr                                    # [n, 3, 3]
    z = xw[:, 2]
    dpi = np.zeros((len(z), 2, 3))
    dpi[:, 0, 0] = fx / z
    dpi[:, 0, 2] = -fx * xw[:, 0] / (z * z)
    dpi[:, 1, 1] = fy / z
    dpi[:, 1, 2] = -fy * xw[:, 1] / (z * z)
    Ac = np.concatenate([dpi @ dXw_dw, dpi], axis=2)              # d/dt = identity
    Ap = dpi @ R
    return Ac, Ap


def check_obs_records(p, rec, threads=4, rtol=1e-12, cams=None, xyz=None):
    """Every per-observation record of the engine's Jacobian pass (rho' M, rho' b, rho/2) against the oracle's
    dual-number rows: J = -w [gx gy] A  =>  Jc^T Jc = Ac^T M Ac, Jc^T Jp = Ac^T M Ap, Jp^T Jp = Ap^T M Ap,
    Jc^T r = -Ac^T b, Jp^T r = -Ap^T b.  Ap has rank 2, so the 3x3 block alone already determines M (and the
    3-vector b).  Tolerance: relative to the largest entry of each block."""
    bp = oracle.block_products(p, autodiff=True, threads=threads, cams=cams, xyz=xyz)
    Ac, Ap = projection_jacobians_numpy(p, cams, xyz)
    M = np.zeros((p.n_obs, 2, 2))
    M[:, 0, 0], M[:, 0, 1], M[:, 1, 0], M[:, 1, 1] = rec[:, 0], rec[:, 1], rec[:, 1], rec[:, 2]
    b = rec[:, 3:5]
    AcT, ApT = Ac.transpose(0, 2, 1), Ap.transpose(0, 2, 1)
    mine = dict(JcJc=AcT @ M @ Ac, JcJp=AcT @ M @ Ap, JpJp=ApT @ M @ Ap,
                Jcr=-np.einsum("nij,nj->ni", AcT, b), Jpr=-np.einsum("nij,nj->ni", ApT, b))
    worst = {}
    for k, v in mine.items():
        ref = bp[k]
        scale = np.abs(ref).reshape(p.n_obs, -1).max(1)
        err = np.abs(v - ref).reshape(p.n_obs, -1).max(1)
        ok = err <= rtol * scale + 1e-300
        assert ok.all(), (k, int((~ok).sum()), float((err / np.maximum(scale, 1e-300)).max()))
        worst[k] = float((err / np.maximum(scale, 1e-300)).max())
    return worst
