"""Where the oracle's Levenberg-Marquardt loop ends, against a third-party optimiser (scipy.optimize.least_squares on
tiny problems -- the end point, not the path).

Everything the optimiser sees is restated here in vectorised numpy, independently of oracle/pba_oracle.cpp: scipy's rotation for
AngleAxisRotatePoint (photobundle.cc:700), the pinhole projection (calibration.h:34-38), the truncating / clamping bilinear rule
with float32 weights (sample_eigen.h:34-102) and the reference's Jacobian convention -- the derivative of a sampled intensity is
the SAMPLED gradient plane times d(u, v) (sample_eigen.h:108-126, jet_extras.h:95-108), not the derivative of the interpolant, so
both optimisers look for the same fixed point J^T r = 0.  Only camera 0 is constant (photobundle.cc:809-813): the global scale is a
gauge freedom, so the minimum is compared through its COST and through gauge-free quantities (the re-projections)."""
import numpy as np
import pytest
pytest.importorskip("scipy")      # a box without scipy skips these cross-checks instead of failing collection
from scipy.optimize import least_squares      # noqa: E402
from scipy.spatial.transform import Rotation      # noqa: E402

from oracle import oracle
from photobundle_amd import synthetic


def _axis(c, size):
    """LinearInitAxis (sample_eigen.h:34-52), vectorised: indices i0, i1 and the float32 weight d of texel i0."""
    c = c.astype(np.float32)
    i = np.trunc(c).astype(np.int64)
    lo, hi = i < 0, i > size - 2
    i0 = np.where(lo, 0, np.where(hi, size - 1, i))
    i1 = np.where(lo, 0, np.where(hi, size - 1, i + 1))
    d = np.where(lo | hi, np.float32(1.0), (i + 1).astype(np.float32) - c).astype(np.float32)
    return i0, i1, d


def _sample(plane, y, x):
    """SampleLinear (sample_eigen.h:56-102): dy (dx a11 + (1 - dx) a12) + (1 - dy) (dx a21 + (1 - dx) a22), stored as float."""
    rows, cols = plane.shape
    x0, x1, dx = _axis(x, cols)
    y0, y1, dy = _axis(y, rows)
    dx, dy = dx.astype(np.float64), dy.astype(np.float64)        # the blend runs in double on float taps
    top = dx * plane[y0, x0] + (1.0 - dx) * plane[y0, x1]
    bot = dx * plane[y1, x0] + (1.0 - dx) * plane[y1, x1]
    return (dy * top + (1.0 - dy) * bot).astype(np.float32).astype(np.float64)


class _Restatement:
    def __init__(self, p):
        self.p = p
        R = p.radius
        yy, xx = np.meshgrid(np.arange(-R, R + 1), np.arange(-R, R + 1), indexing="ij")
        self.oy, self.ox = yy.ravel().astype(np.float64), xx.ravel().astype(np.float64)      # row-major patch (photobundle.cc:714-716)
        self.free = [s for s in range(p.cams.shape[0]) if s != p.fixed_slot]
        self.col = {s: 6 * k for k, s in enumerate(self.free)}
        self.n_cam = 6 * len(self.free)

    def unpack(self, theta):
        cams = self.p.cams.copy()
        for s in self.free:
            cams[s] = theta[self.col[s]:self.col[s] + 6]
        return cams, theta[self.n_cam:].reshape(-1, 3)

    def pack(self, cams, xyz):
        return np.concatenate([np.concatenate([cams[s] for s in self.free]), xyz.ravel()])

    def _geometry(self, cams, xyz):
        p = self.p
        fx, fy, cx, cy = p.K
        X = xyz[p.obs_point]
        Rm = Rotation.from_rotvec(cams[:, :3]).as_matrix()[p.obs_slot]
        Xc = np.einsum("oij,oj->oi", Rm, X) + cams[p.obs_slot, 3:]
        u = fx * Xc[:, 0] / Xc[:, 2] + cx
        v = fy * Xc[:, 1] / Xc[:, 2] + cy
        return Rm, X, Xc, u, v

    def residuals(self, theta):
        p = self.p
        cams, xyz = self.unpack(theta)
        _, _, _, u, v = self._geometry(cams, xyz)
        r = np.empty((p.n_obs, self.ox.size))
        for s in range(cams.shape[0]):
            m = p.obs_slot == s
            I = _sample(p.planes[s, 0], (v[m, None] + self.oy[None, :]), (u[m, None] + self.ox[None, :]))
            r[m] = p.weights[None, :] * (p.desc[p.obs_point[m]] - I)
        return r.ravel()

    def jacobian(self, theta):
        p = self.p
        fx, fy, _, _ = p.K
        cams, xyz = self.unpack(theta)
        Rm, X, Xc, u, v = self._geometry(cams, xyz)
        P = self.ox.size
        J = np.zeros((p.n_obs * P, theta.size))
        # d(u, v) / d(Xc)
        iz = 1.0 / Xc[:, 2]
        dudXc = np.stack([fx * iz, np.zeros_like(iz), -fx * Xc[:, 0] * iz * iz], axis=1)
        dvdXc = np.stack([np.zeros_like(iz), fy * iz, -fy * Xc[:, 1] * iz * iz], axis=1)
        # d(Xc) / d(omega) by central differences of scipy's rotation (independent of the oracle's closed form)
        dXc_dw = np.zeros((p.n_obs, 3, 3))
        for k in range(3):
            h = 1e-6
            cp, cm = cams.copy(), cams.copy()
            cp[:, k] += h
            cm[:, k] -= h
            Rp = Rotation.from_rotvec(cp[:, :3]).as_matrix()[p.obs_slot]
            Rn = Rotation.from_rotvec(cm[:, :3]).as_matrix()[p.obs_slot]
            dXc_dw[:, :, k] = np.einsum("oij,oj->oi", Rp - Rn, X) / (2 * h)
        A_u = np.concatenate([np.einsum("oi,oik->ok", dudXc, dXc_dw), dudXc, np.einsum("oi,oik->ok", dudXc, Rm)], axis=1)   # [obs, 9]
        A_v = np.concatenate([np.einsum("oi,oik->ok", dvdXc, dXc_dw), dvdXc, np.einsum("oi,oik->ok", dvdXc, Rm)], axis=1)
        for s in range(cams.shape[0]):
            m = np.nonzero(p.obs_slot == s)[0]
            yy, xx = v[m, None] + self.oy[None, :], u[m, None] + self.ox[None, :]
            gx, gy = _sample(p.planes[s, 1], yy, xx), _sample(p.planes[s, 2], yy, xx)
            for j, o in enumerate(m):
                rows = slice(o * P, (o + 1) * P)
                Jo = -p.weights[:, None] * (gx[j][:, None] * A_u[o][None, :] + gy[j][:, None] * A_v[o][None, :])
                if s in self.col:
                    J[rows, self.col[s]:self.col[s] + 6] = Jo[:, :6]
                pc = self.n_cam + 3 * p.obs_point[o]
                J[rows, pc:pc + 3] = Jo[:, 6:]
        return J


def _window(seed):
    return synthetic.make_window(n_frames=3, n_points=60, radius=1, size=(96, 128), K=(160.0, 160.0, 64.0, 48.0),
                                 rot_deg=0.05, trans=0.01, depth_noise=0.005, seed_offset=seed)


def test_restated_residuals_are_the_oracles():
    p = _window(3)
    rs = _Restatement(p)
    theta = rs.pack(p.cams, p.xyz)
    r = rs.residuals(theta).reshape(p.n_obs, -1)
    J = rs.jacobian(theta)
    for obs in range(0, p.n_obs, 11):
        ro, jc, jp = oracle.eval_block(p, obs, autodiff=True)
        # scipy's rotation and Ceres' Rodrigues form differ in the last bits of (u, v); once rounded to float (sample_eigen.h:117-118)
        # that is at most a float ulp of the coordinate, i.e. ~1e-5 of a grey level per unit gradient
        assert np.abs(r[obs] - ro).max() <= 1e-3 and np.mean(r[obs] == ro) >= 0.3
        P = r.shape[1]
        Jp = J[obs * P:(obs + 1) * P, rs.n_cam + 3 * p.obs_point[obs]:rs.n_cam + 3 * p.obs_point[obs] + 3]
        assert np.allclose(Jp, jp, rtol=1e-9, atol=1e-9 * np.abs(jp).max())
        s = p.obs_slot[obs]
        if s in rs.col:
            Jc = J[obs * P:(obs + 1) * P, rs.col[s]:rs.col[s] + 6]
            assert np.allclose(Jc, jc, rtol=1e-6, atol=1e-6 * np.abs(jc).max())
    assert 0.5 * float(r.ravel() @ r.ravel()) == pytest.approx(oracle.cost(p)[0], rel=1e-6)


@pytest.mark.parametrize("seed", [3, 8, 5])
def test_end_point_matches_scipy_least_squares(seed):
    """Same start, same residuals, same Jacobian convention, a third-party trust-region loop.  The cost surface is piecewise (texel
    cells, float-rounded coordinates) and the Jacobian is not its derivative, so both loops stop where their trust region has shrunk
    below a float ulp of (u, v) -- not at a stationary point: restarted from each other's end point either gains a few per cent
    (measured: 0-7 %).  What CAN be asserted: both descend into the same basin -- costs within 5 % of each other, the median point
    lands within 0.1 px of the same place in every frame."""
    p = _window(seed)
    rs = _Restatement(p)
    theta0 = rs.pack(p.cams, p.xyz)
    res = oracle.solve(p, oracle.default_options(max_num_iterations=400, function_tolerance=1e-14, gradient_tolerance=1e-14,
                                                 parameter_tolerance=1e-14))
    sp = least_squares(rs.residuals, theta0, jac=rs.jacobian, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15,
                       max_nfev=2000)
    assert res["final_cost"] < 0.85 * res["initial_cost"] and sp.cost < 0.85 * res["initial_cost"]
    assert abs(sp.cost - res["final_cost"]) <= 0.05 * res["final_cost"], (sp.cost, res["final_cost"], sp.status, res["message"])
    # gauge-free comparison (only camera 0 is constant: the global scale is free): where every point lands in every frame
    cs, xs = rs.unpack(sp.x)
    _, _, _, u1, v1 = rs._geometry(cs, xs)
    _, _, _, u2, v2 = rs._geometry(res["cams"], res["xyz"])
    assert np.median(np.hypot(u1 - u2, v1 - v2)) < 0.1      # pixels
    # the restated cost at the oracle's end point is the oracle's own final cost
    r = rs.residuals(rs.pack(res["cams"], res["xyz"]))
    assert 0.5 * float(r @ r) == pytest.approx(res["final_cost"], rel=1e-6)
