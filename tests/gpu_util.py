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


def pose_rmse(a, b):
    d = a[1:] - b[1:]
    return float(np.sqrt(np.mean(d[:, :3] ** 2))), float(np.sqrt(np.mean(d[:, 3:] ** 2)))


def camera_centres(cams):
    """World positions C = -R(w)^T t of world->camera parameters [w, t] (reference photobundle.cc:774-778: the optimised
    blocks are the INVERSES of the world poses)."""
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(cams[:, :3]).as_matrix()
    return -np.einsum("nji,nj->ni", R, cams[:, 3:])


def pose_rmse_gauge_fixed(a, b):
    """Pose distance with the window's FREE GAUGE taken out.  Only camera 0 is constant (photobundle.cc:809-813) and the points
    are free, so rotation and translation of the world are pinned by camera 0 but its SCALE about camera 0's centre is not
    observable: (C_i - C_0, X_p - C_0) -> s (C_i - C_0, X_p - C_0) leaves every residual unchanged.  Two solves that end in
    the same optimum up to that gauge differ by a scale of the camera centres only.  Returns (rotation RMSE [rad] -- rotations
    are gauge-invariant --, RMSE of the camera centres [m] after the least-squares scale s* of `a` onto `b`, s* - 1), over
    the free cameras."""
    Ca, Cb = camera_centres(a), camera_centres(b)
    da, db = Ca[1:] - Ca[0], Cb[1:] - Cb[0]
    s = float((da * db).sum() / (da * da).sum())
    rot = float(np.sqrt(np.mean((a[1:, :3] - b[1:, :3]) ** 2)))
    return rot, float(np.sqrt(np.mean((s * da - db) ** 2))), s - 1.0


def referee_parity(q, twins, res, tag):
    """Long solves of this problem are CHAOTIC in the rounding: the objective is piecewise bilinear in u8 images and the
    window has a free scale gauge (only camera 0 is constant, photobundle.cc:809-813), so double-precision evaluations of
    the SAME algorithm that differ only in rounding drift apart after a handful of iterations (the relative cost
    distance grows ~100x per iteration once it starts, then saturates near 1e-3; different twins stop after 63 .. 500
    iterations).  The arbiter is the oracle's REFEREE mode `q` (oracle/pba_oracle.h: extended_precision -- every sum and
    the whole linear algebra in x87 extended precision, same double / fp32 residual blocks): `twins` are plain double
    runs of the oracle (dual numbers, analytic Jacobian), `res` is the engine.  Per iteration i:
      * engine-to-referee distance <= 2 x the largest twin-to-referee distance seen up to iteration i + 1 (an amplifier
        of ~100x per iteration makes "one iteration later" the natural granularity), with a floor of 1e-9 (the trace
        tolerance of every other parity test) for the first 10 iterations (their trace length) and 1e-7 afterwards (r4;
        it was 1e-5 -- a floor that would have hidden a 1e-6 defect in a late iteration).
        The later floor is for a DISCRETE event the twins almost never see: sample positions are rounded to float
        (sample_eigen.h:117-118), so a parameter vector that differs by 1e-12 from the referee's now and then rounds ONE
        of ~10^5 - 10^7 positions to the neighbouring float; the cost then differs by ~1e-9 and the gap grows from there.
        The twins share the referee's formulas and stay within 1e-15 of its parameters; the engine (analytic M / b sums,
        FMA, reciprocal refinements, L D L^T) sits ~1e-12 away, which makes the event ~100x likelier per iteration.
        Measured on the second window of configs[0]: 15+ digits for 16 iterations, 9 at the 17th, 5 - 6 from the 28th on.
        That every cost along the engine's OWN trajectory is the reference's value at that point is checked separately
        (trajectory_consistency: the oracle re-evaluates the engine's states),
      * identical accept / reject decisions for as long as every twin's decisions equal the referee's and the costs have
        not separated (distances <= 1e-6: once the traces are 1e-4 apart a different decision is not a defect).
    At the end: the engine converged (termination_type 0), its final cost is within 2 x the twins' distance of the
    referee's or below it, its poses within 2 x the twins' pose distance (+ the north_star 1e-5) of the referee's.
    Returns the number of leading iterations in which the engine is within 1e-9 of the referee."""
    qi, gi = q["iterations"], res["iterations"]
    n = min([len(qi), len(gi)] + [len(t["iterations"]) for t in twins])
    d_tw = [max(abs(t["iterations"][i]["cost"] - qi[i]["cost"]) / qi[i]["cost"] for t in twins) for i in range(n)]
    d_en = [abs(gi[i]["cost"] - qi[i]["cost"]) / qi[i]["cost"] for i in range(n)]
    same, tight, worst = True, 0, 0.0

    def digits(d):     # one character per iteration: agreement with the referee in decimal digits (f = 15+, 0 = none)
        return "".join("f" if x <= 1e-15 else "%x" % max(0, min(15, int(-np.log10(x)))) for x in d)

    for i in range(n):
        # (late iterations: one more iteration of look-ahead -- measured on the well-initialised configs[1] window, r4: engine 6.9e-7 at
        # iteration 44 where the twins reach 3.0e-7 at 45 and 2e-6 at 46; both wander inside the same amplification band)
        run = max(d_tw[:min(n, i + (2 if i < 10 else 3))])
        same = same and all(t["iterations"][i]["step_is_successful"] == qi[i]["step_is_successful"] for t in twins)
        same = same and max(run, d_en[i]) <= 1e-6
        if same:
            assert gi[i]["step_is_successful"] == qi[i]["step_is_successful"] and gi[i]["step_is_valid"] == qi[i]["step_is_valid"], (tag, i)
        floor = 1e-9 if i < 10 else 1e-7
        assert d_en[i] <= max(floor, 2.0 * run), (tag, i, d_en[i], d_tw[i], run, digits(d_en), digits(d_tw))
        worst = max(worst, d_en[i] / max(floor, run))
        if tight == i and d_en[i] <= 1e-9:
            tight = i + 1
    tw_tight = 0
    while tw_tight < n and d_tw[tw_tight] <= 1e-9:
        tw_tight += 1
    fq = q["final_cost"]
    fc_tw = max(abs(t["final_cost"] - fq) / fq for t in twins)
    fc_en = abs(res["final_cost"] - fq) / fq
    pose_tw = [max(pose_rmse(t["cams"], q["cams"])[k] for t in twins) for k in (0, 1)]
    pose_en = pose_rmse(res["cams"], q["cams"])
    # the same distances in a gauge-fixed metric (scale of the camera centres about camera 0 removed)
    gf_en = pose_rmse_gauge_fixed(res["cams"], q["cams"])
    gf_tw = [pose_rmse_gauge_fixed(t["cams"], q["cams"]) for t in twins]
    print("%s: GAUGE-FIXED pose distance to the referee at convergence (scale about camera 0 removed): engine rot %.2e rad, centres %.2e m "
          "(scale drift %.2e); twins rot %s, centres %s m (scale drift %s)"
          % (tag, gf_en[0], gf_en[1], gf_en[2], ["%.2e" % g[0] for g in gf_tw], ["%.2e" % g[1] for g in gf_tw], ["%.1e" % g[2] for g in gf_tw]))
    print("%s: digits of agreement with the referee per iteration: engine %s | twins %s" % (tag, digits(d_en), digits(d_tw)))
    print("%s: iterations referee %d / twins %s / engine %d; within 1e-9 of the referee: engine %d iterations, twins %d; largest "
          "engine / twin distance ratio %.2f; final cost engine %.8e referee %.8e twins %s (relative distance engine %.2e, twins %.2e); "
          "pose RMSE to the referee: engine rot %.2e rad trans %.2e m, twins %.2e / %.2e"
          % (tag, len(qi) - 1, [len(t["iterations"]) - 1 for t in twins], len(gi) - 1, tight, tw_tight, worst, res["final_cost"], fq,
             ["%.8e" % t["final_cost"] for t in twins], fc_en, fc_tw, pose_en[0], pose_en[1], pose_tw[0], pose_tw[1]))
    assert res["termination_type"] == 0, res["message"]
    assert tight >= min(tw_tight, n, 10) - 1, (tag, tight, tw_tight)  # over the first 10 iterations the engine stays tight about as long as the double oracle does
    assert fc_en <= max(1e-5, 2.0 * fc_tw) or res["final_cost"] <= min([fq] + [t["final_cost"] for t in twins])
    assert pose_en[0] <= 2.0 * pose_tw[0] + 1e-5 and pose_en[1] <= 2.0 * pose_tw[1] + 1e-5
    # gauge-fixed: no farther from the referee than twice the worst double-precision run of the oracle itself (+ the bar)
    assert gf_en[1] <= 2.0 * max(g[1] for g in gf_tw) + 1e-5, (tag, gf_en, gf_tw)
    referee_parity.last_gauge_fixed = (gf_en, gf_tw)
    return tight


def trajectory_consistency(p, engine, ks, options_fn, rtol=1e-12):
    """The engine's cost after k iterations against the ORACLE's cost evaluated at the engine's own state after those k
    iterations (cameras + points read back through pba_get_state): whatever path the trust-region loop took, every
    point on it carries the reference's objective value.  Returns the largest relative difference."""
    worst = 0.0
    for k in ks:
        engine.load(p)
        res = engine.solve(options_fn(k))
        c_ref, _ = oracle.cost(p, cams=res["cams"], xyz=res["xyz"], threads=8)
        d = abs(res["final_cost"] - c_ref) / c_ref
        assert d <= rtol, (k, res["final_cost"], c_ref, d)
        worst = max(worst, d)
    return worst


# ---- extended-precision arbiter of one LM step ---------------------------------------------------------------------
def _inv3(A):
    a, b, c, d, e, f, g, h, i = [A[:, r, s] for r in range(3) for s in range(3)]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    out = np.stack([e * i - f * h, c * h - b * i, b * f - c * e, f * g - d * i, a * i - c * g, c * d - a * f,
                    d * h - e * g, b * g - a * h, a * e - b * d], 1).reshape(-1, 3, 3)
    return out / det[:, None, None]


def _chol_solve(S, r):
    n = len(r)
    L = np.zeros_like(S)
    for j in range(n):
        L[j, j] = np.sqrt(S[j, j] - (L[j, :j] * L[j, :j]).sum())
        L[j + 1:, j] = (S[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    y = np.zeros_like(r)
    for j in range(n):
        y[j] = (r[j] - (L[j, :j] * y[:j]).sum()) / L[j, j]
    x = np.zeros_like(r)
    for j in range(n - 1, -1, -1):
        x[j] = (y[j] - (L[j + 1:, j] * x[j + 1:]).sum()) / L[j, j]
    return x


def block_system(p, bp, radius, scale, dt):
    """The Jacobi-scaled, damped normal equations of one Ceres LM step from per-block products `bp` in numpy dtype `dt`: scaling
    1 / (1 + sqrt(diag)) (fixed from the first call: pass the returned `scale` back in), damping clip(diag, 1e-6, 1e32) / radius.
    Returns (dict(Hcc [n_free, 6, 6], Hpp [n_points, 3, 3], E [n_obs, 6, 3], gc, gp, oc, op, m, sc, sp), scale)."""
    free = [c for c in range(p.n_frames) if c != p.fixed_slot]
    col = -np.ones(p.n_frames, int)
    col[free] = np.arange(len(free))
    nf, npt = len(free), p.n_points
    oc, op = col[p.obs_slot], p.obs_point
    m = oc >= 0
    Hcc, gc = np.zeros((nf, 6, 6), dt), np.zeros((nf, 6), dt)
    Hpp, gp = np.zeros((npt, 3, 3), dt), np.zeros((npt, 3), dt)
    np.add.at(Hcc, oc[m], bp["JcJc"][m].astype(dt))
    np.add.at(gc, oc[m], bp["Jcr"][m].astype(dt))
    np.add.at(Hpp, op, bp["JpJp"].astype(dt))
    np.add.at(gp, op, bp["Jpr"].astype(dt))
    if scale is None:
        scale = (1 / (1 + np.sqrt(np.einsum("nii->ni", Hcc))), 1 / (1 + np.sqrt(np.einsum("nii->ni", Hpp))))
    sc, sp = scale
    Hcc, gc = Hcc * sc[:, :, None] * sc[:, None, :], gc * sc
    Hpp, gp = Hpp * sp[:, :, None] * sp[:, None, :], gp * sp
    for H in (Hcc, Hpp):
        k = H.shape[1]
        H[:, np.arange(k), np.arange(k)] += np.clip(np.einsum("nii->ni", H), dt(1e-6), dt(1e32)) / dt(radius)
    E = bp["JcJp"].astype(dt) * sc[np.maximum(oc, 0)][:, :, None] * sp[op][:, None, :]
    E[~m] = 0
    return dict(Hcc=Hcc, Hpp=Hpp, E=E, gc=gc, gp=gp, oc=oc, op=op, m=m, sc=sc, sp=sp), scale


def block_step(p, bp, radius, scale, dt):
    """One Ceres LM step from per-block products `bp` (oracle.block_products or record_blocks) in numpy dtype `dt` (np.longdouble = x87
    extended precision on the x86 hosts used here; numpy.linalg has no longdouble, hence the hand-written 3x3 inverses and Cholesky):
    block_system, point elimination, dense solve, back-substitution.  Returns (camera step [n_free, 6], scale, S, point step [n_points, 3])."""
    y, scale = block_system(p, bp, radius, scale, dt)
    nf, npt = len(y["Hcc"]), p.n_points
    oc, op, m, E = y["oc"], y["op"], y["m"], y["E"]
    Ci = _inv3(y["Hpp"])
    n = 6 * nf
    S, rhs = np.zeros((n, n), dt), y["gc"].reshape(-1).copy()
    for a in range(nf):
        S[6 * a:6 * a + 6, 6 * a:6 * a + 6] = y["Hcc"][a]
    begin = np.searchsorted(op, np.arange(npt + 1))
    for pt in range(npt):
        o = np.arange(begin[pt], begin[pt + 1])
        o = o[m[o]]
        if not len(o):
            continue
        W = E[o] @ Ci[pt]                                      # [k, 6, 3]
        idx = (6 * oc[o][:, None] + np.arange(6)[None]).reshape(-1)
        S[np.ix_(idx, idx)] -= np.einsum("aij,bkj->aibk", W, E[o]).reshape(len(idx), len(idx))
        rhs[idx] -= (W @ y["gp"][pt]).reshape(-1)
    yc = _chol_solve(S, rhs).reshape(nf, 6)
    t = y["gp"].copy()                                         # y_p = C^-1 (g_p - E^T y_c)
    np.subtract.at(t, op[m], np.einsum("nij,ni->nj", E[m], yc[oc[m]]))
    yp = np.einsum("nij,nj->ni", Ci, t)
    return -(yc * y["sc"]), scale, S, -(yp * y["sp"])


def backward_error(p, bp, radius, scale, d_c, d_p):
    """Backward error of a step (d_c [n_free, 6], d_p [n_points, 3], unscaled) as a solution of the FULL scaled, damped normal equations
    H y = g built from `bp`, residual in x87 extended precision, normwise within the camera rows and within the point rows:
    max(max_i |H y - g|_i / max_i (|H| |y| + |g|)_i) over the two groups (a purely componentwise measure is not bounded for an
    elimination-based solver: it reaches 1e-9 for the float64 restatement on some windows; one norm over all rows lets the point rows
    swamp the camera rows).  A stable solver sits at 1e-16 .. 1e-13 whatever the conditioning; a wrong damping, scale or block shows
    at its own size."""
    dt = np.longdouble
    y, _ = block_system(p, bp, radius, scale, dt)
    oc, op, m, E = y["oc"], y["op"], y["m"], y["E"]
    yc, yp = -(d_c.astype(dt) / y["sc"]), -(d_p.astype(dt) / y["sp"])
    rc = np.einsum("nij,nj->ni", y["Hcc"], yc) - y["gc"]
    np.add.at(rc, oc[m], np.einsum("nij,nj->ni", E[m], yp[op[m]]))
    rp = np.einsum("nij,nj->ni", y["Hpp"], yp) - y["gp"]
    np.add.at(rp, op[m], np.einsum("nij,ni->nj", E[m], yc[oc[m]]))
    ac = np.einsum("nij,nj->ni", np.abs(y["Hcc"]), np.abs(yc)) + np.abs(y["gc"])
    np.add.at(ac, oc[m], np.einsum("nij,nj->ni", np.abs(E[m]), np.abs(yp[op[m]])))
    ap = np.einsum("nij,nj->ni", np.abs(y["Hpp"]), np.abs(yp)) + np.abs(y["gp"])
    np.add.at(ap, op[m], np.einsum("nij,ni->nj", np.abs(E[m]), np.abs(yc[oc[m]])))
    return float(max(np.abs(rc).max() / ac.max(), np.abs(rp).max() / ap.max()))


def record_blocks(p, rec, cams, xyz, dt):
    """The per-block products the ENGINE's Jacobian-pass records stand for (rho' M, rho' b per observation; check_obs_records pins them to
    the oracle's rows at 1e-12), formed in dtype `dt` with the independent numpy projection Jacobians."""
    Ac, Ap = [a.astype(dt) for a in projection_jacobians_numpy(p, cams, xyz)]
    M = np.zeros((p.n_obs, 2, 2), dt)
    M[:, 0, 0], M[:, 0, 1], M[:, 1, 0], M[:, 1, 1] = rec[:, 0], rec[:, 1], rec[:, 1], rec[:, 2]
    b = rec[:, 3:5].astype(dt)
    AcT, ApT = Ac.transpose(0, 2, 1), Ap.transpose(0, 2, 1)
    return dict(JcJc=AcT @ M @ Ac, JcJp=AcT @ M @ Ap, JpJp=ApT @ M @ Ap, Jcr=-np.einsum("nij,nj->ni", AcT, b), Jpr=-np.einsum("nij,nj->ni", ApT, b))


def step_accuracy(p, iterations):
    """Accuracy of the engine's LM step, iteration by iteration along its own path (radius 1e4 x 3^i, every step accepted = the path
    of the first iterations of a solve).  Everything downstream of the Jacobian-pass records (point elimination, reduced system,
    scaling, damping, L D L^T, both substitutions) is judged on the ENGINE's OWN records at that state, formed in x87 extended precision:
      * backward error of the engine's (camera, point) step in the full normal equations, next to that of a float64 restatement
        (block_step) of the same elimination -- the conditioning-free measure;
      * forward error of the camera step against the extended-precision step, again next to the float64 restatement's (on these
        windows the 3x3 point blocks at 0.01 m baselines amplify rounding to 1e-10 .. 1e-5 of the step for ANY double algorithm).
    Returns per iteration dict(it, cond, bwd_engine, bwd_f64, bwd_f64_band, fwd_engine, fwd_f64, fwd_f64_band, data_shift); *_band = the largest
    backward / forward error of five double-precision evaluations (records as they are + four one-ulp copies); data_shift = how far the 1e-15
    differences between the engine's and the oracle's evaluation of the blocks move the extended-precision step."""
    free = [c for c in range(p.n_frames) if c != p.fixed_slot]
    rows = []
    with make_engine(p) as e:
        e.linearize()
        scale_x = scale_d = scale_o = None
        for it in range(iterations):
            c0, x0 = [a.copy() for a in e.get_state()]
            rec = e.obs_records()
            bp = oracle.block_products(p, autodiff=True, threads=8, cams=c0, xyz=x0)
            radius = 1e4 * 3.0 ** it
            e.step(radius, init_scale=(it == 0))
            e.accept()
            e.linearize()
            c1, x1 = e.get_state()
            d_e, dp_e = (c1 - c0)[free], x1 - x0
            bx = record_blocks(p, rec, c0, x0, np.longdouble)
            d_x, scale_x1, S, _ = block_step(p, bx, radius, scale_x, np.longdouble)
            scale_d_prev = scale_d
            d_d, scale_d, _, dp_d = block_step(p, record_blocks(p, rec, c0, x0, np.float64), radius, scale_d, np.float64)
            d_o, scale_o, _, _ = block_step(p, bp, radius, scale_o, np.longdouble)
            nrm = np.abs(d_x).max()
            # The forward error of ONE double-precision evaluation is a draw from [0, conditioning x rounding]: on window 84 of the
            # sweep the same restatement sat at 2.8e-9 on the round-4 build's records and at 6.0e-11 on round 5's (the records differ in
            # the last bits), the engine at 4.6e-9 both times.  The band a double algorithm may use is therefore taken from an ENSEMBLE:
            # the restatement on the records as they are and on four copies moved by one ulp (seeded signs).
            # The same holds for the BACKWARD error from the second iteration on (r6, window 671 of the 1 000-case sweep, 13 frames x 11x11,
            # cond(S) 433, iteration 1: the restatement sits at 8.3e-14 on the round-6 build's records and at 1.6e-12 on round 5's -- last-bit
            # differences of the records -- the engine at 1.6e-12 and 5.7e-13: profiles/r06/arbiter_case_671.txt), so its band comes from the
            # same five evaluations.
            fwd_draws = [float(np.abs(d_d - d_x).max() / nrm)]
            bwd_draws = [backward_error(p, bx, radius, scale_x, d_d, dp_d)]
            rng = np.random.default_rng(1000 * it + 17)
            for _ in range(4):
                rec_k = rec * (1.0 + np.ldexp(1.0, -52) * rng.choice([-1.0, 1.0], size=rec.shape))
                d_k, _, _, dp_k = block_step(p, record_blocks(p, rec_k, c0, x0, np.float64), radius, scale_d_prev, np.float64)
                fwd_draws.append(float(np.abs(d_k - d_x).max() / nrm))
                bwd_draws.append(backward_error(p, bx, radius, scale_x, d_k, dp_k))
            rows.append(dict(it=it, cond=float(np.linalg.cond(S.astype(np.float64))),
                             bwd_engine=backward_error(p, bx, radius, scale_x, d_e, dp_e), bwd_f64=bwd_draws[0], bwd_f64_band=max(bwd_draws),
                             fwd_engine=float(np.abs(d_e - d_x).max() / nrm), fwd_f64=fwd_draws[0], fwd_f64_band=max(fwd_draws),
                             data_shift=float(np.abs(d_o - d_x).max() / nrm)))
            scale_x = scale_x1
    return rows
