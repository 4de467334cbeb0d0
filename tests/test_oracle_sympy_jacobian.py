"""CPU: the projection Jacobian A = d(u, v) / d(w, t, X) -- what the reference gets from ceres::Jet through jet_extras.h:87-111 and
the engine evaluates analytically -- derived SYMBOLICALLY (sympy) from the expressions of ceres::AngleAxisRotatePoint and
Calibration::project (reference src/photobundle.cc:696-706, src/calibration.h:34-38) and compared with the oracle's dual-number rows
at machine precision (the central-difference check of test_oracle_solver.py stops at 1e-6).  SURVEY.md 8c names this cross-check."""
import numpy as np
import pytest
sp = pytest.importorskip("sympy")      # a box without sympy skips this cross-check instead of failing collection

from oracle import oracle
from photobundle_amd import synthetic

SMALL = dict(size=(120, 160), K=(200.0, 200.0, 80.0, 60.0))


def _symbolic_A(small_angle):
    w0, w1, w2, t0, t1, t2, X0, X1, X2, fx, fy, cx, cy = sp.symbols("w0 w1 w2 t0 t1 t2 X0 X1 X2 fx fy cx cy", real=True)
    w = sp.Matrix([w0, w1, w2]); X = sp.Matrix([X0, X1, X2]); t = sp.Matrix([t0, t1, t2])
    if small_angle:
        # AngleAxisRotatePoint for theta^2 <= DBL_EPSILON: p + w x p -- the derivative of the code as written
        xw = X + w.cross(X) + t
    else:
        th = sp.sqrt(w0 ** 2 + w1 ** 2 + w2 ** 2)
        a = w / th
        xw = X * sp.cos(th) + a.cross(X) * sp.sin(th) + a * (a.dot(X)) * (1 - sp.cos(th)) + t
    u = fx * xw[0] / xw[2] + cx
    v = fy * xw[1] / xw[2] + cy
    theta = [w0, w1, w2, t0, t1, t2, X0, X1, X2]
    A = sp.Matrix([[sp.diff(u, q) for q in theta], [sp.diff(v, q) for q in theta]])
    args = [w0, w1, w2, t0, t1, t2, X0, X1, X2, fx, fy, cx, cy]
    return sp.lambdify(args, A, modules="mpmath"), sp.lambdify(args, sp.Matrix([u, v]), modules="mpmath")


@pytest.mark.parametrize("small_angle", [False, True])
def test_dual_number_rows_equal_the_symbolic_jacobian(small_angle):
    import mpmath
    mpmath.mp.dps = 40
    forms = {True: _symbolic_A(True), False: _symbolic_A(False)}
    p = synthetic.make_window(n_frames=4, n_points=60, radius=2, seed_offset=21, **SMALL)
    cams = p.cams.copy()
    if small_angle:
        cams[:, :3] *= 1e-10          # theta^2 far below DBL_EPSILON: the first-order branch of the rotation
    R = p.radius
    worst = 0.0
    for obs in range(0, p.n_obs, 7):
        pt, slot = p.obs_point[obs], p.obs_slot[obs]
        cam, X = cams[slot], p.xyz[pt]
        r, jc, jp = oracle.eval_block(p, obs, autodiff=True, cams=cams)
        # (the branch ceres::AngleAxisRotatePoint takes for this camera: the constant camera of a window is the identity)
        fA, fuv = forms[bool(float(cam[:3] @ cam[:3]) <= np.finfo(np.float64).eps)]
        args = [mpmath.mpf(float(x)) for x in list(cam) + list(X) + list(p.K)]
        A = np.array(fA(*args).tolist(), dtype=np.float64)
        uv = np.array([float(x) for x in fuv(*args)])
        i = 0
        for y in range(-R, R + 1):
            for x in range(-R, R + 1):
                s = oracle.sample_linear(p.planes[slot], np.float32(uv[1] + y), np.float32(uv[0] + x))
                Ji = -p.weights[i] * (float(s[1]) * A[0] + float(s[2]) * A[1])
                J = np.concatenate([jc[i], jp[i]])
                scale = np.abs(Ji).max()
                if scale > 0:
                    worst = max(worst, np.abs(J - Ji).max() / scale)
                assert np.allclose(J, Ji, rtol=0.0, atol=1e-12 * scale + 1e-300), (obs, i, J, Ji)
                i += 1
    print("worst relative difference between dual-number and symbolic rows:", worst)
    assert worst <= 1e-12
