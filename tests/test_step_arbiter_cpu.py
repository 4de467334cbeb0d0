"""The extended-precision step arbiter of the random sweep (tests/gpu_util.py::block_step) against the oracle's own trust-region
loop on CPU: the arbiter restates one Ceres LM step (Jacobi scaling, clamped LM diagonal, point elimination, dense solve) from the
oracle's per-block products, so its camera step at the initial point must be the first step of oracle.solve -- in float64 and in
x87 extended precision alike, up to the rounding of a well-conditioned little window."""
import numpy as np

from oracle import oracle
from photobundle_amd import synthetic

from gpu_util import block_step, backward_error


def _window(**kw):
    return synthetic.make_window(n_frames=5, n_points=60, radius=2, size=(96, 128), K=(150.0, 150.0, 64.0, 48.0), **kw)


def test_block_step_is_the_first_step_of_the_oracle_loop():
    for kw in (dict(), dict(huber=0.5), dict(visibility="causal", gaussian=True)):
        p = _window(**kw)
        bp = oracle.block_products(p, autodiff=True)
        ref = oracle.solve(p, oracle.default_options(max_num_iterations=1, use_autodiff=1))
        assert ref["iterations"][1]["step_is_successful"]
        free = [c for c in range(p.n_frames) if c != p.fixed_slot]
        want = (ref["cams"] - p.cams)[free]
        for dt, tol in ((np.float64, 1e-9), (np.longdouble, 1e-9)):
            got, scale, S, got_p = block_step(p, bp, 1e4, None, dt)
            assert got.dtype == dt and S.shape == (6 * len(free), 6 * len(free))
            assert np.abs(got.astype(np.float64) - want).max() <= tol * np.abs(want).max(), (kw, dt)
            assert np.abs(got_p.astype(np.float64) - (ref["xyz"] - p.xyz)).max() <= 1e-8 * np.abs(ref["xyz"] - p.xyz).max(), (kw, dt)
            # the restatement is backward stable in its own precision, and the conditioning-free measure sees a 1e-6 slip of the damping
            assert backward_error(p, bp, 1e4, scale, got, got_p) <= (5e-13 if dt is np.float64 else 5e-16)
        scale = block_step(p, bp, 1e4, None, np.longdouble)[1]
        own = backward_error(p, bp, 1e4, scale, want, ref["xyz"] - p.xyz)                                # the oracle's own loop
        assert own <= 5e-13
        assert backward_error(p, bp, 1e4 * (1 + 1e-6), scale, want, ref["xyz"] - p.xyz) >= 1e-11 > 2 * own


def test_extended_precision_is_wider_than_double_here():
    # the arbiter's "exact" leg only means something where numpy's longdouble is the 80-bit x87 format (x86 hosts)
    assert np.finfo(np.longdouble).eps < 1e-18
