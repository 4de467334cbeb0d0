"""The extended-precision step arbiter of the random sweep (tests/gpu_util.py::block_step) against the oracle's own trust-region
loop on CPU: the arbiter restates one Ceres LM step (Jacobi scaling, clamped LM diagonal, point elimination, dense solve) from the
oracle's per-block products, so its camera step at the initial point must be the first step of oracle.solve -- in float64 and in
x87 extended precision alike, up to the rounding of a well-conditioned little window."""
import numpy as np

from oracle import oracle
from photobundle_amd import synthetic

from gpu_util import block_step


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
            got, scale, S = block_step(p, bp, 1e4, None, dt)
            assert got.dtype == dt and S.shape == (6 * len(free), 6 * len(free))
            assert np.abs(got.astype(np.float64) - want).max() <= tol * np.abs(want).max(), (kw, dt)


def test_extended_precision_is_wider_than_double_here():
    # the arbiter's "exact" leg only means something where numpy's longdouble is the 80-bit x87 format (x86 hosts)
    assert np.finfo(np.longdouble).eps < 1e-18
