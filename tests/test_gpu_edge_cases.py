"""Edge cases of the hot path through the C-ABI: ragged visibility (1..16 observations per point, points that straddle
the wave boundary of a Schur tile), a one-point problem, no constant camera, a free camera that nobody observes.
Reference behaviour: every (point, frame) pair handed to AddResidualBlock is a block of its own, nothing is rejected
(photobundle.cc:791-804, :725-726); SetParameterBlockConstant is optional (:809-813)."""
import copy
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, EngineError, default_solver_options

from gpu_util import dense_system, make_engine, reference_step

pytestmark = pytest.mark.gpu

SMALL = dict(size=(120, 200), K=(250.0, 250.0, 100.0, 60.0))


def _subsample(p, keep_mask):
    """WindowProblem with a subset of the observations (every point keeps >= 1)."""
    q = copy.copy(p)
    q.obs_point = p.obs_point[keep_mask].copy()
    q.obs_slot = p.obs_slot[keep_mask].copy()
    assert len(np.unique(q.obs_point)) == p.n_points
    return q


def _ragged(n_frames, n_points, seed):
    p = synthetic.make_window(n_frames=n_frames, n_points=n_points, radius=2, seed_offset=seed, **SMALL)
    rng = np.random.default_rng(seed)
    keep = np.zeros(p.n_obs, bool)
    begin = np.searchsorted(p.obs_point, np.arange(p.n_points + 1))
    for pt in range(p.n_points):
        n = begin[pt + 1] - begin[pt]
        k = int(rng.integers(1, n + 1))                     # 1 .. n_frames observations
        keep[begin[pt] + rng.choice(n, size=k, replace=False)] = True
    return _subsample(p, keep)


def _check_against_oracle(p, iterations=8):
    lin = oracle.linearize(p, blocks=False)
    with make_engine(p) as e:
        cost = e.linearize()
        rec = e.obs_records()
        assert np.isclose(cost, lin["cost"], rtol=1e-12)
        assert np.allclose(rec[:, 5], 0.5 * lin["block_sqnorm"], rtol=1e-12)
        res = e.solve(default_solver_options(max_num_iterations=iterations))
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=iterations))
    assert len(ref["iterations"]) == len(res["iterations"]), (ref["message"], res["message"])
    for a, b in zip(ref["iterations"], res["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"], a["iteration"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-9), a["iteration"]
    assert np.abs(res["cams"] - ref["cams"]).max() <= 1e-5
    return ref, res


@pytest.mark.parametrize("n_frames,seed", [(16, 1), (8, 2), (5, 3)])
def test_ragged_visibility(n_frames, seed):
    p = _ragged(n_frames, 300, seed)
    counts = np.bincount(p.obs_point)
    assert counts.min() == 1 and counts.max() >= n_frames - 1
    _check_against_oracle(p)


def test_ragged_reduced_system_matches_dense_algebra():
    p = _ragged(16, 60, 7)
    J, r, n_cam = dense_system(p)
    ref = reference_step(J, r, n_cam, 1e4)
    with make_engine(p) as e:
        e.linearize()
        e.step(1e4, init_scale=True)
        S, rhs = e.reduced_system()
    assert np.abs(S - ref["S"]).max() <= 1e-9 * np.abs(ref["S"]).max()
    assert np.abs(rhs - ref["rhs"]).max() <= 1e-9 * np.abs(ref["rhs"]).max()


def test_single_point():
    p = synthetic.make_window(n_frames=3, n_points=1, radius=2, seed_offset=4, **SMALL)
    assert p.n_obs == 3
    _check_against_oracle(p, iterations=5)


def test_no_constant_camera():
    """fixed_slot = -1: all cameras free (gauge freedom is absorbed by the LM damping, as in Ceres)."""
    p = synthetic.make_window(n_frames=4, n_points=200, radius=2, seed_offset=6, **SMALL)
    p.fixed_slot = -1
    _check_against_oracle(p, iterations=6)


def test_unobserved_free_camera():
    """A free camera without residual blocks keeps its pose (its block of the reduced system is the LM diagonal only)."""
    p = synthetic.make_window(n_frames=5, n_points=200, radius=2, seed_offset=8, **SMALL)
    q = _subsample(p, p.obs_slot != 3)
    ref, res = _check_against_oracle(q, iterations=6)
    assert np.array_equal(res["cams"][3], q.cams[3])
    # ... and is not a parameter block of the Ceres program (it never reaches AddResidualBlock, photobundle.cc:791-804):
    # |x| of the parameter-tolerance test runs over the observed free cameras and the points only
    with make_engine(q) as e:
        e.linearize()
        info = e.step(1e4, init_scale=True)
    big = copy.copy(q)
    big.cams = q.cams.copy()
    big.cams[3, 3:] = 1e9          # counted in |x| it would end the solve at the first parameter-tolerance test
    with make_engine(big) as e:
        res_big = e.solve(default_solver_options(max_num_iterations=6))
    assert [i["cost"] for i in res_big["iterations"]] == [i["cost"] for i in res["iterations"]]
    live = [c for c in range(q.n_frames) if c != q.fixed_slot and c != 3]
    expect = np.sqrt((q.cams[live] ** 2).sum() + (q.xyz ** 2).sum())
    assert np.isclose(info["x_norm"], expect, rtol=1e-13), (info["x_norm"], expect)


def test_rejected_inputs(small_window_edge=None):
    p = synthetic.make_window(n_frames=3, n_points=20, radius=2, seed_offset=9, **SMALL)
    e = Engine(120, 200, p.K, 2, 3)
    try:
        for s_ in range(3):
            e.set_frame(s_, p.images[s_])
        with pytest.raises(EngineError):       # empty problem
            e.set_problem(p.xyz[:0], p.desc[:0], p.obs_point[:0], p.obs_slot[:0], p.weights)
        bad = p.obs_slot.copy(); bad[0] = 7
        with pytest.raises(EngineError):       # slot outside the window
            e.set_problem(p.xyz, p.desc, p.obs_point, bad, p.weights)
        e.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights)
        with pytest.raises(EngineError):       # a single frame cannot form a window
            e.set_cameras(p.cams[:1], 0)
    finally:
        e.close()


def test_solve_call_order_is_checked_before_anything_is_launched():
    """include/pba.h: PBA_ERR_STATE for a solve before set_problem / set_cameras (both drivers), for an observation whose
    slot has no camera, and for a slot that never received a frame; the three setters may come in any order."""
    p = synthetic.make_window(n_frames=3, n_points=20, radius=2, seed_offset=9, **SMALL)
    o = default_solver_options(max_num_iterations=3)
    e = Engine(120, 200, p.K, 2, 3)
    try:
        with pytest.raises(EngineError, match="call order"):
            e.solve(o)                                          # nothing set
        e.set_cameras(p.cams, 0)
        with pytest.raises(EngineError, match="call order"):
            e.solve(o)                                          # cameras but no problem
        e.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights)
        with pytest.raises(EngineError, match="no frame"):
            e.solve(o)                                          # frames missing
        for s_ in range(3):
            e.set_frame(s_, p.images[s_])
        a = e.solve(o)                                          # cameras -> problem -> frames: fine
        e.set_cameras(p.cams[:2], 0)
        with pytest.raises(EngineError, match="slot 2"):
            e.solve(o)                                          # observations of slot 2, two cameras
    finally:
        e.close()
    e = Engine(120, 200, p.K, 2, 3)
    try:
        e.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights)
        with pytest.raises(EngineError, match="call order"):
            e.solve(o)                                          # problem but no cameras
        with pytest.raises(EngineError, match="call order"):
            e.linearize()
        e.load(p)
        b = e.solve(o)
        # a second window on the same handle after a solve that may have ended on either parity
        e.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights)
        e.set_cameras(p.cams, 0)
        c = e.solve(o)
        e.set_cameras(p.cams, 0)
        e.set_problem(p.xyz, p.desc, p.obs_point, p.obs_slot, p.weights)
        d = e.solve(o)
    finally:
        e.close()
    for r in (b, c, d):
        assert r["final_cost"] == a["final_cost"] and np.array_equal(r["cams"], a["cams"]) and np.array_equal(r["xyz"], a["xyz"])


def test_rejected_first_step_is_logged_as_rejected():
    """A window whose FIRST trust-region step is rejected (case 66 of the seeded sweep in test_gpu_random_shapes.py): the
    asynchronous driver logs iteration zero and decides iteration one in the same device call, and the iteration-one
    record used to inherit step_is_successful = 1 from iteration zero when that step was rejected (the trajectory itself
    was right).  Both drivers against the oracle, flag by flag."""
    import test_gpu_random_shapes as sweep
    p = sweep._make(sweep._draw_cases(67)[66])
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=5))
    assert ref["iterations"][1]["step_is_successful"] == 0 and ref["iterations"][1]["step_is_valid"] == 1
    for async_on in ("1", "0"):
        os.environ["PBA_ASYNC"] = async_on
        try:
            with make_engine(p) as e:
                res = e.solve(default_solver_options(max_num_iterations=5))
        finally:
            os.environ.pop("PBA_ASYNC", None)
        assert len(res["iterations"]) == len(ref["iterations"])
        for a, b in zip(ref["iterations"], res["iterations"]):
            assert (a["step_is_valid"], a["step_is_successful"]) == (b["step_is_valid"], b["step_is_successful"]), (async_on, a["iteration"])
            assert np.isclose(a["cost"], b["cost"], rtol=1e-9) and np.isclose(a["relative_decrease"], b["relative_decrease"], rtol=1e-6, atol=1e-12)
        assert res["num_successful_steps"] == ref["num_successful_steps"]



@pytest.mark.parametrize("n_it", [0, 5])
def test_non_finite_initial_point_is_an_evaluation_failure_and_leaves_no_stale_flag(n_it):
    """the zero-iteration solve decides in the last workgroup of the gradient-only pass, which reads the
    non-finite flag another workgroup of the same launch wrote -- a plain store there is not visible across XCDs.  A NaN descriptor
    must end the solve as Ceres does ("Initial residual and Jacobian evaluation failed."), and a clean solve on the same engine
    right after must not inherit the flag."""
    p = synthetic.make_window(n_frames=4, n_points=400, radius=2, **SMALL)
    bad = copy.copy(p)
    bad.desc = p.desc.copy()
    bad.desc.reshape(p.n_points, -1)[123, 7] = np.nan      # a non-finite residual in every block of point 123
    opts = default_solver_options(max_num_iterations=n_it)
    with Engine(SMALL["size"][0], SMALL["size"][1], p.K, p.radius, p.n_frames, huber=p.huber) as e:
        for rep in range(3):
            e.load(bad)
            r = e.solve(opts)
            assert "evaluation failed" in r["message"], r["message"]
            assert len(r["iterations"]) == 0
            e.load(p)
            r = e.solve(opts)
            assert "evaluation failed" not in r["message"], r["message"]
            assert len(r["iterations"]) >= 1 and np.isfinite(r["iterations"][0]["cost"])
