"""-m gpu: parity of the HIP engine (through the C-ABI) against the CPU oracle on identical seeded inputs.

Tolerances (fp64 path; the image samples themselves are bit-exact by construction):
  per-observation structure tensors / costs      1e-12 relative
  reduced camera system S, rhs                    1e-9  relative to the largest entry
  per-iteration LM cost                           1e-9  relative, identical accept/reject sequence
  refined poses                                   1e-5  absolute (north_star bar), typically ~1e-9
"""
import os

import numpy as np
import pytest

from oracle import oracle
from photobundle_amd import synthetic
from photobundle_amd.engine import default_solver_options

from gpu_util import check_obs_records, dense_system, make_engine, reference_step

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

SMALL = dict(size=(120, 160), K=(200.0, 200.0, 80.0, 60.0))


def test_device_planes_are_bit_exact(small_window):
    with make_engine(small_window) as e:
        for s in range(small_window.n_frames):
            assert np.array_equal(e.get_frame_planes(s), oracle.planes_from_u8(small_window.images[s]))


def _check_records(p, e):
    cost = e.linearize()
    rec = e.obs_records()
    lin = oracle.linearize(p, blocks=False)
    s = lin["block_sqnorm"]
    a = p.huber
    rho = np.where((a > 0) & (s > a * a), 2 * a * np.sqrt(s) - a * a, s)
    assert np.allclose(rec[:, 5], 0.5 * rho, rtol=1e-12, atol=0)
    assert np.isclose(cost, lin["cost"], rtol=1e-12)
    # columns 0-4 (rho' M, rho' b): every block of J^T J and J^T r they generate, per observation, against the oracle's
    # dual-number rows
    check_obs_records(p, rec)
    return cost


@pytest.mark.parametrize("radius,huber,gaussian,vis", [(2, 0.0, False, "dense"), (1, 0.05, False, "causal"),
                                                     (3, 0.0, True, "dense"), (5, 0.05, False, "dense")])
def test_linearisation_and_reduced_system(radius, huber, gaussian, vis):
    p = synthetic.make_window(n_frames=4, n_points=60, radius=radius, huber=huber, gaussian=gaussian, visibility=vis,
                              seed_offset=radius, **SMALL)
    J, r, n_cam = dense_system(p)
    ref = reference_step(J, r, n_cam, 1e4)
    with make_engine(p) as e:
        _check_records(p, e)
        info = e.step(1e4, init_scale=True)
        S, rhs = e.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-9 * np.abs(ref["S"]).max()
        assert np.abs(rhs - ref["rhs"]).max() <= 1e-9 * np.abs(ref["rhs"]).max()
        assert np.isclose(info["gradient_max_norm"], np.abs(ref["gradient"]).max(), rtol=1e-10)
        assert np.isclose(info["gradient_norm"], np.linalg.norm(ref["gradient"]), rtol=1e-10)
        assert np.isclose(info["model_cost_change"], ref["model_cost_change"], rtol=1e-7)
        assert np.isclose(info["step_norm"], np.linalg.norm(ref["delta"]), rtol=1e-7)
        x = np.concatenate([np.delete(p.cams, p.fixed_slot, 0).reshape(-1), p.xyz.reshape(-1)])
        assert np.isclose(info["x_norm"], np.linalg.norm(x), rtol=1e-13)
        # candidate point and its cost
        cams_c = p.cams.copy()
        free = [c for c in range(p.n_frames) if c != p.fixed_slot]
        cams_c[free] += ref["delta"][:n_cam].reshape(-1, 6)
        xyz_c = p.xyz + ref["delta"][n_cam:].reshape(-1, 3)
        c_ref, _ = oracle.cost(p, cams=cams_c, xyz=xyz_c)
        assert np.isclose(info["candidate_cost"], c_ref, rtol=1e-6)
        e.accept()
        cams_g, xyz_g = e.get_state()
        assert np.abs(cams_g - cams_c).max() <= 1e-7 * max(1.0, np.abs(ref["delta"][:n_cam]).max())
        assert np.abs(xyz_g - xyz_c).max() <= 1e-6 * max(1.0, np.abs(ref["delta"][n_cam:]).max())
        # the candidate pass and a fresh linearisation at the same point agree: bit for bit per observation; the total
        # may differ in the last place when the two passes ran on different workgroup grids (PBA_SPECULATE=0 / PBA_FUSE=0)
        assert np.isclose(e.linearize(), info["candidate_cost"], rtol=1e-14, atol=0.0)
        c_here, _ = oracle.cost(p, cams=cams_g, xyz=xyz_g)
        assert np.isclose(info["candidate_cost"], c_here, rtol=1e-12)


def _compare_traces(p, max_it, pose_tol=1e-5):
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=max_it))
    with make_engine(p) as e:
        res = e.solve(default_solver_options(max_num_iterations=max_it))
    ri, gi = ref["iterations"], res["iterations"]
    assert len(ri) == len(gi), (ref["message"], res["message"])
    for a, b in zip(ri, gi):
        assert a["iteration"] == b["iteration"]
        assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"], a["iteration"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-9), a["iteration"]
        assert np.isclose(a["trust_region_radius"], b["trust_region_radius"], rtol=1e-6)
        assert np.isclose(a["gradient_max_norm"], b["gradient_max_norm"], rtol=1e-6)
        assert np.isclose(a["gradient_norm"], b["gradient_norm"], rtol=1e-6), a["iteration"]   # the last one: gradient-only pass
        if a["iteration"] > 0 and a["step_is_valid"]:
            assert np.isclose(a["step_norm"], b["step_norm"], rtol=1e-5)
    assert res["termination_type"] == ref["termination_type"]
    assert np.isclose(res["initial_cost"], ref["initial_cost"], rtol=1e-12)
    assert np.isclose(res["final_cost"], ref["final_cost"], rtol=1e-9)
    assert res["num_successful_steps"] == ref["num_successful_steps"]
    assert res["num_residuals"] == ref["num_residuals"]
    assert np.abs(res["cams"] - ref["cams"]).max() <= pose_tol
    rmse = np.sqrt(np.mean((res["cams"][1:] - ref["cams"][1:]) ** 2))
    assert rmse <= pose_tol
    assert np.array_equal(res["cams"][p.fixed_slot], p.cams[p.fixed_slot])
    return ref, res


def test_lm_trajectory_matches_oracle(small_window):
    _compare_traces(small_window, 25)


def test_lm_trajectory_huber_causal(small_window_huber):
    _compare_traces(small_window_huber, 25)


def test_lm_trajectory_to_convergence():
    p = synthetic.make_window(n_frames=4, n_points=400, radius=2, rot_deg=0.03, trans=0.005, depth_noise=0.002,
                              seed_offset=11, **SMALL)
    ref, res = _compare_traces(p, 60)
    gt = p.meta["cams_gt"]
    assert np.linalg.norm(res["cams"][1:, :3] - gt[1:, :3]) < np.linalg.norm(p.cams[1:, :3] - gt[1:, :3])


def test_points_outside_the_image_take_the_clamped_path():
    """Out-of-image / behind-the-border projections are not rejected by the reference (photobundle.cc:725-726); the
    sampler clamps (sample_eigen.h:38-51).  Push points around so many patches straddle the border."""
    p = synthetic.make_window(n_frames=3, n_points=80, radius=2, seed_offset=5, **SMALL)
    rng = np.random.default_rng(0)
    p.xyz = p.xyz + rng.normal(0, 1.0, p.xyz.shape) * np.array([3.0, 2.0, 0.0])
    p.xyz[0] = [1e3, 0.0, 1.0]        # far right of the image
    p.xyz[1] = [0.0, 0.0, -2.0]       # behind the camera
    p.xyz[2] = [-0.41, -0.31, 1.0]    # u ~ -2: clamp-to-zero columns
    lin = oracle.linearize(p, blocks=False)
    with make_engine(p) as e:
        cost = e.linearize()
        rec = e.obs_records()
    assert np.isclose(cost, lin["cost"], rtol=1e-12)
    assert np.allclose(rec[:, 5], 0.5 * lin["block_sqnorm"], rtol=1e-12)
    check_obs_records(p, rec)       # gradients of clamped / border taps included


def test_run_to_run_determinism(small_window):
    out = []
    for _ in range(2):
        with make_engine(small_window) as e:
            res = e.solve(default_solver_options(max_num_iterations=10))
            out.append((res["cams"].copy(), res["xyz"].copy(), res["final_cost"]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


def test_sixteen_frames_reduced_system():
    p = synthetic.make_window(n_frames=16, n_points=40, radius=2, size=(120, 200), K=(250.0, 250.0, 100.0, 60.0),
                              seed_offset=2, visibility="causal")
    J, r, n_cam = dense_system(p)
    ref = reference_step(J, r, n_cam, 1e4)
    with make_engine(p) as e:
        e.linearize()
        e.step(1e4, init_scale=True)
        S, rhs = e.reduced_system()
    assert S.shape == (90, 90)
    assert np.abs(S - ref["S"]).max() <= 1e-9 * np.abs(ref["S"]).max()
    assert np.abs(rhs - ref["rhs"]).max() <= 1e-9 * np.abs(ref["rhs"]).max()


@pytest.mark.parametrize("n_frames", [6, 9, 11, 12, 14])
def test_lm_trace_other_window_lengths(n_frames):
    """Every instantiation family of the reduced-system solve: k_solve_wave<NF> (n <= 60) and k_solve_wave2<NF>
    (60 < n <= 90) against the oracle's trust-region trace."""
    p = synthetic.make_window(n_frames=n_frames, n_points=60, radius=2, size=(120, 200), K=(250.0, 250.0, 100.0, 60.0),
                              seed_offset=3 + n_frames, visibility="causal")
    _compare_traces(p, 12)


def test_error_paths(small_window):
    from photobundle_amd.engine import Engine, EngineError
    p = small_window
    e = Engine(120, 160, p.K, 2, 4)
    with pytest.raises(EngineError, match="call order"):
        e.linearize()
    with pytest.raises(EngineError):
        e.set_problem(p.xyz, p.desc, p.obs_point[::-1].copy(), p.obs_slot, p.weights)   # not grouped by point
    e.close()


def test_final_pass_that_decides_and_flushes_matches_the_separate_kernels(small_window):
    """r5: at the iteration limit the gradient-only pass of a single-rank solve decides and flushes in its own last workgroup
    (k_reduce_solve with a Fin block) instead of k_decide + k_flush behind it, and the trust-region state is initialised by the first
    kernel of the solve: same log, same state, bit for bit, as the round-4 launch sequence (PBA_FUSE_FINAL=0 in a fresh process)."""
    import json, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import json, sys
        sys.path.insert(0, %r)
        from photobundle_amd import synthetic
        from photobundle_amd.engine import Engine, default_solver_options
        out = []
        for n_it in (0, 1, 4, 7):
            p = synthetic.make_window(n_frames=4, n_points=300, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0))
            with Engine(120, 160, p.K, p.radius, p.n_frames, huber=p.huber) as e:
                e.load(p)
                r = e.solve(default_solver_options(max_num_iterations=n_it, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
            out.append([[i["cost"].hex(), i["gradient_max_norm"].hex(), i["gradient_norm"].hex(), i["trust_region_radius"].hex(), i["step_is_successful"]]
                        for i in r["iterations"]] + [r["cams"].tobytes().hex(), r["final_cost"].hex(), r["message"]])
        print("RESULT" + json.dumps(out))
    """ % ROOT)
    runs = []
    for fuse in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PBA_FUSE_FINAL=fuse))
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:]))
    assert runs[0] == runs[1]
    assert len(runs[0][3]) == 8 + 3          # iteration 0 + 7 logged steps (+ cameras, final cost, message)
