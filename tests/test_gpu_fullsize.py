"""-m gpu: parity at the BASELINE.json size (configs[1]: 8 frames, 50k points, 5x5 patches, 1241x376 frames).

The oracle still finishes a handful of LM iterations in seconds at this size, so the headline claim is checked
directly: same accept/reject decisions, per-iteration cost to 1e-9 relative, pose RMSE <= 1e-5 (north_star).  Plus
size-independent properties: run-to-run bit determinism and monotone cost over accepted steps."""
import numpy as np
import pytest

from gpu_util import pose_rmse, referee_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_window():
    import referee_cache as rc
    return rc.windows()["configs1"]()


@pytest.mark.timeout(1200)
def test_configs1_parity_with_oracle(full_window):
    from oracle import oracle
    from photobundle_amd.engine import default_solver_options
    from gpu_util import check_obs_records, make_engine
    p = full_window
    assert p.n_obs == 400000
    c_ref, sq = oracle.cost(p)
    with make_engine(p, keep_reduced_system=False) as e:
        c = e.linearize()
        assert np.isclose(c, c_ref, rtol=1e-12)
        rec = e.obs_records()
        assert np.allclose(rec[:, 5], 0.5 * sq, rtol=1e-12)
        # all 400k Jacobian-pass records (M, b) against the oracle's dual-number rows
        worst = check_obs_records(p, rec, threads=8)
        print("configs[1] record check, worst relative block errors:", worst)
    n_it = 4
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=n_it, num_threads=8))
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options(max_num_iterations=n_it))
    assert len(res["iterations"]) == len(ref["iterations"]) == n_it + 1
    for a, b in zip(ref["iterations"], res["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-9), (a["iteration"], a["cost"], b["cost"])
        assert np.isclose(a["gradient_max_norm"], b["gradient_max_norm"], rtol=1e-6)
    rmse = np.sqrt(np.mean((res["cams"][1:] - ref["cams"][1:]) ** 2))
    assert rmse <= 1e-5, rmse
    assert np.abs(res["cams"] - ref["cams"]).max() <= 1e-5
    assert res["num_residuals"] == 400000 * 25


@pytest.mark.timeout(600)
def test_configs1_determinism_and_monotone_cost(full_window):
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    o = default_solver_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    runs = []
    for _ in range(2):
        with make_engine(full_window, keep_reduced_system=False) as e:
            runs.append(e.solve(o))
    a, b = runs
    assert a["final_cost"] == b["final_cost"] and np.array_equal(a["cams"], b["cams"]) and np.array_equal(a["xyz"], b["xyz"])
    cost = a["iterations"][0]["cost"]
    for it in a["iterations"][1:]:
        if it["step_is_successful"]:
            assert it["cost"] < cost
            cost = it["cost"]
    assert a["final_cost"] == cost < a["initial_cost"]


@pytest.mark.timeout(900)
def test_sixteen_frame_window_large():
    """configs[3] shape scaled to one GPU's test budget: 16 frames, 20k points (320k residual blocks), 90x90 system."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    p = synthetic.make_window(n_frames=16, n_points=20000, radius=2)
    with make_engine(p) as e:
        c = e.linearize()
        c_ref, _ = oracle.cost(p)
        assert np.isclose(c, c_ref, rtol=1e-12)
        e.step(1e4, init_scale=True)
        S, rhs = e.reduced_system()
        assert S.shape == (90, 90) and np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
        assert np.linalg.eigvalsh(S).min() > 0
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=2, num_threads=8))
    with make_engine(p) as e:
        res = e.solve(default_solver_options(max_num_iterations=2))
    for a, b in zip(ref["iterations"], res["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-9)
    assert np.abs(res["cams"] - ref["cams"]).max() <= 1e-5


@pytest.mark.timeout(1500)
def test_configs4_shape_huber_11x11():
    """configs[4] shape: 8 frames, 50k points, 11x11 patches, Huber a = 0.05 (48.4 M residuals).  The oracle still does
    the cost and ONE LM iteration at this size; beyond that the check is the size-independent monotone-cost property."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    p = synthetic.make_window(n_frames=8, n_points=50000, radius=5, huber=0.05)
    assert p.n_obs == 400000 and p.patch_len == 121
    c_ref, sq = oracle.cost(p, threads=8)
    with make_engine(p, keep_reduced_system=False) as e:
        c = e.linearize()
        assert np.isclose(c, c_ref, rtol=1e-12)
        rec = e.obs_records()
        a = p.huber
        rho = np.where(sq > a * a, 2 * a * np.sqrt(sq) - a * a, sq)
        assert np.allclose(rec[:, 5], 0.5 * rho, rtol=1e-12)
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=1, num_threads=8, use_autodiff=0))
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options(max_num_iterations=6))
    for a_, b_ in zip(ref["iterations"], res["iterations"][:2]):
        assert a_["step_is_successful"] == b_["step_is_successful"]
        assert np.isclose(a_["cost"], b_["cost"], rtol=1e-9)
    cost = res["iterations"][0]["cost"]
    for it in res["iterations"][1:]:
        if it["step_is_successful"]:
            assert it["cost"] < cost
            cost = it["cost"]
    assert res["num_residuals"] == 400000 * 121


def _trace_parity(ref, res, pose_tol=1e-5):
    ri, gi = ref["iterations"], res["iterations"]
    assert len(ri) == len(gi), (len(ri), len(gi), ref["message"], res["message"])
    for a, b in zip(ri, gi):
        assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"], a["iteration"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-9), (a["iteration"], a["cost"], b["cost"])
        assert np.isclose(a["trust_region_radius"], b["trust_region_radius"], rtol=1e-6), a["iteration"]
        assert np.isclose(a["gradient_max_norm"], b["gradient_max_norm"], rtol=1e-5), a["iteration"]
    assert res["termination_type"] == ref["termination_type"], (ref["message"], res["message"])
    assert res["num_successful_steps"] == ref["num_successful_steps"]
    assert np.isclose(res["final_cost"], ref["final_cost"], rtol=1e-9)
    free = [c for c in range(ref["cams"].shape[0])][1:]
    d = res["cams"][free] - ref["cams"][free]
    rmse_rot = np.sqrt(np.mean(d[:, :3] ** 2))
    rmse_t = np.sqrt(np.mean(d[:, 3:] ** 2))
    assert rmse_rot <= pose_tol and rmse_t <= pose_tol, (rmse_rot, rmse_t)
    assert np.abs(d).max() <= pose_tol
    return rmse_rot, rmse_t



def _oracle_runs(p, window, ulp_twins=False):
    """The referee + double-precision runs of the oracle: dual numbers, analytic Jacobian and (ulp_twins) the analytic one
    with every point coordinate moved one ulp up / down -- four samples of where a double-precision solve of this window
    may end up.  These are the slow part of this file (~100 s of host time per test): tests/referee_cache.py serves them
    from the committed tests/golden/referee_traces.json when the entry was computed for exactly this window, and runs the
    oracle live otherwise."""
    import referee_cache as rc
    q = rc.solve(p, window + "/referee")
    twins = [rc.solve(p, window + "/twin_autodiff"), rc.solve(p, window + "/twin_analytic")]
    if ulp_twins:
        twins += [rc.solve(p, window + "/twin_ulp_up"), rc.solve(p, window + "/twin_ulp_down")]
    return q, twins


@pytest.mark.timeout(2400)
def test_configs1_parity_to_convergence(full_window):
    """configs[1] at full size with the reference's solver options (tolerances on, photobundle.cc:738-761), run until a
    tolerance terminates the solve (~70-100 iterations), against the extended-precision referee with two double-precision
    oracle runs fixing the noise floor (see _referee_parity).  Then the north_star pose bar on this (poor-init, headline)
    window: solves truncated after k iterations are compared pose by pose -- measured: the engine holds 1e-5 (rotation
    in rad AND translation in m, RMSE over the free cameras) for the first 8 iterations (8e-9 rad / 3e-7 m after 8), the
    double-precision oracle itself only for the first 6 (5e-5 m after 8); from iteration ~10 on the rounding noise of
    any double-precision evaluation is amplified past the bar."""
    from oracle import oracle
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    p = full_window
    import referee_cache as rc
    q, twins = _oracle_runs(p, "configs1", ulp_twins=True)
    assert q["termination_type"] == 0 and len(q["iterations"]) >= 10, q["message"]
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options())
        tight = referee_parity(q, twins, res, "configs[1] to convergence")
        assert tight >= 5
        from gpu_util import trajectory_consistency
        worst_c = trajectory_consistency(p, e, (10, 40), lambda k_: default_solver_options(max_num_iterations=k_))
        print("configs[1]: oracle cost at the engine's own states after 10 / 40 iterations: largest relative difference %.1e" % worst_c)
        hold_engine = hold_twin = 0
        for k in (2, 4, 6, 8):
            qk = rc.solve(p, "configs1/referee_%d" % k)
            tk = rc.solve(p, "configs1/autodiff_%d" % k)
            e.load(p)
            rk = e.solve(default_solver_options(max_num_iterations=k))
            pe, pt = pose_rmse(rk["cams"], qk["cams"]), pose_rmse(tk["cams"], qk["cams"])
            print("configs[1], %d iterations: pose RMSE to the referee: engine rot %.2e rad trans %.2e m; double oracle %.2e / %.2e"
                  % (k, pe[0], pe[1], pt[0], pt[1]))
            if max(pe) <= 1e-5 and hold_engine == k - 2:
                hold_engine = k
            if max(pt) <= 1e-5 and hold_twin == k - 2:
                hold_twin = k
            assert pe[0] <= 2.0 * pt[0] + 1e-12 and pe[1] <= 2.0 * pt[1] + 1e-12
        print("configs[1]: the 1e-5 pose bar holds for the first %d iterations (engine) / %d (double-precision oracle)" % (hold_engine, hold_twin))
        assert hold_engine >= 6 and hold_engine >= hold_twin


@pytest.mark.timeout(2400)
def test_configs1_well_initialised_window_to_convergence():
    """Same shape, the "good VO" regime (small pose / depth perturbation): everything stays together for ~20 iterations,
    so 20 iterations are compared at the tight tolerances against the plain oracle, then the run to convergence as above."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    import referee_cache as rc
    p = rc.windows()["configs1_good"]()
    q, twins = _oracle_runs(p, "configs1_good")
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options(max_num_iterations=150))
        tight = referee_parity(q, twins, res, "configs[1], well initialised, to convergence")
        assert tight >= 20
        # AT CONVERGENCE even this window is decided by where the slowly converging tail stops (referee 95 iterations, the double
        # oracle runs 68 and 150, r4 engine 81): the double-precision runs end 1e-5 rad / 5e-4 m from the referee, the engine
        # 5e-6 rad / 2.3e-4 m (r3's build happened to stop at the referee's point: 2e-9 / 1e-7).  Asserted: no farther than the
        # oracle's own double-precision runs, in the raw and in the gauge-fixed metric (the 2x envelope is in referee_parity)
        gf_en, gf_tw = referee_parity.last_gauge_fixed
        assert gf_en[0] <= max(g[0] for g in gf_tw) + 1e-12 and gf_en[1] <= max(g[1] for g in gf_tw) + 1e-12, (gf_en, gf_tw)
        n_it = 12
        e.load(p)
        res12 = e.solve(default_solver_options(max_num_iterations=n_it))
    ref12 = rc.solve(p, "configs1_good/autodiff_12")
    rr, rt = _trace_parity(ref12, res12)
    print("configs[1], well initialised, %d iterations: pose RMSE rot %.3e rad, trans %.3e m" % (n_it, rr, rt))


@pytest.mark.timeout(2400)
def test_configs4_ten_iterations_against_oracle():
    """configs[4] shape (8 frames, 50k points, 11x11, Huber 0.05): 10 LM iterations.  The second oracle run that fixes
    the noise floor sees the same inputs with the points moved by one ulp."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    import referee_cache as rc
    p = rc.windows()["configs4"]()
    n_it = 10
    q = rc.solve(p, "configs4/referee_10")
    ref = rc.solve(p, "configs4/analytic_10")
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options(max_num_iterations=n_it))
    assert len(ref["iterations"]) == n_it + 1 == len(res["iterations"]) == len(q["iterations"])
    run = 0.0
    for i, (a, b, c) in enumerate(zip(q["iterations"], ref["iterations"], res["iterations"])):
        assert a["step_is_successful"] == b["step_is_successful"] == c["step_is_successful"]
        run = max(run, abs(b["cost"] - a["cost"]) / a["cost"])
        assert abs(c["cost"] - a["cost"]) / a["cost"] <= max(1e-12, 2.0 * run), (i, c["cost"], a["cost"], run)
    pe, pt = pose_rmse(res["cams"], q["cams"]), pose_rmse(ref["cams"], q["cams"])
    print("configs[4] %d iterations: pose RMSE to the referee: engine rot %.2e rad trans %.2e m, double oracle %.2e / %.2e" % (n_it, pe[0], pe[1], pt[0], pt[1]))
    assert max(pe) <= 1e-5 and pe[0] <= 2.0 * pt[0] + 1e-12 and pe[1] <= 2.0 * pt[1] + 1e-12


@pytest.fixture(scope="module")
def window_configs3():
    from photobundle_amd import synthetic
    return synthetic.make_window(n_frames=16, n_points=200000, radius=2, dense_births=(0, 8))


@pytest.mark.timeout(2400)
def test_configs3_full_shape_on_one_gpu(window_configs3):
    """configs[3] at its stated shape on ONE GPU: 16 frames x 200k points = 3.2 M residual blocks (80 M residuals),
    90x90 reduced system.  Cost and every Jacobian-pass record against the oracle, 3 LM iterations of trace parity."""
    from oracle import oracle
    from photobundle_amd.engine import default_solver_options
    from gpu_util import check_obs_records, make_engine
    p = window_configs3
    assert p.n_obs == 3200000 and p.n_frames == 16
    c_ref, sq = oracle.cost(p, threads=8)
    with make_engine(p, keep_reduced_system=False) as e:
        c = e.linearize()
        assert np.isclose(c, c_ref, rtol=1e-12)
        rec = e.obs_records()
        assert np.allclose(rec[:, 5], 0.5 * sq, rtol=1e-12)
        worst = check_obs_records(p, rec, threads=8)
        print("configs[3] record check, worst relative block errors:", worst)
    del rec, sq
    n_it = 3
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=n_it, num_threads=8, use_autodiff=0))
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options(max_num_iterations=n_it))
    rr, rt = _trace_parity(ref, res)
    assert res["num_residuals"] == 3200000 * 25
    print("configs[3] %d iterations: pose RMSE rot %.3e rad, trans %.3e m" % (n_it, rr, rt))


@pytest.mark.timeout(1200)
def test_configs3_determinism_and_monotone_cost(window_configs3):
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    o = default_solver_options(max_num_iterations=8, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    runs = []
    for _ in range(2):
        with make_engine(window_configs3, keep_reduced_system=False) as e:
            runs.append(e.solve(o))
    a, b = runs
    assert a["final_cost"] == b["final_cost"] and np.array_equal(a["cams"], b["cams"]) and np.array_equal(a["xyz"], b["xyz"])
    cost = a["iterations"][0]["cost"]
    for it in a["iterations"][1:]:
        if it["step_is_successful"]:
            assert it["cost"] < cost
            cost = it["cost"]
    assert a["final_cost"] == cost < a["initial_cost"]


@pytest.mark.timeout(2400)
def test_configs1_shape_multichannel_descriptors():
    """BASELINE configs[1] shape (8 frames x 50k points, 5x5) with IntensityAndGradient descriptors (3 channels, 75
    residuals per block; reference photobundle.cc:229-245, :708-723) on the fused + asynchronous pipeline: cost and all
    400k Jacobian-pass records against the oracle, 4 LM iterations of trace parity, bit determinism."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import check_obs_records, make_engine
    p = synthetic.make_window(n_frames=8, n_points=50000, radius=2, channel_fn=synthetic.channel_fn("IntensityAndGradient"))
    assert p.n_obs == 400000 and p.channels == 3 and p.desc.shape[1] == 75
    c_ref, sq = oracle.cost(p, threads=8)
    with make_engine(p, keep_reduced_system=False) as e:
        c = e.linearize()
        assert np.isclose(c, c_ref, rtol=1e-12)
        rec = e.obs_records()
        assert np.allclose(rec[:, 5], 0.5 * sq, rtol=1e-12)
        worst = check_obs_records(p, rec, threads=8)
        print("configs[1] shape, 3 channels: record check, worst relative block errors:", worst)
    del rec, sq
    n_it = 4
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=n_it, num_threads=8, use_autodiff=0))
    runs = []
    for _ in range(2):
        with make_engine(p, keep_reduced_system=False) as e:
            runs.append(e.solve(default_solver_options(max_num_iterations=n_it)))
    res = runs[0]
    assert np.array_equal(res["cams"], runs[1]["cams"]) and res["final_cost"] == runs[1]["final_cost"]
    rr, rt = _trace_parity(ref, res)
    assert res["num_residuals"] == 400000 * 75
    print("configs[1] shape, 3 channels, %d iterations: pose RMSE rot %.3e rad, trans %.3e m" % (n_it, rr, rt))
