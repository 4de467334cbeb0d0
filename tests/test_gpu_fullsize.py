"""-m gpu: parity at the BASELINE.json size (configs[1]: 8 frames, 50k points, 5x5 patches, 1241x376 frames).

The oracle still finishes a handful of LM iterations in seconds at this size, so the headline claim is checked
directly: same accept/reject decisions, per-iteration cost to 1e-9 relative, pose RMSE <= 1e-5 (north_star).  Plus
size-independent properties: run-to-run bit determinism and monotone cost over accepted steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_window():
    from photobundle_amd import synthetic
    return synthetic.make_window(n_frames=8, n_points=50000, radius=2)


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


def _pose_rmse(a, b):
    d = a[1:] - b[1:]
    return float(np.sqrt(np.mean(d[:, :3] ** 2))), float(np.sqrt(np.mean(d[:, 3:] ** 2)))


def _noise_floor_parity(ref, alts, res, tag):
    """Long solves of this problem are CHAOTIC in the rounding: the objective is piecewise bilinear in u8 images and the
    window has a free scale gauge (only camera 0 is constant, photobundle.cc:809-813), so double-precision CPU
    evaluations of the SAME algorithm that differ only in rounding (`ref` = dual-number oracle, `alts` = further oracle
    runs: analytic Jacobian, inputs moved by one ulp either way) drift apart after a handful of iterations.  At
    configs[1] five such twins stop after 63 .. 98 iterations, by function OR parameter tolerance, with final costs up to
    2.4e-3 apart and translations up to a few mm apart.  That spread is the noise floor of the reference itself; the
    engine (one more rounding of the same algorithm) is held to it:
      * while all twins still agree with ref to 1e-9 in cost (the deterministic prefix) the engine matches ref to 1e-9
        with identical accept / reject decisions and trust-region radii,
      * afterwards its distance to ref stays within 20x the largest twin-ref distance seen so far,
      * at the end it has converged (termination_type 0) to a cost that is within 2x the twins' spread of its nearest
        twin -- or below every twin's --, with poses within 2x the twins' pose spread (+ the north_star 1e-5) of its
        nearest twin."""
    if isinstance(alts, dict):
        alts = [alts]
    ri, gi = ref["iterations"], res["iterations"]
    floor, prefix = 0.0, 0
    rows = []
    for i in range(min([len(ri), len(gi)] + [len(a["iterations"]) for a in alts])):
        a, g = ri[i], gi[i]
        floor = max([floor] + [abs(a["cost"] - b["iterations"][i]["cost"]) / a["cost"] for b in alts])
        dg = abs(a["cost"] - g["cost"]) / a["cost"]
        rows.append((i, floor, dg))
        if floor <= 1e-9:
            prefix = i + 1
            assert dg <= 1e-9, (tag, i, a["cost"], g["cost"])
            assert a["step_is_successful"] == g["step_is_successful"] and a["step_is_valid"] == g["step_is_valid"], (tag, i)
            assert np.isclose(a["trust_region_radius"], g["trust_region_radius"], rtol=1e-6), (tag, i)
        else:
            assert dg <= 20.0 * floor, (tag, i, floor, dg)
    assert prefix >= 4, (tag, prefix, rows[:8])
    twins = [ref] + list(alts)
    fcs = np.array([t["final_cost"] for t in twins])
    fc_spread = float((fcs.max() - fcs.min()) / fcs.min())
    fc_near = float(np.min(np.abs(fcs - res["final_cost"]) / fcs))
    pose_spread = [0.0, 0.0]
    for i in range(len(twins)):
        for j in range(i + 1, len(twins)):
            r, t = _pose_rmse(twins[i]["cams"], twins[j]["cams"])
            pose_spread = [max(pose_spread[0], r), max(pose_spread[1], t)]
    pose_near = min((_pose_rmse(t["cams"], res["cams"]) for t in twins), key=lambda rt: rt[0] / max(pose_spread[0], 1e-30) + rt[1] / max(pose_spread[1], 1e-30))
    print("%s: iterations of the %d CPU twins %s / engine %d; deterministic prefix %d iterations; final cost: engine %.8e, twins %s "
          "(spread %.3e, engine to nearest twin %.3e); pose RMSE engine to nearest twin rot %.3e rad trans %.3e m (twins' spread %.3e / %.3e)"
          % (tag, len(twins), [len(t["iterations"]) - 1 for t in twins], len(gi) - 1, prefix, res["final_cost"],
             ["%.8e" % f for f in fcs], fc_spread, fc_near, pose_near[0], pose_near[1], pose_spread[0], pose_spread[1]))
    assert res["termination_type"] == ref["termination_type"], (res["message"], ref["message"])
    assert fc_near <= 2.0 * fc_spread + 1e-9 or res["final_cost"] <= fcs.min()
    assert pose_near[0] <= 2.0 * pose_spread[0] + 1e-5 and pose_near[1] <= 2.0 * pose_spread[1] + 1e-5
    return prefix


@pytest.mark.timeout(1800)
def test_configs1_parity_to_convergence(full_window):
    """configs[1] at full size with the reference's solver options (tolerances on, photobundle.cc:738-761), run until a
    tolerance terminates the solve (~80-100 iterations), against the dual-number oracle; see _noise_floor_parity."""
    from oracle import oracle
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    p = full_window
    ref = oracle.solve(p, oracle.default_options(num_threads=8, use_autodiff=1))
    o0 = oracle.default_options(num_threads=8, use_autodiff=0)
    alts = [oracle.solve(p, o0), oracle.solve(p, o0, xyz=np.nextafter(p.xyz, np.inf)), oracle.solve(p, o0, xyz=np.nextafter(p.xyz, -np.inf))]
    assert ref["termination_type"] == 0 and len(ref["iterations"]) >= 10, ref["message"]
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options())
    _noise_floor_parity(ref, alts, res, "configs[1] to convergence")


@pytest.mark.timeout(1800)
def test_configs1_well_initialised_window_to_convergence():
    """Same shape, the "good VO" regime (small pose / depth perturbation): the three solves stay together for ~40
    iterations, so 30 iterations are compared at the tight tolerances, then the run to convergence as above."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    p = synthetic.make_window(n_frames=8, n_points=50000, radius=2, rot_deg=0.02, trans=0.003, depth_noise=0.002)
    ref = oracle.solve(p, oracle.default_options(num_threads=8, use_autodiff=1))
    o0 = oracle.default_options(num_threads=8, use_autodiff=0)
    alts = [oracle.solve(p, o0), oracle.solve(p, o0, xyz=np.nextafter(p.xyz, np.inf)), oracle.solve(p, o0, xyz=np.nextafter(p.xyz, -np.inf))]
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options())
        prefix = _noise_floor_parity(ref, alts, res, "configs[1], well initialised, to convergence")
        n_it = min(30, prefix - 1)
        assert n_it >= 10
        e.load(p)
        res30 = e.solve(default_solver_options(max_num_iterations=n_it))
    ref30 = oracle.solve(p, oracle.default_options(num_threads=8, use_autodiff=1, max_num_iterations=n_it))
    rr, rt = _trace_parity(ref30, res30)
    print("configs[1], well initialised, %d iterations: pose RMSE rot %.3e rad, trans %.3e m" % (n_it, rr, rt))


@pytest.mark.timeout(2400)
def test_configs4_ten_iterations_against_oracle():
    """configs[4] shape (8 frames, 50k points, 11x11, Huber 0.05): 10 LM iterations.  The second oracle run that fixes
    the noise floor sees the same inputs with the points moved by one ulp."""
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    p = synthetic.make_window(n_frames=8, n_points=50000, radius=5, huber=0.05)
    n_it = 10
    o = oracle.default_options(max_num_iterations=n_it, num_threads=8, use_autodiff=0)
    ref = oracle.solve(p, o)
    alts = [oracle.solve(p, o, xyz=np.nextafter(p.xyz, np.inf)), oracle.solve(p, o, xyz=np.nextafter(p.xyz, -np.inf))]
    with make_engine(p, keep_reduced_system=False) as e:
        res = e.solve(default_solver_options(max_num_iterations=n_it))
    assert len(ref["iterations"]) == n_it + 1 == len(res["iterations"])
    _noise_floor_parity(ref, alts, res, "configs[4] 10 iterations")
    for a, b in zip(ref["iterations"], res["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"]


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
