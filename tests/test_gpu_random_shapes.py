"""Seeded random sweep over the shapes of the hot path through the C-ABI: window length 3..16, patch radius 1..5,
Huber on/off, Gaussian weights, image size, number of points, dense / causal / ragged visibility, strength of the pose
perturbation (strong ones push patches over the image border = the clamped sampler path, sample_eigen.h:38-51).  Every
case checks (a) all per-observation records of the Jacobian pass against the oracle's dual-number rows, (b) the first LM
iterations (cost, accept / reject, radius) against the oracle's trust-region loop, (c) the camera parameters after
those iterations.  The shapes are drawn once from a fixed seed; the list is printed so that a failure can be replayed."""
import copy
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle
from photobundle_amd import synthetic
from photobundle_amd.engine import default_solver_options

from gpu_util import trajectory_consistency, check_obs_records, make_engine, step_accuracy

pytestmark = pytest.mark.gpu


def _draw_cases(n, seed=20260928):
    rng = np.random.default_rng(seed)
    rng_ch = np.random.default_rng(seed + 1)      # separate stream: the shapes of the round-2 sweep stay what they were
    cases = []
    for k in range(n):
        rows = int(rng.integers(72, 160))
        cols = int(rng.integers(120, 260))
        f = float(rng.uniform(180.0, 320.0))
        cases.append(dict(
            n_frames=int(rng.integers(3, 17)), radius=int(rng.integers(1, 6)), n_points=int(rng.integers(40, 700)),
            size=(rows, cols), K=(f, f * float(rng.uniform(0.95, 1.05)), cols / 2.0 + float(rng.uniform(-8, 8)), rows / 2.0 + float(rng.uniform(-5, 5))),
            visibility=str(rng.choice(["dense", "causal"])), ragged=bool(rng.random() < 0.4),
            huber=float(rng.choice([0.0, 0.0, 0.05, 0.5, 5.0])), gaussian=bool(rng.random() < 0.3),
            rot_deg=float(rng.choice([0.05, 0.1, 0.5, 2.0])), trans=float(rng.choice([0.01, 0.02, 0.1, 0.4])),
            seed_offset=100 + k))
        # descriptor channels of the residual blocks (reference Options::descriptorType): mostly Intensity, some
        # IntensityAndGradient (3) / BitPlanes (8) -- the fused multi-channel kernel with the asynchronous driver
        cases[-1]["channels"] = int(rng_ch.choice([1, 1, 1, 1, 3, 8]))
    return cases


CASES = _draw_cases(int(os.environ.get("PBA_RANDOM_CASES", "24")))      # the first 24 of the seeded sequence by default; more on request
if os.environ.get("PBA_RANDOM_FIRST"):      # replay aid: skip the head of the sequence (the rarity check below then skips itself)
    CASES = CASES[int(os.environ["PBA_RANDOM_FIRST"]):]


def _make(c):
    ch_fn = synthetic.channel_fn({3: "IntensityAndGradient", 8: "BitPlanes"}[c["channels"]]) if c.get("channels", 1) > 1 else None
    p = synthetic.make_window(n_frames=c["n_frames"], n_points=c["n_points"], radius=c["radius"], size=c["size"], K=c["K"],
                              visibility=c["visibility"], huber=c["huber"], gaussian=c["gaussian"], rot_deg=c["rot_deg"],
                              trans=c["trans"], seed_offset=c["seed_offset"], channel_fn=ch_fn)
    if c["ragged"]:
        rng = np.random.default_rng(c["seed_offset"])
        begin = np.searchsorted(p.obs_point, np.arange(p.n_points + 1))
        keep = np.zeros(p.n_obs, bool)
        for pt in range(p.n_points):
            n = begin[pt + 1] - begin[pt]
            keep[begin[pt] + rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False)] = True
        q = copy.copy(p)
        q.obs_point, q.obs_slot = p.obs_point[keep].copy(), p.obs_slot[keep].copy()
        p = q
    return p


ARBITRATED = []      # seed offsets of the cases of this session that needed the step arbiter
RAN = []             # seed offsets of the cases that ran to the end in THIS process (the rarity check below needs all of them)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", CASES, ids=["%02d-f%d-r%d-c%d-%s%s" % (c["seed_offset"] - 100, c["n_frames"], c["radius"], c["channels"], c["visibility"], "-ragged" if c["ragged"] else "")
                                             for c in CASES])
def test_random_shape(case):
    print("case:", case)
    p = _make(case)
    iterations = 5
    lin = oracle.linearize(p, blocks=False)
    with make_engine(p) as e:
        cost = e.linearize()
        rec = e.obs_records()
        assert np.isclose(cost, lin["cost"], rtol=1e-12)
        s = lin["block_sqnorm"]
        a = case["huber"]
        rho = s if a <= 0.0 else np.where(s <= a * a, s, 2.0 * a * np.sqrt(s) - a * a)     # ceres::HuberLoss
        assert np.allclose(rec[:, 5], 0.5 * rho, rtol=1e-12, atol=0.0)
        worst = check_obs_records(p, rec)
        res = e.solve(default_solver_options(max_num_iterations=iterations))
        # whatever path the loop took, the objective at the engine's own states is the reference's (oracle cost there)
        consistency = trajectory_consistency(p, e, [2, iterations], lambda k: default_solver_options(max_num_iterations=k), rtol=1e-11)
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=iterations, use_autodiff=1))
    alt = oracle.solve(p, oracle.default_options(max_num_iterations=iterations, use_autodiff=0), xyz=np.nextafter(p.xyz, np.inf))
    # Ill-conditioned little windows amplify rounding from the first iterations on (see _noise_floor_parity in
    # test_gpu_fullsize.py), and here in discrete jumps: the sampler rounds the projection to float
    # (sample_eigen.h:117-118), so a different summation order stays invisible until one observation's (float)u crosses
    # a rounding boundary -- then the cost moves by ~1e-9..1e-7 relative at once.  The oracle against itself with the
    # points moved by ONE ulp (analytic instead of dual-number Jacobian) shows the same jumps at other iterations.
    # Bar: 1e-9 for the first two steps, then 1e-5, or 20x / 50x the twin's distance where that is larger; identical accept /
    # reject decisions.  One flipped coordinate is worth ~4e-10 of the cost at these sizes (more at 11x11 patches), so now and
    # then the engine is a few flips away from the oracle before the one-ulp twin is: such a case must then be explained by the
    # extended-precision step arbiter (below), and is reported.
    floor, rows = 0.0, []
    for i, (a, b, g) in enumerate(zip(ref["iterations"], alt["iterations"], res["iterations"])):
        floor = max(floor, abs(a["cost"] - b["cost"]) / a["cost"])
        dg = abs(a["cost"] - g["cost"]) / a["cost"]
        rows.append((i, floor, dg, a["step_is_successful"], g["step_is_successful"]))
    print("rows (iteration, ref-twin distance, engine-ref distance, accepted ref / engine):", rows)
    assert len(ref["iterations"]) == len(res["iterations"]), (ref["message"], res["message"])
    def violations(rows_):
        bad = []
        for i, fl, dg, sa, sg in rows_:
            assert sa == sg, rows_
            # 1e-9 is the bar of the early iterations, 1e-5 of the later ones, or 20x / 50x the noise floor where that is larger
            if dg > (max(1e-9, 20.0 * fl) if i <= 2 else max(1e-5, 50.0 * fl)):
                bad.append((i, dg, fl))
        return bad

    bad = violations(rows)
    explained = False

    def arbiter(n_it):
        acc = step_accuracy(p, n_it)
        for r in acc:
            # backward error of the engine's step in the full normal equations (x87 extended-precision residual on the engine's own
            # records, normwise within the camera rows and within the point rows) against that of the float64 restatement of the same
            # elimination: over 130 (window, iteration) samples of the sweep the two track each other -- engine 1e-17 .. 2.3e-11, float64
            # 1e-17 .. 4.7e-11, ratio <= 4.2 (profiles/r04/random_sweep_r3_vs_r4.txt); a 1e-6 slip of the damping shows as 5e-11 where
            # the restatement sits at 1e-15 (tests/test_step_arbiter_cpu.py)
            # (r6: against the BAND of five double-precision evaluations, like the forward error below -- one draw of the restatement's own
            # backward error moves by 20x with the last bits of the records: gpu_util.step_accuracy)
            assert r["bwd_engine"] <= 10.0 * r["bwd_f64_band"] + 1e-14, r
            # ... and its FORWARD error (camera step against the extended-precision step of the same records) stays within a stated
            # factor of the float64 restatement's at every arbitrated iteration: the pose bar that remains when the oracle-twin bars
            # are left (VERDICT r4 #4b) -- a step that is farther from the exact one than any double algorithm would be is a defect
            # (the band of an ENSEMBLE of five double-precision evaluations -- one draw alone is anything between 0 and the
            # conditioning: gpu_util.step_accuracy)
            assert r["fwd_engine"] <= 10.0 * r["fwd_f64_band"] + 1e-13, r
        return acc

    if bad:
        # The one-ulp twin of the ORACLE has not moved where the engine has: the twin runs the oracle's own solver code, so its rounding
        # is strongly correlated with the oracle's and it under-states the band (round 3: 1 such case in 200; the round-4 build 3 in 160
        # with these bars; the round-3 build on the same box sits on the same side of the bars in one of them and inside in the others:
        # profiles/r04/random_sweep_r3_vs_r4.txt).  Independent arbiter (gpu_util.step_accuracy): at the engine's own states its (camera,
        # point) step must solve the full normal equations of its own Jacobian-pass records -- which check_obs_records pins to the
        # oracle's rows -- with the backward error (x87 extended-precision residual) of a float64 restatement of the same
        # elimination.  The FORWARD error of any double-precision step on these windows is 1e-10 .. 1e-5 of the step (3x3 point
        # blocks at 0.01 m baselines; printed next to the float64 restatement's), which is 1e-8 pixels and flips a float-rounded patch
        # position now and then: what separates such a trajectory from the oracle's is amplification, not a wrong step.  The objective
        # itself is pinned by the consistency check above (oracle cost at the engine's own states, 1e-11).  Every use is reported.
        assert consistency <= 1e-11
        ARBITRATED.append(case["seed_offset"])
        # hard ceilings that no arbitration lifts: costs within 1e-4 of the oracle's after five iterations, 1e-2 on the
        # windows perturbed by 2 degrees / 0.4 m (far outside the north_star's regime; window 157 of the 240-case sweep: cond(S) 4e6,
        # 2.9e-5 at the fourth and 5.7e-4 at the fifth iteration with backward errors at the float64 restatement's -- the round-3 build,
        # whose steps have the same forward errors, happened to stay at 2e-8: profiles/r04/random_sweep_r3_vs_r4.txt)
        hard_case = case["rot_deg"] >= 2.0 or case["trans"] >= 0.4
        assert max(dg for _, dg, _ in bad) <= (1e-2 if hard_case else 1e-4), bad
        print("STEP ARBITER used by case %s: violations of the oracle-twin bars %s; per iteration: %s" % (case, bad, arbiter(iterations)))
        explained = True
    elif case["seed_offset"] % 8 == 3:
        arbiter(3)       # (the arbiter itself stays exercised on windows that pass)
    print("oracle cost at the engine's own states: largest relative difference %.1e" % consistency)
    cam_floor = np.abs(alt["cams"] - ref["cams"]).max()
    # the 2-degree / 0.4 m perturbations on these little images are far outside the north_star's regime: five iterations
    # amplify rounding to 1e-5 .. 1e-4 in the poses there (the one-ulp twin shows the same), so the bar follows the twin
    hard = case["rot_deg"] >= 2.0 or case["trans"] >= 0.4
    cam_dist = np.abs(res["cams"] - ref["cams"]).max()
    bar = (50.0 * cam_floor + 3e-4) if hard else (3.0 * cam_floor + 1e-5)
    if explained:
        # costs that have separated by 1e-8 .. 1e-5 put the cameras on different branches of the amplification: the twin envelope no
        # longer applies, a ceiling of 10x the plain bar does (the per-iteration forward-error bar of the arbiter is the sharp one)
        print("camera distance to the oracle after %d iterations: %.2e (one-ulp twin of the oracle: %.2e, ceiling %.2e)" % (iterations, cam_dist, cam_floor, 10.0 * bar))
        assert cam_dist <= 10.0 * bar
    else:
        assert cam_dist <= bar
    print("worst record error:", worst)
    RAN.append(case["seed_offset"])


def test_arbitrated_cases_stay_rare():
    """The arbiter is for the odd window whose one-ulp twin under-states the rounding band (3 of 240 on the round-4 build): a build
    that needs it for more than 1 case in 40 (and more than one case at all) has a precision problem, whatever the arbiter says."""
    print("cases that needed the step arbiter: %s of %d (%d ran in this process)" % (ARBITRATED, len(CASES), len(RAN)))
    # under pytest-xdist, -k selections or reordering this process has seen only some of the cases: a ceiling on a subset would pass
    # vacuously, so the check says so instead
    if sorted(RAN) != sorted(c["seed_offset"] for c in CASES):
        pytest.skip("only %d of %d cases ran in this process: the rarity ceiling needs the whole sweep" % (len(RAN), len(CASES)))
    assert len(ARBITRATED) <= max(1, len(CASES) // 40), ARBITRATED
