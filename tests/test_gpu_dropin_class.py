"""-m gpu: the C++ drop-in class (photobundle_amd/host, reference API of src/photobundle.h) driven exactly like the
reference's apps/run_kitti.cc, checked against a numpy restatement of the front-end + the CPU oracle solve."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "photobundle_amd", "bin", "run_kitti")


def _write_sequence(tmp, n_frames, size, K):
    from photobundle_amd import synthetic
    tex = synthetic.Texture()
    T_gt = synthetic.make_trajectory(n_frames)
    local, _ = synthetic.perturb_local_poses(T_gt, rot_deg=0.03, trans=0.005)
    imgs, depths = [], []
    for i, T in enumerate(T_gt):
        im, z = synthetic.render_frame(T, K, size, tex)
        z = np.where(np.isfinite(z), z, -1.0).astype(np.float32)
        imgs.append(im)
        depths.append(z)
        with open(os.path.join(tmp, "image_%06d.pgm" % i), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (size[1], size[0]))
            f.write(im.tobytes())
        z.tofile(os.path.join(tmp, "depth_%06d.bin" % i))
    with open(os.path.join(tmp, "calib.txt"), "w") as f:
        f.write("%r %r %r %r 0.5372\n" % tuple(K))
    with open(os.path.join(tmp, "init.txt"), "w") as f:
        for T in local:
            f.write(" ".join("%.17g" % v for v in T[:3, :].reshape(-1)) + "\n")
    return imgs, depths, local


def _read_results(path):
    """Parses run_kitti -r: one record per Result the class returned."""
    out, cur = [], None
    for line in open(path):
        t = line.split()
        if t[0] == "result":
            cur = dict(frame=int(t[2]), n_poses=int(t[4]), it=[], refined=[], original=[])
            out.append(cur)
        elif t[0] == "cost":
            cur.update(initial=float(t[1]), final=float(t[2]), fixed=float(t[3]), steps=int(t[5]), residuals=int(t[7]))
        elif t[0] == "message":
            cur["message"] = line[len("message "):].rstrip("\n")
        elif t[0] == "it":
            cur["it"].append([int(t[1]), int(t[2]), int(t[3])] + [float(v) for v in t[4:]])
        elif t[0] == "pt":
            v = [float(x) for x in t[1:]]
            cur["refined"].append(v[:3])
            cur["original"].append(v[3:])
    for r in out:
        r["refined"] = np.array(r["refined"]).reshape(-1, 3)
        r["original"] = np.array(r["original"]).reshape(-1, 3)
    return out


@pytest.mark.timeout(900)
def test_run_kitti_matches_emulated_reference_pipeline(tmp_path):
    from oracle.frontend import Emulator
    assert os.path.exists(RUN), "build photobundle_amd/bin/run_kitti first (__graft_entry__.build())"
    size, K = (120, 160), (200.0, 200.0, 80.0, 60.0)
    n_frames, window, radius, max_points = 6, 4, 1, 4096  # no top-N cut: nth_element is unspecified among saliency ties
    tmp = str(tmp_path)
    imgs, depths, local = _write_sequence(tmp, n_frames, size, K)
    cfg = os.path.join(tmp, "test.cfg")
    with open(cfg, "w") as f:
        f.write("# reference config keys (config/kitti_stereo.cfg) + DataDirectory\n")
        f.write("DataDirectory = %s\nTrajectory = %s/init.txt\n" % (tmp, tmp))
        f.write("maxNumPoints = %d\nslidingWindowSize = %d\npatchRadius = %d\nminScore = 0.65\nrobustThreshold = 0.05\nverbose = 0\n"
                % (max_points, window, radius))
    out = os.path.join(tmp, "refined.txt")
    dump = os.path.join(tmp, "results.txt")
    r = subprocess.run([RUN, "-c", cfg, "-o", out, "-r", dump, "-p"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    refined = np.loadtxt(out).reshape(-1, 3, 4)
    assert refined.shape[0] == n_frames

    emu = Emulator(K, size, window, radius, max_points, min_score=0.65, huber=0.05)
    for im, z, T in zip(imgs, depths, local):
        emu.add_frame(im, z, T)
    assert len(emu.results) == n_frames - window + 1 and all(r_["n_points"] > 20 for r_ in emu.results)
    ref = np.stack([T[:3, :] for T in emu.T_w])
    # the class ran the same windows: "Using N points (M residual blocks)" lines must agree with the emulation
    import re
    used = [tuple(int(t) for t in m.groups()) for m in re.finditer(r"Using (\d+) points \((\d+) residual blocks\)", r.stderr)]
    assert used == [(r_["n_points"], r_["n_obs"]) for r_ in emu.results], (used, emu.results)
    assert np.abs(refined - ref).max() <= 1e-5, np.abs(refined - ref).max()
    # and the optimisation did something: the refined trajectory differs from plain chaining of the initial poses
    from photobundle_amd import se3
    chained = np.stack([T[:3, :] for T in se3.chain_local_poses(local)])
    assert np.abs(refined - chained).max() > 1e-6
    # every field of every Result the class handed back (reference photobundle.cc:857-875) against the emulation
    got = _read_results(dump)
    assert [g["frame"] for g in got] == list(range(window - 1, n_frames))
    assert sum(len(g["refined"]) for g in got) > 0
    for g, e in zip(got, emu.results):
        assert g["n_poses"] == e["n_poses"]                                      # the whole trajectory so far (:858)
        assert g["residuals"] == e["num_residuals"] == e["n_obs"] * (2 * radius + 1) ** 2
        assert g["steps"] == e["num_successful_steps"]
        assert np.isclose(g["initial"], e["initial_cost"], rtol=1e-12) and np.isclose(g["final"], e["final_cost"], rtol=1e-9)
        assert g["fixed"] == 0.0
        assert g["message"] == e["message"] or g["message"].split(".")[0] == e["message"].split(".")[0], (g["message"], e["message"])
        assert len(g["it"]) == len(e["it"])
        for a, b in zip(g["it"], e["it"]):
            assert a[0] == b["iteration"] and a[1] == b["step_is_valid"] and a[2] == b["step_is_successful"]
            assert np.isclose(a[3], b["cost"], rtol=1e-9) and np.isclose(a[8], b["trust_region_radius"], rtol=1e-6)
            assert np.isclose(a[5], b["gradient_max_norm"], rtol=1e-6)
        assert g["refined"].shape == e["refined"].shape
        # originalPoints are the back-projections of addFrame (never touched by the solver), refinedPoints the solver's
        assert np.abs(g["original"] - e["original"]).max() <= 1e-12 * max(1.0, np.abs(e["original"]).max())
        assert np.abs(g["refined"] - e["refined"]).max() <= 1e-6 * max(1.0, np.abs(e["refined"]).max())
        moved = np.abs(g["refined"] - g["original"]).max(1) > 0
        assert moved.any()


@pytest.mark.timeout(1500)
def test_pyramid_path_matches_emulation(tmp_path):
    """configs[2] (photobundle_pyramid path): 2-level coarse-to-fine through run_kitti (numLevels = 2)."""
    from oracle.frontend import PyramidEmulator
    size, K = (160, 224), (280.0, 280.0, 112.0, 80.0)
    n_frames, window, radius, max_points = 5, 3, 1, 100000
    tmp = str(tmp_path)
    imgs, depths, local = _write_sequence(tmp, n_frames, size, K)
    cfg = os.path.join(tmp, "pyr.cfg")
    with open(cfg, "w") as f:
        f.write("DataDirectory = %s\nTrajectory = %s/init.txt\n" % (tmp, tmp))
        f.write("numLevels = 2\nmaxNumPoints = %d\nslidingWindowSize = %d\npatchRadius = %d\nminScore = 0.65\nrobustThreshold = 0.05\nverbose = 0\n"
                % (max_points, window, radius))
    out = os.path.join(tmp, "refined.txt")
    r = subprocess.run([RUN, "-c", cfg, "-o", out, "-p"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("Pyramid level 1") == n_frames and r.stderr.count("Pyramid level 0") == n_frames
    refined = np.loadtxt(out).reshape(-1, 3, 4)
    emu = PyramidEmulator(2, K, size, window=window, radius=radius, max_points=max_points, min_score=0.65, huber=0.05)
    for im, z, T in zip(imgs, depths, local):
        fine = emu.add_frame(im, z, T)
    assert len(fine.results) == n_frames - window + 1
    ref = np.stack([T[:3, :] for T in fine.T_w])
    assert np.abs(refined - ref).max() <= 1e-5, np.abs(refined - ref).max()


@pytest.mark.timeout(1500)
def test_pyramid_three_levels_matches_emulation(tmp_path):
    """configs[2] path with numLevels = 3 (192x256 -> 96x128 -> 48x64) against the numpy emulation + oracle solves."""
    from oracle.frontend import PyramidEmulator
    size, K = (192, 256), (320.0, 320.0, 128.0, 96.0)
    n_frames, window, radius, max_points = 5, 3, 1, 100000
    tmp = str(tmp_path)
    imgs, depths, local = _write_sequence(tmp, n_frames, size, K)
    cfg = os.path.join(tmp, "pyr3.cfg")
    with open(cfg, "w") as f:
        f.write("DataDirectory = %s\nTrajectory = %s/init.txt\n" % (tmp, tmp))
        f.write("numLevels = 3\nmaxNumPoints = %d\nslidingWindowSize = %d\npatchRadius = %d\nminScore = 0.65\nrobustThreshold = 0.05\nverbose = 0\n"
                % (max_points, window, radius))
    out = os.path.join(tmp, "refined.txt")
    r = subprocess.run([RUN, "-c", cfg, "-o", out, "-p"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    for lvl in range(3):
        assert r.stderr.count("Pyramid level %d" % lvl) == n_frames
    refined = np.loadtxt(out).reshape(-1, 3, 4)
    emu = PyramidEmulator(3, K, size, window=window, radius=radius, max_points=max_points, min_score=0.65, huber=0.05)
    for im, z, T in zip(imgs, depths, local):
        fine = emu.add_frame(im, z, T)
    assert len(fine.results) == n_frames - window + 1
    assert all(len(e_.results) == n_frames - window + 1 and all(r_["n_points"] > 10 for r_ in e_.results) for e_ in emu.emus)
    ref = np.stack([T[:3, :] for T in fine.T_w])
    assert np.abs(refined - ref).max() <= 1e-5, np.abs(refined - ref).max()


def _pose_errors(T, T_gt):
    rot, tr = [], []
    for a, b in zip(T, T_gt):
        d = np.linalg.inv(b) @ np.vstack([a, [0, 0, 0, 1]])
        rot.append(np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)))
        tr.append(np.linalg.norm(d[:3, 3]))
    return np.array(rot), np.array(tr)


@pytest.mark.timeout(2400)
def test_configs2_full_size_three_level_pyramid(tmp_path):
    """BASELINE configs[2] at its stated shape: 3-level pyramid (1241x376, 621x188, 311x94), 8-frame window, ~50k points
    in the finest window, through PhotometricBundleAdjustmentPyr / run_kitti.  The emulator cannot cover this size, so
    the checks are size-independent properties: every level ran for every frame, each Result's cost went down, the
    refined trajectory is closer to the ground truth than the chained initial poses, and a second run reproduces the
    output bit for bit."""
    from photobundle_amd import se3, synthetic
    size, K = synthetic.KITTI_SIZE, synthetic.KITTI_K
    n_frames, window, radius = 10, 8, 2
    tmp = str(tmp_path)
    imgs, depths, local = _write_sequence(tmp, n_frames, size, K)
    T_gt = synthetic.make_trajectory(n_frames)
    cfg = os.path.join(tmp, "cfg2.cfg")
    with open(cfg, "w") as f:
        f.write("DataDirectory = %s\nTrajectory = %s/init.txt\n" % (tmp, tmp))
        # the synthetic scene has only ~12k saliency maxima per frame; without the suppression (nonMaxSuppRadius = 0) the
        # 16000 most salient valid pixels of every frame become points, which fills the finest 8-frame window to ~50k
        f.write("numLevels = 3\nmaxNumPoints = 16000\nnonMaxSuppRadius = 0\nslidingWindowSize = %d\npatchRadius = %d\nminScore = 0.75\n"
                "robustThreshold = 0.05\nverbose = 0\n" % (window, radius))
    outs = []
    for k in range(2):
        out = os.path.join(tmp, "refined%d.txt" % k)
        dump = os.path.join(tmp, "results%d.txt" % k)
        r = subprocess.run([RUN, "-c", cfg, "-o", out, "-r", dump, "-p"], capture_output=True, text=True, timeout=1000)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((open(out).read(), open(dump).read(), r.stderr))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]          # run-to-run determinism of the whole pipeline
    err = outs[0][2]
    for lvl in range(3):
        assert err.count("Pyramid level %d" % lvl) == n_frames
    import re
    used = [tuple(int(t) for t in m.groups()) for m in re.finditer(r"Using (\d+) points \((\d+) residual blocks\)", err)]
    assert len(used) == 3 * (n_frames - window + 1)
    finest = max(u[0] for u in used[2::3])       # every third line is the finest level (coarse -> fine per frame)
    print("configs[2]: windows (points, residual blocks) per optimisation:", used)
    assert finest >= 40000, used                                              # the "50k points" window of configs[2]
    got = _read_results(os.path.join(tmp, "results0.txt"))
    assert len(got) == n_frames - window + 1
    for g in got:
        assert g["final"] < g["initial"] and g["steps"] >= 2 and g["n_poses"] == g["frame"] + 1
        assert g["residuals"] % ((2 * radius + 1) ** 2) == 0
    refined = np.loadtxt(os.path.join(tmp, "refined0.txt")).reshape(-1, 3, 4)
    chained = np.stack([T[:3, :] for T in se3.chain_local_poses(local)])
    r_ref, t_ref = _pose_errors(refined, T_gt)
    r_ini, t_ini = _pose_errors(chained, T_gt)
    print("configs[2]: mean rotation error %.3e -> %.3e rad, mean translation error %.3e -> %.3e m" % (r_ini.mean(), r_ref.mean(), t_ini.mean(), t_ref.mean()))
    assert r_ref.mean() < 0.5 * r_ini.mean() and t_ref.mean() < 0.5 * t_ini.mean()


@pytest.mark.timeout(2400)
def test_configs2_full_size_windows_replayed_through_the_oracle(tmp_path):
    """BASELINE configs[2] at its stated shape AGAINST THE ORACLE: the windows the three-level pyramid class hands to its
    engines (PBA_DUMP_WINDOWS: cameras, points, descriptors, observation lists exactly as passed to pba_set_problem / pba_set_cameras) are
    solved again by the CPU oracle on the images of their level (cv::pyrDown restatement, calibration halved per level: reference
    src/photobundle_pyramid.cc:9-69) -- the coarsest (311x94) and the finest (1241x376, ~50k points, ~400k residual blocks) window of two
    frames, four iterations each as tests/test_gpu_fullsize.py does at configs[1] / configs[3]: identical decisions, costs at 1e-9,
    cameras at 1e-5 for the engine (same window through the C-ABI), and the finest level's own Result (the class's iteration log as
    run_kitti -r wrote it) carries the oracle's costs."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import oracle
    from oracle.frontend import pyr_down_u8
    from photobundle_amd import imgproc, synthetic
    from photobundle_amd.engine import default_solver_options
    from photobundle_amd.problem import WindowProblem
    from gpu_util import make_engine
    from test_gpu_configs0 import _read_window
    size, K = synthetic.KITTI_SIZE, synthetic.KITTI_K
    n_frames, window, radius, n_it = 9, 8, 2, 4
    tmp = str(tmp_path)
    imgs, depths, local = _write_sequence(tmp, n_frames, size, K)
    cfg = os.path.join(tmp, "cfg2.cfg")
    with open(cfg, "w") as f:
        f.write("DataDirectory = %s\nTrajectory = %s/init.txt\n" % (tmp, tmp))
        f.write("numLevels = 3\nmaxNumPoints = 16000\nnonMaxSuppRadius = 0\nslidingWindowSize = %d\npatchRadius = %d\nminScore = 0.75\n"
                "robustThreshold = 0.05\nverbose = 0\n" % (window, radius))
    dump_dir = os.path.join(tmp, "windows")
    os.makedirs(dump_dir)
    res_file = os.path.join(tmp, "results.txt")
    r = subprocess.run([RUN, "-c", cfg, "-o", os.path.join(tmp, "refined.txt"), "-r", res_file, "-p"], capture_output=True, text=True, timeout=1000,
                       env=dict(os.environ, PBA_DUMP_WINDOWS=dump_dir, PBA_DUMP_TAG_SIZE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    levels = [(size[0], size[1])]
    for _ in range(2):
        levels.append(((levels[-1][0] + 1) // 2, (levels[-1][1] + 1) // 2))
    assert levels == [(376, 1241), (188, 621), (94, 311)]
    names = sorted(os.listdir(dump_dir))
    assert names == sorted("window_%06d_%dx%d.bin" % (fid, c_, r_) for fid in range(window - 1, n_frames) for (r_, c_) in levels), names
    results = {g["frame"]: g for g in _read_results(res_file)}
    # image pyramid of every frame (cv::pyrDown on u8, oracle/frontend.py) and its planes
    pyr = [[im] for im in imgs]
    for p_ in pyr:
        for _ in range(2):
            p_.append(pyr_down_u8(p_[-1]))
    for fid in (window - 1, n_frames - 1):
        for lvl in (2, 0):                      # coarsest, finest
            rows, cols = levels[lvl]
            w = _read_window(os.path.join(dump_dir, "window_%06d_%dx%d.bin" % (fid, cols, rows)))
            assert (w["window"], w["radius"], w["id_end"], w["id_start"]) == (window, radius, fid, fid - window + 1)
            Kl = tuple(v * 0.5 ** lvl for v in K)                                 # Calibration::pyrDown: K * 0.5 per level
            frames = [pyr[w["id_start"] + ((s_ - w["id_start"]) % window)][lvl] for s_ in range(window)]      # slot = id % window
            assert frames[0].shape == (rows, cols)
            planes = np.stack([imgproc.planes_from_u8(im) for im in frames])
            fixed = w["first_slot"]
            p = WindowProblem(K=Kl, radius=radius, planes=planes, cams=w["cams"], xyz=w["xyz"], desc=w["desc"], obs_point=w["obs_point"],
                              obs_slot=w["obs_slot"], weights=w["weights"], huber=w["huber"], fixed_slot=fixed, images=np.stack(frames))
            if lvl == 0:
                assert p.n_points >= 40000 and p.n_obs >= 200000, (p.n_points, p.n_obs)      # the "50k points" window of configs[2] (causal visibility: 4-5 blocks per point)
            ref = oracle.solve(p, oracle.default_options(max_num_iterations=n_it, use_autodiff=0))
            with make_engine(p, keep_reduced_system=False) as e:
                res = e.solve(default_solver_options(max_num_iterations=n_it))
            assert len(res["iterations"]) == len(ref["iterations"]) == n_it + 1
            worst = 0.0
            for a, b in zip(ref["iterations"], res["iterations"]):
                assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"]
                assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(a["cost"]), (fid, lvl, a["cost"], b["cost"])
                worst = max(worst, abs(a["cost"] - b["cost"]) / abs(a["cost"]))
            cam_err = float(np.abs(res["cams"] - ref["cams"]).max())
            assert cam_err <= 1e-5, (fid, lvl, cam_err)
            print("configs[2] frame %d level %d (%dx%d, %d points, %d residual blocks): %d iterations, decisions identical, costs within %.1e, "
                  "cameras within %.1e of the oracle" % (fid, lvl, cols, rows, p.n_points, p.n_obs, n_it, worst, cam_err))
            if lvl == 0:
                # the class's own Result of this frame (finest level): same window, same engine -- its log must carry the oracle's costs too
                g = results[fid]
                assert g["residuals"] == p.n_obs * (2 * radius + 1) ** 2
                assert np.isclose(g["initial"], ref["initial_cost"], rtol=1e-12)
                for a, b in zip(ref["iterations"], g["it"][:n_it + 1]):
                    assert b[0] == a["iteration"] and b[2] == a["step_is_successful"]
                    assert np.isclose(b[3], a["cost"], rtol=1e-9), (fid, b[3], a["cost"])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("descriptor", ["IntensityAndGradient", "BitPlanes"])
def test_multichannel_descriptor_types_match_emulation(tmp_path, descriptor):
    """Options::descriptorType = IntensityAndGradient / BitPlanes (reference photobundle.cc:225-248) through the class and
    run_kitti: channel images, saliency over all channels, C patches per descriptor, C (2R+1)^2 residuals per block."""
    from oracle.frontend import Emulator
    size, K = (120, 160), (200.0, 200.0, 80.0, 60.0)
    n_frames, window, radius, max_points = 6, 4, 1, 4096
    tmp = str(tmp_path)
    imgs, depths, local = _write_sequence(tmp, n_frames, size, K)
    cfg = os.path.join(tmp, "mc.cfg")
    with open(cfg, "w") as f:
        f.write("DataDirectory = %s\nTrajectory = %s/init.txt\ndescriptorType = %s\n" % (tmp, tmp, descriptor))
        f.write("maxNumPoints = %d\nslidingWindowSize = %d\npatchRadius = %d\nminScore = 0.65\nrobustThreshold = 0.05\nverbose = 0\n"
                % (max_points, window, radius))
    out, dump = os.path.join(tmp, "refined.txt"), os.path.join(tmp, "results.txt")
    r = subprocess.run([RUN, "-c", cfg, "-o", out, "-r", dump, "-p"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    refined = np.loadtxt(out).reshape(-1, 3, 4)
    emu = Emulator(K, size, window, radius, max_points, min_score=0.65, huber=0.05, descriptor_type=descriptor)
    for im, z, T in zip(imgs, depths, local):
        emu.add_frame(im, z, T)
    import re
    used = [tuple(int(t) for t in m.groups()) for m in re.finditer(r"Using (\d+) points \((\d+) residual blocks\)", r.stderr)]
    assert used == [(r_["n_points"], r_["n_obs"]) for r_ in emu.results], (used, [(r_["n_points"], r_["n_obs"]) for r_ in emu.results])
    assert all(r_["n_points"] > 20 for r_ in emu.results)
    C = {"IntensityAndGradient": 3, "BitPlanes": 8}[descriptor]
    got = _read_results(dump)
    for g, e in zip(got, emu.results):
        assert g["residuals"] == e["num_residuals"] == e["n_obs"] * C * (2 * radius + 1) ** 2
        # (solves run to convergence: rounding differences grow along the LM path, final costs agree to ~1e-7)
        assert np.isclose(g["initial"], e["initial_cost"], rtol=1e-12) and np.isclose(g["final"], e["final_cost"], rtol=1e-6)
    ref = np.stack([T[:3, :] for T in emu.T_w])
    assert np.abs(refined - ref).max() <= 1e-5, np.abs(refined - ref).max()
