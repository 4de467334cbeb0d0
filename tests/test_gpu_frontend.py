"""-m gpu: the device front-end (pba_frontend_visibility / _candidates / _descriptors; reference src/photobundle.cc:505-603) against
the numpy restatement of the same reference lines (oracle/frontend.py: interp2 / ZnccPatch_ in float arithmetic, the
oracle's channel planes for the saliency, ExtractPatch) -- bit for bit -- and through the drop-in class: run_kitti with the device
front-end (default) and with the host one (PBA_HOST_FRONTEND=1) write byte-identical trajectories and Result dumps."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "photobundle_amd", "bin", "run_kitti")
KINDS = {"Intensity": 1, "IntensityAndGradient": 3, "BitPlanes": 8}


def _image(rng, size):
    r, c = size
    yy, xx = np.mgrid[0:r, 0:c]
    img = (127 + 60 * np.sin(xx / 6.0) * np.cos(yy / 4.0) + 40 * np.sin((xx + 2 * yy) / 9.0)).astype(np.float64)
    img += rng.normal(0, 6, size)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _engine(size, kind, radius=1):
    from photobundle_amd.engine import Engine
    return Engine(rows=size[0], cols=size[1], max_frames=2, radius=radius, K=(200.0, 200.0, size[1] / 2.0, size[0] / 2.0), channels=KINDS[kind])


@pytest.mark.parametrize("size", [(96, 131), (376, 1241)])
def test_zncc_visibility_matches_the_float_restatement(size):
    from oracle.frontend import Zncc
    rng = np.random.default_rng(size[0])
    img = _image(rng, size)
    n = 600 if size[0] > 200 else 300
    B = 2
    # stored patches: integer positions of a slightly different image (noise), tested at sub-pixel positions of `img`
    other = np.clip(img.astype(np.int32) + rng.integers(-9, 10, size), 0, 255).astype(np.uint8)
    x0 = rng.integers(B + 3, size[1] - B - 4, n)
    y0 = rng.integers(B + 3, size[0] - B - 4, n)
    stored = [Zncc(other, float(x), float(y)) for x, y in zip(x0, y0)]
    uv = np.stack([x0 + rng.uniform(-0.49, 0.49, n), y0 + rng.uniform(-0.49, 0.49, n)], 1)
    uv[:8] = np.stack([x0[:8], y0[:8]], 1).astype(np.float64)            # exact integer positions too
    uv[8, :] = (B + 0.2, B - 0.3)                                          # the patch reaches row -1: fill value
    uv[9, :] = (size[1] - B - 1.2, size[0] - B - 1.6)                      # far corner of the admissible region
    rc = np.stack([np.floor(uv[:, 1] + 0.5), np.floor(uv[:, 0] + 0.5)], 1).astype(np.int32)
    pats = np.array([np.concatenate([z.data, [z.norm]]) for z in stored], np.float32)
    want_score = np.array([stored[i].score(Zncc(img, uv[i, 0], uv[i, 1])) for i in range(n)])
    e = _engine(size, "Intensity")
    e.set_frame(1, img)
    # the float values themselves: patch, norm and score of every point
    probe = e.frontend_zncc_probe(uv, pats)
    new = [Zncc(img, uv[i, 0], uv[i, 1]) for i in range(n)]
    want = np.array([np.concatenate([z.data, [z.norm]]) for z in new], np.float32)
    bad = np.nonzero((probe[:, :26] != want).any(1))[0]
    assert len(bad) == 0, (len(bad), bad[:5], probe[bad[0], :26], want[bad[0]])
    assert np.array_equal(probe[:, 26], want_score.astype(np.float32)), np.abs(probe[:, 26] - want_score).max()
    for thr in (0.3, 0.65, 0.9):
        hit = e.frontend_visibility(uv, rc, pats, thr, 1)
        assert np.array_equal(hit.astype(bool), want_score > thr), (thr, int((hit.astype(bool) != (want_score > thr)).sum()))
    # thresholds placed exactly ON scores: `>` must come out false there (the float score itself is reproduced, not just its side)
    for i in range(0, n, 37):
        hit = e.frontend_visibility(uv[i:i + 1], rc[i:i + 1], pats[i:i + 1], float(want_score[i]), 1)
        assert hit[0] == 0
        hit = e.frontend_visibility(uv[i:i + 1], rc[i:i + 1], pats[i:i + 1], float(np.nextafter(np.float32(want_score[i]), np.float32(-2))), 1)
        assert hit[0] == (1 if want_score[i] > -1.0 else 0)
    e.close()


def _candidates_numpy(planes_mc, depth, mask, border, nms, dmin, dmax):
    rows, cols = depth.shape
    sal = (np.abs(planes_mc[1]) + np.abs(planes_mc[2])).astype(np.float32)
    for k in range(1, planes_mc.shape[0] // 3):
        sal = (sal + (np.abs(planes_mc[3 * k + 1]) + np.abs(planes_mc[3 * k + 2])).astype(np.float32)).astype(np.float32)
    ok = (depth.astype(np.float64) >= dmin) & (depth.astype(np.float64) <= dmax)
    if nms > 0:
        ok &= mask.astype(bool) & ~(sal < 0)
        pad = np.pad(sal, nms, constant_values=-np.inf)
        for dr in range(-nms, nms + 1):
            for dc in range(-nms, nms + 1):
                if dr or dc:
                    ok &= ~(pad[nms + dr:nms + dr + rows, nms + dc:nms + dc + cols] >= sal)
    region = np.zeros_like(ok)
    region[border:rows - border - 1, border:cols - border - 1] = True
    ys, xs = np.nonzero(ok & region)                                        # row-major
    return sal, xs, ys


@pytest.mark.parametrize("kind", ["Intensity", "IntensityAndGradient", "BitPlanes"])
@pytest.mark.parametrize("size,nms", [((96, 131), 1), ((376, 1241), 1), ((64, 80), 2), ((48, 64), 0)])
def test_candidates_and_descriptors_match_numpy(kind, size, nms):
    from oracle import oracle
    from photobundle_amd import imgproc
    rng = np.random.default_rng(7 * size[0] + nms)
    img = _image(rng, size)
    img[10:14, 20:40] = 200                                   # plateaus: ties must NOT be strict maxima
    depth = rng.uniform(0.5, 60.0, size).astype(np.float32)
    depth[rng.random(size) < 0.2] = -1.0                      # invalid depth
    e = _engine(size, kind, radius=2)
    if kind == "Intensity":
        e.set_frame(1, img)
    else:
        e.set_frame_descriptor(1, img, kind)
    channels = oracle.descriptor_channels(img, kind)
    planes = oracle.channel_planes(channels)
    border = 3
    # re-observed points: their blocks leave the mask
    n = 40
    rc = np.stack([rng.integers(border, size[0] - border - 1, n), rng.integers(border, size[1] - border - 1, n)], 1).astype(np.int32)
    uv = rc[:, ::-1].astype(np.float64)
    z = np.zeros((n, 26), np.float32)
    z[:, 25] = 1.0
    z[:, 0] = 1.0
    hit = e.frontend_visibility(uv, rc, z, -2.0, 1)           # every point "hits" (score > -2)
    assert hit.all()
    mask = np.ones(size, np.uint8)
    for r, c in rc:
        mask[r - 1:r + 2, c - 1:c + 2] = 0
    got = e.frontend_candidates(1, depth, 1.0, 50.0, nms, border)
    sal, xs, ys = _candidates_numpy(planes, depth, mask, border, nms, 1.0, 50.0)
    assert len(got) == len(xs) and len(xs) > 10
    assert np.array_equal(got["x"], xs) and np.array_equal(got["y"], ys)
    assert np.array_equal(got["saliency"], sal[ys, xs])
    # the mask is per frame: a scan without a preceding visibility call sees none
    again = e.frontend_candidates(1, depth, 1.0, 50.0, nms, border)
    _, xs2, ys2 = _candidates_numpy(planes, depth, np.ones(size, np.uint8), border, nms, 1.0, 50.0)
    assert len(again) == len(xs2) and (nms == 0 or len(xs2) >= len(xs))
    # descriptors at the selected pixels (+ pixels at the very border: clamped indices)
    sel = np.stack([xs[::7], ys[::7]], 1)
    sel = np.concatenate([sel, [[0, 0], [size[1] - 1, size[0] - 1], [1, size[0] - 2]]]).astype(np.int32)
    d = e.frontend_descriptors(1, sel)
    want = np.stack([imgproc.extract_patches(channels[k], sel, 2) for k in range(channels.shape[0])], 1)
    assert np.array_equal(d.astype(np.float64), want)
    e.close()


def _run(cfg, out, res, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([RUN, "-c", cfg, "-o", out, "-p", "-r", res], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read(), open(res).read(), r.stderr


@pytest.mark.timeout(1500)
def test_class_output_is_identical_with_the_host_front_end(tmp_path):
    from test_gpu_dropin_class import _write_sequence
    size, K = (160, 224), (280.0, 280.0, 112.0, 80.0)
    tmp = str(tmp_path)
    _write_sequence(tmp, 6, size, K)
    base = "DataDirectory = %s\nTrajectory = %s/init.txt\nmaxNumPoints = 2048\nslidingWindowSize = 3\npatchRadius = 1\nminScore = 0.65\nrobustThreshold = 0.05\nverbose = 0\n" % (tmp, tmp)
    cases = {"i": "", "pyr": "numLevels = 3\n", "ig": "descriptorType = IntensityAndGradient\n", "bp": "descriptorType = BitPlanes\n",
             "nms2": "nonMaxSuppRadius = 2\nmaskBlockRadius = 2\npatchRadius = 2\n"}
    for name, extra in cases.items():
        cfg = os.path.join(tmp, name + ".cfg")
        with open(cfg, "w") as f:
            f.write(base + extra)
        dev = _run(cfg, os.path.join(tmp, name + "_dev.txt"), os.path.join(tmp, name + "_dev.res"), {})
        host = _run(cfg, os.path.join(tmp, name + "_host.txt"), os.path.join(tmp, name + "_host.res"), {"PBA_HOST_FRONTEND": "1"})
        assert dev[0] == host[0] and len(dev[0].splitlines()) >= 3, name
        assert dev[1] == host[1], name                                       # every Result field of every window
        # same front-end statistics line by line ("updated %d ... new %d": photobundle.cc:593-595)
        upd = lambda s: [l for l in s.splitlines() if l.startswith("updated ")]
        assert upd(dev[2]) == upd(host[2]) and len(upd(dev[2])) == 6 * (3 if name == "pyr" else 1), name


def test_frontend_refuses_a_stale_u8_stage_and_a_border_without_interior():
    """the ZNCC reads the u8 image of the frame uploaded last; a frame that arrived as float channels (or none at all)
    leaves nothing to read -> PBA_ERR_STATE instead of scores of a stale image.  And the candidate scan checks its border."""
    from photobundle_amd.engine import EngineError
    size = (96, 131)
    rng = np.random.default_rng(5)
    img = _image(rng, size)
    uv = np.array([[40.0, 40.0]]); rc = np.array([[40, 40]], np.int32); pats = np.zeros((1, 26), np.float32)
    e = _engine(size, "Intensity")
    with pytest.raises(EngineError):                       # nothing uploaded yet
        e.frontend_visibility(uv, rc, pats, 0.5, 1)
    with pytest.raises(EngineError):
        e.frontend_zncc_probe(uv, pats)
    e.set_frame(0, img)
    e.frontend_visibility(uv, rc, pats, 0.5, 1)            # fine after a u8 upload
    with pytest.raises(EngineError):                       # border 48 leaves no interior in 96 rows
        e.frontend_candidates(0, np.ones(size, np.float32), 0.1, 100.0, 1, 48)
    e.close()
    e3 = _engine(size, "IntensityAndGradient")
    e3.set_frame_descriptor(0, img, "IntensityAndGradient")
    e3.frontend_visibility(uv, rc, pats, 0.5, 1)           # the descriptor producer keeps the u8 stage
    ch = np.zeros((3,) + size, np.float32)
    e3.set_frame_channels(1, ch)                            # float channels: no u8 image behind this frame
    with pytest.raises(EngineError):
        e3.frontend_visibility(uv, rc, pats, 0.5, 1)
    e3.close()
