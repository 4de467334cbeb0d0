"""-m gpu: the device-side producers (pba_set_frame_descriptor_u8, pba_set_frame_pyr_down) against the host code they
replace -- photobundle_amd/host/imgproc.h (DescriptorFrame::Create, reference src/photobundle.cc:225-248,
src/imgproc.cc:109-245) and photobundle_pyramid.cc pyrDownU8 (reference src/photobundle_pyramid.cc:45-52) -- bit for bit,
and through the drop-in class (same output with the host producers switched back on)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "photobundle_amd", "bin", "run_kitti")
KINDS = {"IntensityAndGradient": (1, 3), "BitPlanes": (2, 8)}


def _host():
    L = C.CDLL(os.path.join(ROOT, "tests", "native", "libhost_probe.so"))
    L.pb_descriptor_channels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.pb_descriptor_channels.restype = C.c_int
    L.pb_pyr_down_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pb_pyr_down_u8.restype = None
    return L


def _image(rng, size):
    img = rng.integers(0, 256, size=size, dtype=np.uint8)
    r, c = size
    yy, xx = np.mgrid[0:r, 0:c]
    smooth = (127 + 90 * np.sin(xx / 7.0) * np.cos(yy / 5.0)).astype(np.uint8)
    img[: r // 2, :] = smooth[: r // 2, :]          # half texture with plateaus and ties (census >=), half noise
    img[r // 4: r // 4 + 3, : c // 3] = 255         # saturated run against the fixed-point rounding / clamp
    img[r // 4 + 4: r // 4 + 6, : c // 3] = 0
    return img


def _engine(size, channels=1, max_frames=2):
    from photobundle_amd.engine import Engine
    return Engine(rows=size[0], cols=size[1], max_frames=max_frames, radius=1, K=(200.0, 200.0, size[1] / 2.0, size[0] / 2.0), channels=channels)


def _host_channels(L, img, kind):
    code, c = KINDS[kind]
    out = np.empty((c,) + img.shape, np.float32)
    assert L.pb_descriptor_channels(img.ctypes.data, img.shape[0], img.shape[1], code, out.ctypes.data) == c
    return out


@pytest.mark.parametrize("kind", ["IntensityAndGradient", "BitPlanes"])
@pytest.mark.parametrize("size", [(37, 53), (376, 1241), (16, 16)])
def test_descriptor_channels_on_device_equal_imgproc_h(kind, size):
    from oracle import oracle
    from photobundle_amd import imgproc
    L = _host()
    rng = np.random.default_rng(11 + size[0])
    img = _image(rng, size)
    ref = _host_channels(L, img, kind)
    assert np.array_equal(ref, imgproc.descriptor_channels(img, kind))       # the numpy mirror bench.py uses
    planes = imgproc.channel_planes(ref).reshape(ref.shape[0], 3, *size)
    # ... and the ORACLE's DescriptorFrame::Create (oracle/pba_oracle.cpp: census imgproc.cc:126-197, the two GaussianBlur
    # restatements, imgradient imgproc.cc:27-95): the device channels and their gradient planes are checked against it
    # directly below, at every size incl. 376 x 1241, not only through the product's own host code
    o_ch = oracle.descriptor_channels(img, kind)
    o_planes = oracle.channel_planes(o_ch).reshape(o_ch.shape[0], 3, *size)
    assert np.array_equal(o_ch, ref) and np.array_equal(o_planes, planes)
    c = KINDS[kind][1]
    dev, via_host = _engine(size, channels=c), _engine(size, channels=c)
    dev.set_frame_descriptor(1, img, kind)
    via_host.set_frame_channels(1, ref)
    assert np.array_equal(dev.get_frame_channels(1), ref) and np.array_equal(via_host.get_frame_channels(1), ref)   # what the class's front-end reads back
    for k in range(c):
        got = dev.get_frame_channel(1, k)
        assert np.array_equal(got[0], o_ch[k]), (kind, k, np.abs(got[0] - o_ch[k]).max())      # device vs oracle
        assert np.array_equal(got, o_planes[k])
        assert np.array_equal(got[0], ref[k]), (kind, k, np.abs(got[0] - ref[k]).max())
        assert np.array_equal(got, planes[k])
        assert np.array_equal(got, via_host.get_frame_channel(1, k))
    dev.close(); via_host.close()


def test_bitplanes_without_smoothing():
    """sigma <= 0 skips the 3x3 / 5x5 smoothing (imgproc.h:124-141 with sigma_ct / sigma_bp <= 0).  Against the oracle's pieces."""
    from oracle import oracle as imgproc
    size = (41, 67)
    img = _image(np.random.default_rng(3), size)
    e = _engine(size, channels=8)
    for sct, sbp in [(0.0, 1.5), (1.0, 0.0), (0.0, 0.0), (0.7, 2.0)]:
        src = imgproc.gaussian_blur_u8_3x3(img, sct) if sct > 0 else img
        cen = imgproc.census(src)
        e.set_frame_descriptor(0, img, "BitPlanes", sigma_ct=sct, sigma_bp=sbp)
        for b in range(8):
            plane = ((cen >> b) & 1).astype(np.float32)
            want = imgproc.gaussian_blur_f32_5x5(plane, sbp) if sbp > 0 else plane
            assert np.array_equal(e.get_frame_channel(0, b)[0], want), (sct, sbp, b)
    e.close()


def test_descriptor_producer_argument_checks():
    from photobundle_amd.engine import EngineError
    size = (24, 32)
    img = _image(np.random.default_rng(0), size)
    e3, e1 = _engine(size, channels=3), _engine(size, channels=1)
    with pytest.raises(EngineError):
        e3.set_frame_descriptor(0, img, "BitPlanes")            # 8 channels into an engine created for 3
    with pytest.raises(EngineError):
        e1.set_frame_descriptor(0, img, "IntensityAndGradient")
    e1.set_frame_descriptor(0, img, "Intensity")                  # = pba_set_frame_u8
    assert np.array_equal(e1.get_frame_planes(0)[0], img.astype(np.float32))
    with pytest.raises(EngineError):
        e1.get_frame_channel(0, 0)                                # single-channel engine
    e3.close(); e1.close()


@pytest.mark.parametrize("size", [(376, 1241), (37, 53), (64, 64), (33, 35)])
def test_pyr_down_on_device_equals_host(size):
    """Three levels, level to level on the device: every level's u8 image equals pyrDownU8 of the previous one, and the
    planes the coarse engine samples equal those of an engine that was handed the host's image."""
    from oracle import oracle
    L = _host()
    img = _image(np.random.default_rng(size[1]), size)
    sizes = [size]
    for _ in range(2):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    engines = [_engine(s) for s in sizes]
    engines[0].set_frame(1, img)
    prev = img
    for lvl in (1, 2):
        want = np.empty(sizes[lvl], np.uint8)
        L.pb_pyr_down_u8(prev.ctypes.data, prev.shape[0], prev.shape[1], want.ctypes.data)
        got = engines[lvl].set_frame_pyr_down(0, engines[lvl - 1], 1 if lvl == 1 else 0)
        assert np.array_equal(got, oracle.pyr_down_u8(prev)), lvl            # device vs the oracle's cv::pyrDown restatement
        assert np.array_equal(engines[lvl].get_frame_planes(0), oracle.planes_from_u8(got))   # and the planes the level samples
        assert np.array_equal(got, want), (lvl, np.abs(got.astype(int) - want).max())
        ref = _engine(sizes[lvl])
        ref.set_frame(0, want)
        assert np.array_equal(engines[lvl].get_frame_planes(0), ref.get_frame_planes(0))
        ref.close()
        prev = want
    # asynchronous form (no image back): same planes
    again = _engine(sizes[1])
    again.set_frame_pyr_down(1, engines[0], 1, want_image=False)
    assert np.array_equal(again.get_frame_planes(1), engines[1].get_frame_planes(0))
    again.close()
    for e in engines:
        e.close()


def test_pyr_down_argument_checks():
    from photobundle_amd.engine import EngineError
    fine, wrong, ok = _engine((40, 50)), _engine((20, 26)), _engine((20, 25))
    with pytest.raises(EngineError):
        ok.set_frame_pyr_down(0, fine, 0)                         # nothing in the finer slot yet
    fine.set_frame(0, _image(np.random.default_rng(1), (40, 50)))
    with pytest.raises(EngineError):
        wrong.set_frame_pyr_down(0, fine, 0)                      # 20x26 is not the next level of 40x50
    ok.set_frame_pyr_down(0, fine, 0)
    for e in (fine, wrong, ok):
        e.close()


def _run(cfg, out, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([RUN, "-c", cfg, "-o", out, "-p"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


@pytest.mark.timeout(1500)
def test_class_output_is_identical_with_host_producers(tmp_path):
    """run_kitti with the device producers (default) and with the host ones (PBA_HOST_PYRAMID / PBA_HOST_CHANNELS):
    byte-identical refined trajectories -- the device builds the same frames."""
    from test_gpu_dropin_class import _write_sequence
    size, K = (160, 224), (280.0, 280.0, 112.0, 80.0)
    tmp = str(tmp_path)
    _write_sequence(tmp, 5, size, K)
    base = "DataDirectory = %s\nTrajectory = %s/init.txt\nmaxNumPoints = 4096\nslidingWindowSize = 3\npatchRadius = 1\nminScore = 0.65\nrobustThreshold = 0.05\nverbose = 0\n" % (tmp, tmp)
    cases = {"pyr": ("numLevels = 3\n", "PBA_HOST_PYRAMID"), "ig": ("descriptorType = IntensityAndGradient\n", "PBA_HOST_CHANNELS"),
             "bp": ("descriptorType = BitPlanes\n", "PBA_HOST_CHANNELS")}
    for name, (extra, env) in cases.items():
        cfg = os.path.join(tmp, name + ".cfg")
        with open(cfg, "w") as f:
            f.write(base + extra)
        dev = _run(cfg, os.path.join(tmp, name + "_dev.txt"), {})
        host = _run(cfg, os.path.join(tmp, name + "_host.txt"), {env: "1"})
        assert dev == host and len(dev.splitlines()) >= 3, name
