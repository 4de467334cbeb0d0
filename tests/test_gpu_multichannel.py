"""-m gpu: multi-channel descriptors (reference DescriptorFrame::Create, photobundle.cc:225-248: IntensityAndGradient =
3 channels, BitPlanes = 8) through the C-ABI against the CPU oracle.  A residual block then has C (2R+1)^2 rows, channel
major; the patch weights restart with every channel (:714-721) and the robust loss sees the whole block.
Tolerances as in test_gpu_parity.py (the channel samples themselves are bit-exact by construction)."""
import numpy as np
import pytest

from oracle import oracle
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, EngineError, default_solver_options

from gpu_util import check_obs_records, dense_system, make_engine, reference_step

pytestmark = pytest.mark.gpu
SMALL = dict(size=(120, 160), K=(200.0, 200.0, 80.0, 60.0))


def _channels(kind):
    def fn(img):
        ch = oracle.descriptor_channels(img, kind)
        return ch, oracle.channel_planes(ch)
    return fn


def _window(kind, radius=2, huber=0.0, gaussian=False, vis="dense", n_points=80, seed=0):
    return synthetic.make_window(n_frames=4, n_points=n_points, radius=radius, huber=huber, gaussian=gaussian, visibility=vis,
                                 seed_offset=seed, channel_fn=_channels(kind), **SMALL)


@pytest.mark.parametrize("kind,radius,huber,gaussian,vis", [("IntensityAndGradient", 2, 0.0, False, "dense"),
                                                            ("IntensityAndGradient", 1, 0.5, True, "causal"),
                                                            ("BitPlanes", 2, 0.0, False, "dense"),
                                                            ("BitPlanes", 3, 0.05, False, "causal"),
                                                            ("BitPlanes", 5, 0.0, False, "dense")])
def test_linearisation_and_reduced_system(kind, radius, huber, gaussian, vis):
    p = _window(kind, radius, huber, gaussian, vis, n_points=60, seed=radius)
    assert p.channels == {"IntensityAndGradient": 3, "BitPlanes": 8}[kind] and p.desc.shape[1] == p.channels * p.patch_len
    lin = oracle.linearize(p, blocks=False)
    J, r, n_cam = dense_system(p)
    ref = reference_step(J, r, n_cam, 1e4)
    with make_engine(p) as e:
        cost = e.linearize()
        rec = e.obs_records()
        s, a = lin["block_sqnorm"], p.huber
        rho = np.where((a > 0) & (s > a * a), 2 * a * np.sqrt(s) - a * a, s)
        assert np.isclose(cost, lin["cost"], rtol=1e-12)
        assert np.allclose(rec[:, 5], 0.5 * rho, rtol=1e-12, atol=0)
        check_obs_records(p, rec)
        info = e.step(1e4, init_scale=True)
        S, rhs = e.reduced_system()
        assert np.abs(S - ref["S"]).max() <= 1e-9 * np.abs(ref["S"]).max()
        assert np.abs(rhs - ref["rhs"]).max() <= 1e-9 * np.abs(ref["rhs"]).max()
        assert np.isclose(info["gradient_max_norm"], np.abs(ref["gradient"]).max(), rtol=1e-10)


@pytest.mark.parametrize("kind,huber", [("IntensityAndGradient", 0.0), ("BitPlanes", 0.05)])
def test_lm_trace_matches_oracle(kind, huber):
    p = _window(kind, 2, huber, n_points=300, seed=7)
    n_it = 15
    ref = oracle.solve(p, oracle.default_options(max_num_iterations=n_it))
    with make_engine(p) as e:
        res = e.solve(default_solver_options(max_num_iterations=n_it))
    assert len(ref["iterations"]) == len(res["iterations"]), (ref["message"], res["message"])
    # rounding differences grow along the LM path (see tests/test_gpu_fullsize.py::_noise_floor_parity): the first
    # iterations are held to 1e-9, the rest of the trace to the looser bound, the refined poses to the north-star bar
    for a, b in zip(ref["iterations"], res["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"], a["iteration"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-9 if a["iteration"] <= 4 else 1e-6), a["iteration"]
        assert np.isclose(a["trust_region_radius"], b["trust_region_radius"], rtol=1e-6 if a["iteration"] <= 4 else 1e-3)
    assert res["num_residuals"] == ref["num_residuals"] == p.n_obs * p.channels * p.patch_len
    assert np.abs(res["cams"] - ref["cams"]).max() <= 1e-5


def test_border_patches_take_the_clamped_path():
    p = _window("IntensityAndGradient", 2, n_points=80, seed=5)
    rng = np.random.default_rng(0)
    p.xyz = p.xyz + rng.normal(0, 1.0, p.xyz.shape) * np.array([3.0, 2.0, 0.0])
    lin = oracle.linearize(p, blocks=False)
    with make_engine(p) as e:
        cost = e.linearize()
        rec = e.obs_records()
    assert np.isclose(cost, lin["cost"], rtol=1e-12)
    assert np.allclose(rec[:, 5], 0.5 * lin["block_sqnorm"], rtol=1e-12)
    check_obs_records(p, rec)


def test_frame_entry_points_are_checked():
    p = _window("IntensityAndGradient", 2, n_points=20, seed=9)
    e = Engine(120, 160, p.K, 2, 4, channels=3)
    try:
        with pytest.raises(EngineError, match="channels"):
            e.set_frame(0, p.images[0])                     # u8 frames belong to the single-channel engine
        with pytest.raises(EngineError):
            e._check(e._L.pba_set_frame_channels_f32(e._h, 0, 8, p.channel_images[0].ctypes.data), "n_channels mismatch")
    finally:
        e.close()
    e = Engine(120, 160, p.K, 2, 4)
    try:
        with pytest.raises(EngineError):
            e._check(e._L.pba_set_frame_channels_f32(e._h, 0, 3, p.channel_images[0].ctypes.data), "single-channel engine")
    finally:
        e.close()
