"""-m gpu: a systematic sweep of patch positions over every image border and corner, single- and multi-channel.
An identity camera with K = (1, 1, 0, 0) projects the point (u, v, 1) onto (u, v) exactly, so the footprint origin of a
patch can be put on, one pixel inside and one pixel outside every limit the kernels branch on: the clamping rules of
LinearInitAxis (reference src/sample_eigen.h:34-52), the zero one-pixel border of the gradient images
(src/imgproc.cc:27-95), the regular walk / per-tap path split of k_sample (4 x 4 corner, right / bottom overhang) and of
k_sample_mc (interior footprints with a halo of values).  Every record against the oracle's rows at 1e-12."""
import numpy as np
import pytest

from oracle import oracle
from photobundle_amd.problem import WindowProblem

from gpu_util import check_obs_records, make_engine

pytestmark = pytest.mark.gpu
ROWS, COLS = 24, 32


def _positions(size, radius):
    """Patch centres whose first tap lands around 0, 1, 2, 4 (left / top limits) and whose last tap lands around size - 3 ..
    size + 1 (right / bottom limits), with a few fractional parts each; plus plain interior ones."""
    out = []
    for first_tap in (-1.5, -0.25, 0.0, 0.5, 1.0, 1.25, 2.0, 2.75, 3.5, 4.0, 4.5):
        out.append(first_tap + radius)
    for last_tap in (size - 4.5, size - 3.0, size - 2.5, size - 2.0, size - 1.75, size - 1.0, size - 0.5, size + 0.0, size + 1.25):
        out.append(last_tap - radius)
    out += [size / 2.0 + 0.37, size / 2.0 - 3.0]
    return out


def _problem(radius, channels, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(ROWS, COLS), dtype=np.uint8)
    img[5:12, 8:20] = (np.add.outer(np.arange(7), np.arange(12)) * 9 % 256).astype(np.uint8)
    if channels == 1:
        planes1 = oracle.planes_from_u8(img)
        chan = None
    else:
        kind = "IntensityAndGradient" if channels == 3 else "BitPlanes"
        chan = oracle.descriptor_channels(img, kind)
        planes1 = oracle.channel_planes(chan)
    us, vs = _positions(COLS, radius), _positions(ROWS, radius)
    xyz = np.array([[u, v, 1.0] for v in vs for u in us])
    n = xyz.shape[0]
    P = (2 * radius + 1) ** 2
    # (descriptors are patches of float channel images, photobundle.cc:466-479: the engine keeps them as float)
    desc = rng.uniform(0.0, 255.0 if channels == 1 else 1.0, size=(n, channels * P)).astype(np.float32).astype(np.float64)
    obs_point = np.repeat(np.arange(n, dtype=np.int32), 2)
    obs_slot = np.tile(np.array([0, 1], np.int32), n)
    return WindowProblem(K=(1.0, 1.0, 0.0, 0.0), radius=radius, planes=np.stack([planes1, planes1]), cams=np.zeros((2, 6)), xyz=xyz,
                         desc=desc, obs_point=obs_point, obs_slot=obs_slot, weights=np.ones(P), huber=0.0, fixed_slot=0,
                         images=np.stack([img, img]), channels=channels,
                         channel_images=None if chan is None else np.stack([chan, chan]))


@pytest.mark.parametrize("channels", [1, 3, 8])
@pytest.mark.parametrize("radius", [1, 2, 3])
def test_every_border_and_corner(radius, channels):
    p = _problem(radius, channels, seed=10 * radius + channels)
    lin = oracle.linearize(p, blocks=False)
    with make_engine(p) as e:
        cost = e.linearize()
        rec = e.obs_records()
        # the second slot's camera moves: the sweep also goes through the fused pipeline (back-substitution + candidate pass)
        e.step(1e4, init_scale=True)
    assert np.isfinite(cost) and np.isclose(cost, lin["cost"], rtol=1e-12), (cost, lin["cost"])
    assert np.allclose(rec[:, 5], 0.5 * lin["block_sqnorm"], rtol=1e-12, atol=0.0)
    check_obs_records(p, rec)
