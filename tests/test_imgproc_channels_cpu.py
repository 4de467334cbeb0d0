"""CPU: the host-side channel producers of photobundle_amd/imgproc.py (used by bench.py --channels; no oracle in the
product path) against the oracle's restatement of DescriptorFrame::Create (reference src/photobundle.cc:225-248,
src/imgproc.cc:109-245) -- bit for bit."""
import numpy as np
import pytest

from oracle import oracle
from photobundle_amd import imgproc, synthetic


@pytest.mark.parametrize("kind", ["IntensityAndGradient", "BitPlanes"])
def test_channels_equal_the_oracle(kind):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    img[5:20, 7:30] = (np.add.outer(np.arange(15), np.arange(23)) * 5 % 256).astype(np.uint8)
    ch = imgproc.descriptor_channels(img, kind)
    ref = oracle.descriptor_channels(img, kind)
    assert ch.shape == ref.shape and ch.dtype == np.float32
    assert np.array_equal(ch, ref)
    assert np.array_equal(imgproc.channel_planes(ch), oracle.channel_planes(ref))


def test_inverse_depth_rays_reproduce_the_points():
    p = synthetic.make_window(n_frames=4, n_points=50, radius=1, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0), visibility="causal")
    rays, rho = synthetic.inverse_depth_rays(p)
    assert (rho > 0).all()
    assert np.abs(rays[:, :3] + rays[:, 3:] / rho[:, None] - p.xyz).max() < 1e-9
