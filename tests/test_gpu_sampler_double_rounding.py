"""-m gpu: the engine's sampler on the double-rounding cases of tests/golden/sampler_double_rounding.json (inputs where
a fused multiply-add in the vertical blend of sample_eigen.h:82-83 would flip the float result), through
pba_sample_frame: packed u8 frames (mode 0) and float channel images (mode 1), bit for bit."""
import numpy as np
import pytest

from test_sampler_double_rounding_cpu import bits32, load_cases, plane, position

pytestmark = pytest.mark.gpu
K = (1.0, 1.0, 0.0, 0.0)


def test_u8_frames():
    from photobundle_amd.engine import Engine
    e = Engine(rows=8, cols=8, max_frames=2, radius=1, K=K)
    n = 0
    for c in load_cases():
        if c["mode"] != 0:
            continue
        y, x = position(c)
        e.set_frame(0, plane(c).astype(np.uint8))
        got = e.sample_frame(0, [y], [x])[0]
        assert bits32(got[0]) == c["expected"], (c, got)
        n += 1
    assert n >= 5
    e.close()


def test_float_channels():
    from photobundle_amd.engine import Engine
    e = Engine(rows=8, cols=8, max_frames=2, radius=1, K=K, channels=3)
    n = 0
    for c in load_cases():
        if c["mode"] != 1:
            continue
        y, x = position(c)
        ch = np.zeros((3, 8, 8), np.float32)
        ch[1] = plane(c)
        e.set_frame_channels(1, ch)
        got = e.sample_frame(1, [y], [x], channel=1)[0]
        assert bits32(got[0]) == c["expected"], (c, got)
        assert np.array_equal(e.sample_frame(1, [y], [x], channel=0)[0], np.zeros(3, np.float32))
        n += 1
    assert n >= 20
    e.close()


def test_probe_equals_the_oracle_on_random_positions():
    """pba_sample_frame against oracle.sample_linear: interior, border-clamped and out-of-image positions."""
    from oracle import oracle
    from photobundle_amd.engine import Engine
    rng = np.random.default_rng(4)
    rows, cols = 40, 56
    img = rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)
    e = Engine(rows=rows, cols=cols, max_frames=2, radius=1, K=K)
    e.set_frame(0, img)
    planes = oracle.planes_from_u8(img)
    y = rng.uniform(-3, rows + 3, 4000).astype(np.float32)
    x = rng.uniform(-3, cols + 3, 4000).astype(np.float32)
    y[:500] = rng.uniform(0, 4, 500).astype(np.float32)          # the corner the regular walk hands to the per-tap path
    x[:500] = rng.uniform(0, 4, 500).astype(np.float32)
    got = e.sample_frame(0, y, x)
    want = np.stack([oracle.sample_linear(planes, yy, xx) for yy, xx in zip(y, x)])
    assert np.array_equal(got, want)
    e.close()
