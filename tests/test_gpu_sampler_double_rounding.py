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


def _tap_positions(case):
    """Float tap coordinates of a 3x3 patch whose FIRST tap is the case's position: the engine forms them as
    (float)(u + (double)(j - R)) (photobundle.cc:715-717) with u = 3 - dx, v = 3 - dy."""
    u, v = 3.0 - c_frac(case, "kx"), 3.0 - c_frac(case, "ky")
    xs = [np.float32(u + float(j - 1)) for j in range(3)]
    ys = [np.float32(v + float(j - 1)) for j in range(3)]
    return u, v, ys, xs


def c_frac(case, key):
    return case[key] * 2.0 ** -23


@pytest.mark.parametrize("mode", [0, 1])
def test_walk_and_irregular_paths_reproduce_the_oracle_samples_exactly(mode):
    """The kernels that run in a solve, not the probe: an identity camera with K = (1, 1, 0, 0) projects the point
    (3 - dx, 3 - dy, 1) onto itself, so the first tap of its 3x3 patch is the golden position.  With the descriptor set to
    the ORACLE's samples at the nine taps every residual is exactly zero -- iff the engine's samples are the same bits.
    Float channels (mode 1) take the regular walk of k_sample_mc; u8 frames (mode 0) sit in the 4x4 corner that k_sample
    routes to the per-tap path."""
    from oracle import oracle
    from photobundle_amd.engine import Engine
    C = 3 if mode else 1
    e = Engine(rows=8, cols=8, max_frames=2, radius=1, K=K, channels=C)
    n = 0
    for c in load_cases():
        if c["mode"] != mode:
            continue
        u, v, ys, xs = _tap_positions(c)
        assert ys[0] == position(c)[0] and xs[0] == position(c)[1]
        base = plane(c)
        base[3:5, 1:5] = base[1:3, 1:3].mean()             # something non-trivial under the other taps
        if mode:
            ch = np.zeros((3, 8, 8), np.float32)
            ch[1] = base
            for s in (0, 1):
                e.set_frame_channels(s, ch)
            planes = [oracle.channel_planes(ch)[3 * k: 3 * k + 3] for k in range(3)]
        else:
            img = base.astype(np.uint8)
            for s in (0, 1):
                e.set_frame(s, img)
            planes = [oracle.planes_from_u8(img)]
        desc = np.array([[float(oracle.sample_linear(pl, yy, xx)[0]) for yy in ys for xx in xs] for pl in planes]).reshape(1, -1)
        e.set_problem(np.array([[u, v, 1.0]]), desc, [0, 0], [0, 1], np.ones(9))
        e.set_cameras(np.zeros((2, 6)), fixed_slot=0)
        cost = e.linearize()
        rec = e.obs_records()
        assert cost == 0.0 and np.all(rec[:, 5] == 0.0), (c, cost, rec[:, 5])
        # ... and the test has teeth: the fused value in the descriptor instead gives a non-zero residual
        wrong = c["fused_dy_top"] if c["fused_dy_top"] != c["expected"] else c["fused_omdy_bot"]
        desc2 = desc.copy()
        desc2[0, (9 if mode else 0)] = np.frombuffer(np.uint32(wrong).tobytes(), np.float32)[0]
        e.set_problem(np.array([[u, v, 1.0]]), desc2, [0, 0], [0, 1], np.ones(9))
        e.set_cameras(np.zeros((2, 6)), fixed_slot=0)
        assert e.linearize() > 0.0
        n += 1
    assert n >= 5
    e.close()
