"""Known-answer tests that pin the CPU oracle.  The reference ships no tests, so every
expected value below is derived from the cited reference lines or from an independent implementation (scipy)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import oracle
from photobundle_amd import imgproc, se3


def test_gradient_ramp():
    # imgproc.cc:34-43,79-94: interior 0.5*(I[x+1]-I[x-1]); first/last row and column exactly zero
    rows, cols = 9, 12
    img = (3 * np.arange(cols)[None, :] + 5 * np.arange(rows)[:, None]).astype(np.uint8)
    pl = oracle.planes_from_u8(img)
    assert np.array_equal(pl[0], img.astype(np.float32))
    assert np.all(pl[1][1:-1, 1:-1] == 3.0) and np.all(pl[2][1:-1, 1:-1] == 5.0)
    for g in (pl[1], pl[2]):
        assert np.all(g[0] == 0) and np.all(g[-1] == 0) and np.all(g[:, 0] == 0) and np.all(g[:, -1] == 0)
    assert np.array_equal(pl, imgproc.planes_from_u8(img))


def test_gradient_random_matches_host_mirror():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(oracle.planes_from_u8(img), imgproc.planes_from_u8(img))


def _planes(rows=8, cols=10, seed=0):
    rng = np.random.default_rng(seed)
    return oracle.planes_from_u8(rng.integers(0, 256, (rows, cols), dtype=np.uint8))


def test_sampler_integer_coords_return_pixel():
    pl = _planes()
    for (y, x) in [(0, 0), (3, 4), (6, 8)]:
        s = oracle.sample_linear(pl, y, x)
        assert s[0] == pl[0, y, x] and s[1] == pl[1, y, x] and s[2] == pl[2, y, x]


def test_sampler_bilinear_interior():
    pl = _planes()
    y, x = 2.25, 3.5
    dy, dx = 3 - y, 4 - x
    for k in range(3):
        a = pl[k]
        exp = dy * (dx * a[2, 3] + (1 - dx) * a[2, 4]) + (1 - dy) * (dx * a[3, 3] + (1 - dx) * a[3, 4])
        assert oracle.sample_linear(pl, y, x)[k] == np.float32(exp)


def test_sampler_border_rules():
    # sample_eigen.h:38-51: truncation toward zero => x in (-1, 0) EXTRAPOLATES (ix = 0, dx = 1 - x > 1);
    # x <= -1 clamps to column 0; ix > size-2 clamps to the last column.
    pl = _planes()
    rows, cols = pl.shape[1:]
    a = pl[0]
    x = -0.5
    dx = 1.0 - x
    exp = dx * a[2, 0] + (1 - dx) * a[2, 1]
    assert oracle.sample_linear(pl, 2, x)[0] == np.float32(exp)
    assert oracle.sample_linear(pl, 2, -1.0)[0] == a[2, 0]
    assert oracle.sample_linear(pl, 2, -7.3)[0] == a[2, 0]
    assert oracle.sample_linear(pl, 2, cols - 1)[0] == a[2, cols - 1]
    assert oracle.sample_linear(pl, 2, cols - 0.5)[0] == a[2, cols - 1]
    assert oracle.sample_linear(pl, 2, 1e9)[0] == a[2, cols - 1]
    assert oracle.sample_linear(pl, rows + 3, 4)[0] == a[rows - 1, 4]
    # x in (cols-2, cols-1): ix = cols-2 is NOT > size-2 => regular interpolation
    x = cols - 1.25
    exp = 0.25 * a[2, cols - 2] + 0.75 * a[2, cols - 1]
    assert oracle.sample_linear(pl, 2, x)[0] == np.float32(exp)
    # NaN / overflow follow the x86 cvttss2si convention (INT_MIN => clamp to 0)
    assert oracle.sample_linear(pl, 2, np.nan)[0] == a[2, 0]
    assert oracle.sample_linear(pl, 2, 1e20)[0] == a[2, 0]


@pytest.mark.parametrize("seed", range(5))
def test_angle_axis_against_scipy(seed):
    rng = np.random.default_rng(seed)
    aa = rng.normal(0, 1.0, 3)
    pt = rng.normal(0, 5.0, 3)
    R = Rotation.from_rotvec(aa).as_matrix()
    assert np.allclose(oracle.angle_axis_rotate_point(aa, pt), R @ pt, rtol=0, atol=1e-13)
    assert np.allclose(oracle.angle_axis_to_rotation_matrix(aa), R, atol=1e-14)
    back = oracle.rotation_matrix_to_angle_axis(R)
    assert np.allclose(back, Rotation.from_matrix(R).as_rotvec(), atol=1e-12)
    assert np.allclose(se3.angle_axis_to_matrix(aa), R, atol=1e-14)
    assert np.allclose(se3.matrix_to_angle_axis(R), back, atol=1e-14)


def test_angle_axis_small_angle_branch():
    # theta^2 <= DBL_EPSILON: first-order p + w x p
    aa = np.array([1e-9, -2e-9, 3e-9])
    pt = np.array([1.0, 2.0, 3.0])
    assert np.array_equal(oracle.angle_axis_rotate_point(aa, pt), pt + np.cross(aa, pt))
    R = oracle.angle_axis_to_rotation_matrix(aa)
    assert R[0, 0] == 1.0 and R[1, 0] == aa[2] and R[0, 1] == -aa[2]
    assert np.allclose(oracle.rotation_matrix_to_angle_axis(np.eye(3)), 0.0)


def test_rotation_round_trip_near_pi():
    aa = np.array([0.0, 3.1, 0.2])
    R = oracle.angle_axis_to_rotation_matrix(aa)
    assert np.allclose(oracle.rotation_matrix_to_angle_axis(R), aa, atol=1e-12)


def test_patch_weights_and_extract_patch():
    assert np.array_equal(oracle.make_patch_weights(2), np.ones(25))
    w = oracle.make_patch_weights(2, True)
    assert abs(w.sum() - 1.0) < 1e-15 and w[12] == w.max() and np.allclose(w, imgproc.make_patch_weights(2, True), atol=1e-17)
    pl = _planes(12, 14, 3)
    for (u, v) in [(5, 6), (0, 0), (13, 11), (1, 10)]:
        d = oracle.extract_patch(pl[0], u, v, 2)
        assert np.array_equal(d, imgproc.extract_patches(pl[0], [(u, v)], 2)[0])
    # interior patch is the raw pixels, row-major (photobundle.cc:472-477)
    assert np.array_equal(oracle.extract_patch(pl[0], 5, 6, 1), pl[0][5:8, 4:7].reshape(-1).astype(np.float64))
