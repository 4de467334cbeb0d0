"""CPU: the oracle's restatements of the OpenCV calls on the path (cv::pyrDown, cv::resize, the two cv::GaussianBlur of
DescriptorFrame::Create; reference src/photobundle_pyramid.cc:45-56, src/imgproc.cc:109-245) against an INDEPENDENT code base,
scipy.ndimage.  OpenCV itself is absent here (the restatements follow its documented behaviour and stay unpinned against it); what
these tests rule out is a shared mistake in weights, border rule, sampling grid or rounding between the oracle, the host library and
the device producers, which all derive from the same reading of that documentation."""
import numpy as np
import pytest
ndimage = pytest.importorskip("scipy.ndimage")      # a box without scipy skips instead of failing collection

from oracle import oracle


def _img(rng, rows, cols):
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = 127 + 70 * np.sin(xx / 5.0) * np.cos(yy / 3.0) + rng.normal(0, 12, (rows, cols))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(9, 13), (64, 83), (120, 161), (376, 1241)])
def test_pyr_down_u8_against_ndimage(size):
    """[1 4 6 4 1] / 16 on both axes, BORDER_REFLECT_101 (= ndimage 'mirror'), integer sums, (sum + 128) >> 8, every second pixel."""
    rng = np.random.default_rng(size[0])
    img = _img(rng, *size)
    w = np.array([1, 4, 6, 4, 1], np.int64)
    h = ndimage.correlate1d(img.astype(np.int64), w, axis=1, mode="mirror")
    v = ndimage.correlate1d(h, w, axis=0, mode="mirror")
    want = ((v + 128) >> 8)[::2, ::2].astype(np.uint8)
    got = oracle.pyr_down_u8(img)
    assert got.shape == ((size[0] + 1) // 2, (size[1] + 1) // 2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("size", [(9, 13), (64, 83), (376, 1241)])
def test_resize_bilinear_f32_against_map_coordinates(size):
    """cv::resize INTER_LINEAR: source coordinate (d + 0.5) * scale - 0.5, taps clamped at the border (ndimage 'nearest')."""
    rng = np.random.default_rng(size[1])
    z = rng.uniform(0.5, 60.0, size).astype(np.float32)
    dr, dc = (size[0] + 1) // 2, (size[1] + 1) // 2
    # (OpenCV rounds the source coordinate to float before it splits it into tap and weight: at 1241 columns that alone moves the
    # result by 3e-3 against double coordinates, so the independent implementation gets the same float-rounded coordinates)
    ys = ((np.arange(dr) + 0.5) * (size[0] / dr) - 0.5).astype(np.float32).astype(np.float64)
    xs = ((np.arange(dc) + 0.5) * (size[1] / dc) - 0.5).astype(np.float32).astype(np.float64)
    yy, xx = np.meshgrid(np.clip(ys, 0, size[0] - 1), np.clip(xs, 0, size[1] - 1), indexing="ij")
    want = ndimage.map_coordinates(z.astype(np.float64), [yy, xx], order=1, mode="nearest")
    got = oracle.resize_bilinear_f32(z, dr, dc)
    assert got.shape == (dr, dc)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()        # float arithmetic of the restatement vs double


@pytest.mark.parametrize("sigma", [0.8, 1.5])
def test_gaussian_blur_f32_5x5_against_ndimage(sigma):
    """cv::GaussianBlur(32F, Size(5, 5), sigma): normalised 5-tap Gaussian on both axes, BORDER_REFLECT_101."""
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 255, (57, 91)).astype(np.float32)
    k = np.exp(-0.5 * (np.arange(-2, 3) / sigma) ** 2)
    k /= k.sum()
    want = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    got = oracle.gaussian_blur_f32_5x5(img, sigma)
    assert np.abs(got - want).max() <= 1e-4


@pytest.mark.parametrize("sigma", [0.8, 1.0])
def test_gaussian_blur_u8_3x3_against_ndimage(sigma):
    """cv::GaussianBlur(8U, Size(3, 3), sigma): OpenCV's (2.4 / 3.x) 8-bit path works with the kernel in 8-bit fixed point,
    cvRound(k * 256) per coefficient; where those sum to 256 -- the reference's sigma_ct = 1 (src/imgproc.h:44-46) and 0.8 -- the result is
    the rounded real-valued blur to within one grey level.  (At sigma = 0.5 the coefficients are 27, 201, 27 = 255 and that OpenCV is
    itself 0.8 % dark; the reference never goes there.)"""
    rng = np.random.default_rng(11)
    img = _img(rng, 64, 83)
    k = np.exp(-0.5 * (np.arange(-1, 2) / sigma) ** 2)
    k /= k.sum()
    want = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    got = oracle.gaussian_blur_u8_3x3(img, sigma).astype(np.float64)
    assert np.abs(got - want).max() <= 1.0 + 1e-9
    assert np.abs(got - np.rint(want)).mean() <= 0.1                      # and almost everywhere the nearest integer


def test_image_gradient_against_ndimage():
    """imgradient_ (src/imgproc.cc:27-95): central differences x 0.5 in the interior, zero on the one-pixel border."""
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 255, (40, 53)).astype(np.float32)
    gx, gy = oracle.imgradient_f32(img)
    wx = ndimage.correlate1d(img.astype(np.float64), np.array([-0.5, 0.0, 0.5]), axis=1, mode="nearest")
    wy = ndimage.correlate1d(img.astype(np.float64), np.array([-0.5, 0.0, 0.5]), axis=0, mode="nearest")
    assert np.allclose(gx[1:-1, 1:-1], wx[1:-1, 1:-1], atol=1e-4) and np.allclose(gy[1:-1, 1:-1], wy[1:-1, 1:-1], atol=1e-4)
    assert not gx[0].any() and not gx[-1].any() and not gx[:, 0].any() and not gx[:, -1].any()
    assert not gy[0].any() and not gy[-1].any() and not gy[:, 0].any() and not gy[:, -1].any()
