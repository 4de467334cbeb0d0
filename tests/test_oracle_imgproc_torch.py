"""CPU: a THIRD, independent implementation of two OpenCV conventions the pyramid path rests on (reference src/photobundle_pyramid.cc:45-56),
PyTorch's own CPU kernels: F.interpolate(mode="bilinear", align_corners=False) -- the half-pixel sampling grid (d + 0.5) * scale - 0.5 with
border taps clamped that cv::resize INTER_LINEAR uses -- and F.conv2d over an F.pad(mode="reflect") image -- BORDER_REFLECT_101, the border
rule of cv::pyrDown / cv::GaussianBlur.  Neither shares code with the oracle or with scipy.ndimage (tests/test_oracle_imgproc_scipy.py);
OpenCV itself is absent from this image, so the oracle's image operators stay unpinned against it."""
import numpy as np
import pytest
torch = pytest.importorskip("torch")
import torch.nn.functional as F

from oracle import oracle


@pytest.mark.parametrize("size", [(9, 13), (64, 83), (188, 621), (376, 1241)])
def test_resize_bilinear_f32_against_torch_interpolate(size):
    rng = np.random.default_rng(size[1])
    z = rng.uniform(0.5, 60.0, size).astype(np.float32)
    dr, dc = (size[0] + 1) // 2, (size[1] + 1) // 2
    # (torch derives the scale from the sizes, in[i] / out[i], like cv::resize with fx = fy = 0)
    want = F.interpolate(torch.from_numpy(z)[None, None].double(), size=(dr, dc), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
    got = oracle.resize_bilinear_f32(z, dr, dc)
    assert got.shape == (dr, dc)
    # OpenCV (and the oracle) round the source coordinate to float before splitting it into tap and weight; torch in double does not: the
    # weights differ by 6e-8 x the coordinate (<= 1241), the values by that x the local contrast: measured 3e-7 (9 x 13) .. 4.9e-5 (376 x 1241) of
    # the range at the worst pixel, 5e-8 .. 3.2e-6 on average (the scipy test hands the independent side the float-rounded coordinates: 2e-5)
    assert np.abs(got - want).max() <= 1.5e-7 * max(size) * np.abs(want).max()
    assert np.abs(got - want).mean() <= 1e-8 * max(size) * np.abs(want).max()


@pytest.mark.parametrize("size", [(9, 13), (64, 83), (376, 1241)])
def test_pyr_down_u8_against_torch_conv(size):
    rng = np.random.default_rng(size[0])
    yy, xx = np.mgrid[0:size[0], 0:size[1]]
    img = np.clip(np.rint(127 + 70 * np.sin(xx / 5.0) * np.cos(yy / 3.0) + rng.normal(0, 12, size)), 0, 255).astype(np.uint8)
    w = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0], dtype=torch.float64)
    k2 = (w[:, None] * w[None, :])[None, None]                                        # exact in double: integers <= 36 x 255 x 256
    x = F.pad(torch.from_numpy(img.astype(np.float64))[None, None], (2, 2, 2, 2), mode="reflect")
    s = F.conv2d(x, k2, stride=2)[0, 0].numpy()                                         # sums at every second pixel
    want = np.floor((s + 128.0) / 256.0).astype(np.uint8)
    got = oracle.pyr_down_u8(img)
    assert got.shape == want.shape == ((size[0] + 1) // 2, (size[1] + 1) // 2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("sigma", [0.8, 1.5])
def test_gaussian_blur_f32_5x5_against_torch_conv(sigma):
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 255, (57, 91)).astype(np.float32)
    k = torch.exp(-0.5 * (torch.arange(-2, 3, dtype=torch.float64) / sigma) ** 2)
    k = k / k.sum()
    x = F.pad(torch.from_numpy(img)[None, None].double(), (2, 2, 2, 2), mode="reflect")
    want = F.conv2d(x, (k[:, None] * k[None, :])[None, None])[0, 0].numpy()
    got = oracle.gaussian_blur_f32_5x5(img, sigma)
    assert np.abs(got - want).max() <= 1e-4


def test_sample_linear_interior_against_torch_grid_sample():
    """SampleLinear / SampleWithDerivative (reference src/sample_eigen.h:33-126) inside the image: pixel centres at integer coordinates, the four
    neighbours blended with (1 - d, d) -- F.grid_sample(align_corners=True) on the three planes (I, Gx, Gy).  The reference's float / double mix
    moves the result by 1e-5 of the range at most; its rule OUTSIDE [0, n - 1] (truncation towards zero, then clamped taps: positions in
    (-1, 0) extrapolate) is its own and is pinned by the known-answer tests (tests/test_oracle_kat.py), not here."""
    rng = np.random.default_rng(5)
    rows, cols = 37, 53
    pl = oracle.planes_from_u8(rng.integers(0, 256, (rows, cols)).astype(np.uint8))
    ys, xs = rng.uniform(0, rows - 1, 3000), rng.uniform(0, cols - 1, 3000)
    got = np.stack([oracle.sample_linear(pl, y, x) for y, x in zip(ys, xs)]).astype(np.float64)
    gx = 2.0 * torch.from_numpy(np.float32(xs).astype(np.float64)) / (cols - 1) - 1.0       # (the sampler receives float coordinates)
    gy = 2.0 * torch.from_numpy(np.float32(ys).astype(np.float64)) / (rows - 1) - 1.0
    want = F.grid_sample(torch.from_numpy(pl.astype(np.float64))[None], torch.stack([gx, gy], -1)[None, None], mode="bilinear",
                         padding_mode="border", align_corners=True)[0, :, 0, :].T.numpy()
    assert np.abs(got - want).max() <= 1e-4
