"""Host mirrors of the image-plane producers the hot path consumes (numpy; used to build test/bench inputs).

reference: src/imgproc.cc:27-95 (imgradient), src/photobundle.cc:231 (u8 -> float channel), :466-479 (ExtractPatch),
:617-644 (MakePatchWeights)."""
import numpy as np


def planes_from_u8(img):
    """[rows, cols] u8 -> [3, rows, cols] f32 = (I, Gx, Gy); border rows/cols of the gradients are zero."""
    I = np.asarray(img).astype(np.float32)
    gx = np.zeros_like(I)
    gy = np.zeros_like(I)
    gx[1:-1, 1:-1] = np.float32(0.5) * (I[1:-1, 2:] - I[1:-1, :-2])
    gy[1:-1, 1:-1] = np.float32(0.5) * (I[2:, 1:-1] - I[:-2, 1:-1])
    return np.stack([I, gx, gy])


def extract_patches(I, uv, radius):
    """Integer-pixel descriptors: [n, 2] int (u, v) -> [n, (2R+1)^2] f64, indices clamped like ExtractPatch."""
    rows, cols = I.shape
    uv = np.asarray(uv, dtype=np.int64)
    off = np.arange(-radius, radius + 1)
    r = np.clip(uv[:, 1, None] + off[None, :], radius, rows - radius - 1)   # [n, 2R+1]
    c = np.clip(uv[:, 0, None] + off[None, :], radius, cols - radius - 1)
    return I[r[:, :, None], c[:, None, :]].reshape(len(uv), -1).astype(np.float64)


def make_patch_weights(radius, gaussian=False):
    n = (2 * radius + 1) ** 2
    if not gaussian:
        return np.ones(n)
    off = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 * (off[:, None] ** 2 + off[None, :] ** 2)).reshape(-1)
    return w / w.sum()


# ---- multi-channel descriptors (reference DescriptorFrame::Create, src/photobundle.cc:225-248) -------------------------
def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def _gaussian_kernel_f32(n, sigma):
    """cv::getGaussianKernel(n, sigma, CV_32F): exp in double, stored as float, normalised by the sum of the stored floats."""
    x = np.arange(n) - (n - 1) * 0.5
    k = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    return (k.astype(np.float64) * (1.0 / k.astype(np.float64).sum())).astype(np.float32)


def census(img):
    """src/imgproc.cc:126-197: bit b set when the b-th neighbour (row-major 3x3 order without the centre) >= centre."""
    I = np.asarray(img, dtype=np.uint8)
    out = np.zeros_like(I)
    c = I[1:-1, 1:-1]
    shifts = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
    acc = np.zeros(c.shape, dtype=np.uint8)
    for b, (dy, dx) in enumerate(shifts):
        nb = I[1 + dy:I.shape[0] - 1 + dy, 1 + dx:I.shape[1] - 1 + dx]
        acc |= ((nb >= c).astype(np.uint8) << b)
    out[1:-1, 1:-1] = acc
    return out


def gaussian_blur_u8_3x3(img, sigma):
    """cv::GaussianBlur(8U, Size(3, 3), sigma): 8-bit fixed-point kernel, integer passes, (sum + 2^15) >> 16, reflect-101."""
    I = np.asarray(img, dtype=np.int64)
    rows, cols = I.shape
    k = np.rint(_gaussian_kernel_f32(3, sigma).astype(np.float64) * 256.0).astype(np.int64)
    xs = np.arange(cols)
    tmp = k[0] * I[:, _reflect101(xs - 1, cols)] + k[1] * I + k[2] * I[:, _reflect101(xs + 1, cols)]
    ys = np.arange(rows)
    v = k[0] * tmp[_reflect101(ys - 1, rows)] + k[1] * tmp + k[2] * tmp[_reflect101(ys + 1, rows)]
    return np.clip((v + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def gaussian_blur_f32_5x5(img, sigma):
    """cv::GaussianBlur(32F, Size(5, 5), sigma): separable symmetric form k2 c + k1 (l1 + r1) + k0 (l2 + r2) in float."""
    I = np.asarray(img, dtype=np.float32)
    rows, cols = I.shape
    k = _gaussian_kernel_f32(5, sigma)
    xs, ys = np.arange(cols), np.arange(rows)
    t = I * k[2]
    t = t + (I[:, _reflect101(xs - 1, cols)] + I[:, _reflect101(xs + 1, cols)]) * k[1]
    t = t + (I[:, _reflect101(xs - 2, cols)] + I[:, _reflect101(xs + 2, cols)]) * k[0]
    o = t * k[2]
    o = o + (t[_reflect101(ys - 1, rows)] + t[_reflect101(ys + 1, rows)]) * k[1]
    o = o + (t[_reflect101(ys - 2, rows)] + t[_reflect101(ys + 2, rows)]) * k[0]
    return o.astype(np.float32)


def descriptor_channels(img, kind):
    """u8 frame -> [C, rows, cols] f32 channel images: "Intensity" (1), "IntensityAndGradient" (3: I, 0.5 central
    differences of the u8 image) or "BitPlanes" (8: census of the smoothed frame, each bit plane blurred)."""
    I = np.asarray(img, dtype=np.uint8)
    if kind == "BitPlanes":
        c = census(gaussian_blur_u8_3x3(I, 1.0))
        return np.stack([gaussian_blur_f32_5x5(((c >> b) & 1).astype(np.float32), 1.5) for b in range(8)])
    f = I.astype(np.float32)
    if kind == "Intensity":
        return f[None]
    gx, gy = np.zeros_like(f), np.zeros_like(f)
    gx[1:-1, 1:-1] = np.float32(0.5) * (f[1:-1, 2:] - f[1:-1, :-2])
    gy[1:-1, 1:-1] = np.float32(0.5) * (f[2:, 1:-1] - f[:-2, 1:-1])
    return np.stack([f, gx, gy])


def channel_planes(channels):
    """[C, rows, cols] -> [3 C, rows, cols]: every channel followed by its own gradient images (photobundle.cc:172-175)."""
    out = []
    for ch in channels:
        out.extend(_planes_f32(ch))
    return np.stack(out)


def _planes_f32(ch):
    I = np.asarray(ch, dtype=np.float32)
    gx, gy = np.zeros_like(I), np.zeros_like(I)
    gx[1:-1, 1:-1] = np.float32(0.5) * (I[1:-1, 2:] - I[1:-1, :-2])
    gy[1:-1, 1:-1] = np.float32(0.5) * (I[2:, 1:-1] - I[:-2, 1:-1])
    return [I, gx, gy]
