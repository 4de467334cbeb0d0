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
