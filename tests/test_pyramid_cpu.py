"""CPU checks of the pyramid helpers of the drop-in host library (cv::pyrDown / cv::resize semantics as documented in
photobundle_amd/host/photobundle_pyramid.h) against the numpy restatement used by the GPU pyramid test."""
import ctypes as C
import os

import numpy as np

from oracle.frontend import pyr_down_u8, resize_bilinear_f32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    return C.CDLL(os.path.join(ROOT, "tests", "native", "libhost_probe.so"))


def test_pyr_down_matches_numpy_and_known_answers():
    L = _lib()
    rng = np.random.default_rng(0)
    for rows, cols in [(8, 8), (9, 13), (120, 161), (2, 3)]:
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        out = np.zeros(((rows + 1) // 2, (cols + 1) // 2), np.uint8)
        L.pb_pyr_down_u8(img.ctypes.data_as(C.c_void_p), rows, cols, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, pyr_down_u8(img))
    # constant image stays constant (kernel sums to 256), size rule (cols+1)/2
    img = np.full((7, 10), 77, np.uint8)
    out = np.zeros((4, 5), np.uint8)
    L.pb_pyr_down_u8(img.ctypes.data_as(C.c_void_p), 7, 10, out.ctypes.data_as(C.c_void_p))
    assert np.all(out == 77)
    # interior pixel by hand: 1-D weights [1 4 6 4 1] on both axes
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    out = np.zeros((5, 5), np.uint8)
    L.pb_pyr_down_u8(img.ctypes.data_as(C.c_void_p), 9, 9, out.ctypes.data_as(C.c_void_p))
    assert out[2, 2] == (255 * 36 + 128) >> 8 and out[1, 2] == (255 * 6 + 128) >> 8 and out[0, 0] == 0


def test_resize_bilinear_matches_numpy_and_box_average():
    L = _lib()
    rng = np.random.default_rng(1)
    for rows, cols in [(8, 8), (9, 13), (120, 161)]:
        z = rng.uniform(-0.1, 50.0, (rows, cols)).astype(np.float32)
        dr, dc = (rows + 1) // 2, (cols + 1) // 2
        out = np.zeros((dr, dc), np.float32)
        L.pb_resize_bilinear_f32(z.ctypes.data_as(C.c_void_p), rows, cols, dr, dc, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, resize_bilinear_f32(z, dr, dc))
    # exact 2:1: the bilinear rule degenerates to the 2x2 box average
    z = rng.uniform(0, 10, (8, 12)).astype(np.float32)
    out = np.zeros((4, 6), np.float32)
    L.pb_resize_bilinear_f32(z.ctypes.data_as(C.c_void_p), 8, 12, 4, 6, out.ctypes.data_as(C.c_void_p))
    box = 0.25 * (z[0::2, 0::2] + z[0::2, 1::2] + z[1::2, 0::2] + z[1::2, 1::2])
    assert np.allclose(out, box, rtol=1e-6)
