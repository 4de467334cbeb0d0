"""ctypes binding of the CPU oracle (oracle/libpba_oracle.so).  TEST INFRASTRUCTURE ONLY.

The oracle restates the reference hot path (reference src/photobundle.cc:669-736, 738-761, 764-829,
src/sample_eigen.h:33-126, src/jet_extras.h:87-111, src/calibration.h:34-38, src/imgproc.cc:27-95) and the
Ceres 1.x behaviour it relies on.  PARITY UNPINNED (the reference cannot be built here and ships no tests);
see oracle/pba_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/libpba_oracle.so with g++ (recipe: oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


class _Problem(C.Structure):
    _fields_ = [
        ("rows", C.c_int32), ("cols", C.c_int32),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("n_frames", C.c_int32), ("radius", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32),
        ("fixed_slot", C.c_int32), ("n_channels", C.c_int32),
        ("huber", C.c_double),
        ("planes", C.c_void_p), ("desc", C.c_void_p), ("weights", C.c_void_p),
        ("obs_point", C.c_void_p), ("obs_slot", C.c_void_p),
        ("cams", C.c_void_p), ("xyz", C.c_void_p),
    ]


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("num_threads", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("use_autodiff", C.c_int32), ("extended_precision", C.c_int32),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_nonmonotonic", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double), ("eta", C.c_double), ("step_size", C.c_double),
        ("line_search_function_evaluations", C.c_int32), ("line_search_gradient_evaluations", C.c_int32),
        ("line_search_iterations", C.c_int32), ("linear_solver_iterations", C.c_int32),
        ("iteration_time_in_seconds", C.c_double), ("step_solver_time_in_seconds", C.c_double),
        ("cumulative_time_in_seconds", C.c_double),
        ("model_cost_change", C.c_double), ("candidate_cost", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_iterations", C.c_int32), ("num_residuals", C.c_int32), ("num_residual_blocks", C.c_int32),
        ("termination_type", C.c_int32),
        ("total_time_in_seconds", C.c_double),
        ("num_jacobian_passes", C.c_int64), ("num_cost_passes", C.c_int64),
        ("message", C.c_char * 256),
    ]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpba_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_solve.restype = C.c_int
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Holder:
    """Keeps the numpy buffers referenced by an oracle_problem alive."""

    def __init__(self, prob, cams=None, xyz=None):
        self.planes = np.ascontiguousarray(prob.planes, dtype=np.float32)
        self.desc = _f64(prob.desc)
        self.weights = _f64(prob.weights)
        self.obs_point = np.ascontiguousarray(prob.obs_point, dtype=np.int32)
        self.obs_slot = np.ascontiguousarray(prob.obs_slot, dtype=np.int32)
        self.cams = _f64(prob.cams if cams is None else cams).copy()
        self.xyz = _f64(prob.xyz if xyz is None else xyz).copy()
        n_frames, three, rows, cols = self.planes.shape
        n_ch = int(getattr(prob, "channels", 1) or 1)
        assert three == 3 * n_ch                       # per channel: I, Gx, Gy
        R = int(prob.radius)
        Ppix = (2 * R + 1) ** 2
        P = n_ch * Ppix                                # residuals of one block
        assert self.desc.shape == (self.xyz.shape[0], P)
        assert self.weights.shape == (Ppix,)
        assert self.cams.shape == (n_frames, 6)
        assert np.all(np.diff(self.obs_point) >= 0), "observations must be grouped by point"
        s = _Problem()
        s.rows, s.cols = rows, cols
        s.fx, s.fy, s.cx, s.cy = [float(v) for v in prob.K]
        s.n_frames, s.radius = n_frames, R
        s.n_points, s.n_obs = self.xyz.shape[0], self.obs_point.shape[0]
        s.fixed_slot = int(prob.fixed_slot)
        s.n_channels = n_ch
        s.huber = float(prob.huber)
        s.planes, s.desc, s.weights = _ptr(self.planes), _ptr(self.desc), _ptr(self.weights)
        s.obs_point, s.obs_slot = _ptr(self.obs_point), _ptr(self.obs_slot)
        s.cams, s.xyz = _ptr(self.cams), _ptr(self.xyz)
        self.c = s
        self.P = P


def default_options(**kw):
    o = Options()
    lib().oracle_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def planes_from_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    out = np.empty((3, rows, cols), np.float32)
    lib().oracle_planes_from_u8(_ptr(img), rows, cols, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]))
    return out


def imgradient_f32(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    rows, cols = img.shape
    gx = np.empty_like(img)
    gy = np.empty_like(img)
    lib().oracle_imgradient_f32(_ptr(img), rows, cols, _ptr(gx), _ptr(gy))
    return gx, gy


DESCRIPTOR_TYPES = {"Intensity": 0, "IntensityAndGradient": 1, "BitPlanes": 2}


def descriptor_channels(img, descriptor_type):
    """u8 frame -> float channel images [C, rows, cols] (DescriptorFrame::Create, photobundle.cc:225-248)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape
    t = DESCRIPTOR_TYPES[descriptor_type] if isinstance(descriptor_type, str) else int(descriptor_type)
    n = lib().oracle_num_channels(t)
    out = np.empty((n, rows, cols), np.float32)
    lib().oracle_descriptor_channels(_ptr(img), rows, cols, t, _ptr(out))
    return out


def channel_planes(channels):
    """[C, rows, cols] channel images -> [3C, rows, cols]: every channel followed by its own Gx, Gy."""
    channels = np.ascontiguousarray(channels, dtype=np.float32)
    n, rows, cols = channels.shape
    out = np.empty((3 * n, rows, cols), np.float32)
    lib().oracle_channel_planes(_ptr(channels), n, rows, cols, _ptr(out))
    return out


def census(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib().oracle_census(_ptr(img), img.shape[0], img.shape[1], _ptr(out))
    return out


def gaussian_blur_u8_3x3(img, sigma):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib().oracle_gaussian_blur_u8_3x3(_ptr(img), img.shape[0], img.shape[1], C.c_double(float(sigma)), _ptr(out))
    return out


def gaussian_blur_f32_5x5(img, sigma):
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.empty_like(img)
    lib().oracle_gaussian_blur_f32_5x5(_ptr(img), img.shape[0], img.shape[1], C.c_double(float(sigma)), _ptr(out))
    return out


# ---- image / depth pyramid (reference src/photobundle_pyramid.cc:45-56; BASELINE configs[2]) --------------------------------
# The reference hands both to OpenCV, which is absent here: restated from OpenCV's documented behaviour in numpy integer /
# float32 arithmetic (doubly unpinned, like the two GaussianBlur restatements above -- see pba_oracle.h).
def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def pyr_down_u8(img):
    """cv::pyrDown(8U) as called at photobundle_pyramid.cc:45-47: separable [1 4 6 4 1] / 16 at the even pixels,
    BORDER_REFLECT_101, integer sums, (sum + 128) >> 8, destination size ((cols + 1) / 2, (rows + 1) / 2)."""
    img = np.asarray(img)
    rows, cols = img.shape
    w = np.array([1, 4, 6, 4, 1], np.int64)
    drows, dcols = (rows + 1) // 2, (cols + 1) // 2
    xs = 2 * np.arange(dcols)[:, None] + np.arange(-2, 3)[None, :]
    h = (img.astype(np.int64)[:, _reflect101(xs, cols)] * w).sum(-1)                 # [rows, dcols]
    ys = 2 * np.arange(drows)[:, None] + np.arange(-2, 3)[None, :]
    v = (h[_reflect101(ys, rows), :] * w[None, :, None]).sum(1)                      # [drows, dcols]
    return ((v + 128) >> 8).astype(np.uint8)


def resize_bilinear_f32(src, drows, dcols):
    """cv::resize(32F, dsize) with its default INTER_LINEAR, as called on the depth map at photobundle_pyramid.cc:54-56 (the
    comment there says "nearest neighbor"; the call is bilinear): source coordinate (d + 0.5) * scale - 0.5 in float, the
    horizontal pass then the vertical one, float arithmetic, border taps clamped."""
    src = np.asarray(src)
    rows, cols = src.shape
    f32 = np.float32

    def axis(n_src, n_dst):
        f = ((np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5).astype(f32)
        i = np.floor(f).astype(np.int64)
        a = (f - i.astype(f32)).astype(f32)
        lo = i < 0
        a[lo] = 0
        i[lo] = 0
        hi = i >= n_src - 1
        a[hi] = 0
        i[hi] = n_src - 1
        return i, a
    ix, ax = axis(cols, dcols)
    iy, ay = axis(rows, drows)
    x1 = np.minimum(ix + 1, cols - 1)
    y1 = np.minimum(iy + 1, rows - 1)
    s = src.astype(f32)
    a0 = (f32(1) - ax).astype(f32)
    h0 = (s[iy][:, ix] * a0 + s[iy][:, x1] * ax).astype(f32)
    h1 = (s[y1][:, ix] * a0 + s[y1][:, x1] * ax).astype(f32)
    b1 = ay[:, None]
    return (h0 * (f32(1) - b1) + h1 * b1).astype(f32)


def sample_linear(planes, y, x):
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    _, rows, cols = planes.shape
    out = np.zeros(3, np.float32)
    lib().oracle_sample_linear(_ptr(planes[0]), _ptr(planes[1]), _ptr(planes[2]), rows, cols,
                               C.c_float(float(np.float32(y))), C.c_float(float(np.float32(x))), _ptr(out))
    return out


def angle_axis_rotate_point(aa, pt):
    aa, pt = _f64(aa), _f64(pt)
    out = np.zeros(3)
    lib().oracle_angle_axis_rotate_point(_ptr(aa), _ptr(pt), _ptr(out))
    return out


def angle_axis_to_rotation_matrix(aa):
    aa = _f64(aa)
    R = np.zeros(9)
    lib().oracle_angle_axis_to_rotation_matrix(_ptr(aa), _ptr(R))
    return R.reshape(3, 3).T.copy()  # column-major storage -> numpy row-major matrix


def rotation_matrix_to_angle_axis(R):
    Rc = _f64(np.asarray(R).T).copy()  # to column-major storage
    aa = np.zeros(3)
    lib().oracle_rotation_matrix_to_angle_axis(_ptr(Rc), _ptr(aa))
    return aa


def make_patch_weights(radius, gaussian=False):
    w = np.zeros((2 * radius + 1) ** 2)
    lib().oracle_make_patch_weights(int(radius), int(bool(gaussian)), _ptr(w))
    return w


def extract_patch(I, u, v, radius):
    I = np.ascontiguousarray(I, dtype=np.float32)
    rows, cols = I.shape
    d = np.zeros((2 * radius + 1) ** 2)
    lib().oracle_extract_patch(_ptr(I), rows, cols, int(u), int(v), int(radius), _ptr(d))
    return d


def eval_block(prob, obs, autodiff=True, jac=True, cams=None, xyz=None):
    h = Holder(prob, cams, xyz)
    r = np.zeros(h.P)
    jc = np.zeros((h.P, 6)) if jac else None
    jp = np.zeros((h.P, 3)) if jac else None
    lib().oracle_eval_block(C.byref(h.c), int(obs), int(autodiff), _ptr(r),
                            _ptr(jc) if jac else None, _ptr(jp) if jac else None)
    return r, jc, jp


def linearize(prob, autodiff=True, threads=4, cams=None, xyz=None, blocks=True):
    """Returns dict(cost, block_sqnorm, grad_cams, grad_pts, U, V, W) at (cams, xyz)."""
    h = Holder(prob, cams, xyz)
    n_c, n_p, n_o = h.c.n_frames, h.c.n_points, h.c.n_obs
    cost = C.c_double(0.0)
    out = dict(block_sqnorm=np.zeros(n_o), grad_cams=np.zeros((n_c, 6)), grad_pts=np.zeros((n_p, 3)))
    if blocks:
        out.update(U=np.zeros((n_c, 6, 6)), V=np.zeros((n_p, 3, 3)), W=np.zeros((n_o, 6, 3)))
    lib().oracle_linearize(C.byref(h.c), int(autodiff), int(threads), C.byref(cost), _ptr(out["block_sqnorm"]),
                           _ptr(out["grad_cams"]), _ptr(out["grad_pts"]),
                           _ptr(out["U"]) if blocks else None, _ptr(out["V"]) if blocks else None,
                           _ptr(out["W"]) if blocks else None)
    out["cost"] = cost.value
    return out


def cost(prob, threads=4, cams=None, xyz=None):
    h = Holder(prob, cams, xyz)
    c = C.c_double(0.0)
    sq = np.zeros(h.c.n_obs)
    lib().oracle_linearize(C.byref(h.c), 1, int(threads), C.byref(c), _ptr(sq), None, None, None, None, None)
    return c.value, sq


def block_products(prob, autodiff=True, threads=4, cams=None, xyz=None):
    """Per residual block: dict(JcJc [n,6,6], JcJp [n,6,3], JpJp [n,3,3], Jcr [n,6], Jpr [n,3]) of the corrected rows."""
    h = Holder(prob, cams, xyz)
    out = np.zeros((h.c.n_obs, 72))
    lib().oracle_block_products(C.byref(h.c), int(autodiff), int(threads), _ptr(out))
    return dict(JcJc=out[:, :36].reshape(-1, 6, 6), JcJp=out[:, 36:54].reshape(-1, 6, 3), JpJp=out[:, 54:63].reshape(-1, 3, 3),
                Jcr=out[:, 63:69], Jpr=out[:, 69:72])


ITER_FIELDS = [f for f, _ in Iteration._fields_]


def solve(prob, options=None, cams=None, xyz=None, max_iterations_out=512):
    """Runs the Ceres-faithful LM.  Returns dict(cams, xyz, summary fields..., iterations=[dict])."""
    h = Holder(prob, cams, xyz)
    o = options or default_options()
    s = Summary()
    its = (Iteration * max_iterations_out)()
    lib().oracle_solve(C.byref(h.c), C.byref(o), C.byref(s), its, max_iterations_out)
    res = {f: getattr(s, f) for f, _ in Summary._fields_}
    res["message"] = s.message.decode()
    res["iterations"] = [{f: getattr(its[i], f) for f in ITER_FIELDS} for i in range(s.num_iterations)]
    res["cams"] = h.cams
    res["xyz"] = h.xyz
    return res
