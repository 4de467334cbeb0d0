"""Synthetic sliding windows of the benchmark shapes (no KITTI imagery on either box).

Scene: Lambertian textured ground plane (y = +1.65 m) and fronto wall (z = +40 m) in camera-0 coordinates, texture =
sum of random 2-D sinusoids, so every frame is exactly photo-consistent.  Trajectory: forward motion 0.7 m/frame with
small yaw / lateral noise (matches the regime of reference data/kitti_init_poor/00.txt).  Initial poses: ground-truth
local poses perturbed (the "poor VO" regime).  Points are drawn from gradient-weighted pixel sites, back-projected
with noisy depth; descriptors are integer-pixel patches of the birth frame (reference src/photobundle.cc:466-479).
"""
import hashlib
import os

import numpy as np

from . import imgproc, se3
from .problem import WindowProblem

KITTI_K = (718.856, 718.856, 607.1928, 185.2157)   # fx fy cx cy, KITTI seq 00 (what reference dataset.cc:250-251 loads)
KITTI_SIZE = (376, 1241)                            # rows, cols

SEED_TEXTURE = 20260928
SEED_TRAJ = 20260929
SEED_INIT = 20260930
SEED_POINTS = 20260931

GROUND_Y = 1.65
WALL_Z = 40.0


class Texture:
    def __init__(self, n_waves=64, seed=SEED_TEXTURE):
        rng = np.random.default_rng(seed)
        self.lam = rng.uniform(0.2, 4.0, n_waves)
        ang = rng.uniform(0.0, 2 * np.pi, n_waves)
        self.dir = np.stack([np.cos(ang), np.sin(ang)], 1)
        self.phase = rng.uniform(0.0, 2 * np.pi, n_waves)
        self.amp = self.lam / self.lam.sum()
        self.sigma = np.sqrt(0.5 * np.sum(self.amp ** 2))

    def __call__(self, a, b):
        """plane coordinates (metres) -> intensity in [16, 240] (float64, before u8 rounding)."""
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        out = np.zeros(a.shape)
        flat_a, flat_b, flat_o = a.reshape(-1), b.reshape(-1), out.reshape(-1)
        step = 1 << 16
        for s in range(0, flat_a.size, step):
            aa, bb = flat_a[s:s + step, None], flat_b[s:s + step, None]
            arg = 2 * np.pi * (aa * self.dir[None, :, 0] + bb * self.dir[None, :, 1]) / self.lam[None, :] + self.phase
            flat_o[s:s + step] = np.sin(arg) @ self.amp
        f = np.clip(out / (3.0 * self.sigma), -1.0, 1.0)
        return 128.0 + 112.0 * f


def _yaw(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _small_rot(w):
    return se3.angle_axis_to_matrix(w)


def make_trajectory(n_frames, seed=SEED_TRAJ):
    """Ground-truth camera->world poses; frame 0 is the world frame."""
    rng = np.random.default_rng(seed)
    T = [np.eye(4)]
    for _ in range(1, n_frames):
        d = np.eye(4)
        d[:3, :3] = _yaw(np.deg2rad(rng.normal(0.0, 0.2)))
        d[:3, 3] = [rng.normal(0.0, 0.01), 0.0, 0.7]
        T.append(T[-1] @ d)
    return T


def perturb_local_poses(T_w, seed=SEED_INIT, rot_deg=0.1, trans=0.02):
    """GT world poses -> perturbed LOCAL poses (the addFrame() `T` argument), then re-chained world poses."""
    rng = np.random.default_rng(seed)
    local = [np.eye(4)]
    for i in range(1, len(T_w)):
        Tl = np.linalg.inv(T_w[i]) @ T_w[i - 1]          # trajectory.cc:7-16 inverted: T_w_i = T_w_{i-1} inv(T_i)
        P = np.eye(4)
        P[:3, :3] = _small_rot(np.deg2rad(rng.normal(0.0, rot_deg, 3)))
        P[:3, 3] = rng.normal(0.0, trans, 3)
        local.append(P @ Tl)
    return local, se3.chain_local_poses(local)


def render_frame(T_wc, K, size, tex):
    """Returns (u8 image, fp32 depth map) of the two-plane scene seen from camera->world pose T_wc."""
    rows, cols = size
    fx, fy, cx, cy = K
    xs, ys = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    d_cam = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)
    R, o = T_wc[:3, :3], T_wc[:3, 3]
    d_w = d_cam @ R.T
    with np.errstate(divide="ignore", invalid="ignore"):
        s_g = (GROUND_Y - o[1]) / d_w[..., 1]
        s_w = (WALL_Z - o[2]) / d_w[..., 2]
    s_g = np.where(s_g > 0, s_g, np.inf)
    s_w = np.where(s_w > 0, s_w, np.inf)
    ground = s_g < s_w
    s = np.where(ground, s_g, s_w)
    Xw = o[None, None, :] + s[..., None] * d_w
    val = np.where(ground, tex(Xw[..., 0], Xw[..., 2]), tex(Xw[..., 0] + 17.0, Xw[..., 1] + 5.0))
    img = np.clip(np.rint(val), 0, 255).astype(np.uint8)
    return img, s.astype(np.float32)


def project(K, T_cw, X):
    Xc = X @ T_cw[:3, :3].T + T_cw[:3, 3]
    fx, fy, cx, cy = K
    return np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy], 1), Xc[:, 2]


def make_window(n_frames=8, n_points=50000, radius=2, size=KITTI_SIZE, K=KITTI_K, visibility="dense",
                huber=0.0, gaussian=False, depth_noise=0.01, rot_deg=0.1, trans=0.02, seed_offset=0,
                point_seed_offset=0, dense_births=(0,), channel_fn=None):
    """Builds a WindowProblem of the named shape.

    visibility = "dense": every point is born in frame 0 and observed in every frame (sites whose ground-truth
                          projection leaves the margin in any frame are not drawn) -> n_obs = n_frames * n_points.
                 "causal": points are born uniformly over frames 0..n_frames-3 and observed from birth on while
                          inside the margin (mirrors the selection rule of reference photobundle.cc:789).
    point_seed_offset only changes the drawn points (multi-GPU shards share frames and cameras).
    channel_fn: None (Intensity) or a callable u8 frame -> (float channel images [C, rows, cols], planes [3C, rows, cols])
                of a multi-channel descriptor (reference photobundle.cc:229-245), supplied by the caller.
    dense_births: frames the "dense" sites (and their descriptors) are taken from; every point is still observed in ALL
                  frames.  One frame does not hold 200k sites that stay inside a 16-frame window (BASELINE configs[3]):
                  that shape uses (0, 8).
    """
    # PBA_WINDOW_CACHE=<dir> (dev tool: A/B timing loops on the GPU box spend 20-40 s per process generating the same window): the
    # finished problem is kept there as an .npz of plain arrays (np.load with allow_pickle=False: nothing in the file can execute),
    # keyed by every argument AND by the sources that generate it (an edit to the generator never reuses a stale window); the
    # directory is created private to the user.  Windows with a caller-supplied channel_fn are not cached.
    cache_file = None
    if os.environ.get("PBA_WINDOW_CACHE") and channel_fn is None:
        key = repr((n_frames, n_points, radius, tuple(size), tuple(K), visibility, huber, gaussian, depth_noise, rot_deg, trans,
                    seed_offset, point_seed_offset, tuple(dense_births), _generator_fingerprint()))
        cache_file = os.path.join(os.environ["PBA_WINDOW_CACHE"], "window_%s.npz" % hashlib.sha1(key.encode()).hexdigest()[:16])
        if os.path.exists(cache_file):
            return _window_from_npz(cache_file)
    prob = _make_window(n_frames, n_points, radius, size, K, visibility, huber, gaussian, depth_noise, rot_deg, trans, seed_offset,
                        point_seed_offset, dense_births, channel_fn)
    if cache_file is not None:
        os.makedirs(os.path.dirname(cache_file), mode=0o700, exist_ok=True)
        tmp = cache_file + ".%d.tmp.npz" % os.getpid()
        _window_to_npz(prob, tmp)
        os.replace(tmp, cache_file)
    return prob


_ARRAYS = ("planes", "cams", "xyz", "desc", "obs_point", "obs_slot", "weights", "images")
_META_ARRAYS = ("cams_gt", "T_gt", "T_init", "local_init", "depths")


def _generator_fingerprint():
    h = hashlib.sha1()
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("synthetic.py", "imgproc.py", "se3.py", "problem.py"):
        with open(os.path.join(here, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _window_to_npz(p, path):
    d = {k: getattr(p, k) for k in _ARRAYS}
    d.update({"meta_" + k: np.asarray(p.meta[k]) for k in _META_ARRAYS})
    d["scalars"] = np.array([p.radius, p.fixed_slot, p.channels], np.int64)
    d["K"] = np.array(p.K, np.float64)
    d["huber"] = np.array([p.huber], np.float64)
    d["visibility"] = np.frombuffer(str(p.meta["visibility"]).encode(), np.uint8)
    with open(path, "wb") as f:
        np.savez(f, **d)


def _window_from_npz(path):
    with np.load(path, allow_pickle=False) as z:
        radius, fixed_slot, channels = (int(v) for v in z["scalars"])
        meta = {k: z["meta_" + k] for k in _META_ARRAYS}
        meta["visibility"] = bytes(z["visibility"]).decode()
        return WindowProblem(K=tuple(float(v) for v in z["K"]), radius=radius, huber=float(z["huber"][0]), fixed_slot=fixed_slot,
                             channels=channels, meta=meta, **{k: z[k] for k in _ARRAYS})


def _make_window(n_frames, n_points, radius, size, K, visibility, huber, gaussian, depth_noise, rot_deg, trans, seed_offset,
                 point_seed_offset, dense_births, channel_fn):
    rows, cols = size
    tex = Texture(seed=SEED_TEXTURE + seed_offset)
    T_gt = make_trajectory(n_frames, SEED_TRAJ + seed_offset)
    local_init, T_init = perturb_local_poses(T_gt, SEED_INIT + seed_offset, rot_deg, trans)
    images, depths = [], []
    for T in T_gt:
        im, z = render_frame(T, K, size, tex)
        images.append(im)
        depths.append(z)
    images = np.stack(images)
    planes = np.stack([imgproc.planes_from_u8(im) for im in images])
    n_ch, channel_images, planes_mc = 1, None, None
    if channel_fn is not None:
        both = [channel_fn(im) for im in images]
        channel_images = np.stack([b[0] for b in both])
        planes_mc = np.stack([b[1] for b in both])
        n_ch = channel_images.shape[1]

    rng = np.random.default_rng(SEED_POINTS + seed_offset + 1000 * point_seed_offset)
    margin = radius + 2
    fx, fy, cx, cy = K
    T_cw_gt = [np.linalg.inv(T) for T in T_gt]

    births = list(dense_births) if visibility == "dense" else list(range(0, max(1, n_frames - 2)))
    per_birth = [n_points // len(births) + (1 if i < n_points % len(births) else 0) for i in range(len(births))]
    xyz_all, desc_all, obs_p, obs_s = [], [], [], []
    base = 0
    for b, n_b in zip(births, per_birth):
        sal = np.abs(planes[b, 1]) + np.abs(planes[b, 2])
        w = np.zeros_like(sal, dtype=np.float64)
        w[margin:rows - margin, margin:cols - margin] = sal[margin:rows - margin, margin:cols - margin] + 1e-3
        z_gt = depths[b].astype(np.float64)
        w[~np.isfinite(z_gt)] = 0.0
        ys, xs = np.nonzero(w > 0)
        Xc = np.stack([(xs - cx) / fx * z_gt[ys, xs], (ys - cy) / fy * z_gt[ys, xs], z_gt[ys, xs]], 1)
        Xw_gt = Xc @ T_gt[b][:3, :3].T + T_gt[b][:3, 3]
        vis = np.zeros((len(xs), n_frames), bool)
        for f in range(0 if visibility == "dense" else b, n_frames):
            uv, zc = project(K, T_cw_gt[f], Xw_gt)
            vis[:, f] = (zc > 0.1) & (uv[:, 0] >= margin) & (uv[:, 0] <= cols - 1 - margin) & \
                        (uv[:, 1] >= margin) & (uv[:, 1] <= rows - 1 - margin)
        if visibility == "dense":
            keep = vis.all(1)
        else:
            keep = vis[:, b:].sum(1) >= 3
        ys, xs, vis = ys[keep], xs[keep], vis[keep]
        pw = w[ys, xs]
        if n_b > len(xs):
            raise ValueError("not enough candidate sites: %d < %d" % (len(xs), n_b))
        sel = rng.choice(len(xs), size=n_b, replace=False, p=pw / pw.sum())
        sel.sort()
        ys, xs, vis = ys[sel], xs[sel], vis[sel]
        z = z_gt[ys, xs] * (1.0 + depth_noise * rng.standard_normal(n_b))
        Xc = np.stack([(xs - cx) / fx * z, (ys - cy) / fy * z, z], 1)
        # photobundle.cc:560: X = T_w * (z * K^-1 * [x y 1]) with the CURRENT (initial) estimate of the birth pose
        Xw = Xc @ T_init[b][:3, :3].T + T_init[b][:3, 3]
        xyz_all.append(Xw)
        if n_ch == 1:
            desc_all.append(imgproc.extract_patches(planes[b, 0], np.stack([xs, ys], 1), radius))
        else:     # one patch per channel, channel-major (photobundle.cc:597-603)
            desc_all.append(np.concatenate([imgproc.extract_patches(channel_images[b, k], np.stack([xs, ys], 1), radius)
                                            for k in range(n_ch)], axis=1))
        pi, fi = np.nonzero(vis)
        obs_p.append(pi + base)
        obs_s.append(fi)
        base += n_b

    obs_point = np.concatenate(obs_p).astype(np.int32)
    obs_slot = np.concatenate(obs_s).astype(np.int32)
    order = np.lexsort((obs_slot, obs_point))
    cams = np.stack([se3.pose_to_params(np.linalg.inv(T)) for T in T_init])     # photobundle.cc:774-778
    cams_gt = np.stack([se3.pose_to_params(np.linalg.inv(T)) for T in T_gt])
    return WindowProblem(
        channels=n_ch, channel_images=channel_images,
        K=tuple(K), radius=radius, planes=planes if n_ch == 1 else planes_mc, cams=cams, xyz=np.concatenate(xyz_all),
        desc=np.concatenate(desc_all), obs_point=obs_point[order], obs_slot=obs_slot[order],
        weights=imgproc.make_patch_weights(radius, gaussian), huber=huber, fixed_slot=0, images=images,
        meta=dict(cams_gt=cams_gt, T_gt=T_gt, T_init=T_init, local_init=local_init, depths=np.stack(depths),
                  visibility=visibility))


def channel_fn(kind):
    """make_window(channel_fn=...) for Options::descriptorType = "IntensityAndGradient" / "BitPlanes" (host producers of
    photobundle_amd/imgproc.py)."""
    def fn(img):
        ch = imgproc.descriptor_channels(img, kind)
        return ch, imgproc.channel_planes(ch)
    return fn


def inverse_depth_rays(p):
    """World ray of every point through the camera of its first observation (initial pose): X = o + d / rho with
    rho = 1 / depth in that camera.  Input of pba_set_inverse_depth (the north star's SE(3) x inverse-depth mode)."""
    first = np.searchsorted(p.obs_point, np.arange(p.n_points))
    slot = p.obs_slot[first]
    rays, rho = np.zeros((p.n_points, 6)), np.zeros(p.n_points)
    for s in np.unique(slot):
        T_cw = se3.params_to_pose(p.cams[s])          # world -> camera
        R, t = T_cw[:3, :3], T_cw[:3, 3]
        m = slot == s
        Xc = p.xyz[m] @ R.T + t
        rays[m, :3] = -R.T @ t
        rays[m, 3:] = (Xc / Xc[:, 2:3]) @ R
        rho[m] = 1.0 / Xc[:, 2]
    return rays, rho
