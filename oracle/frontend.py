"""TEST INFRASTRUCTURE (oracle of SURVEY.md 8 row f1) -- numpy restatement of the reference FRONT-END, never imported by the product.

PhotometricBundleAdjustment::addFrame / optimize (reference src/photobundle.cc:482-615, :764-876), statement by statement:
  * `_interp2`            src/photobundle.cc:262-294   (float arithmetic, the four border cases)
  * `Zncc`                src/photobundle.cc:296-361   (interpolateFixedPatch :299-312, ZnccPatch_<2, float>: mean, norm, score)
  * `Emulator.add_frame`  src/photobundle.cc:482-615   (trajectory push :485-487, visibility update :505-542 with std::round and the
                                                        mask block :536-538, saliency map :550 -> :213-221, candidate selection :555-573
                                                        with IsLocalMax_ src/imgproc.h:176-212, top-N :578-585, ExtractPatch :466-479,
                                                        :597-603, ring buffer :605-612)
  * `Emulator._optimize`  src/photobundle.cc:764-876   (window assembly :774-816 -> WindowProblem, solved by the C++ oracle
                                                        oracle/pba_oracle.cpp, write-back :841-875, eviction :851, :888-905)
  * `PyramidEmulator`     src/photobundle_pyramid.cc:37-69 as intended (see photobundle_amd/host/photobundle_pyramid.h), levels from
                                                        oracle.pyr_down_u8 / resize_bilinear_f32
It rebuilds, frame by frame, the window problems the C++ drop-in class (photobundle_amd/host) must assemble, solves each with the
CPU oracle and returns the refined trajectory; tests/test_gpu_frontend.py also holds the device front-end kernels to `Zncc` bit for
bit.  Like the rest of oracle/: parity unpinned (the reference cannot be built here and ships no vectors).  (Until round 4 this file
was tests/frontend_emulation.py.)"""
import numpy as np

from oracle import oracle
from photobundle_amd import imgproc, se3
from photobundle_amd.problem import WindowProblem


def _interp2(I, xf, yf, fill=np.float32(0.0)):
    # reference photobundle.cc:262-294 (float arithmetic)
    xf = np.float32(xf)
    yf = np.float32(yf)
    max_cols, max_rows = I.shape[1] - 1, I.shape[0] - 1
    xi, yi = int(np.floor(xf)), int(np.floor(yf))
    xf = np.float32(xf - np.float32(xi))
    yf = np.float32(yf - np.float32(yi))
    f32 = np.float32
    if 0 <= xi < max_cols and 0 <= yi < max_rows:
        # float products / sums inside the brackets, double only through the `1.0 - yf` factor, result rounded to float
        wx = f32(1.0 - float(xf))
        top = f32(f32(f32(I[yi, xi]) * wx) + f32(f32(I[yi, xi + 1]) * xf))
        bot = f32(f32(f32(I[yi + 1, xi]) * wx) + f32(f32(I[yi + 1, xi + 1]) * xf))
        return f32((1.0 - float(yf)) * float(top) + float(f32(yf * bot)))
    if xi == max_cols and yi < max_rows:
        return fill if xf > 0 else np.float32((1.0 - float(yf)) * float(I[yi, xi]) + float(yf) * float(I[yi + 1, xi]))
    if yi == max_rows and xi < max_cols:
        return fill if yf > 0 else np.float32((1.0 - float(xf)) * float(I[yi, xi]) + float(xf) * float(I[yi, xi + 1]))
    if xi == max_cols and yi == max_rows:
        return fill if (xf > 0 or yf > 0) else np.float32(I[yi, xi])
    return fill


class Zncc:
    def __init__(self, I, u, v):
        x, y = np.float32(u), np.float32(v)
        d = np.array([_interp2(I, np.float32(c) + x, np.float32(r) + y) for r in range(-2, 3) for c in range(-2, 3)],
                     dtype=np.float32)
        s = np.float32(0.0)
        for k in d:
            s = np.float32(s + k)
        mean = np.float32(s / np.float32(25))
        d = (d - mean).astype(np.float32)
        n2 = np.float32(0.0)
        for k in d:
            n2 = np.float32(n2 + np.float32(k * k))
        self.data, self.norm = d, np.float32(np.sqrt(n2))

    def score(self, o):
        d = np.float32(self.norm * o.norm)
        if not d > 1e-6:
            return -1.0
        dot = np.float32(0.0)
        for a, b in zip(self.data, o.data):
            dot = np.float32(dot + np.float32(a * b))
        return float(np.float32(dot / d))


class Point:
    def __init__(self, X, fid):
        self.X = X.copy()
        self.X0 = X.copy()           # ScenePoint::_X_original (photobundle.cc:380)
        self.f = [fid]
        self.patch = None
        self.desc = None
        self.saliency = 0.0


class Emulator:
    def __init__(self, K, size, window, radius, max_points, min_score=0.75, huber=0.05, mask_radius=1,
                 max_frame_distance=1, nms=1, min_depth=0.01, max_depth=1000.0, max_iterations=500,
                 descriptor_type="Intensity"):
        self.descriptor_type = descriptor_type
        self.K, self.size = K, size
        self.window, self.radius, self.max_points = window, radius, max_points
        self.min_score, self.huber, self.mask_radius = min_score, huber, mask_radius
        self.max_frame_distance, self.nms = max_frame_distance, nms
        self.min_depth, self.max_depth = min_depth, max_depth
        self.max_iterations = max_iterations
        self.frame_id = 0
        self.T_w = []          # world poses by frame id
        self.frames = []       # (id, u8 image) ring
        self.points = []
        self.results = []

    def add_frame(self, img, depth, T_local):
        fx, fy, cx, cy = self.K
        rows, cols = self.size
        Ti = np.linalg.inv(T_local)
        self.T_w.append(Ti if not self.T_w else self.T_w[-1] @ Ti)
        T_w = self.T_w[-1]
        T_c = np.linalg.inv(T_w)
        I = img
        # DescriptorFrame::Create (photobundle.cc:225-248) + the per-channel gradients (:172-175)
        channels = oracle.descriptor_channels(img, self.descriptor_type)
        planes_mc = oracle.channel_planes(channels)
        planes = planes_mc[:3]
        B = max(self.mask_radius, max(2, self.radius))
        max_rows, max_cols = rows - B - 1, cols - B - 1
        mask = np.ones((rows, cols), np.uint16)
        Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        for pt in self.points:
            if self.frame_id - pt.f[-1] <= self.max_frame_distance:
                Xc = T_c[:3, :3] @ pt.X + T_c[:3, 3]
                p = Kmat @ Xc
                uv = (1.0 / p[2]) * p[:2]
                # std::round: half away from zero
                r = int(np.floor(uv[1] + 0.5)) if uv[1] >= 0 else -int(np.floor(-uv[1] + 0.5))
                c = int(np.floor(uv[0] + 0.5)) if uv[0] >= 0 else -int(np.floor(-uv[0] + 0.5))
                if B <= r < max_rows and B <= c <= max_cols:
                    if pt.patch.score(Zncc(I, uv[0], uv[1])) > self.min_score:
                        pt.f.append(self.frame_id)
                        mask[r - self.mask_radius:r + self.mask_radius + 1, c - self.mask_radius:c + self.mask_radius + 1] = 0
        sal = (np.abs(planes_mc[1]) + np.abs(planes_mc[2])).astype(np.float32)      # computeSaliencyMap (:213-221)
        for k in range(1, channels.shape[0]):
            sal = (sal + (np.abs(planes_mc[3 * k + 1]) + np.abs(planes_mc[3 * k + 2])).astype(np.float32)).astype(np.float32)
        new = []
        Kinv = np.linalg.inv(Kmat)
        n = self.nms
        for y in range(B, max_rows):
            for x in range(B, max_cols):
                z = depth[y, x]
                if not (self.min_depth <= z <= self.max_depth):
                    continue
                v = sal[y, x]
                if n > 0:
                    if not mask[y, x]:
                        continue
                    win = sal[y - n:y + n + 1, x - n:x + n + 1].copy()
                    win[n, n] = -np.inf
                    if np.any(win >= v):
                        continue
                ray = Kinv @ np.array([float(x), float(y), 1.0])
                X = T_w[:3, :3] @ (float(z) * ray) + T_w[:3, 3]
                p = Point(X, self.frame_id)
                p.patch = Zncc(I, float(x), float(y))
                p.saliency = float(v)
                p.xy = (x, y)
                new.append(p)
        if len(new) > self.max_points:
            # nth_element keeps an unspecified subset among ties; the test scene has no ties at the cut
            new.sort(key=lambda q: -q.saliency)
            new = new[:self.max_points]
        if new:
            xy = np.array([q.xy for q in new])
            d = np.concatenate([imgproc.extract_patches(channels[k], xy, self.radius) for k in range(channels.shape[0])], axis=1)
            for q, dd in zip(new, d):
                q.desc = dd
        self.points.extend(new)
        self.frames.append((self.frame_id, img, channels, planes_mc))
        if len(self.frames) > self.window:
            self.frames.pop(0)
        if len(self.frames) == self.window:
            self._optimize()
        self.frame_id += 1

    def _optimize(self):
        start, end = self.frames[0][0], self.frames[-1][0]
        W = self.window
        cams = np.zeros((W, 6))
        images = np.zeros((W,) + tuple(self.size), np.uint8)
        n_ch = self.frames[0][2].shape[0]
        planes = np.zeros((W, 3 * n_ch) + tuple(self.size), np.float32)
        for fid, img, _, pl in self.frames:
            cams[fid % W] = se3.pose_to_params(np.linalg.inv(self.T_w[fid]))
            images[fid % W] = img
            planes[fid % W] = pl
        sel = [p for p in self.points if len(p.f) >= 3 and p.f[0] >= start]
        obs_p, obs_s = [], []
        for i, p in enumerate(sel):
            for s in sorted(f % W for f in p.f if start <= f <= end):
                obs_p.append(i)
                obs_s.append(s)
        if sel:
            prob = WindowProblem(K=self.K, radius=self.radius, planes=planes, channels=n_ch,
                                 cams=cams, xyz=np.stack([p.X for p in sel]), desc=np.stack([p.desc for p in sel]),
                                 obs_point=np.array(obs_p, np.int32), obs_slot=np.array(obs_s, np.int32),
                                 weights=imgproc.make_patch_weights(self.radius), huber=self.huber,
                                 fixed_slot=(start % W) if (start % W) in obs_s else -1,     # photobundle.cc:809-815
                                 images=images)
            res = oracle.solve(prob, oracle.default_options(max_num_iterations=self.max_iterations))
            for p, X in zip(sel, res["xyz"]):
                p.X = X.copy()
            for fid in [f[0] for f in self.frames]:
                self.T_w[fid] = np.linalg.inv(se3.params_to_pose(res["cams"][fid % W]))
            # Result write-back, photobundle.cc:857-875: the points that leave the window (refFrameId <= frame_id_start)
            gone = [p for p in self.points if p.f[0] <= start]
            self.results.append(dict(n_points=len(sel), n_obs=len(obs_p), initial_cost=res["initial_cost"],
                                     final_cost=res["final_cost"], iterations=len(res["iterations"]),
                                     num_successful_steps=res["num_successful_steps"], num_residuals=res["num_residuals"],
                                     message=res["message"], it=res["iterations"], n_poses=len(self.T_w),
                                     refined=np.array([p.X for p in gone]).reshape(-1, 3),
                                     original=np.array([p.X0 for p in gone]).reshape(-1, 3)))
        self.points = [p for p in self.points if p.f[0] > start]


# ---- coarse-to-fine wrapper (intended behaviour of reference src/photobundle_pyramid.cc, see host/photobundle_pyramid.h) --
def _reflect101(i, n):
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    i = np.abs(i)
    return np.where(i >= n, 2 * n - 2 - i, i)


# cv::pyrDown / cv::resize restatements live with the rest of the oracle (oracle/oracle.py: pyr_down_u8, resize_bilinear_f32)
from oracle.oracle import pyr_down_u8, resize_bilinear_f32  # noqa: E402,F401


class PyramidEmulator:
    def __init__(self, levels, K, size, **kw):
        self.levels = levels
        self.emus = []
        fx, fy, cx, cy = K
        rows, cols = size
        for _ in range(levels):
            self.emus.append(Emulator((fx, fy, cx, cy), (rows, cols), **kw))
            fx, fy, cx, cy = 0.5 * fx, 0.5 * fy, 0.5 * cx, 0.5 * cy      # Calibration::pyrDown: K * 0.5, K(2,2) = 1
            rows, cols = (rows + 1) // 2, (cols + 1) // 2

    def add_frame(self, img, depth, T_local):
        ims, zs = [img], [depth.astype(np.float32)]
        for l in range(1, self.levels):
            ims.append(pyr_down_u8(ims[-1]))
            zs.append(resize_bilinear_f32(zs[-1], *ims[-1].shape))
        T = T_local
        for l in range(self.levels - 1, -1, -1):
            emu = self.emus[l]
            n_before = len(emu.results)
            emu.add_frame(ims[l], zs[l], T)
            if len(emu.results) > n_before and len(emu.T_w) >= 2:
                T = np.linalg.inv(emu.T_w[-1]) @ emu.T_w[-2]
        return self.emus[0]
