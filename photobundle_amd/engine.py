"""Python mirror of the engine's C-ABI (include/pba.h): thin, no compute.  The product path is
WindowProblem -> Engine.load() -> Engine.solve(); everything numerical happens in libpba_hip.so on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EngineUnavailable  # noqa: F401


class EngineError(RuntimeError):
    pass


def default_solver_options(**kw):
    o = _lib.SolverOptions()
    _lib.lib().pba_default_solver_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One engine = one GPU = one shard of points (all cameras and frames replicated)."""

    def __init__(self, rows, cols, K, radius, max_frames, huber=0.0, device=0, keep_reduced_system=False, precision="exact",
                 channels=1):
        self._L = _lib.lib()
        cfg = _lib.Config()
        cfg.rows, cfg.cols, cfg.max_frames, cfg.radius = int(rows), int(cols), int(max_frames), int(radius)
        cfg.fx, cfg.fy, cfg.cx, cfg.cy = [float(v) for v in K]
        cfg.huber = float(huber)
        cfg.device = int(device)
        # precision: "exact" (reference-exact sampler, default) | "fp32" | "bf16" (configs[4] tolerance sweep, include/pba.h)
        cfg.flags = (1 if keep_reduced_system else 0) | ({"exact": 0, "fp32": 1, "bf16": 2}[precision] << 1)
        cfg.channels = int(channels)        # descriptor channels (include/pba.h): > 1 takes float channel images per frame
        self._h = C.c_void_p()
        rc = self._L.pba_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise EngineError("pba_create: %s" % self._L.pba_status_string(rc).decode())
        self.cfg = cfg
        self.n_points = self.n_obs = self.n_frames = 0
        self._cb = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.pba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s: %s (%s)" % (what, self._L.pba_status_string(rc).decode(),
                                               self._L.pba_last_error(self._h).decode()))

    # ---- data ------------------------------------------------------------------------------------------
    def set_frame(self, slot, image_u8):
        img = np.ascontiguousarray(image_u8, dtype=np.uint8)
        assert img.shape == (self.cfg.rows, self.cfg.cols)
        self._check(self._L.pba_set_frame_u8(self._h, int(slot), _ptr(img)), "pba_set_frame_u8")

    def set_frame_channels(self, slot, channels_f32):
        ch = np.ascontiguousarray(channels_f32, dtype=np.float32)
        assert ch.shape == (self.cfg.channels, self.cfg.rows, self.cfg.cols)
        self._check(self._L.pba_set_frame_channels_f32(self._h, int(slot), ch.shape[0], _ptr(ch)), "pba_set_frame_channels_f32")

    DESCRIPTORS = {"Intensity": 0, "IntensityAndGradient": 1, "BitPlanes": 2}

    def set_frame_descriptor(self, slot, image_u8, kind, sigma_ct=1.0, sigma_bp=1.5):
        """The descriptor channels of `kind` built on the device from the u8 frame (pba_set_frame_descriptor_u8)."""
        img = np.ascontiguousarray(image_u8, dtype=np.uint8)
        assert img.shape == (self.cfg.rows, self.cfg.cols)
        self._check(self._L.pba_set_frame_descriptor_u8(self._h, int(slot), _ptr(img), self.DESCRIPTORS[kind], float(sigma_ct), float(sigma_bp)),
                    "pba_set_frame_descriptor_u8")

    def get_frame_channels(self, slot):
        """[C, rows, cols] f32: the channel images of a multi-channel slot (pba_get_frame_channels_f32)."""
        out = np.empty((self.cfg.channels, self.cfg.rows, self.cfg.cols), np.float32)
        self._check(self._L.pba_get_frame_channels_f32(self._h, int(slot), _ptr(out)), "pba_get_frame_channels_f32")
        return out

    def set_frame_pyr_down(self, slot, finer, finer_slot, want_image=True):
        """cv::pyrDown of a frame of the finer level's engine into this one, device to device; returns the u8 image."""
        out = np.empty((self.cfg.rows, self.cfg.cols), np.uint8) if want_image else None
        self._check(self._L.pba_set_frame_pyr_down(self._h, int(slot), finer._h, int(finer_slot), _ptr(out) if want_image else None),
                    "pba_set_frame_pyr_down")
        return out

    # ---- device front-end (pba_frontend_*: reference photobundle.cc:505-603 on the frame in the ring) --------------------------
    def frontend_visibility(self, uv, rc, patches26, min_score, mask_radius):
        """ZNCC test of tracked points on the most recently uploaded u8 frame; returns the hit flags [n] (and stamps the mask)."""
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        rc = np.ascontiguousarray(rc, dtype=np.int32).reshape(-1, 2)
        pt = np.ascontiguousarray(patches26, dtype=np.float32).reshape(-1, 26)
        n = len(uv)
        assert len(rc) == n and len(pt) == n
        hit = np.zeros(n, np.uint8)
        self._check(self._L.pba_frontend_visibility(self._h, n, _ptr(uv), _ptr(rc), _ptr(pt), float(min_score), int(mask_radius), _ptr(hit)),
                    "pba_frontend_visibility")
        return hit

    def frontend_zncc_probe(self, uv, patches26):
        """Test hook: [n, 27] = the new frame's zero-mean 5x5 patch at uv, its norm, the score against patches26."""
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        pt = np.ascontiguousarray(patches26, dtype=np.float32).reshape(-1, 26)
        out = np.zeros((len(uv), 27), np.float32)
        self._check(self._L.pba_frontend_zncc_probe(self._h, len(uv), _ptr(uv), _ptr(pt), _ptr(out)), "pba_frontend_zncc_probe")
        return out

    def frontend_candidates(self, slot, depth, min_depth, max_depth, nms_radius, border):
        """Candidate scan of the frame in `slot`; returns a structured array (saliency f4, x i4, y i4) in row-major order."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        assert depth.shape == (self.cfg.rows, self.cfg.cols)
        n = C.c_int32(0)
        self._check(self._L.pba_frontend_candidates(self._h, int(slot), _ptr(depth), float(min_depth), float(max_depth), int(nms_radius),
                                                    int(border), C.byref(n)), "pba_frontend_candidates")
        out = np.zeros(n.value, dtype=np.dtype([("saliency", np.float32), ("x", np.int32), ("y", np.int32)]))
        self._check(self._L.pba_frontend_get_candidates(self._h, _ptr(out) if n.value else None, n.value), "pba_frontend_get_candidates")
        return out

    def frontend_descriptors(self, slot, xy):
        """Integer-pixel patches at xy [n][2] = (x, y) of the frame in `slot`: [n, C, (2R+1)^2] f32."""
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        P = (2 * self.cfg.radius + 1) ** 2
        out = np.zeros((len(xy), max(1, self.cfg.channels), P), np.float32)
        self._check(self._L.pba_frontend_descriptors(self._h, int(slot), len(xy), _ptr(xy), _ptr(out)), "pba_frontend_descriptors")
        return out

    def get_frame_channel(self, slot, channel):
        out = np.empty((3, self.cfg.rows, self.cfg.cols), np.float32)
        self._check(self._L.pba_get_frame_channel(self._h, int(slot), int(channel), _ptr(out[0]), _ptr(out[1]), _ptr(out[2])),
                    "pba_get_frame_channel")
        return out

    def sample_frame(self, slot, y, x, channel=0):
        """The engine's sampler at float positions: [n, 3] = value, Gx, Gy (pba_sample_frame)."""
        y = np.ascontiguousarray(y, np.float32).ravel()
        x = np.ascontiguousarray(x, np.float32).ravel()
        assert y.shape == x.shape
        out = np.empty((y.shape[0], 3), np.float32)
        self._check(self._L.pba_sample_frame(self._h, int(slot), int(channel), y.shape[0], _ptr(y), _ptr(x), _ptr(out)), "pba_sample_frame")
        return out

    def get_frame_planes(self, slot):
        out = np.empty((3, self.cfg.rows, self.cfg.cols), np.float32)
        self._check(self._L.pba_get_frame_planes(self._h, int(slot), _ptr(out[0]), _ptr(out[1]), _ptr(out[2])),
                    "pba_get_frame_planes")
        return out

    def set_problem(self, xyz, desc, obs_point, obs_slot, weights):
        xyz = np.ascontiguousarray(xyz, np.float64)
        desc = np.ascontiguousarray(desc, np.float64)
        op = np.ascontiguousarray(obs_point, np.int32)
        os_ = np.ascontiguousarray(obs_slot, np.int32)
        w = np.ascontiguousarray(weights, np.float64)
        self._check(self._L.pba_set_problem(self._h, xyz.shape[0], _ptr(xyz), _ptr(desc), op.shape[0], _ptr(op),
                                            _ptr(os_), _ptr(w)), "pba_set_problem")
        self.n_points, self.n_obs = xyz.shape[0], op.shape[0]

    def set_cameras(self, cams, fixed_slot=0):
        cams = np.ascontiguousarray(cams, np.float64)
        self._check(self._L.pba_set_cameras(self._h, _ptr(cams), cams.shape[0], int(fixed_slot)), "pba_set_cameras")
        self.n_frames = cams.shape[0]
        self.n_free = self.n_frames - (1 if fixed_slot >= 0 else 0)

    def load(self, prob):
        """Uploads a WindowProblem (frames from prob.images)."""
        if self.cfg.channels > 1:
            assert prob.channel_images is not None and prob.channels == self.cfg.channels
            for s in range(prob.n_frames):
                self.set_frame_channels(s, prob.channel_images[s])
        else:
            assert prob.images is not None, "the engine consumes u8 frames (addFrame's input), not float planes"
            for s in range(prob.n_frames):
                self.set_frame(s, prob.images[s])
        self.set_problem(prob.xyz, prob.desc, prob.obs_point, prob.obs_slot, prob.weights)
        self.set_cameras(prob.cams, prob.fixed_slot)
        return self

    def set_inverse_depth(self, rays, rho):
        """Inverse-depth variant (include/pba.h): rays [n_points, 6] = world origin + direction, rho [n_points] > 0."""
        rays = np.ascontiguousarray(rays, np.float64)
        rho = np.ascontiguousarray(rho, np.float64)
        assert rays.shape == (self.n_points, 6) and rho.shape == (self.n_points,)
        self._check(self._L.pba_set_inverse_depth(self._h, _ptr(rays), _ptr(rho)), "pba_set_inverse_depth")

    def get_points_world(self):
        xyz = np.zeros((self.n_points, 3))
        self._check(self._L.pba_get_points_world(self._h, _ptr(xyz)), "pba_get_points_world")
        return xyz

    def get_state(self):
        cams = np.zeros((self.n_frames, 6))
        xyz = np.zeros((self.n_points, 3))
        self._check(self._L.pba_get_state(self._h, _ptr(cams), _ptr(xyz)), "pba_get_state")
        return cams, xyz

    # ---- primitive passes --------------------------------------------------------------------------------
    def linearize(self, want_cost=True):
        c = C.c_double(0.0)
        self._check(self._L.pba_linearize(self._h, C.byref(c) if want_cost else None), "pba_linearize")
        return c.value

    def step(self, radius, init_scale=False, options=None):
        o = options or default_solver_options()
        info = _lib.StepInfo()
        self._check(self._L.pba_step(self._h, float(radius), int(bool(init_scale)), C.byref(o), C.byref(info)), "pba_step")
        return {f: getattr(info, f) for f, _ in _lib.StepInfo._fields_}

    def accept(self):
        self._check(self._L.pba_accept(self._h), "pba_accept")

    def reduced_system(self):
        n = C.c_int32(0)
        nn = 6 * self.n_free
        S = np.zeros((nn, nn))
        rhs = np.zeros(nn)
        self._check(self._L.pba_get_reduced_system(self._h, _ptr(S), _ptr(rhs), C.byref(n)), "pba_get_reduced_system")
        assert n.value == nn
        return S, rhs

    def obs_records(self):
        rec = np.zeros((self.n_obs, 6))
        self._check(self._L.pba_get_obs_records(self._h, _ptr(rec)), "pba_get_obs_records")
        return rec

    # ---- the LM loop ---------------------------------------------------------------------------------------
    def solve(self, options=None, max_iterations_out=512, fetch_state=True):
        res = self.unpack_solve(*self.solve_raw(options, max_iterations_out))
        if fetch_state:
            res["cams"], res["xyz"] = self.get_state()
        return res

    def solve_raw(self, options=None, max_iterations_out=512, buffers=None):
        """pba_solve and nothing else: returns the C structs (summary, iteration array) as filled by the library.
        `buffers` = a (summary, iteration array) pair from an earlier call to reuse (timing loops)."""
        o = options or default_solver_options()
        s, its = buffers if buffers is not None else self.solve_buffers(max_iterations_out)
        self._check(self._L.pba_solve(self._h, C.byref(o), C.byref(s), its, len(its)), "pba_solve")
        return s, its

    @staticmethod
    def solve_buffers(max_iterations_out=512):
        return _lib.SolverSummary(), (_lib.IterationSummary * max_iterations_out)()

    @staticmethod
    def unpack_solve(s, its):
        res = {f: getattr(s, f) for f, _ in _lib.SolverSummary._fields_}
        res["message"] = s.message.decode()
        res["iterations"] = [{f: getattr(its[i], f) for f, _ in _lib.IterationSummary._fields_}
                             for i in range(s.num_iterations)]
        return res

    # ---- multi-GPU -------------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * 128)()
        rc = _lib.lib().pba_comm_unique_id(buf)
        if rc != 0:
            raise EngineError("pba_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf)

    def comm_init_rccl(self, unique_id, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._check(self._L.pba_comm_init_rccl(self._h, buf, int(rank), int(world)), "pba_comm_init_rccl")

    def comm_enable_peer_exchange(self):
        """Collective.  Returns the transport name afterwards ("rccl+peer" / "callback+peer" when the device-side exchange is on)."""
        self._check(self._L.pba_comm_enable_peer_exchange(self._h), "pba_comm_enable_peer_exchange")
        return self.comm_transport()

    def solve_driver(self):
        """Which driver ran the last solve: "resident" (one cooperative launch), "pipelined", "host-stepped" or "none"."""
        return self._L.pba_solve_driver(self._h).decode()

    def comm_rank_count(self):
        return int(self._L.pba_comm_rank_count(self._h))

    def comm_transport(self):
        self._L.pba_comm_transport.restype = C.c_char_p
        return self._L.pba_comm_transport(self._h).decode()

    def comm_init_callback(self, fn, rank, world):
        """fn(numpy float64 view, op) must all-reduce in place (op 0 = sum, 1 = max)."""
        def tramp(ptr, n, op, ctx):
            try:
                a = np.ctypeslib.as_array(ptr, shape=(n,))
                fn(a, int(op))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _lib.ALLREDUCE_FN(tramp)
        self._check(self._L.pba_comm_init_callback(self._h, self._cb, None, int(rank), int(world)), "pba_comm_init_callback")

    # ---- counters ----------------------------------------------------------------------------------------------
    def reset_counters(self):
        self._check(self._L.pba_reset_counters(self._h), "pba_reset_counters")

    def set_profiling(self, mode):
        """0 off, 1 HIP events around every kernel (host-stepped driver), 2 device time stamps in the asynchronous pipeline."""
        self._check(self._L.pba_set_profiling(self._h, int(mode)), "pba_set_profiling")

    def counters(self):
        c = _lib.Counters()
        self._check(self._L.pba_get_counters(self._h, C.byref(c)), "pba_get_counters")
        return {f: getattr(c, f) for f, _ in _lib.Counters._fields_}
