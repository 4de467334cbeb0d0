"""Host-side container for one sliding-window problem at the inner seam the engine replaces
(reference src/photobundle.cc:784-829: what optimize() hands to ceres::Problem / ceres::Solve)."""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class WindowProblem:
    """Everything `optimize()` assembles before the solve.

    K            (fx, fy, cx, cy)                         calibration.h:34-38
    images       [n_frames, rows, cols] u8                the frames of the window (DescriptorFrame channel 0 source)
    planes       [n_frames, 3, rows, cols] f32            I, Gx, Gy (photobundle.cc:231, :172-175; imgproc.cc:27-95)
    cams         [n_frames, 6] f64                        angle-axis + t of the world->camera pose (photobundle.cc:774-778)
    xyz          [n_points, 3] f64                        world points (photobundle.cc:795)
    desc         [n_points, P] f64                        reference descriptors (photobundle.cc:597-603)
    obs_point    [n_obs] i32, obs_slot [n_obs] i32        one residual block per entry, grouped by point (:791-804)
    weights      [P] f64                                  MakePatchWeights (:617-644)
    huber        robustThreshold (:797-798), <= 0 disables
    fixed_slot   the constant camera (:809-813)
    """
    K: tuple
    radius: int
    planes: np.ndarray
    cams: np.ndarray
    xyz: np.ndarray
    desc: np.ndarray
    obs_point: np.ndarray
    obs_slot: np.ndarray
    weights: np.ndarray
    huber: float = 0.0
    fixed_slot: int = 0
    images: np.ndarray = None
    meta: dict = field(default_factory=dict)
    channels: int = 1            # descriptor channels C: planes is [n_frames, 3C, rows, cols], desc [n_points, C P]
    channel_images: np.ndarray = None   # [n_frames, C, rows, cols] f32, what pba_set_frame_channels_f32 consumes (C > 1)

    @property
    def n_frames(self):
        return self.planes.shape[0]

    @property
    def n_points(self):
        return self.xyz.shape[0]

    @property
    def n_obs(self):
        return self.obs_point.shape[0]

    @property
    def patch_len(self):
        return (2 * self.radius + 1) ** 2

    def shard(self, rank, world):
        """Point-sharded view for rank `rank` of `world`: contiguous point ranges balanced by
        observation count; cameras, K, weights and frames are replicated."""
        lo, hi = shard_bounds(self.obs_point, self.n_points, rank, world)
        o_lo = int(np.searchsorted(self.obs_point, lo, side="left"))
        o_hi = int(np.searchsorted(self.obs_point, hi, side="left"))
        return WindowProblem(
            K=self.K, radius=self.radius, planes=self.planes, cams=self.cams.copy(),
            xyz=self.xyz[lo:hi].copy(), desc=self.desc[lo:hi], obs_point=(self.obs_point[o_lo:o_hi] - lo).astype(np.int32),
            obs_slot=self.obs_slot[o_lo:o_hi], weights=self.weights, huber=self.huber, fixed_slot=self.fixed_slot,
            images=self.images, meta=dict(self.meta, shard=(rank, world), point_range=(lo, hi)), channels=self.channels,
            channel_images=self.channel_images)


def shard_bounds(obs_point, n_points, rank, world):
    """[lo, hi) point range of `rank`: split the observation list into `world` equal parts and cut at point
    boundaries, so every rank gets whole points and about the same number of residual blocks."""
    obs_point = np.asarray(obs_point)
    n_obs = obs_point.shape[0]
    if world <= 1:
        return 0, n_points

    def cut(r):
        if r <= 0:
            return 0
        if r >= world:
            return n_points
        o = (n_obs * r) // world
        if o >= n_obs:
            return n_points
        return int(obs_point[o])
    return cut(rank), cut(rank + 1)
