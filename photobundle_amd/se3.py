"""Pose <-> parameter conversions of the hot path in numpy (host side, fp64).

reference: src/photobundle.cc:646-667 (PoseToParams / ParamsToPose) with the Ceres rotation.h conventions
(matrix -> quaternion -> angle-axis, theta^2 > eps branch), src/trajectory.cc:7-16 (local -> world chaining)."""
import numpy as np

_EPS = np.finfo(np.float64).eps


def angle_axis_to_matrix(aa):
    aa = np.asarray(aa, dtype=np.float64)
    t2 = float(aa @ aa)
    if t2 > _EPS:
        t = np.sqrt(t2)
        w = aa / t
        c, s = np.cos(t), np.sin(t)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        return c * np.eye(3) + s * K + (1 - c) * np.outer(w, w)
    return np.array([[1, -aa[2], aa[1]], [aa[2], 1, -aa[0]], [-aa[1], aa[0], 1]])


def matrix_to_angle_axis(R):
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr >= 0.0:
        t = np.sqrt(tr + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    s2 = q[1] ** 2 + q[2] ** 2 + q[3] ** 2
    if s2 > 0.0:
        s = np.sqrt(s2)
        two_theta = 2.0 * (np.arctan2(-s, -q[0]) if q[0] < 0.0 else np.arctan2(s, q[0]))
        k = two_theta / s
    else:
        k = 2.0
    return q[1:] * k


def pose_to_params(T):
    """4x4 -> [w(3), t(3)]."""
    T = np.asarray(T, dtype=np.float64)
    return np.concatenate([matrix_to_angle_axis(T[:3, :3]), T[:3, 3]])


def params_to_pose(p):
    T = np.eye(4)
    T[:3, :3] = angle_axis_to_matrix(p[:3])
    T[:3, 3] = p[3:6]
    return T


def chain_local_poses(T_local):
    """trajectory.cc:7-16: T_w_i = T_w_{i-1} * inv(T_i), T_w_0 = inv(T_0)."""
    out = []
    for T in T_local:
        Ti = np.linalg.inv(T)
        out.append(Ti if not out else out[-1] @ Ti)
    return out
