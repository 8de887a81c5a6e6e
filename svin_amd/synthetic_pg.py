"""Synthetic pose graphs for the global optimisation path (BASELINE config #5).

A robot flies `laps` laps of a closed 3-D curve; the "SVIn" poses handed to the pose graph are dead-reckoned from
noisy relative motions (they drift), loop edges connect a keyframe of a later lap to the keyframe of an earlier lap
at the same place, measured as PnP would deliver them (pose_graph/src/pose_graph/Keyframe.cpp:495-500):
    relative_t = R_old^T (t_cur - t_old),  relative_q = R_old^T R_cur,  relative_yaw = yaw_cur - yaw_old
from the TRUE geometry plus a little noise.  Quaternions are [x y z w]; yaw/pitch/roll in degrees (Utils.h:71-103).
"""
from dataclasses import dataclass

import numpy as np


def ypr2R(ypr_deg):
    y, p, r = np.deg2rad(ypr_deg)
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


def R2ypr(R):
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = np.arctan2(n[1], n[0])
    p = np.arctan2(-n[2], n[0] * np.cos(y) + n[1] * np.sin(y))
    r = np.arctan2(a[0] * np.sin(y) - a[1] * np.cos(y), -o[0] * np.sin(y) + o[1] * np.cos(y))
    return np.rad2deg(np.array([y, p, r]))


def R2q(R):
    """Eigen::Quaterniond(Matrix3d), [x y z w]."""
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


def q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@dataclass
class PoseGraphSpec:
    n: int
    t_true: np.ndarray      # n x 3
    R_true: np.ndarray      # n x 3 x 3
    t_svin: np.ndarray      # drifted input poses
    q_svin: np.ndarray      # n x 4 [x y z w]
    loops: dict             # index -> (loop_index, rel_t, rel_q, rel_yaw_deg)
    sequence: np.ndarray    # n ints


def make_pose_graph(n=400, laps=4, loop_every=25, seed=7, radius=8.0, drift_yaw_deg=0.03, drift_t=0.004,
                    loop_noise_t=0.002, loop_noise_deg=0.02):
    rng = np.random.default_rng(seed)
    per_lap = n // laps
    t_true = np.zeros((n, 3))
    R_true = np.zeros((n, 3, 3))
    for k in range(n):
        ph = 2 * np.pi * (k % per_lap) / per_lap
        t_true[k] = [radius * np.cos(ph), 0.6 * radius * np.sin(ph), 1.0 * np.sin(2 * ph)]
        heading = np.rad2deg(np.arctan2(0.6 * radius * np.cos(ph), -radius * np.sin(ph)))
        R_true[k] = ypr2R([heading, 4.0 * np.sin(ph), 3.0 * np.cos(2 * ph)])
    # dead reckoning with noisy relative motions
    t_svin = np.zeros((n, 3))
    R_svin = np.zeros((n, 3, 3))
    t_svin[0], R_svin[0] = t_true[0], R_true[0]
    for k in range(1, n):
        dR = R_true[k - 1].T @ R_true[k]
        dt = R_true[k - 1].T @ (t_true[k] - t_true[k - 1])
        dR = dR @ ypr2R(rng.normal(0, [drift_yaw_deg, 0.2 * drift_yaw_deg, 0.2 * drift_yaw_deg]))
        dt = dt + rng.normal(0, drift_t, 3)
        R_svin[k] = R_svin[k - 1] @ dR
        t_svin[k] = t_svin[k - 1] + R_svin[k - 1] @ dt
    q_svin = np.stack([R2q(R) for R in R_svin])
    loops = {}
    for k in range(per_lap, n):
        if (k % loop_every) != 0:
            continue
        old = k - per_lap * rng.integers(1, k // per_lap + 1)
        Ro, Rc = R_true[old], R_true[k] @ ypr2R(rng.normal(0, loop_noise_deg, 3))
        rel_t = Ro.T @ (t_true[k] - t_true[old]) + rng.normal(0, loop_noise_t, 3)
        rel_q = R2q(Ro.T @ Rc)
        yaw = R2ypr(Rc)[0] - R2ypr(Ro)[0]
        yaw = yaw - 360.0 if yaw > 180.0 else (yaw + 360.0 if yaw < -180.0 else yaw)
        loops[k] = (int(old), rel_t, rel_q, float(yaw))
    return PoseGraphSpec(n, t_true, R_true, t_svin, q_svin, loops, np.ones(n, int))


def feed(pg, spec, upto=None):
    upto = spec.n if upto is None else upto
    for k in range(upto):
        pg.add_keyframe(k, int(spec.sequence[k]), spec.t_svin[k], spec.q_svin[k], spec.loops.get(k))
    earliest = min((v[0] for k, v in spec.loops.items() if k < upto), default=0)
    return earliest, upto - 1


def align_error(T_est, spec, upto=None):
    """RMS position error after aligning the first optimised keyframe frame to the truth (the gauge is fixed there)."""
    upto = spec.n if upto is None else upto
    return float(np.sqrt(np.mean(np.sum((T_est[:upto] - spec.t_true[:upto]) ** 2, axis=1))))
