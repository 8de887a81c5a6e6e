"""Seeded synthetic sliding windows.

Pure numpy data generation: a smooth trajectory, an ideal IMU sampled from it (plus the
TestEstimator-style noise model), a stereo rig with the constants of the shipped YAML files
(/root/reference/config/config_fpga_p2_euroc.yaml, config_stereorig_v2.yaml -- values only),
landmarks in the union of the frusta and their noisy projections.  The same `WindowSpec`
drives the oracle, the product and the benchmark through `feed()`.
"""
from dataclasses import dataclass, field
import time

import numpy as np

DIST_NONE, DIST_RADTAN, DIST_EQUIDISTANT, DIST_RADTAN8 = 0, 1, 2, 3


# ----------------------------------------------------------------------------- small SE(3) helpers
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def rotvec_to_quat(a):
    th = np.linalg.norm(a)
    if th < 1e-12:
        return np.array([0.5 * a[0], 0.5 * a[1], 0.5 * a[2], 1.0])
    return np.r_[np.sin(th / 2) * a / th, np.cos(th / 2)]


def T_from_matrix(M):
    M = np.asarray(M, float).reshape(4, 4)
    return np.r_[M[:3, 3], R_to_quat(M[:3, :3])]


def pose_oplus(T, delta):
    """okvis Transformation::oplus: r += dr; q = dq(dalpha) * q."""
    q = quat_mul(rotvec_to_quat(np.asarray(delta[3:])), T[3:])
    return np.r_[T[:3] + delta[:3], q / np.linalg.norm(q)]


def pose_inverse_apply(T, p_w):
    R = quat_to_R(T[3:])
    return (p_w - T[:3]) @ R  # == R^T (p - r) for row-vectors


# ----------------------------------------------------------------------------- camera model (numpy, vectorised)
def distort(model, k, u):
    u0, u1 = u[..., 0], u[..., 1]
    if model == DIST_NONE:
        return u, np.ones(u.shape[:-1], bool)
    if model == DIST_RADTAN:
        k1, k2, p1, p2 = k[:4]
        rho = u0 * u0 + u1 * u1
        rad = k1 * rho + k2 * rho * rho
        d0 = u0 + u0 * rad + 2 * p1 * u0 * u1 + p2 * (rho + 2 * u0 * u0)
        d1 = u1 + u1 * rad + 2 * p2 * u0 * u1 + p1 * (rho + 2 * u1 * u1)
        return np.stack([d0, d1], -1), np.ones(u.shape[:-1], bool)
    if model == DIST_EQUIDISTANT:
        k1, k2, k3, k4 = k[:4]
        r = np.sqrt(u0 * u0 + u1 * u1)
        th = np.arctan(r)
        th2 = th * th
        thd = th * (1 + k1 * th2 + k2 * th2 ** 2 + k3 * th2 ** 3 + k4 * th2 ** 4)
        s = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
        return np.stack([s * u0, s * u1], -1), np.ones(u.shape[:-1], bool)
    if model == DIST_RADTAN8:
        k1, k2, p1, p2, k3, k4, k5, k6 = k[:8]
        rho = u0 * u0 + u1 * u1
        rad = (1 + ((k3 * rho + k2) * rho + k1) * rho) / (1 + ((k6 * rho + k5) * rho + k4) * rho)
        d0 = u0 * rad + 2 * p1 * u0 * u1 + p2 * (rho + 2 * u0 * u0)
        d1 = u1 * rad + 2 * p2 * u0 * u1 + p1 * (rho + 2 * u1 * u1)
        return np.stack([d0, d1], -1), rho <= 9.0
    raise ValueError(model)


def project(cam, p_c):
    """p_c: (..., 3) points in the camera frame -> (uv, visible)"""
    z = p_c[..., 2]
    zs = np.where(np.abs(z) < 1e-12, 1.0, z)
    u = p_c[..., :2] / zs[..., None]
    d, ok = distort(cam["model"], np.asarray(cam["dist"], float), u)
    uv = np.stack([cam["intr"][0] * d[..., 0] + cam["intr"][2], cam["intr"][1] * d[..., 1] + cam["intr"][3]], -1)
    vis = ok & (z > 0.2) & (uv[..., 0] >= 0) & (uv[..., 0] < cam["width"]) & (uv[..., 1] >= 0) & (uv[..., 1] < cam["height"])
    # radial-tangential models fold back far outside the image: restrict to a sane field of view
    vis &= (np.abs(u[..., 0]) < 1.2) & (np.abs(u[..., 1]) < 1.0)
    return uv, vis


# ----------------------------------------------------------------------------- rigs
def euroc_rig():
    c0 = dict(model=DIST_RADTAN, intr=[458.654880721, 457.296696463, 367.215803962, 248.37534061],
              dist=[-0.28340811217, 0.0739590738929, 0.000193595028569, 1.76187114545e-05], width=752, height=480,
              T_SC=T_from_matrix([0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975,
                                  0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
                                  -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949,
                                  0, 0, 0, 1]))
    c1 = dict(model=DIST_RADTAN, intr=[457.587426604, 456.13442556, 379.99944652, 255.238185386],
              dist=[-0.283683654496, 0.0745128430929, -0.000104738949098, -3.55590700274e-05], width=752, height=480,
              T_SC=T_from_matrix([0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556,
                                  0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024,
                                  -0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038,
                                  0, 0, 0, 1]))
    imu = dict(a_max=176.0, g_max=7.8, sigma_g_c=12.0e-4, sigma_a_c=8.0e-3, sigma_bg=0.03, sigma_ba=0.1,
               sigma_gw_c=4.0e-6, sigma_aw_c=4.0e-5, tau=3600.0, g=9.81007, a0=[0.0, 0.0, 0.0], rate=200)
    return [c0, c1], imu, [0.0, 0.0, 0.0, 0.0]


def rig_v2():
    c0 = dict(model=DIST_RADTAN, intr=[1156.5188534683703, 1156.5772666173266, 763.2083316463371, 617.2779853849335],
              dist=[-0.17473019446863114, 0.10643290428040156, 0.005210777878907861, -0.00028664918860079295],
              width=1600, height=1200,
              T_SC=T_from_matrix([-0.999951484924370, -0.007271683630496, -0.006644577844233, 0.095860371617174,
                                  0.007454284822327, -0.999583407112654, -0.027882716202835, -0.002288837387091,
                                  -0.006439055469378, -0.027930894046525, 0.999589117448978, -0.023754113917685,
                                  0, 0, 0, 1]))
    c1 = dict(model=DIST_RADTAN, intr=[1158.625855755729, 1156.0604864187183, 765.5609812846063, 588.6683184401453],
              dist=[-0.173831269260396, 0.10747272137157605, 0.004231076633773206, -0.0026692219494915187],
              width=1600, height=1200,
              T_SC=T_from_matrix([-0.999982297266498, -0.005818312404541, -0.001245951195096, -0.043066851727302,
                                  0.005847412971625, -0.999675635403190, -0.024787733715850, -6.630840216796008e-4,
                                  -0.001101324274080, -0.024794580496386, 0.999691960487255, -0.023978804210993,
                                  0, 0, 0, 1]))
    imu = dict(a_max=176.0, g_max=7.8, sigma_g_c=0.0016017, sigma_a_c=0.0071376, sigma_bg=0.03, sigma_ba=0.1,
               sigma_gw_c=0.0000165, sigma_aw_c=0.0002874, tau=3600.0, g=9.81007, a0=[0.0, 0.0, 0.0], rate=100)
    return [c0, c1], imu, [0.0, 0.0, 1.0e-8, 1.0e-8]


T_SSO_RIG_V2 = T_from_matrix([0.0, 1.0, 0.0, 0.015995, -1.0, 0.0, 0.0, 0.125, 0.0, 0.0, 1.0, 0.128, 0, 0, 0, 1])


def test_rig(extr_case=0):
    """The rig of okvis_ceres/test/TestEstimator.cpp:52-214 (equidistant test cameras, identity / 0.1 m baseline)."""
    cams = []
    for r in ([0.0, 0.0, 0.0], [0.0, 0.1, 0.0]):
        cams.append(dict(model=DIST_EQUIDISTANT, intr=[350.0, 360.0, 378.0, 238.0], dist=[-0.21, 0.14, 0.0006, 0.0003],
                         width=752, height=480, T_SC=np.r_[r, 0.0, 0.0, 0.0, 1.0]))
    imu = dict(a_max=1000.0, g_max=1000.0, sigma_g_c=6.0e-4, sigma_a_c=2.0e-3, sigma_bg=0.03, sigma_ba=0.1,
               sigma_gw_c=3.0e-6, sigma_aw_c=2.0e-5, tau=3600.0, g=9.81, a0=[0.0, 0.0, 0.0], rate=100)
    c = extr_case
    sig = [1.0e-3 * (c % 2), 1.0e-4 * (c % 2), 1e-8 * (c // 2), 1e-7 * (c // 2)]
    if c == 4:  # numerically benign online-calibration case (the 1e-8 cases carry 1e16-scale information)
        sig = [1.0e-3, 1.0e-4, 1.0e-4, 1.0e-3]
    return cams, imu, sig


# ----------------------------------------------------------------------------- trajectory + IMU
class Trajectory:
    """p(t): constant velocity plus gentle sinusoids; attitude: level start, sinusoidal roll/pitch/yaw (<= ~0.2 rad)."""

    def __init__(self, g, speed=1.0, rot_amp=0.2, wobble=0.3, axis=0):
        self.g = g
        self.speed = speed
        self.rot_amp = rot_amp
        self.wobble = wobble
        self.axis = axis   # world axis of the main motion: 0 = sideways to the cameras, 2 = along the optical axes

    def p(self, t):
        w = self.wobble
        return np.roll(np.array([self.speed * t + w * np.sin(0.7 * t), w * np.sin(0.9 * t + 0.3) - w * np.sin(0.3),
                                 0.5 * w * np.sin(1.1 * t)]), self.axis)

    def R(self, t):
        a = self.rot_amp
        roll, pitch, yaw = a * np.sin(0.8 * t), 0.7 * a * np.sin(0.6 * t + 0.5) - 0.7 * a * np.sin(0.5), a * np.sin(0.5 * t)
        cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
        return Rz @ Ry @ Rx

    def v(self, t, h=1e-5):
        return (self.p(t + h) - self.p(t - h)) / (2 * h)

    def acc(self, t, h=1e-4):
        return (self.p(t + h) - 2 * self.p(t) + self.p(t - h)) / (h * h)

    def omega_body(self, t, h=1e-5):
        Rd = (self.R(t + h) - self.R(t - h)) / (2 * h)
        W = self.R(t).T @ Rd
        return np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) * 0.5

    def imu(self, t):
        """ideal (gyro, accel) in the sensor frame: a_S = C_SW (p'' + g e_z)."""
        return self.omega_body(t), self.R(t).T @ (self.acc(t) + np.array([0.0, 0.0, self.g]))

    def T_WS(self, t):
        return np.r_[self.p(t), R_to_quat(self.R(t))]


@dataclass
class WindowSpec:
    cameras: list
    extr_sigmas: list
    imu_params: dict
    stamps: np.ndarray            # (P, 2) uint32
    T_WS_true: np.ndarray         # (P, 7)
    sb_true: np.ndarray           # (P, 9)
    T_WS_init: np.ndarray         # (P, 7) perturbed initial guesses
    sb_init: np.ndarray
    keyframe: np.ndarray          # (P,) bool
    imu_t: np.ndarray             # (M, 2) uint32
    imu_meas: np.ndarray          # (M, 6)
    lm_true: np.ndarray           # (L, 4)
    lm_init: np.ndarray           # (L, 4)
    obs_lm: np.ndarray            # (N,) landmark index
    obs_frame: np.ndarray         # (N,) frame index
    obs_cam: np.ndarray           # (N,) camera index
    obs_uv: np.ndarray            # (N, 2)
    obs_size: np.ndarray          # (N,)
    sonar: list = field(default_factory=list)   # per frame: None or (range, heading)
    depth: list = field(default_factory=list)   # per frame: None or depth value
    first_depth: float = 0.0
    T_SSo: np.ndarray = None
    seed: int = 0

    @property
    def P(self):
        return len(self.stamps)

    @property
    def L(self):
        return len(self.lm_true)

    @property
    def N(self):
        return len(self.obs_lm)


def stamp_of(t0_sec, t):
    ns = int(round(t * 1e9))
    return np.array([t0_sec + ns // 1_000_000_000, ns % 1_000_000_000], np.uint32)


def make_window(P=10, L=2000, n_obs=20000, seed=20250629, rig="euroc", frame_dt=0.5, pixel_noise=1.0, kp_size=8.0,
                imu_noise=True, pose_noise=(0.05, 0.01), lm_noise=0.1, depth_range=(2.0, 15.0), sonar=False, depth=False,
                keyframe_every=1, t0_sec=1000, traj=None, sonar_patch=(8, 30)):
    """Build one seeded synthetic window (config #2 defaults: 10 KF / 2 000 landmarks / 20 000 residuals).

    imu_noise: True = discrete white noise sigma/sqrt(dt); "testestimator" = the reference test's model
    (uniform[-1,1] * sigma * sqrt(dt), TestEstimator.cpp:90-96); False = none.  traj: Trajectory kwargs."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if rig == "euroc":
        cams, imu_params, sig = euroc_rig()
    elif rig == "rig_v2":
        cams, imu_params, sig = rig_v2()
    elif rig.startswith("test"):
        cams, imu_params, sig = test_rig(int(rig[4:] or 0))
    else:
        raise ValueError(rig)
    if traj is None and sonar:
        traj = dict(axis=2)   # along the optical axes: the side-looking sonar sweeps what the cameras saw earlier
    traj = Trajectory(imu_params["g"], **(traj or {}))
    rate = imu_params["rate"]
    times = np.arange(P) * frame_dt
    # IMU stream covering [ -2/rate, T + 2/rate ]
    n_imu = int(round(times[-1] * rate)) + 5
    imu_times = (np.arange(n_imu) - 2) / rate
    imu_meas = np.zeros((n_imu, 6))
    dt = 1.0 / rate
    for i, t in enumerate(imu_times):
        w, a = traj.imu(t)
        if imu_noise == "testestimator":
            w = w + rng.uniform(-1, 1, 3) * imu_params["sigma_g_c"] * np.sqrt(dt)
            a = a + rng.uniform(-1, 1, 3) * imu_params["sigma_a_c"] * np.sqrt(dt)
        elif imu_noise:
            w = w + rng.normal(size=3) * imu_params["sigma_g_c"] / np.sqrt(dt)
            a = a + rng.normal(size=3) * imu_params["sigma_a_c"] / np.sqrt(dt)
        imu_meas[i, :3], imu_meas[i, 3:] = w, a
    imu_t = np.stack([stamp_of(t0_sec + 1, t) for t in imu_times])  # +1 s so that t=-2/rate stays positive
    stamps = np.stack([stamp_of(t0_sec + 1, t) for t in times])
    T_true = np.stack([traj.T_WS(t) for t in times])
    sb_true = np.zeros((P, 9))
    for k, t in enumerate(times):
        sb_true[k, :3] = traj.v(t)
    # landmarks: sample in the frustum of camera 0 of a random frame
    lm = np.zeros((L, 4))
    for l in range(L):
        k = rng.integers(P)
        cam = cams[0]
        uvn = np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.45, 0.45)])
        z = rng.uniform(*depth_range)
        p_c = np.r_[uvn * z, z]
        p_s = quat_to_R(cam["T_SC"][3:]) @ p_c + cam["T_SC"][:3]
        lm[l, :3] = quat_to_R(T_true[k, 3:]) @ p_s + T_true[k, :3]
        lm[l, 3] = 1.0
    # sonar (config #3): every frame gets one (range, heading) return and a *visual patch* -- a cluster of landmarks
    # around the sonar point T_WS * T_SSo * [r cos h, r sin h, 0] (Estimator.cpp:265-316 gathers the landmarks inside a
    # +-0.1 m box around that point, evaluated at the IMU-predicted pose).  The patch landmarks take the place of the
    # last ones of the table, so L stays what the caller asked for.
    sonar_meas, patch_slots = [None] * P, np.zeros(0, int)
    if sonar:
        n_patch = [int(rng.integers(sonar_patch[0], sonar_patch[1] + 1)) for _ in range(P)]
        assert sum(n_patch) < L, "not enough landmarks for the sonar patches"
        slot = L - sum(n_patch)
        patch_slots = np.arange(slot, L)
        for k in range(P):
            rge, hdg = rng.uniform(0.6, 1.6), rng.uniform(-np.pi, np.pi)
            R_ws = quat_to_R(T_true[k, 3:])
            p_s = quat_to_R(T_SSO_RIG_V2[3:]) @ np.array([rge * np.cos(hdg), rge * np.sin(hdg), 0.0]) + T_SSO_RIG_V2[:3]
            p_w = R_ws @ p_s + T_true[k, :3]
            lm[slot:slot + n_patch[k], :3] = p_w + rng.uniform(-0.03, 0.03, size=(n_patch[k], 3))
            slot += n_patch[k]
            sonar_meas[k] = (float(rge + 0.005 * rng.normal()), float(hdg))
    # observations: all successful projections, landmark-major, truncated to n_obs
    obs = []
    for k in range(P):
        Rws = quat_to_R(T_true[k, 3:])
        p_s = (lm[:, :3] - T_true[k, :3]) @ Rws
        for c, cam in enumerate(cams):
            Rsc = quat_to_R(cam["T_SC"][3:])
            p_c = (p_s - cam["T_SC"][:3]) @ Rsc
            uv, vis = project(cam, p_c)
            idx = np.nonzero(vis)[0]
            obs.append(np.stack([idx.astype(float), np.full(len(idx), float(k)), np.full(len(idx), float(c)), uv[idx, 0],
                                 uv[idx, 1]], 1))
    obs = np.concatenate(obs)
    order = np.lexsort((obs[:, 2], obs[:, 1], obs[:, 0]))
    obs = obs[order]
    if n_obs is not None and len(obs) > n_obs:
        keep = np.sort(rng.choice(len(obs), n_obs, replace=False))
        obs = obs[keep]
    N = len(obs)
    uv = obs[:, 3:5] + pixel_noise * rng.normal(size=(N, 2))
    # keypoints originate as float32 in the reference (cv::KeyPoint)
    uv = uv.astype(np.float32).astype(np.float64)
    T_init = T_true.copy()
    sb_init = sb_true.copy()
    for k in range(P):
        d = np.r_[rng.normal(size=3) * pose_noise[0], rng.normal(size=3) * pose_noise[1]]
        if k == 0:
            d[:] = 0  # first pose is pinned by the 1e8 prior at its initial value
        T_init[k] = pose_oplus(T_true[k], d)
        sb_init[k, :3] += rng.normal(size=3) * 0.02
    lm_init = lm.copy()
    lm_init[:, :3] += lm_noise * rng.normal(size=(L, 3))
    if len(patch_slots):   # the patch landmarks are close to the vehicle and well triangulated: cm-level error
        lm_init[patch_slots, :3] = lm[patch_slots, :3] + min(lm_noise, 0.01) * rng.normal(size=(len(patch_slots), 3))
    spec = WindowSpec(cameras=cams, extr_sigmas=sig, imu_params=imu_params, stamps=stamps, T_WS_true=T_true,
                      sb_true=sb_true, T_WS_init=T_init, sb_init=sb_init,
                      keyframe=(np.arange(P) % keyframe_every == 0), imu_t=imu_t, imu_meas=imu_meas, lm_true=lm,
                      lm_init=lm_init, obs_lm=obs[:, 0].astype(np.int64), obs_frame=obs[:, 1].astype(np.int64),
                      obs_cam=obs[:, 2].astype(np.int64), obs_uv=uv, obs_size=np.full(N, kp_size), seed=seed)
    spec.sonar = [None] * P
    spec.depth = [None] * P
    if depth:
        spec.first_depth = 0.0
        # DepthError: e = z_WS - (first_depth - depth)  ->  depth = first_depth - z
        spec.depth = [float(spec.first_depth - T_true[k, 2] + 0.01 * rng.normal()) for k in range(P)]
    if sonar:
        spec.T_SSo = T_SSO_RIG_V2.copy()
        spec.sonar = sonar_meas
    return spec


def feed(est, spec, optimize_each=0, perturb=True, frames=None, on_frame=None, timing=None):
    """Drive an estimator-like object (oracle or product) with a WindowSpec.

    The estimator API is the okvis::Estimator mirror: new_id / add_camera / add_imu / add_states /
    add_landmark / add_observation / set_T_WS / set_speed_and_bias / optimize.
    Returns (frame_ids, landmark_ids).
    """
    for cam in spec.cameras:
        est.add_camera(cam["model"], cam["intr"], cam["dist"], cam["width"], cam["height"], spec.extr_sigmas)
    est.add_imu(spec.imu_params)
    if spec.T_SSo is not None:
        est.set_sonar_extrinsics(spec.T_SSo)
    T_SC = np.stack([c["T_SC"] for c in spec.cameras])
    lm_ids = [est.new_id() for _ in range(spec.L)]
    for l in range(spec.L):
        est.add_landmark(lm_ids[l], spec.lm_init[l] if perturb else spec.lm_true[l])
    by_frame = [np.nonzero(spec.obs_frame == k)[0] for k in range(spec.P)]
    frame_ids = []
    kp_counter = {}
    imu_sec = spec.imu_t[:, 0].astype(np.float64) - float(spec.imu_t[0, 0]) + 1e-9 * spec.imu_t[:, 1]
    frm_sec = spec.stamps[:, 0].astype(np.float64) - float(spec.imu_t[0, 0]) + 1e-9 * spec.stamps[:, 1]
    margin = 2.5 / spec.imu_params["rate"]
    # The sonar patch of a frame is selected inside add_states at the IMU-predicted pose (Estimator.cpp:265-316), i.e.
    # from the previous frame's *current* estimate.  The reference runs with optimised states at that point; in batch
    # mode (all frames added, then one optimize) the states therefore stay at the truth while the window is built and
    # the seeded perturbation is applied once every frame is in.
    has_sonar = any(x is not None for x in (spec.sonar or []))
    defer = perturb and has_sonar and on_frame is None and not optimize_each
    for k in range(spec.P if frames is None else frames):
        fid = est.new_id()
        frame_ids.append(fid)
        # the deque handed to addStates covers [previous frame, this frame] with a small margin
        lo = frm_sec[k - 1] - margin if k > 0 else frm_sec[0] - margin
        sel = (imu_sec >= lo) & (imu_sec <= frm_sec[k] + margin)
        imu_t_k, imu_m_k = spec.imu_t[sel], spec.imu_meas[sel]
        son = [spec.sonar[k]] if spec.sonar and spec.sonar[k] is not None else None
        dep = [spec.depth[k]] if spec.depth and spec.depth[k] is not None else None
        t0 = time.perf_counter()
        ok = est.add_states(fid, (int(spec.stamps[k, 0]), int(spec.stamps[k, 1])), 400, T_SC, imu_t_k, imu_m_k,
                            bool(spec.keyframe[k]), son, dep, spec.first_depth)
        t1 = time.perf_counter()
        assert ok, "add_states failed for frame %d" % k
        if perturb and not defer:
            if k > 0:
                est.set_T_WS(fid, spec.T_WS_init[k])
            est.set_speed_and_bias(fid, spec.sb_init[k])
        else:
            if k > 0:
                est.set_T_WS(fid, spec.T_WS_true[k])
            est.set_speed_and_bias(fid, spec.sb_true[k])
        if timing is not None:
            timing.setdefault("add_states_s", []).append(t1 - t0)
            timing.setdefault("set_states_s", []).append(time.perf_counter() - t1)
        if hasattr(est, "add_observations") and len(by_frame[k]):   # the product's batched form of the same calls
            idx = by_frame[k]
            cams = spec.obs_cam[idx].astype(np.uint64)
            kps = np.zeros(len(idx), np.uint64)
            for c in np.unique(cams):
                m = cams == c
                kps[m] = np.arange(int(m.sum()), dtype=np.uint64)
            lids = np.array([lm_ids[int(j)] for j in spec.obs_lm[idx]], np.uint64)
            t0 = time.perf_counter()
            est.add_observations(lids, np.full(len(idx), fid, np.uint64), cams, kps, spec.obs_uv[idx], spec.obs_size[idx])
            if timing is not None:
                timing.setdefault("add_observations_s", []).append(time.perf_counter() - t0)
                timing.setdefault("add_observations_n", []).append(len(idx))
        else:
            t0 = time.perf_counter()
            for i in by_frame[k]:
                c = int(spec.obs_cam[i])
                kp = kp_counter.get((k, c), 0)
                kp_counter[(k, c)] = kp + 1
                est.add_observation(lm_ids[int(spec.obs_lm[i])], fid, c, kp, spec.obs_uv[i], float(spec.obs_size[i]))
            if timing is not None:   # (includes this loop's own Python overhead)
                timing.setdefault("add_observations_s", []).append(time.perf_counter() - t0)
                timing.setdefault("add_observations_n", []).append(len(by_frame[k]))
        if optimize_each:
            est.optimize(optimize_each, 1, False)
        if on_frame is not None:
            on_frame(k, fid)
    if defer:
        for k, fid in enumerate(frame_ids):
            if k > 0:
                est.set_T_WS(fid, spec.T_WS_init[k])
            est.set_speed_and_bias(fid, spec.sb_init[k])
    return frame_ids, lm_ids
