"""The cost of tests/golden/tiny_window.npz in plain numpy, written from the definitions (pinhole + radial-tangential distortion,
ReprojectionError e = z - h(T_SC^-1 T_WS^-1 p) weighted by sqrt(information), Ceres' CauchyLoss(1) per 2-vector block, cost =
sum 0.5 rho) -- no code of the oracle or of the library in it.  Used to check solutions on reduced pose manifolds
(okvis_ceres/src/PoseManifold.cpp:173-466) for STATIONARITY in the directions the manifold leaves free."""
import numpy as np


def quat_mul(a, b):   # (x, y, z, w), Hamilton product a * b
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def rot(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def exp_q(a):   # unit quaternion of the rotation vector a
    t = np.linalg.norm(a)
    s = 0.5 if t < 1e-12 else np.sin(0.5 * t) / t
    return np.r_[s * np.asarray(a, float), np.cos(0.5 * t)]


def pose_oplus(T, d6):
    """Transformation::oplus (kinematics/implementation/Transformation.hpp:217-228): r += dr, q <- exp(dalpha) * q"""
    q = quat_mul(exp_q(d6[3:]), T[3:])
    return np.r_[T[:3] + d6[:3], q / np.linalg.norm(q)]


class TinyWindow:
    def __init__(self, g):
        self.g = g
        self.intr, self.k = g["intr"], g["dist"]
        self.T_SC, self.T0, self.uv = g["T_SC"], g["T0"], g["uv"]
        self.w = np.sqrt(64.0 / float(g["size"]) ** 2)

    def project(self, T, p, c):
        pS = rot(T[3:]).T @ (p - T[:3])
        pC = rot(self.T_SC[c][3:]).T @ (pS - self.T_SC[c][:3])
        x, y = pC[0] / pC[2], pC[1] / pC[2]
        k1, k2, p1, p2 = self.k[:4]
        r2 = x * x + y * y
        rad = 1 + k1 * r2 + k2 * r2 * r2
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        return np.array([self.intr[0] * xd + self.intr[2], self.intr[1] * yd + self.intr[3]])

    def cost(self, T1, lm):
        c = 0.0
        for f, T in enumerate((self.T0, T1)):
            for cam in range(2):
                for l in range(len(lm)):
                    e = self.w * (self.uv[f, cam, l] - self.project(T, lm[l], cam))
                    c += 0.5 * np.log1p(e @ e)
        return c

    def gradient(self, T1, lm, h=1e-6):
        """central differences in the 6 tangent directions of the pose (oplus) and the 3 L landmark coordinates"""
        gp = np.zeros(6)
        for k in range(6):
            d = np.zeros(6)
            d[k] = h
            gp[k] = (self.cost(pose_oplus(T1, d), lm) - self.cost(pose_oplus(T1, -d), lm)) / (2 * h)
        gl = np.zeros_like(lm)
        for l in range(lm.shape[0]):
            for k in range(3):
                a, b = lm.copy(), lm.copy()
                a[l, k] += h
                b[l, k] -= h
                gl[l, k] = (self.cost(T1, a) - self.cost(T1, b)) / (2 * h)
        return gp, gl
