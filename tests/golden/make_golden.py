#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from an INDEPENDENT derivation.

Nothing here calls the oracle or the product: the residual functions are written again from the
mathematical definitions (mpmath, 60 significant digits) and every Jacobian is a high-precision central
difference of those functions along the manifold retractions (step 1e-25 => ~1e-40 truncation error), so the
expected values do not depend on any analytic Jacobian formula.  The converged tiny-window states come from
scipy.optimize.least_squares (an independent minimiser, loss='cauchy' == ceres::CauchyLoss(1)).

Definitions restated (reference files, for the reader):
  residual / parameter order  okvis_ceres/include/okvis/ceres/implementation/ReprojectionError.hpp:85-137
  pinhole + distortions       okvis_cv/include/okvis/cameras/implementation/{PinholeCamera,RadialTangentialDistortion,
                              EquidistantDistortion,RadialTangentialDistortion8}.hpp
  pose retraction             okvis_kinematics/.../Transformation.hpp:206-217  (r += dr; q = dq(dalpha) * q)
  PoseError / RelativePoseError / SpeedAndBiasError   okvis_ceres/src/{PoseError,RelativePoseError,SpeedAndBiasError}.cpp

Run:  python tests/golden/make_golden.py      (writes error_terms.npz, tiny_window.npz)
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))
H = mp.mpf(10) ** -25


# ----------------------------------------------------------------------------- mp helpers
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz]


def qconj(q):
    return [-q[0], -q[1], -q[2], q[3]]


def qrot(q, v):
    """rotate v by the (unit or non-unit) quaternion through Eigen's toRotationMatrix formula"""
    x, y, z, w = q
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return [sum(R[i][k] * v[k] for k in range(3)) for i in range(3)]


def qrot_inv(q, v):
    x, y, z, w = q
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return [sum(R[k][i] * v[k] for k in range(3)) for i in range(3)]


def dq_of(alpha):
    n = mp.sqrt(sum(a * a for a in alpha))
    if n == 0:
        return [mp.mpf(0), mp.mpf(0), mp.mpf(0), mp.mpf(1)]
    s = mp.sin(n / 2) / n
    return [s * alpha[0], s * alpha[1], s * alpha[2], mp.cos(n / 2)]


def pose_plus(T, d):
    q = qmul(dq_of(d[3:6]), T[3:7])
    n = mp.sqrt(sum(c * c for c in q))
    return [T[0] + d[0], T[1] + d[1], T[2] + d[2]] + [c / n for c in q]


def distort(model, k, u0, u1):
    if model == 0:
        return u0, u1
    if model == 1:
        k1, k2, p1, p2 = k[:4]
        rho = u0 * u0 + u1 * u1
        rad = k1 * rho + k2 * rho * rho
        return (u0 + u0 * rad + 2 * p1 * u0 * u1 + p2 * (rho + 2 * u0 * u0),
                u1 + u1 * rad + 2 * p2 * u0 * u1 + p1 * (rho + 2 * u1 * u1))
    if model == 2:
        k1, k2, k3, k4 = k[:4]
        r = mp.sqrt(u0 * u0 + u1 * u1)
        th = mp.atan(r)
        thd = th * (1 + k1 * th ** 2 + k2 * th ** 4 + k3 * th ** 6 + k4 * th ** 8)
        s = thd / r if r > mp.mpf("1e-8") else mp.mpf(1)
        return s * u0, s * u1
    k1, k2, p1, p2, k3, k4, k5, k6 = k[:8]
    rho = u0 * u0 + u1 * u1
    rad = (1 + ((k3 * rho + k2) * rho + k1) * rho) / (1 + ((k6 * rho + k5) * rho + k4) * rho)
    return (u0 * rad + 2 * p1 * u0 * u1 + p2 * (rho + 2 * u0 * u0), u1 * rad + 2 * p2 * u0 * u1 + p1 * (rho + 2 * u1 * u1))


def reproj(model, intr, k, T_WS, hp, T_SC, uv, w):
    """w * (uv - project(T_CS T_SW hp))"""
    hw = hp[3]
    dW = [hp[i] - T_WS[i] * hw for i in range(3)]
    pS = qrot_inv(T_WS[3:7], dW)
    dS = [pS[i] - T_SC[i] * hw for i in range(3)]
    pC = qrot_inv(T_SC[3:7], dS)
    if hw < 0:
        pC = [-c for c in pC]
    d0, d1 = distort(model, k, pC[0] / pC[2], pC[1] / pC[2])
    return [w * (uv[0] - (intr[0] * d0 + intr[2])), w * (uv[1] - (intr[1] * d1 + intr[3]))]


def numdiff(f, n):
    """central differences of a vector function of an n-vector perturbation, at 0"""
    cols = []
    for j in range(n):
        dp = [mp.mpf(0)] * n
        dm = [mp.mpf(0)] * n
        dp[j], dm[j] = H, -H
        fp, fm = f(dp), f(dm)
        cols.append([(a - b) / (2 * H) for a, b in zip(fp, fm)])
    return np.array([[float(cols[j][i]) for j in range(n)] for i in range(len(cols[0]))])


def to_mp(a):
    return [mp.mpf(float(x)) for x in a]


def rand_pose(rng, tr, rot):
    a = rng.uniform(-rot, rot, 3)
    th = np.linalg.norm(a)
    return np.r_[rng.uniform(-tr, tr, 3), np.sin(th / 2) * a / th, np.cos(th / 2)]


MODELS = {0: [], 1: [-0.28340811217, 0.0739590738929, 0.000193595028569, 1.76187114545e-05],
          2: [-0.21, 0.14, 0.0006, 0.0003], 3: [-0.16, 0.15, 0.0003, 0.0002, 0.01, 0.02, -0.01, 0.005]}
INTR = [458.654880721, 457.296696463, 367.215803962, 248.37534061]


def golden_error_terms():
    rng = np.random.default_rng(20250629)
    out = {}
    # ---- reprojection: 6 cases per distortion model
    rows = []
    for model, dist in MODELS.items():
        k = to_mp(list(dist) + [0.0] * (8 - len(dist)))
        for c in range(6):
            T_WS, T_SC = rand_pose(rng, 1.0, 0.5), rand_pose(rng, 0.2, 0.2)
            z = rng.uniform(1.0, 8.0)
            pc = np.r_[rng.uniform(-0.4, 0.4, 2) * z, z]
            # world point = T_WS * T_SC * pc  (float, only used to place the point in front of the camera)
            def rot(q, v):
                return np.array([float(x) for x in qrot(to_mp(q), to_mp(v))])
            pS = rot(T_SC[3:], pc) + T_SC[:3]
            pW = rot(T_WS[3:], pS) + T_WS[:3]
            hw = 1.0 if c % 2 == 0 else rng.uniform(0.5, 2.0)
            hp = np.r_[pW * hw, hw]
            uv = np.array([300.0, 200.0]) + rng.normal(size=2) * 40.0
            size = rng.uniform(4.0, 12.0)
            w = mp.sqrt(mp.mpf(64) / (mp.mpf(float(size)) ** 2))
            mT, mh, mE, muv = to_mp(T_WS), to_mp(hp), to_mp(T_SC), to_mp(uv)
            r = reproj(model, to_mp(INTR), k, mT, mh, mE, muv, w)
            Jp = numdiff(lambda d: reproj(model, to_mp(INTR), k, pose_plus(mT, d), mh, mE, muv, w), 6)
            Jl = numdiff(lambda d: reproj(model, to_mp(INTR), k, mT, [mh[0] + d[0], mh[1] + d[1], mh[2] + d[2], mh[3]], mE, muv, w), 3)
            Je = numdiff(lambda d: reproj(model, to_mp(INTR), k, mT, mh, pose_plus(mE, d), muv, w), 6)
            rows.append(dict(model=model, dist=np.r_[dist, np.zeros(8 - len(dist))], T_WS=T_WS, hp=hp, T_SC=T_SC, uv=uv, size=size,
                             r=np.array([float(x) for x in r]), Jp=Jp, Jl=Jl, Je=Je))
    for key in rows[0]:
        out["reproj_" + key] = np.array([row[key] for row in rows])
    out["reproj_intr"] = np.array(INTR)

    # ---- PoseError: e = [r_m - r ; 2 vec(q_m * q^-1)], weighted by sqrt-information (diagonal here)
    rows = []
    for c in range(6):
        Tm, T = rand_pose(rng, 1.0, 0.5), rand_pose(rng, 1.0, 0.5)
        info = rng.uniform(1.0, 100.0, 6)
        sw = [mp.sqrt(mp.mpf(float(v))) for v in info]

        def perr(Tx):
            dq = qmul(to_mp(Tm[3:]), qconj(Tx[3:7]))
            e = [mp.mpf(float(Tm[i])) - Tx[i] for i in range(3)] + [2 * dq[0], 2 * dq[1], 2 * dq[2]]
            return [sw[i] * e[i] for i in range(6)]
        mT = to_mp(T)
        rows.append(dict(Tm=Tm, T=T, info=info, r=np.array([float(x) for x in perr(mT)]),
                         J=numdiff(lambda d: perr(pose_plus(mT, d)), 6)))
    for key in rows[0]:
        out["pose_" + key] = np.array([row[key] for row in rows])

    # ---- RelativePoseError: e = [r1 - r0 ; 2 vec(q1 * q0^-1)], isotropic variances
    rows = []
    for c in range(6):
        T0, T1 = rand_pose(rng, 1.0, 0.5), rand_pose(rng, 1.0, 0.5)
        tv, rv = rng.uniform(1e-3, 1e-1), rng.uniform(1e-4, 1e-2)
        sw = [mp.sqrt(1 / mp.mpf(float(tv)))] * 3 + [mp.sqrt(1 / mp.mpf(float(rv)))] * 3

        def rerr(Ta, Tb):
            dq = qmul(Tb[3:7], qconj(Ta[3:7]))
            e = [Tb[i] - Ta[i] for i in range(3)] + [2 * dq[0], 2 * dq[1], 2 * dq[2]]
            return [sw[i] * e[i] for i in range(6)]
        m0, m1 = to_mp(T0), to_mp(T1)
        rows.append(dict(T0=T0, T1=T1, tv=tv, rv=rv, r=np.array([float(x) for x in rerr(m0, m1)]),
                         J0=numdiff(lambda d: rerr(pose_plus(m0, d), m1), 6), J1=numdiff(lambda d: rerr(m0, pose_plus(m1, d)), 6)))
    for key in rows[0]:
        out["relpose_" + key] = np.array([row[key] for row in rows])
    np.savez(os.path.join(HERE, "error_terms.npz"), **out)
    print("wrote error_terms.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------- tiny window solved by scipy
def golden_tiny_window():
    """2 camera poses (pose 0 pinned by a strong prior), 12 landmarks, fixed identity-like extrinsics, no distortion
    subtleties: an independent minimiser must land on the same fixed point as the oracle / GPU solver."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(7)
    intr = np.array(INTR)
    k = MODELS[1]
    # stereo rig (0.11 m baseline) so that the scale is observable and the minimum is unique
    T_SCs = [np.r_[0.05, -0.02, 0.01, 0.0, 0.0, 0.0, 1.0], np.r_[0.05, 0.09, 0.01, 0.0, 0.0, 0.0, 1.0]]
    T0 = np.r_[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    T1 = np.r_[0.4, 0.05, -0.03, rand_pose(rng, 0.0, 0.08)[3:]]
    L = 12
    lm = np.c_[rng.uniform(-1.5, 1.5, L), rng.uniform(-1.0, 1.0, L), rng.uniform(3.0, 8.0, L)]

    def fquat_rot_inv(q, v):
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        return R.T @ v

    def proj(T, p, c):
        T_SC = T_SCs[c]
        pS = fquat_rot_inv(T[3:], p - T[:3])
        pC = fquat_rot_inv(T_SC[3:], pS - T_SC[:3])
        d0, d1 = distort(1, k, pC[0] / pC[2], pC[1] / pC[2])
        return np.array([intr[0] * float(d0) + intr[2], intr[1] * float(d1) + intr[3]])

    uv = np.zeros((2, 2, L, 2))  # frame, camera, landmark
    for f, T in enumerate((T0, T1)):
        for c in range(2):
            for l in range(L):
                uv[f, c, l] = (proj(T, lm[l], c) + rng.normal(size=2)).astype(np.float32)
    size = 8.0
    w = np.sqrt(64.0 / size ** 2)
    # unknowns: delta(6) for pose 1 around T1_init, landmarks (3 each); pose 0 fixed exactly (prior -> infinity)
    T1_init = np.r_[T1[:3] + rng.normal(size=3) * 0.05, T1[3:]]
    lm_init = lm + rng.normal(size=lm.shape) * 0.1

    def fpose_plus(T, d):
        out = pose_plus(to_mp(T), to_mp(d))
        return np.array([float(x) for x in out])

    def residuals(x):
        T1x = fpose_plus(T1_init, x[:6])
        pts = x[6:].reshape(L, 3)
        res = []
        for f, T in enumerate((T0, T1x)):
            for c in range(2):
                for l in range(L):
                    res.append(w * (uv[f, c, l] - proj(T, pts[l], c)))
        return np.concatenate(res)

    # scipy's robust losses act per scalar residual; ceres' act per 2-vector block -> implement the block loss by hand:
    # minimise sum_i log(1 + |r_i|^2)  ==  least squares on  sqrt(log(1+|r_i|^2)) per block
    def block_cauchy(x):
        r = residuals(x).reshape(-1, 2)
        s = np.sum(r * r, axis=1)
        return np.sqrt(np.log1p(s))

    from scipy.optimize import minimize

    def cost(x):
        return 0.5 * np.sum(block_cauchy(x) ** 2)

    x0 = np.r_[np.zeros(6), lm_init.reshape(-1)]
    # stage 1: plain least squares (fast, lands near the robust minimum); stage 2: quasi-Newton on the robust cost
    sol = least_squares(residuals, x0, method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    x = sol.x
    for _ in range(6):
        res = minimize(cost, x, method="BFGS", options=dict(gtol=1e-10, maxiter=2000))
        x = res.x
    T1_opt = fpose_plus(T1_init, x[:6])
    lm_opt = x[6:].reshape(L, 3)
    np.savez(os.path.join(HERE, "tiny_window.npz"), intr=intr, dist=np.array(k), T_SC=np.stack(T_SCs), T0=T0, T1_init=T1_init,
             lm_init=lm_init, uv=uv, size=size, T1_opt=T1_opt, lm_opt=lm_opt, cost=cost(x))
    print("wrote tiny_window.npz: cost", cost(x), "grad-inf", np.max(np.abs(res.jac)))


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 2 or sys.argv[1] == "terms":
        golden_error_terms()
    if len(sys.argv) < 2 or sys.argv[1] == "window":
        golden_tiny_window()
