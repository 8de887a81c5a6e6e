#!/usr/bin/env python3
"""Golden fixture for the IMU arithmetic from an INDEPENDENT mpmath restatement (50 digits) of the
recurrences -- nothing here calls the oracle or the product.

Definitions restated (reference files, for the reader):
  ImuError::propagation        okvis_ceres/src/ImuError.cpp:266-476   trapezoidal integration of Delta_q, int C, int int C,
                               int a, int int a, the bias sub-Jacobians and the covariance P <- F P F^T + Q per step; the
                               first / last sample are interpolated to t_start / t_end (:319-338); prediction :452-458,
                               covariance of the states :475-483
  ImuError::redoPreintegration :76-263   the same loop with two differences (kept on purpose):
                               dalpha_db_g += C_1 rightJacobian(omega dt) dt  (:189; propagation: dt C_1, :384)
                               sigma2_v = dt sigma_a_c^2                     (:215; propagation: dt sigma_a_c sigma_a_c, same value
                                                                              unless the accelerometer saturates)
  ImuError error vector        :786-791 at a fresh linearisation (Delta_b = 0):
                               e = [C_S0W dp + intint a ; 2 vec(Dq (q1^-1 q0)) ; C_S0W dv + int a ; b0 - b1]
                               chi^2 = e^T P_delta^-1 e  (what the weighted residual's squared norm must equal)
Run:  python tests/golden/make_golden_imu.py      (writes imu.npz)
"""
import os

import mpmath as mp
import numpy as np

from make_golden import qmul, qrot

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def M(rows):
    return mp.matrix(rows)


def quat_to_R(q):
    x, y, z, w = q
    return M([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def cross_mx(v):
    return M([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def right_jacobian(phi):
    n = mp.sqrt(phi[0] ** 2 + phi[1] ** 2 + phi[2] ** 2)
    X = cross_mx(phi)
    if n < mp.mpf("1e-4"):      # the reference switches to the series here
        a, b = mp.mpf(-1) / 2, mp.mpf(1) / 6
    else:
        a, b = -(1 - mp.cos(n)) / n ** 2, (n - mp.sin(n)) / n ** 3
    return mp.eye(3) + a * X + b * (X * X)


def sinc(x):
    return mp.sin(x) / x if abs(x) > mp.mpf("1e-6") else 1 - x ** 2 / 6 + x ** 4 / 120 - x ** 6 / 5040


def integrate(t, gyr, acc, par, bg, ba, t0, t1, redo):
    """t: sample times (mp seconds), returns the pre-integrated quantities"""
    Dq = [mp.mpf(0), mp.mpf(0), mp.mpf(0), mp.mpf(1)]
    Ci, Cdi, cross = mp.zeros(3), mp.zeros(3), mp.zeros(3)
    ai, adi = mp.zeros(3, 1), mp.zeros(3, 1)
    dal, dv, dp = mp.zeros(3), mp.zeros(3), mp.zeros(3)
    P = mp.zeros(15)
    time, Dt, started, used = t0, mp.mpf(0), False, 0
    n = len(t)
    for k in range(n):
        w0, a0 = M(gyr[k]), M(acc[k])
        w1, a1 = (M(gyr[k + 1]), M(acc[k + 1])) if k + 1 < n else (w0, a0)
        nxt = t[k + 1] if k + 1 < n else t1
        dt = nxt - time
        if t1 < nxt:
            interval = nxt - t[k]
            nxt = t1
            dt = nxt - time
            r = dt / interval
            w1, a1 = (1 - r) * w0 + r * w1, (1 - r) * a0 + r * a1
        if dt <= 0:
            continue
        Dt += dt
        if not started:
            started = True
            r = dt / (nxt - t[k])
            w0, a0 = r * w0 + (1 - r) * w1, r * a0 + (1 - r) * a1
        sg, sa = par["sigma_g_c"], par["sigma_a_c"]
        if max(abs(x) for x in list(w0) + list(w1)) > par["g_max"]:
            sg *= 100
        if max(abs(x) for x in list(a0) + list(a1)) > par["a_max"]:
            sa *= 100
        wt, at = (w0 + w1) / 2 - bg, (a0 + a1) / 2 - ba
        th = mp.sqrt(wt[0] ** 2 + wt[1] ** 2 + wt[2] ** 2) * dt / 2
        s = sinc(th) * dt / 2
        dq = [s * wt[0], s * wt[1], s * wt[2], mp.cos(th)]
        Dq1 = qmul(Dq, dq)
        C, C1 = quat_to_R(Dq), quat_to_R(Dq1)
        Cs = C + C1
        Ci1 = Ci + Cs * dt / 2
        ai1 = ai + Cs * at * dt / 2
        adi_step = ai * dt + Cs * at * dt * dt / 4
        Cdi = Cdi + Ci * dt + Cs * dt * dt / 4
        adi = adi + adi_step
        rj = right_jacobian(wt * dt)
        dal = dal + (C1 * rj * dt if redo else C1 * dt)
        nq = sum(c * c for c in dq)
        dq_inv = [-dq[0] / nq, -dq[1] / nq, -dq[2] / nq, dq[3] / nq]
        cross1 = quat_to_R(dq_inv) * cross + rj * dt
        ax = cross_mx(at)
        mix = C * ax * cross + C1 * ax * cross1
        dv1 = dv + mix * dt / 2
        dp_step = dv * dt + mix * dt * dt / 4
        dp = dp + dp_step
        F = mp.eye(15)

        def setb(r0, c0, B):
            for a in range(3):
                for b in range(3):
                    F[r0 + a, c0 + b] = B[a, b]
        setb(0, 3, -cross_mx(adi_step))
        setb(0, 6, mp.eye(3) * dt)
        setb(0, 9, dp_step)
        setb(0, 12, -Ci * dt + Cs * dt * dt / 4)
        setb(3, 9, -C1 * dt)
        setb(6, 3, -cross_mx(Cs * at * dt / 2))
        setb(6, 9, mix * dt / 2)
        setb(6, 12, -Cs * dt / 2)
        P = F * P * F.T
        s2a = dt * sg * sg
        s2v = dt * sa * sa if redo else dt * sa * par["sigma_a_c"]
        s2p = dt * dt * s2v / 2
        for c in range(3):
            P[3 + c, 3 + c] += s2a
            P[6 + c, 6 + c] += s2v
            P[c, c] += s2p
            P[9 + c, 9 + c] += dt * par["sigma_gw_c"] ** 2
            P[12 + c, 12 + c] += dt * par["sigma_aw_c"] ** 2
        Dq, Ci, ai, cross, dv, time = Dq1, Ci1, ai1, cross1, dv1, nxt
        used += 1
        if nxt == t1:
            break
    return dict(Dq=Dq, Ci=Ci, Cdi=Cdi, ai=ai, adi=adi, dal=dal, dv=dv, dp=dp, P=P, Dt=Dt, used=used)


def fl(x):
    if isinstance(x, mp.matrix):
        return np.array([[float(x[i, j]) for j in range(x.cols)] for i in range(x.rows)])
    return np.array([float(v) for v in x])


def main():
    rng = np.random.default_rng(20251001)
    par_f = dict(a_max=176.0, g_max=7.8, sigma_g_c=12.0e-4, sigma_a_c=8.0e-3, sigma_bg=0.03, sigma_ba=0.1, sigma_gw_c=4.0e-6,
                 sigma_aw_c=4.0e-5, tau=3600.0, g=9.81007)
    par = {k: mp.mpf(v) for k, v in par_f.items()}
    rows = []
    for case in range(4):
        rate, n = (200, 30) if case < 3 else (100, 40)
        base_sec = 100 + case
        ns = (np.arange(n) * (1_000_000_000 // rate) + 3_000_000).astype(np.int64)          # sample stamps (exact in ns)
        t0_ns = int(ns[2] + (1_000_000_000 // rate) * 0.37)                                     # between samples 2 and 3
        t1_ns = int(ns[n - 4] + (1_000_000_000 // rate) * 0.61)                                 # between samples n-4 and n-3
        gyr = rng.normal(size=(n, 3)) * 0.3
        acc = rng.normal(size=(n, 3)) * 0.5 + np.array([0.0, 0.0, 9.81])
        if case == 2:
            gyr[10, 1] = 9.0      # gyroscope saturation on one sample: sigma_g_c x 100 on the two steps that use it
        bg, ba = rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.05
        a = rng.uniform(-0.5, 0.5, 3)
        th = np.linalg.norm(a)
        T0 = np.r_[rng.uniform(-1, 1, 3), np.sin(th / 2) * a / th, np.cos(th / 2)]
        v0 = rng.normal(size=3)
        t = [mp.mpf(int(x)) / 10 ** 9 for x in ns]
        t0, t1 = mp.mpf(t0_ns) / 10 ** 9, mp.mpf(t1_ns) / 10 ** 9
        mg, ma = [[mp.mpf(float(x)) for x in r] for r in gyr], [[mp.mpf(float(x)) for x in r] for r in acc]
        mbg, mba = M([mp.mpf(float(x)) for x in bg]), M([mp.mpf(float(x)) for x in ba])
        q0 = [mp.mpf(float(x)) for x in T0[3:]]
        nq = mp.sqrt(sum(c * c for c in q0))
        q0 = [c / nq for c in q0]
        r0, mv0 = M([mp.mpf(float(x)) for x in T0[:3]]), M([mp.mpf(float(x)) for x in v0])
        C0 = quat_to_R(q0)
        gW = M([0, 0, par["g"]])
        # ---- propagation
        pr = integrate(t, mg, ma, par, mbg, mba, t0, t1, redo=False)
        Dt = pr["Dt"]
        r1 = r0 + mv0 * Dt + C0 * pr["adi"] - gW * Dt * Dt / 2
        q1 = qmul(q0, pr["Dq"])
        n1 = mp.sqrt(sum(c * c for c in q1))
        q1 = [c / n1 for c in q1]
        v1 = mv0 + C0 * pr["ai"] - gW * Dt
        Tm = mp.eye(15)
        for blk in range(3):
            for i in range(3):
                for j in range(3):
                    Tm[3 * blk + i, 3 * blk + j] = C0[i, j]
        cov = Tm * pr["P"] * Tm.T
        # ---- the factor: pre-integration in its own flavour, error at states near the prediction
        rd = integrate(t, mg, ma, par, mbg, mba, t0, t1, redo=True)
        dT = rng.normal(size=3) * 0.02
        da = rng.normal(size=3) * 0.01
        T1 = np.r_[fl(r1)[:, 0] + dT, np.array([float(c) for c in qmul([mp.mpf(float(x)) for x in np.r_[np.sin(np.linalg.norm(da) / 2) * da / np.linalg.norm(da), np.cos(np.linalg.norm(da) / 2)]], q1)])]
        sb1 = np.r_[fl(v1)[:, 0] + rng.normal(size=3) * 0.02, bg + rng.normal(size=3) * 1e-3, ba + rng.normal(size=3) * 1e-3]
        mq1 = [mp.mpf(float(x)) for x in T1[3:]]
        nn = mp.sqrt(sum(c * c for c in mq1))
        mq1 = [c / nn for c in mq1]
        mr1, mv1 = M([mp.mpf(float(x)) for x in T1[:3]]), M([mp.mpf(float(x)) for x in sb1[:3]])
        dp_est = r0 - mr1 + mv0 * Dt - gW * Dt * Dt / 2
        dv_est = mv0 - mv1 - gW * Dt
        q1inv = [-mq1[0], -mq1[1], -mq1[2], mq1[3]]
        eq = qmul(rd["Dq"], qmul(q1inv, q0))
        e = list(C0.T * dp_est + rd["adi"]) + [2 * eq[0], 2 * eq[1], 2 * eq[2]] + list(C0.T * dv_est + rd["ai"]) + \
            [mp.mpf(float(bg[i])) - mp.mpf(float(sb1[3 + i])) for i in range(3)] + [mp.mpf(float(ba[i])) - mp.mpf(float(sb1[6 + i])) for i in range(3)]
        e = M(e)
        Ps = (rd["P"] + rd["P"].T) / 2
        chi2 = (e.T * mp.lu_solve(Ps, e))[0, 0]
        stamps = np.stack([np.full(n, base_sec), ns], 1).astype(np.uint32)
        rows.append(dict(imu_t=stamps, imu_m=np.c_[gyr, acc], t0=np.array([base_sec, t0_ns], np.uint32), t1=np.array([base_sec, t1_ns], np.uint32),
                         T0=np.r_[T0[:3], [float(c) for c in q0]], sb0=np.r_[v0, bg, ba], used=pr["used"],
                         T_pred=np.r_[fl(r1)[:, 0], [float(c) for c in q1]], v_pred=fl(v1)[:, 0], cov=fl(cov),
                         integrals=np.r_[fl(pr["adi"])[:, 0], fl(pr["ai"])[:, 0], float(Dt)],
                         T1=np.r_[T1[:3], [float(c) for c in mq1]], sb1=sb1, e=fl(e)[:, 0], P_delta=fl(Ps), chi2=float(chi2),
                         dalpha_db_g=fl(rd["dal"]), dp_db_g=fl(rd["dp"]), C_doubleintegral=fl(rd["Cdi"])))
    out = {"params": np.array([par_f[k] for k in ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c", "tau", "g")])}
    for key in rows[0]:
        if key in ("imu_t", "imu_m"):
            nmax = max(len(r[key]) for r in rows)
            out[key] = np.array([np.vstack([r[key], np.zeros((nmax - len(r[key]), r[key].shape[1]), r[key].dtype)]) for r in rows])
            out["imu_n"] = np.array([len(r["imu_t"]) for r in rows])
        else:
            out[key] = np.array([r[key] for r in rows])
    np.savez(os.path.join(HERE, "imu.npz"), **out)
    print("wrote imu.npz:", len(rows), "cases; steps", [r["used"] for r in rows], "chi2", [r["chi2"] for r in rows])


if __name__ == "__main__":
    main()
