"""Independent 40-digit restatement of MarginalizationError::addResidualBlock (M1): which residuals were linearised, at which
linearisation points and in which ordering is DATA (the oracle's log of its addResidualBlock calls: measurements, weights,
block ids -- no number the oracle computed from them); the arithmetic is done again here in mpmath, from the definitions.

Reference (okvis_ceres/src/MarginalizationError.cpp): every residual is evaluated AT THE LINEARISATION POINTS of its blocks
(:255-270, first-estimate Jacobians), corrected for its loss function exactly as ceres' Corrector does (:283-330), and
    H(i, j) += J_i^T J_j ,   b0(i) -= J_i^T r                                   (:333-382)
with the MINIMAL Jacobians J_i.  Error terms restated (definitions, not code):
  ReprojectionError   r = W (z - project(T_CS T_SW hp)),  W = upper Cholesky factor of the 2x2 information (data)
                      Jacobians: 40-digit central differences along the block's plus (they ARE the reference's minimal
                      Jacobians: Map::isJacobianCorrect is its own test of that)
  PoseError           e = [z.r - r ; 2 vec(z.q x q^-1)],  r = W e,  J = -W [I 0; 0 plus(dq)_3x3]  (= d e / d delta exactly)
  SpeedAndBiasError   e = z - x,  J = -W
  RelativePoseError   (src/RelativePoseError.cpp:79-147; between the extrinsics of consecutive frames, stereo_rig_v2)
                      e = [r_1 - r_0 ; 2 vec(q_1 x q_0^-1)],  J_0 = -W [I 0; 0 plus(dq)_3x3],  J_1 = W [I 0; 0 oplus(dq)_3x3]
  ImuError            the reference's closed form (src/ImuError.cpp:707-800): e and the blocks F0 / F1 evaluated at 40 digits on
                      a 40-digit pre-integration (tests/golden/make_golden_imu.py `integrate`), weighted by the upper Cholesky
                      factor of sym(inv(sym(P_delta))).  Its Jacobians are NOT difference quotients: the reference's analytic
                      blocks are the definition (they linearise the bias dependence of the pre-integrals to first order).
"""
import os
import sys

import mpmath as mp
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as G          # noqa: E402  (mp helpers: qmul, qrot_inv, distort, pose_plus ...)
import make_golden_imu as GI     # noqa: E402  (integrate, quat_to_R, cross_mx)

KIND_REPROJ, KIND_IMU, KIND_POSE, KIND_SB, KIND_RELPOSE = 0, 1, 2, 3, 4      # orc::ErrorTerm::Kind
TYPE_POSE, TYPE_SB, TYPE_LM = 0, 1, 2


def mpv(a):
    return [mp.mpf(float(x)) for x in a]


def normalised_pose(T):
    """okvis::kinematics::Transformation(r, q) normalises q"""
    n = mp.sqrt(sum(c * c for c in T[3:7]))
    return list(T[:3]) + [c / n for c in T[3:7]]


def plus(btype, x, d):
    if btype == TYPE_POSE:
        return G.pose_plus(normalised_pose(x), d)
    if btype == TYPE_SB:
        return [x[i] + d[i] for i in range(9)]
    return [x[0] + d[0], x[1] + d[1], x[2] + d[2], x[3]]


def mdim_of(btype):
    return {TYPE_POSE: 6, TYPE_SB: 9, TYPE_LM: 3}[btype]


def numdiff_blocks(f, xs, types, active):
    """minimal Jacobians of f(x_0, x_1, ...) by central differences (1e-25 at 60 digits) along each active block's plus"""
    h = mp.mpf(10) ** -25
    out = []
    for b, (x, t) in enumerate(zip(xs, types)):
        if not active[b]:
            out.append(None)
            continue
        n = mdim_of(t)
        cols = []
        for j in range(n):
            dp, dm = [mp.mpf(0)] * n, [mp.mpf(0)] * n
            dp[j], dm[j] = h, -h
            xp, xm = list(xs), list(xs)
            xp[b], xm[b] = plus(t, x, dp), plus(t, x, dm)
            fp, fm = f(*xp), f(*xm)
            cols.append([(a - c) / (2 * h) for a, c in zip(fp, fm)])
        out.append(mp.matrix([[cols[j][i] for j in range(n)] for i in range(len(cols[0]))]))
    return out


def reprojection(defn):
    z, W = mpv(defn[0:2]), mpv(defn[2:6])                      # W row-major upper 2x2
    model, intr, k = int(defn[7]), mpv(defn[8:12]), mpv(defn[12:20])

    def f(T_WS, hp, T_SC):
        T_WS, T_SC = normalised_pose(T_WS), normalised_pose(T_SC)
        e = G.reproj(model, intr, k, T_WS, hp, T_SC, z, mp.mpf(1))
        return [W[0] * e[0] + W[1] * e[1], W[2] * e[0] + W[3] * e[1]]
    return f


def quat_plus_3x3(q):     # operators.hpp:91-110, top-left 3x3 of plus(q)
    x, y, z, w = q
    return mp.matrix([[w, -z, y], [z, w, -x], [-y, x, w]])


def quat_mats(q):
    x, y, z, w = q
    P = mp.matrix([[w, -z, y, x], [z, w, -x, y], [-y, x, w, z], [-x, -y, -z, w]])
    O = mp.matrix([[w, z, -y, x], [-z, w, x, y], [y, -x, w, z], [-x, -y, -z, w]])
    return P, O


def qinv(q):
    n = sum(c * c for c in q)
    return [-q[0] / n, -q[1] / n, -q[2] / n, q[3] / n]


def pose_error(defn, x):
    z, W = normalised_pose(mpv(defn[0:7])), mp.matrix(6, 6)
    for a in range(6):
        for b in range(6):
            W[a, b] = mp.mpf(float(defn[7 + 6 * a + b]))
    T = normalised_pose(x)
    dq = G.qmul(z[3:7], qinv(T[3:7]))
    e = mp.matrix([z[0] - T[0], z[1] - T[1], z[2] - T[2], 2 * dq[0], 2 * dq[1], 2 * dq[2]])
    F = -mp.eye(6)
    Q = quat_plus_3x3(dq)
    for a in range(3):
        for b in range(3):
            F[3 + a, 3 + b] = -Q[a, b]
    return W * e, W * F


def relative_pose_error(defn, x0, x1):
    W = mp.matrix(6, 6)
    for a in range(6):
        for b in range(6):
            W[a, b] = mp.mpf(float(defn[6 * a + b]))
    T0, T1 = normalised_pose(x0), normalised_pose(x1)
    dq = G.qmul(T1[3:7], qinv(T0[3:7]))
    e = mp.matrix([T1[0] - T0[0], T1[1] - T0[1], T1[2] - T0[2], 2 * dq[0], 2 * dq[1], 2 * dq[2]])
    P, O = quat_mats(dq)
    J0, J1 = -mp.eye(6), mp.eye(6)
    for a in range(3):
        for b in range(3):
            J0[3 + a, 3 + b] = -P[a, b]
            J1[3 + a, 3 + b] = O[a, b]
    return W * e, [W * J0, W * J1]


def speed_bias_error(defn, x):
    W = mp.matrix(9, 9)
    for a in range(9):
        for b in range(9):
            W[a, b] = mp.mpf(float(defn[9 + 9 * a + b]))
    e = mp.matrix([mp.mpf(float(defn[i])) - x[i] for i in range(9)])
    return W * e, -W


def imu_error(defn, T0, sb0, T1, sb1):
    """src/ImuError.cpp:707-800 at the given parameters; returns (weighted residual, [J0 15x6, J1 15x9, J2 15x6, J3 15x9])"""
    t0 = mp.mpf(int(defn[0])) + mp.mpf(int(defn[1])) / 10 ** 9
    t1 = mp.mpf(int(defn[2])) + mp.mpf(int(defn[3])) / 10 ** 9
    redo = defn[4] != 0.0
    sb_ref = mpv(defn[5:14])
    names = ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c", "tau", "g")
    par = {k: mp.mpf(float(v)) for k, v in zip(names, defn[14:24])}
    n = int(defn[24])
    S = np.asarray(defn[25:25 + 8 * n]).reshape(n, 8)
    t = [mp.mpf(int(r[0])) + mp.mpf(int(r[1])) / 10 ** 9 for r in S]
    gyr, acc = [mpv(r[2:5]) for r in S], [mpv(r[5:8]) for r in S]
    T0, T1 = normalised_pose(T0), normalised_pose(T1)
    Dt = t1 - t0
    Db = [sb0[3 + i] - sb_ref[3 + i] for i in range(6)]
    # :738-748: a gyro-bias change of more than 1e-4 / Delta_t (or a pending redo_) pre-integrates again at THESE biases
    if redo or mp.sqrt(sum(c * c for c in Db[:3])) * Dt > mp.mpf("0.0001"):
        sb_ref = list(sb0)
        Db = [mp.mpf(0)] * 6
    pre = GI.integrate(t, gyr, acc, par, mp.matrix(sb_ref[3:6]), mp.matrix(sb_ref[6:9]), t0, t1, redo=True)
    Ps = (pre["P"] + pre["P"].T) / 2
    info = mp.inverse(Ps)
    info = (info + info.T) / 2
    W = mp.cholesky(info).T                       # squareRootInformation_ = L^T
    C0 = GI.quat_to_R(T0[3:7])
    C0t = C0.T
    gW = mp.matrix([0, 0, par["g"]])
    r0, r1 = mp.matrix(T0[:3]), mp.matrix(T1[:3])
    v0, v1 = mp.matrix(sb0[:3]), mp.matrix(sb1[:3])
    dp_est = r0 - r1 + v0 * Dt - gW * Dt * Dt / 2
    dv_est = v0 - v1 - gW * Dt
    dbg, dba = mp.matrix(Db[:3]), mp.matrix(Db[3:])
    corr = -(pre["dal"] * dbg)
    Dq = G.qmul(G.dq_of([corr[0], corr[1], corr[2]]), pre["Dq"])      # deltaQ(-dalpha_db_g Delta_b_g) * Delta_q
    q0, q1 = T0[3:7], T1[3:7]
    q1inv = qinv(q1)
    F0 = mp.eye(15)

    def setb(Fm, r, c, B):
        for a in range(3):
            for b in range(3):
                Fm[r + a, c + b] = B[a, b]
    setb(F0, 0, 0, C0t)
    setb(F0, 0, 3, C0t * GI.cross_mx(dp_est))
    setb(F0, 0, 6, C0t * Dt)
    setb(F0, 0, 9, pre["dp"])
    setb(F0, 0, 12, -pre["Cdi"])
    Pa, _ = quat_mats(G.qmul(Dq, q1inv))
    _, Ob = quat_mats(q0)
    setb(F0, 3, 3, (Pa * Ob)[0:3, 0:3])
    _, Oc = quat_mats(G.qmul(q1inv, q0))
    _, Od = quat_mats(Dq)
    setb(F0, 3, 9, (Oc * Od)[0:3, 0:3] * (-pre["dal"]))
    setb(F0, 6, 3, C0t * GI.cross_mx(dv_est))
    setb(F0, 6, 6, C0t)
    setb(F0, 6, 9, pre["dv"])
    setb(F0, 6, 12, -pre["Ci"])
    F1 = -mp.eye(15)
    setb(F1, 0, 0, -C0t)
    Pd, _ = quat_mats(Dq)
    Pe, _ = quat_mats(q1inv)
    setb(F1, 3, 3, -(Pd * Ob * Pe)[0:3, 0:3])
    setb(F1, 6, 6, -C0t)
    e = mp.matrix(15, 1)
    top = C0t * dp_est + pre["adi"] + F0[0:3, 9:12] * dbg + F0[0:3, 12:15] * dba
    eq = G.qmul(Dq, G.qmul(q1inv, q0))
    mid = C0t * dv_est + pre["ai"] + F0[6:9, 9:12] * dbg + F0[6:9, 12:15] * dba
    for i in range(3):
        e[i], e[3 + i], e[6 + i] = top[i], 2 * eq[i], mid[i]
    for i in range(6):
        e[9 + i] = sb0[3 + i] - sb1[3 + i]
    return W * e, [W * F0[:, 0:6], W * F0[:, 6:15], W * F1[:, 0:6], W * F1[:, 6:15]]


def cauchy_corrector(r, Js, a):
    """MarginalizationError.cpp:283-330 (ceres' Corrector) with ceres::CauchyLoss(a): rho(s) = a^2 log(1 + s / a^2),
    rho' = 1 / (1 + s / a^2), rho'' = -rho'^2 / a^2.  rho'' <= 0 always, so the branch `(sq_norm == 0) || (rho[2] <= 0)` is
    the one taken: residual and Jacobians are scaled by sqrt(rho') and the second-order term is dropped -- restated in full
    so that the branch is a computed fact, not an assumption."""
    s = sum(x * x for x in r)
    b = a * a
    rho1 = 1 / (1 + s / b)
    rho2 = -(rho1 * rho1) / b
    sq1 = mp.sqrt(rho1)
    if s == 0 or rho2 <= 0:
        scaling, alpha_sq = sq1, mp.mpf(0)
    else:
        D = 1 + 2 * s * rho2 / rho1
        alpha = 1 - mp.sqrt(D)
        scaling, alpha_sq = sq1 / (1 - alpha), alpha / s
    rv = mp.matrix(list(r))
    Jout = [None if J is None else sq1 * (J - alpha_sq * rv * (rv.T * J)) for J in Js]
    return [scaling * x for x in r], Jout


def m1(log, dps=40):
    """log = OracleEstimator.marg_m1_log().  Returns (H, b0) as float arrays in the log's ordering."""
    mp.mp.dps = max(dps, 60)     # the difference quotients need the head-room; results are rounded to double at the end
    blocks = {b["id"]: b for b in log["blocks"]}
    n = max((b["ordering"] + b["mdim"] for b in log["blocks"]), default=0)
    H, b0 = mp.zeros(n), mp.zeros(n, 1)
    for ent in log["log"]:
        bl = [blocks[i] for i in ent["ids"]]
        xs = [mpv(b["lin"][:{TYPE_POSE: 7, TYPE_SB: 9, TYPE_LM: 4}[b["type"]]]) for b in bl]
        types = [b["type"] for b in bl]
        active = [b["mdim"] > 0 for b in bl]
        k = ent["kind"]
        if k == KIND_REPROJ:
            f = reprojection(ent["defn"])
            r = f(*xs)
            Js = numdiff_blocks(f, xs, types, active)
        elif k == KIND_POSE:
            rm, J = pose_error(ent["defn"], xs[0])
            r, Js = list(rm), [J if active[0] else None]
        elif k == KIND_SB:
            rm, J = speed_bias_error(ent["defn"], xs[0])
            r, Js = list(rm), [J if active[0] else None]
        elif k == KIND_IMU:
            rm, Jl = imu_error(ent["defn"], *xs)
            r, Js = list(rm), [J if a else None for J, a in zip(Jl, active)]
        elif k == KIND_RELPOSE:
            rm, Jl = relative_pose_error(ent["defn"], *xs)
            r, Js = list(rm), [J if a else None for J, a in zip(Jl, active)]
        else:
            raise NotImplementedError("mp_m1: error-term kind %d" % k)
        if ent["loss"] == 1:
            r, Js = cauchy_corrector(r, Js, mp.mpf(float(ent["loss_param"])))
        elif ent["loss"] != 0:
            raise NotImplementedError("mp_m1: loss %d" % ent["loss"])
        rv = mp.matrix(list(r))
        for i, (bi, Ji) in enumerate(zip(bl, Js)):
            if Ji is None:
                continue
            oi = bi["ordering"]
            g = Ji.T * rv
            for a in range(bi["mdim"]):
                b0[oi + a] -= g[a]
            for bj, Jj in zip(bl, Js):
                if Jj is None:
                    continue
                oj = bj["ordering"]
                blk = Ji.T * Jj
                for a in range(bi["mdim"]):
                    for c in range(bj["mdim"]):
                        H[oi + a, oj + c] += blk[a, c]
    Hf = np.array([[float(H[i, j]) for j in range(n)] for i in range(n)])
    return Hf, np.array([float(b0[i]) for i in range(n)])


class ExactChain:
    """The marginalisation prior carried from call to call at 40 digits: M1 (above) on top of the previous exact prior, then
    M2 (tests/mp_marg.py) -- every number from the raw definitions, only structure (which residuals, ordering, which
    rows leave) from the oracle's log."""

    def __init__(self):
        self.kept = {}        # block id -> (first row, rows) in the current exact prior
        self.H = np.zeros((0, 0))
        self.b0 = np.zeros(0)

    def m1(self, log):
        """exact system after M1 in the log's ordering"""
        assert log["had_prior"] == bool(self.kept)
        n = max((b["ordering"] + b["mdim"] for b in log["blocks"]), default=0)
        Hn, bn = m1(log)
        assert Hn.shape == (n, n)
        # the previous prior sits in the rows of its blocks, wherever M1's book-keeping moved them
        pos = {b["id"]: b["ordering"] for b in log["blocks"]}
        old = sorted(self.kept.items(), key=lambda kv: kv[1][0])
        idx_new, idx_old = [], []
        for bid, (o, m) in old:
            for k in range(m):
                idx_new.append(pos[bid] + k)
                idx_old.append(o + k)
        if idx_new:
            Hn[np.ix_(idx_new, idx_new)] += self.H[np.ix_(idx_old, idx_old)]
            bn[idx_new] += self.b0[idx_old]
        self.pre = dict(H=Hn, b0=bn, blocks=log["blocks"])
        return Hn, bn

    def m2(self, lm_ranges, dense_ranges):
        """marginalizeOut on the exact system: the rows in the ranges leave; returns mp_marg's result"""
        import mp_marg
        ex = mp_marg.marginalize_mp(self.pre["H"], self.pre["b0"], list(lm_ranges), list(dense_ranges))
        gone = set()
        for r, m in list(lm_ranges) + list(dense_ranges):
            gone.update(range(r, r + m))
        self.kept, o = {}, 0
        for b in sorted(self.pre["blocks"], key=lambda b: b["ordering"]):
            if b["mdim"] > 0 and b["ordering"] not in gone:
                self.kept[b["id"]] = (o, b["mdim"])
                o += b["mdim"]
        self.H, self.b0 = ex["H"], ex["b0"]
        assert self.H.shape[0] == o
        return ex
