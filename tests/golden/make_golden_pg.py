#!/usr/bin/env python3
"""Golden fixture for the pose-graph error terms from an INDEPENDENT mpmath restatement (30 digits) of the
reference's functors -- nothing here calls the oracle or the product; svin_amd.synthetic_pg only supplies the INPUT graph,
which is stored in the fixture.

Definitions restated (reference files, for the reader):
  FourDOFError / FourDOFWeightError   pose_graph/include/pose_graph/PoseGraph.h:134-231
        r = [R(yaw_a, pitch_a, roll_a)^T (t_b - t_a) - t_meas ; normalizeAngle(yaw_b - yaw_a - yaw_meas)]   (degrees; loop: yaw row / 10)
  PoseGraph3dErrorTerm                pose_graph/include/pose_graph/Pose3DError.h:103-147
        r = sqrtInfo [R(q_a)^T (t_b - t_a) - t_meas ; 2 vec(q_meas (q_a^-1 q_b)^-1)]
  problem construction                pose_graph/src/pose_graph/PoseGraph.cpp:262-332 (4 DoF: the 2 previous keyframes of the sequence,
        pitch / roll of node a held at their SVIn values) and :436-489 (6 DoF: the 4 previous keyframes, sqrtInfo diag(20,20,20,100,100,57.3);
        loop edges diag(20,20,20,100,100,100)); loop edges under HuberLoss(0.1), the first keyframe constant
Stored: the input graph; cost at the SVIn poses (only loop edges contribute there); residual vector / cost at a perturbed state;
pre-loss Jacobians of three 4-DoF edges there (40-digit central differences); the minimum of the 4-DoF problem (Gauss-Newton in
mpmath until |gradient| reaches the floor of the 1e-20 differencing step, < 1e-18).
Run:  python tests/golden/make_golden_pg.py     (writes pg.npz)
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from svin_amd import synthetic_pg as spg  # noqa: E402  (input data only)

mp.mp.dps = 40
D2R = mp.pi / 180


def ypr2R(y, p, r):
    y, p, r = y * D2R, p * D2R, r * D2R
    Rz = mp.matrix([[mp.cos(y), -mp.sin(y), 0], [mp.sin(y), mp.cos(y), 0], [0, 0, 1]])
    Ry = mp.matrix([[mp.cos(p), 0, mp.sin(p)], [0, 1, 0], [-mp.sin(p), 0, mp.cos(p)]])
    Rx = mp.matrix([[1, 0, 0], [0, mp.cos(r), -mp.sin(r)], [0, mp.sin(r), mp.cos(r)]])
    return Rz * Ry * Rx


def q2R(q):
    x, y, z, w = q
    return mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R2ypr(R):
    y = mp.atan2(R[1, 0], R[0, 0])
    p = mp.atan2(-R[2, 0], R[0, 0] * mp.cos(y) + R[1, 0] * mp.sin(y))
    r = mp.atan2(R[0, 2] * mp.sin(y) - R[1, 2] * mp.cos(y), -R[0, 1] * mp.sin(y) + R[1, 1] * mp.cos(y))
    return y / D2R, p / D2R, r / D2R


def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz]


def conj(q):
    return [-q[0], -q[1], -q[2], q[3]]


def wrap(d):
    return d - 360 if d > 180 else (d + 360 if d < -180 else d)


def huber(r):
    s = sum(x * x for x in r)
    if s <= mp.mpf("0.01"):
        return list(r)
    k = mp.sqrt(2 * mp.mpf("0.1") * mp.sqrt(s) - mp.mpf("0.01")) / mp.sqrt(s)
    return [x * k for x in r]


class Graph:
    def __init__(self, t_svin, q_svin, loops):
        self.n = len(t_svin)
        self.t0 = [[mp.mpf(float(x)) for x in r] for r in t_svin]
        self.q0 = []
        for q in q_svin:
            qq = [mp.mpf(float(x)) for x in q]
            nn = mp.sqrt(sum(c * c for c in qq))
            self.q0.append([c / nn for c in qq])
        self.ypr0 = [R2ypr(q2R(q)) for q in self.q0]
        self.loops = {int(k): (int(v[0]), [mp.mpf(float(x)) for x in v[1]], [mp.mpf(float(x)) for x in v[2]], mp.mpf(float(v[3])))
                      for k, v in loops.items()}

    def edge4(self, a, b, ya, ta, yb, tb, tm, ym, loop):
        R = ypr2R(ya, self.ypr0[a][1], self.ypr0[a][2])
        d = R.T * mp.matrix([tb[c] - ta[c] for c in range(3)])
        return [d[0] - tm[0], d[1] - tm[1], d[2] - tm[2], wrap(yb - ya - ym) / (10 if loop else 1)]

    def edges(self, six):
        out = []
        for i in range(self.n):
            for j in range(1, (4 if six else 2) + 1):
                if i - j >= 0:
                    out.append((i - j, i, False))
            if i in self.loops:
                out.append((self.loops[i][0], i, True))
        return out

    def meas(self, a, b, loop, six):
        if loop:
            _, rt, rq, ry = self.loops[b]
            return rt, (rq if six else ry)
        Ra = q2R(self.q0[a])
        tm = Ra.T * mp.matrix([self.t0[b][c] - self.t0[a][c] for c in range(3)])
        if six:
            return list(tm), qmul(conj(self.q0[a]), self.q0[b])
        return list(tm), self.ypr0[b][0] - self.ypr0[a][0]

    def residuals4(self, yaw, t):
        out = []
        for a, b, loop in self.edges(False):
            tm, ym = self.meas(a, b, loop, False)
            r = self.edge4(a, b, yaw[a], t[a], yaw[b], t[b], tm, ym, loop)
            out += huber(r) if loop else r
        return out

    def residuals6(self, q, t):
        out = []
        for a, b, loop in self.edges(True):
            tm, qm = self.meas(a, b, loop, True)
            pab = q2R(q[a]).T * mp.matrix([t[b][c] - t[a][c] for c in range(3)])
            dq = qmul(qm, conj(qmul(conj(q[a]), q[b])))
            w = [20, 20, 20, 100, 100, 100 if loop else mp.mpf("57.3")]
            r = [(pab[0] - tm[0]) * w[0], (pab[1] - tm[1]) * w[1], (pab[2] - tm[2]) * w[2], 2 * dq[0] * w[3], 2 * dq[1] * w[4], 2 * dq[2] * w[5]]
            out += huber(r) if loop else r
        return out


def half_sq(r):
    return sum(x * x for x in r) / 2


def fl(v):
    return np.array([float(x) for x in v])


def main():
    spec = spg.make_pose_graph(n=24, laps=2, loop_every=3, seed=41, radius=3.0, drift_yaw_deg=0.3, drift_t=0.02, loop_noise_t=0.01,
                               loop_noise_deg=0.1)
    loops = dict(spec.loops)
    k_out = sorted(loops)[1]                       # one loop measurement 0.5 m off: Huber region
    li, rt, rq, ry = loops[k_out]
    loops[k_out] = (li, rt + np.array([0.5, -0.2, 0.1]), rq, ry + 8.0)
    g = Graph(spec.t_svin, spec.q_svin, loops)
    yaw0 = [g.ypr0[k][0] for k in range(g.n)]
    out = dict(t_svin=spec.t_svin, q_svin=spec.q_svin, loop_cur=np.array(sorted(loops)), loop_old=np.array([loops[k][0] for k in sorted(loops)]),
               loop_t=np.array([loops[k][1] for k in sorted(loops)]), loop_q=np.array([loops[k][2] for k in sorted(loops)]),
               loop_yaw=np.array([loops[k][3] for k in sorted(loops)]), yaw0=fl(yaw0))
    out["cost4_initial"] = float(half_sq(g.residuals4(yaw0, g.t0)))
    out["cost6_initial"] = float(half_sq(g.residuals6(g.q0, g.t0)))
    # ---- a perturbed state (tangent perturbations stored, so the tests apply them through each side's own Plus)
    rng = np.random.default_rng(5)
    d4 = np.c_[rng.normal(0, 2.0, g.n), rng.normal(0, 0.05, (g.n, 3))]      # [dyaw(deg), dt]
    d4[0] = 0
    yaw_p = [wrap(yaw0[k] + mp.mpf(float(d4[k, 0]))) for k in range(g.n)]
    t_p = [[g.t0[k][c] + mp.mpf(float(d4[k, 1 + c])) for c in range(3)] for k in range(g.n)]
    r4 = g.residuals4(yaw_p, t_p)
    out.update(d4=d4, r4_pert=fl(r4), cost4_pert=float(half_sq(r4)))
    # pre-loss Jacobians of three edges at the perturbed state: [yaw, tx, ty, tz] of node a and of node b
    picks = [(10, 11, False), (9, 11, False), (loops[sorted(loops)[2]][0], sorted(loops)[2], True)]
    h = mp.mpf(10) ** -18
    JA, JB, RR = [], [], []
    for a, b, loop in picks:
        tm, ym = g.meas(a, b, loop, False)

        def f(xa, xb):
            return g.edge4(a, b, xa[0], xa[1:], xb[0], xb[1:], tm, ym, loop)
        xa, xb = [yaw_p[a]] + t_p[a], [yaw_p[b]] + t_p[b]
        Ja, Jb = np.zeros((4, 4)), np.zeros((4, 4))
        for c in range(4):
            for which, J in ((0, Ja), (1, Jb)):
                xp, xm = [list(xa), list(xb)], [list(xa), list(xb)]
                xp[which][c] += h
                xm[which][c] -= h
                col = [(p - m) / (2 * h) for p, m in zip(f(*xp), f(*xm))]
                J[:, c] = fl(col)
        JA.append(Ja), JB.append(Jb), RR.append(fl(f(xa, xb)))
    out.update(edge_a=np.array([p[0] for p in picks]), edge_b=np.array([p[1] for p in picks]), edge_loop=np.array([p[2] for p in picks]),
               edge_r=np.array(RR), edge_Ja=np.array(JA), edge_Jb=np.array(JB))
    d6 = np.c_[rng.normal(0, 0.05, (g.n, 3)), rng.normal(0, 0.02, (g.n, 3))]    # [dt, rotation vector applied on the left]
    d6[0] = 0
    q_p = []
    for k in range(g.n):
        v = [mp.mpf(float(x)) for x in d6[k, 3:]]
        nv = mp.sqrt(sum(x * x for x in v))
        dq = [0, 0, 0, 1] if nv == 0 else [mp.sin(nv / 2) * x / nv for x in v] + [mp.cos(nv / 2)]
        q_p.append(qmul(dq, g.q0[k]))
    t6 = [[g.t0[k][c] + mp.mpf(float(d6[k, c])) for c in range(3)] for k in range(g.n)]
    r6 = g.residuals6(q_p, t6)
    out.update(q6_pert=np.array([fl(q) for q in q_p]), t6_pert=np.array([fl(t) for t in t6]), r6_pert=fl(r6), cost6_pert=float(half_sq(r6)))
    # ---- the minimum of the 4-DoF problem: damped Gauss-Newton in mpmath from the SVIn poses
    free = list(range(1, g.n))

    def unpack(x):
        yaw, t = list(yaw0), [list(r) for r in g.t0]
        for i, k in enumerate(free):
            yaw[k] = x[4 * i]
            t[k] = [x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]]
        return yaw, t
    x = []
    for k in free:
        x += [yaw0[k]] + list(g.t0[k])
    hh = mp.mpf(10) ** -20
    for it in range(60):
        r = mp.matrix(g.residuals4(*unpack(x)))
        J = mp.zeros(len(r), len(x))
        for c in range(len(x)):
            xp, xm = list(x), list(x)
            xp[c] += hh
            xm[c] -= hh
            col = (mp.matrix(g.residuals4(*unpack(xp))) - mp.matrix(g.residuals4(*unpack(xm)))) / (2 * hh)
            for i in range(len(r)):
                J[i, c] = col[i]
        grad = J.T * r
        gn = mp.norm(grad)
        cost = half_sq(list(r))
        print("iter", it, "cost", mp.nstr(cost, 18), "|g|", mp.nstr(gn, 5))
        if gn < mp.mpf(10) ** -18:
            break
        step = mp.lu_solve(J.T * J, -grad)
        alpha = mp.mpf(1)
        while True:
            xn = [x[i] + alpha * step[i] for i in range(len(x))]
            if half_sq(g.residuals4(*unpack(xn))) <= cost or alpha < mp.mpf("1e-6"):
                break
            alpha /= 2
        x = xn
    yaw_m, t_m = unpack(x)
    out.update(cost4_min=float(cost), yaw4_min=fl(yaw_m), t4_min=np.array([fl(t) for t in t_m]))
    np.savez(os.path.join(HERE, "pg.npz"), **out)
    print("wrote pg.npz: cost4 initial", out["cost4_initial"], "min", out["cost4_min"], "cost6 initial", out["cost6_initial"])


if __name__ == "__main__":
    main()
