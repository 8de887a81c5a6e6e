"""Pose-graph oracle (oracle/orc_posegraph.cpp) pinned by the reference's own criteria: the reference differentiates
FourDOFError / FourDOFWeightError / PoseGraph3dErrorTerm with ceres::AutoDiffCostFunction (exact), so the analytic
Jacobians must agree with central differences taken through the same manifold Plus; the optimisation must pull a
drifted trajectory back onto the loop-closure constraints."""
import numpy as np
import pytest

from oracle import orc
from svin_amd import synthetic_pg as spg


@pytest.mark.parametrize("six", [False, True])
def test_edge_jacobians_match_central_differences(six):
    spec = spg.make_pose_graph(n=60, laps=3, loop_every=7, seed=3)
    pg = orc.OraclePoseGraph(six_dof=six)
    earliest, cur = spg.feed(pg, spec)
    nodes, nt, ne = pg.build(0, cur)
    d = 6 if six else 4
    worst = 0.0
    checked_loop = False
    for e in range(ne):
        a, b, is_loop, r0, Ja, Jb = pg.eval_edge(e)
        if not is_loop and e % 5:
            continue
        checked_loop |= is_loop
        for node, Jan in ((a, Ja), (b, Jb)):
            Jn = np.zeros((d, d))
            for c in range(d):
                h = 1e-6
                dv = np.zeros(d)
                dv[c] = h
                pg.perturb_node(node, dv)
                rp = pg.eval_edge(e)[3].copy()
                pg.perturb_node(node, -2 * dv)
                rm = pg.eval_edge(e)[3].copy()
                pg.perturb_node(node, dv)
                Jn[:, c] = (rp - rm) / (2 * h)
            worst = max(worst, np.max(np.abs(Jn - Jan)) / max(1.0, np.max(np.abs(Jan))))
    assert checked_loop
    assert worst < 1e-6, worst


@pytest.mark.parametrize("six", [False, True])
def test_loop_closures_pull_the_drifted_trajectory_back(six):
    spec = spg.make_pose_graph(n=240, laps=4, loop_every=12, seed=5)
    pg = orc.OraclePoseGraph(six_dof=six, max_iterations=30)
    earliest, cur = spg.feed(pg, spec)
    before = spg.align_error(spec.t_svin, spec)
    s = pg.optimize(earliest, cur)
    T, Q = pg.poses()
    after = spg.align_error(T, spec)
    print("six" if six else "four", "dof: cost", s["initial_cost"], "->", s["final_cost"], "iterations", s["iterations"],
          "position RMS", before, "->", after)
    assert s["final_cost"] < 0.5 * s["initial_cost"]
    assert after < 0.5 * before


@pytest.mark.parametrize("six", [False, True])
def test_envelope_and_dense_solvers_agree(six):
    spec = spg.make_pose_graph(n=150, laps=3, loop_every=10, seed=9)
    res = []
    for env in (False, True):
        pg = orc.OraclePoseGraph(six_dof=six, envelope=env)
        earliest, cur = spg.feed(pg, spec)
        s = pg.optimize(earliest, cur)
        res.append((s, pg.poses()))
    assert res[0][0]["iterations"] == res[1][0]["iterations"]
    assert abs(res[0][0]["final_cost"] - res[1][0]["final_cost"]) <= 1e-9 * res[0][0]["final_cost"]
    assert np.max(np.abs(res[0][1][0] - res[1][1][0])) < 1e-8


def test_reference_iteration_limits_and_fixed_first_keyframe():
    spec = spg.make_pose_graph(n=120, laps=3, loop_every=10, seed=11)
    for six, limit in ((False, 10), (True, 5)):
        pg = orc.OraclePoseGraph(six_dof=six)
        earliest, cur = spg.feed(pg, spec)
        s = pg.optimize(earliest, cur)
        assert s["iterations"] <= limit                       # PoseGraph.cpp:243 / :416
        T, Q = pg.poses()
        assert np.allclose(T[earliest], spec.t_svin[earliest])  # constant block (PoseGraph.cpp:286-289 / :449-452)
        assert np.allclose(T[:earliest], spec.t_svin[:earliest])  # keyframes before the earliest loop are untouched


@pytest.mark.parametrize("six", [False, True])
def test_drift_moves_later_keyframes_and_reoptimisation_starts_from_svin_poses(six):
    """PoseGraph.cpp:127-132 (addKeyframe applies the drift), :262-275 (the problem is built from getSVInPose),
    :356-375 (drift from the current keyframe, applied to the keyframes after it)"""
    spec = spg.make_pose_graph(n=200, laps=2, loop_every=20, seed=4)
    pg = orc.OraclePoseGraph(six_dof=six)
    for k in range(150):
        pg.add_keyframe(k, 1, spec.t_svin[k], spec.q_svin[k], spec.loops.get(k))
    earliest = min(v[0] for k, v in spec.loops.items() if k <= 120)
    s1 = pg.optimize(earliest, 120)
    T1, Q1 = pg.poses()
    yaw, r, t = pg.drift()
    cur_svin = spec.t_svin[120]
    assert np.allclose(T1[120], r @ cur_svin + t, atol=1e-12)          # t_drift's definition
    assert np.allclose(T1[121:150], spec.t_svin[121:150] @ r.T + t, atol=1e-12)
    assert np.allclose(T1[:earliest], spec.t_svin[:earliest])           # before the earliest loop: untouched
    for k in range(150, 200):                                             # new keyframes arrive drift-corrected
        pg.add_keyframe(k, 1, spec.t_svin[k], spec.q_svin[k], spec.loops.get(k))
    T2, _ = pg.poses()
    assert np.allclose(T2[150:], spec.t_svin[150:] @ r.T + t, atol=1e-12)
    # a second pass over the same range starts from the SVIn poses again: identical summary
    s2 = pg.optimize(earliest, 120)
    assert s2["initial_cost"] == s1["initial_cost"] and s2["final_cost"] == s1["final_cost"]


def _numpy_residuals(spec, upto, six, T, Q_or_yaw, earliest):
    """Independent numpy restatement of the three error terms straight from the reference's functors
    (FourDOFError / FourDOFWeightError PoseGraph.h:134-231, PoseGraph3dErrorTerm Pose3DError.h:103-147) and of the
    problem construction (PoseGraph.cpp:262-332 / :436-489); loop blocks carry Huber(0.1) as sqrt(rho(s)) r / |r|."""
    ypr0 = np.array([spg.R2ypr(spg.q2R(q)) for q in spec.q_svin[:upto]])
    blocks = []

    def huber(r):
        s = float(r @ r)
        if s <= 0.01:
            return r
        return r * np.sqrt(2 * 0.1 * np.sqrt(s) - 0.01) / np.sqrt(s)

    def qmul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])

    def conj(q):
        return np.array([-q[0], -q[1], -q[2], q[3]])

    for i in range(earliest, upto):
        nseq = 4 if six else 2
        for j in range(1, nseq + 1):
            a = i - j
            if a < earliest or spec.sequence[a] != spec.sequence[i]:
                continue
            Ra0 = spg.q2R(spec.q_svin[a])
            t_meas = Ra0.T @ (spec.t_svin[i] - spec.t_svin[a])
            if not six:
                Ra = spg.ypr2R([Q_or_yaw[a], ypr0[a, 1], ypr0[a, 2]])
                r = np.zeros(4)
                r[:3] = Ra.T @ (T[i] - T[a]) - t_meas
                d = Q_or_yaw[i] - Q_or_yaw[a] - (ypr0[i, 0] - ypr0[a, 0])
                r[3] = d - 360 if d > 180 else (d + 360 if d < -180 else d)
                blocks.append(r)
            else:
                q_meas = qmul(conj(spec.q_svin[a]), spec.q_svin[i])
                qa, qb = Q_or_yaw[a], Q_or_yaw[i]
                p_ab = spg.q2R(qa).T @ (T[i] - T[a])
                dq = qmul(q_meas, conj(qmul(conj(qa), qb)))
                r = np.concatenate([p_ab - t_meas, 2 * dq[:3]]) * np.array([20, 20, 20, 100, 100, 57.3])
                blocks.append(r)
        if i in spec.loops and spec.loops[i][0] >= earliest:
            a, rt, rq, ryaw = spec.loops[i]
            if not six:
                Ra = spg.ypr2R([Q_or_yaw[a], ypr0[a, 1], ypr0[a, 2]])
                r = np.zeros(4)
                r[:3] = Ra.T @ (T[i] - T[a]) - rt
                d = Q_or_yaw[i] - Q_or_yaw[a] - ryaw
                r[3] = (d - 360 if d > 180 else (d + 360 if d < -180 else d)) / 10.0
            else:
                qa, qb = Q_or_yaw[a], Q_or_yaw[i]
                p_ab = spg.q2R(qa).T @ (T[i] - T[a])
                dq = qmul(rq, conj(qmul(conj(qa), qb)))
                r = np.concatenate([p_ab - rt, 2 * dq[:3]]) * np.array([20, 20, 20, 100, 100, 100.0])
            blocks.append(huber(r))
    return np.concatenate(blocks)


@pytest.mark.parametrize("six", [False, True])
def test_cost_matches_an_independent_numpy_restatement(six):
    spec = spg.make_pose_graph(n=80, laps=2, loop_every=8, seed=23)
    pg = orc.OraclePoseGraph(six_dof=six)
    earliest, cur = spg.feed(pg, spec)
    pg.build(earliest, cur)
    yaw0 = np.array([spg.R2ypr(spg.q2R(q))[0] for q in spec.q_svin])
    f = _numpy_residuals(spec, spec.n, six, spec.t_svin, spec.q_svin if six else yaw0, earliest)
    assert abs(0.5 * f @ f - pg.cost()) <= 1e-10 * max(1.0, pg.cost())


def test_converged_4dof_solution_is_the_minimum_scipy_finds():
    """fixed-point parity of the Levenberg-Marquardt restatement: run to convergence, it must land on the minimum an
    independent solver (scipy least_squares on the numpy restatement above) finds from the same start"""
    from scipy.optimize import least_squares
    spec = spg.make_pose_graph(n=60, laps=2, loop_every=6, seed=29)
    pg = orc.OraclePoseGraph(six_dof=False, max_iterations=200)
    earliest, cur = spg.feed(pg, spec)
    s = pg.optimize(earliest, cur)
    T, _ = pg.poses()
    n = spec.n
    yaw0 = np.array([spg.R2ypr(spg.q2R(q))[0] for q in spec.q_svin])
    free = np.arange(earliest + 1, n)

    def fun(x):
        yaw, t = yaw0.copy(), spec.t_svin.copy()
        yaw[free] = x[:len(free)]
        t[free] = x[len(free):].reshape(-1, 3)
        return _numpy_residuals(spec, n, False, t, yaw, earliest)

    x0 = np.concatenate([yaw0[free], spec.t_svin[free].ravel()])
    sol = least_squares(fun, x0, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=400)
    print("oracle final cost", s["final_cost"], "iterations", s["iterations"], "scipy cost", sol.cost)
    assert abs(s["final_cost"] - sol.cost) <= 2e-5 * max(sol.cost, 1e-12)
    t_scipy = spec.t_svin.copy()
    t_scipy[free] = sol.x[len(free):].reshape(-1, 3)
    assert np.max(np.abs(T - t_scipy)) < 2e-3


def _feed_golden(pg, g):
    loops = {int(k): (int(o), g["loop_t"][i], g["loop_q"][i], float(g["loop_yaw"][i])) for i, (k, o) in enumerate(zip(g["loop_cur"], g["loop_old"]))}
    for k in range(len(g["t_svin"])):
        pg.add_keyframe(k, 1, g["t_svin"][k], g["q_svin"][k], loops.get(k))
    assert min(v[0] for v in loops.values()) == 0
    return len(g["t_svin"]) - 1


def test_error_terms_match_the_mpmath_fixture():
    """FourDOFError / FourDOFWeightError / PoseGraph3dErrorTerm against tests/golden/pg.npz (40-digit mpmath restatement of
    the reference's functors, tests/golden/make_golden_pg.py): cost at the SVIn poses (one loop edge in the Huber region, yaw
    differences crossing +-180 degrees), residual cost at a perturbed state, and the pre-loss Jacobians of three 4-DoF edges"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pg.npz"))
    pg = orc.OraclePoseGraph(six_dof=False)
    cur = _feed_golden(pg, g)
    nodes, nt, ne = pg.build(0, cur)
    assert nodes == len(g["t_svin"]) and nt == 4 * (nodes - 1)
    assert abs(pg.cost() - g["cost4_initial"]) < 1e-12 * g["cost4_initial"]
    for k in range(1, nodes):
        pg.perturb_node(k, g["d4"][k])
    assert abs(pg.cost() - g["cost4_pert"]) < 1e-12 * g["cost4_pert"]
    found = 0
    for e in range(ne):
        a, b, is_loop, r, Ja, Jb = pg.eval_edge(e)
        for i in range(len(g["edge_a"])):
            if (a, b, is_loop) == (int(g["edge_a"][i]), int(g["edge_b"][i]), bool(g["edge_loop"][i])):
                found += 1
                assert np.max(np.abs(r - g["edge_r"][i])) < 1e-12
                assert np.max(np.abs(Ja - g["edge_Ja"][i])) < 1e-12 and np.max(np.abs(Jb - g["edge_Jb"][i])) < 1e-12
    assert found == len(g["edge_a"])
    # 6 DoF
    pg = orc.OraclePoseGraph(six_dof=True)
    cur = _feed_golden(pg, g)
    nodes, nt, ne = pg.build(0, cur)
    assert abs(pg.cost() - g["cost6_initial"]) < 1e-12 * g["cost6_initial"]
    rng = np.random.default_rng(5)
    rng.normal(0, 2.0, nodes), rng.normal(0, 0.05, (nodes, 3))          # the generator's draws before d6
    d6 = np.c_[rng.normal(0, 0.05, (nodes, 3)), rng.normal(0, 0.02, (nodes, 3))]
    d6[0] = 0
    assert np.allclose(g["t_svin"] + d6[:, :3], g["t6_pert"], atol=1e-15)
    for k in range(1, nodes):
        v = d6[k, 3:]
        half = np.sin(np.linalg.norm(v) / 2) * v / np.linalg.norm(v)    # the oracle's tangent is [sin|d| d/|d|, cos|d|]: d = asin(...)
        d = np.arcsin(np.linalg.norm(half)) * half / np.linalg.norm(half)
        pg.perturb_node(k, np.r_[d6[k, :3], d])
    assert abs(pg.cost() - g["cost6_pert"]) < 1e-11 * g["cost6_pert"]


def test_converged_4dof_solution_is_the_mpmath_minimum():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pg.npz"))
    pg = orc.OraclePoseGraph(six_dof=False, max_iterations=200)
    cur = _feed_golden(pg, g)
    s = pg.optimize(0, cur)
    T, _ = pg.poses()
    print("oracle", s, "mpmath minimum", float(g["cost4_min"]))
    assert s["final_cost"] >= g["cost4_min"] * (1 - 1e-12)
    assert s["final_cost"] - g["cost4_min"] < 2e-5 * g["cost4_min"]
    assert np.max(np.abs(T - g["t4_min"])) < 2e-3
