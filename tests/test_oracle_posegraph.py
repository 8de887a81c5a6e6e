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
