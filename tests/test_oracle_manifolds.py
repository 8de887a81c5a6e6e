"""Reduced pose manifolds in the oracle's solver (Map::resetParameterization, okvis_ceres/src/Map.cpp:513-543; PoseManifold3d / 4d /
2d, src/PoseManifold.cpp:173-466).  The reference only runs them (TestMap.cpp:146-156 solves once with Pose2d and checks nothing),
so the pin is by definition: the solution must be a STATIONARY point of the cost in exactly the directions the manifold leaves free
(gradient by central differences of an independent numpy cost, tests/helpers/tiny_window_cost.py), must not have moved in the
directions it holds, and for Pose3d / Pose4d -- whose reachable sets are fixed submanifolds -- must be the minimum scipy finds over
a global parametrisation of that submanifold."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from oracle import orc  # noqa: E402
import tiny_window_cost as tw  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FREE = {6: [0, 1, 2, 3, 4, 5], 3: [3, 4, 5], 4: [0, 1, 2, 5], 2: [3, 4]}   # PoseManifold.cpp:176-178, :279-282, :375-376


def disturbed_start(g):
    """tiny_window's start has position noise only: add an orientation error so that every direction has something to do"""
    return tw.pose_oplus(g["T1_init"], np.r_[0, 0, 0, 0.02, -0.015, 0.03])


def build_oracle(g, T1_start, manifold):
    m = orc.OracleMap()
    info = 64.0 / float(g["size"]) ** 2
    m.add_param(1, orc.BLOCK_POSE, g["T0"]); m.set_constant(1)
    m.add_param(2, orc.BLOCK_POSE, T1_start)
    for c in range(2):
        m.add_param(3 + c, orc.BLOCK_POSE, g["T_SC"][c]); m.set_constant(3 + c)
    for l in range(len(g["lm_init"])):
        m.add_param(10 + l, orc.BLOCK_HPOINT, np.r_[g["lm_init"][l], 1.0])
        for f, pose in enumerate((1, 2)):
            for c in range(2):
                m.add_reproj(orc.DIST_RADTAN, g["intr"], g["dist"], g["uv"][f, c, l], [[info, 0], [0, info]], orc.LOSS_CAUCHY, pose, 10 + l, 3 + c)
    assert m.reset_parameterization(2, manifold)
    orc.lib().orc_map_set_tolerances(m.h, 1e-16, 1e-16, 1e-16)
    return m


def check_solution(g, T_start, T1, lm, cost, manifold):
    """what any solver on that manifold must deliver (shared with tests/test_gpu_map.py)"""
    win = tw.TinyWindow(g)
    assert abs(win.cost(T1, lm) - cost) < 1e-9 * cost                      # the independent cost agrees with the solver's
    gp, gl = win.gradient(T1, lm)
    free = FREE[manifold]
    held = [k for k in range(6) if k not in free]
    scale = max(1.0, float(np.max(np.abs(win.gradient(T_start, lm)[0]))))
    assert np.max(np.abs(gp[free])) < 1e-5 * scale, (manifold, gp)         # stationary where it may move
    assert np.max(np.abs(gl)) < 1e-5 * scale
    if manifold != 6:
        assert np.max(np.abs(gp[held])) > 1e-3 * scale, (manifold, gp)     # ... and visibly not where it may not (the test has teeth)
    if manifold in (3, 2):
        assert np.array_equal(T1[:3], T_start[:3] / 1.0)                   # position held: never touched
    if manifold == 4:   # rotations about the world z axis compose: the total rotation since the start is one
        dq = tw.quat_mul(T1[3:], np.r_[-T_start[3:6], T_start[6]] / np.linalg.norm(T_start[3:]) ** 2)
        assert abs(dq[0]) < 1e-12 and abs(dq[1]) < 1e-12


@pytest.mark.parametrize("manifold", [3, 4, 2])
def test_reduced_manifold_solution_is_stationary_in_the_free_directions_only(manifold):
    g = np.load(os.path.join(GOLD, "tiny_window.npz"))
    T_start = disturbed_start(g)
    m = build_oracle(g, T_start, manifold)
    s = m.solve(500)
    T1 = m.get_param(2)
    lm = np.stack([m.get_param(10 + l)[:3] for l in range(len(g["lm_init"]))])
    assert s["final_cost"] < s["initial_cost"]
    check_solution(g, T_start / np.r_[1, 1, 1, [np.linalg.norm(T_start[3:])] * 4], T1, lm, s["final_cost"], manifold)


@pytest.mark.parametrize("manifold", [3, 4])
def test_reduced_manifold_minimum_matches_scipy_on_the_submanifold(manifold):
    """Pose3d: all orientations at the start position; Pose4d: any position, orientation = Rz(yaw) * start.  scipy's BFGS over
    (rotation vector | position, yaw) and the landmarks, started at the oracle's answer perturbed, must come back to its cost."""
    from scipy.optimize import minimize
    g = np.load(os.path.join(GOLD, "tiny_window.npz"))
    T_start = disturbed_start(g)
    m = build_oracle(g, T_start, manifold)
    s = m.solve(500)
    win = tw.TinyWindow(g)
    nL = len(g["lm_init"])

    def unpack(x):
        d6 = np.zeros(6)
        d6[FREE[manifold]] = x[:manifold]
        return tw.pose_oplus(T_start, d6), x[manifold:].reshape(nL, 3)

    # (a single oplus from the start reaches every point of either submanifold: all rotations / all positions and yaws)
    # started at the oracle's answer, every coordinate moved by 1e-3: BFGS must come back to it and find nothing lower
    T1 = m.get_param(2)
    lm1 = np.stack([m.get_param(10 + l)[:3] for l in range(nL)])
    dq = tw.quat_mul(T1[3:], np.r_[-T_start[3:6], T_start[6]] / np.linalg.norm(T_start[3:]) ** 2)
    rotvec = 2.0 * np.arctan2(np.linalg.norm(dq[:3]), dq[3]) * dq[:3] / max(np.linalg.norm(dq[:3]), 1e-300)
    d6 = np.r_[T1[:3] - T_start[:3], rotvec]
    rng = np.random.default_rng(manifold)
    x = np.r_[d6[FREE[manifold]], lm1.reshape(-1)]
    assert abs(win.cost(*unpack(x)) - s["final_cost"]) < 1e-9 * s["final_cost"]   # the parametrisation reproduces the oracle's point
    x = x + 1e-3 * rng.standard_normal(x.shape)
    for _ in range(3):
        x = minimize(lambda v: win.cost(*unpack(v)), x, method="BFGS", options=dict(gtol=1e-9, maxiter=400)).x
    c_scipy = win.cost(*unpack(x))
    T_scipy, _ = unpack(x)
    print("manifold", manifold, "oracle", s["final_cost"], "scipy", c_scipy)
    assert s["final_cost"] <= c_scipy * (1 + 1e-9)
    assert abs(s["final_cost"] - c_scipy) < 1e-6 * c_scipy
    assert np.linalg.norm(T1[:3] - T_scipy[:3]) < 1e-3 and min(np.linalg.norm(T1[3:] - T_scipy[3:]), np.linalg.norm(T1[3:] + T_scipy[3:])) < 1e-3


def test_reset_parameterization_argument_checks():
    g = np.load(os.path.join(GOLD, "tiny_window.npz"))
    m = build_oracle(g, g["T1_init"], 6)
    assert not m.reset_parameterization(999, 4)        # Map.cpp:514: unknown block
    assert not m.reset_parameterization(10, 4)         # a landmark cannot take a pose manifold
    assert m.reset_parameterization(2, 4) and m.reset_parameterization(2, 6)
