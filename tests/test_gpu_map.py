"""okvis::ceres::Map as a graph builder on the HIP path (Map.cpp:255-376): parameter blocks and residual blocks
added one by one through the C ABI (svin_ba_map_*), outside any frame -- the shape of the reference's own tests
(okvis_ceres/test/TestMap.cpp, TestHomogeneousPointError.cpp:60-99, TestPoseError) -- and the independent fixed point
tests/golden/tiny_window.npz (scipy least_squares, a solver that shares nothing with this code) on the device solver."""
import os

import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def quat_close(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def test_tiny_window_fixed_point_matches_independent_minimiser(gpu_lib):
    """The only solver-independent pin of optimize(): 2 poses (one constant) / 2 constant extrinsics / 12 landmarks / 48
    Cauchy-robustified reprojection residuals; the HIP path must reach the minimum scipy's least_squares found, to the
    tolerances tests/test_oracle_golden.py holds the oracle to (criterion of the reference: TestEstimator.cpp:209-212)."""
    from svin_amd.estimator import Estimator
    g = np.load(os.path.join(GOLD, "tiny_window.npz"))
    est = Estimator(0)
    for c in range(2):
        est.add_camera(syn.DIST_RADTAN, g["intr"], g["dist"], 752, 480, [0.0, 0.0, 0.0, 0.0])
    size = float(g["size"])
    info = 64.0 / (size * size) * np.eye(2)
    assert est.map_add_parameter_block(1, est.BLOCK_POSE, g["T0"]) and est.set_parameter_block_constant(1)
    assert est.map_add_parameter_block(2, est.BLOCK_POSE, g["T1_init"])
    for c in range(2):
        assert est.map_add_parameter_block(3 + c, est.BLOCK_POSE, g["T_SC"][c]) and est.set_parameter_block_constant(3 + c)
    nL = len(g["lm_init"])
    for l in range(nL):
        assert est.map_add_parameter_block(10 + l, est.BLOCK_HOMOGENEOUS_POINT, np.r_[g["lm_init"][l], 1.0])
        for f, pose in enumerate((1, 2)):
            for c in range(2):
                assert est.map_add_reprojection_error(pose, 10 + l, 3 + c, c, g["uv"][f, c, l], info) != 0
    assert not est.map_add_parameter_block(2, est.BLOCK_POSE, g["T1_init"])          # Map.cpp:257-260: a known id is refused
    est.set_solver_options(1e-16, 1e-16, 1e-16)
    est.optimize(500)
    s = est.summary()
    T1 = est.get_parameter_block(2)
    lm = np.stack([est.get_parameter_block(10 + l)[:3] for l in range(nL)])
    print("gpu cost", s["final_cost"], "scipy cost", float(g["cost"]), "dT", np.linalg.norm(T1[:3] - g["T1_opt"][:3]), "dlm",
          np.max(np.abs(lm - g["lm_opt"])), "iterations", s["iterations"])
    assert s["final_cost"] <= float(g["cost"]) * (1 + 1e-9)
    assert abs(s["final_cost"] - float(g["cost"])) < 1e-4 * float(g["cost"])
    assert np.linalg.norm(T1[:3] - g["T1_opt"][:3]) < 2e-3
    assert quat_close(T1[3:], g["T1_opt"][3:]) < 1e-3
    assert np.max(np.abs(lm - g["lm_opt"])) < 3e-2
    assert np.array_equal(est.get_parameter_block(1), g["T0"] / np.r_[1, 1, 1, [np.linalg.norm(g["T0"][3:])] * 4])  # the constant pose stayed
    # the same problem through the oracle's Map: identical fixed point, to the tolerance the solvers converge to
    from oracle import orc
    m = orc.OracleMap()
    L = orc.lib()
    m.add_param(1, orc.BLOCK_POSE, g["T0"]); m.set_constant(1)
    m.add_param(2, orc.BLOCK_POSE, g["T1_init"])
    for c in range(2):
        m.add_param(3 + c, orc.BLOCK_POSE, g["T_SC"][c]); m.set_constant(3 + c)
    for l in range(nL):
        m.add_param(10 + l, orc.BLOCK_HPOINT, np.r_[g["lm_init"][l], 1.0])
        for f, pose in enumerate((1, 2)):
            for c in range(2):
                m.add_reproj(orc.DIST_RADTAN, g["intr"], g["dist"], g["uv"][f, c, l], info, orc.LOSS_CAUCHY, pose, 10 + l, 3 + c)
    L.orc_map_set_tolerances(m.h, 1e-16, 1e-16, 1e-16)
    so = m.solve(500)
    To = m.get_param(2)
    assert abs(s["final_cost"] - so["final_cost"]) < 1e-9 * so["final_cost"]
    assert np.linalg.norm(T1[:3] - To[:3]) < 1e-6 and quat_close(T1[3:], To[3:]) < 1e-6


def test_homogeneous_point_errors_alone_converge_to_zero(gpu_lib):
    """TestHomogeneousPointError.cpp:57-99: 100 points, one HomogeneousPointError (variance 0.1) each, points disturbed by
    0.2, solve: 'this must converge to zero, since it is not an overdetermined system' (final_cost < 1e-10).  A graph without
    a single pose: the reduced camera system is empty."""
    from svin_amd.estimator import Estimator
    rng = np.random.default_rng(5)
    est = Estimator(0)
    pts, rids = [], []
    for i in range(100):
        p = np.r_[100.0 * rng.uniform(-1, 1, 3), 1.0]
        pts.append(p)
        assert est.map_add_parameter_block(1000 + i, est.BLOCK_HOMOGENEOUS_POINT, p)
        rid = est.add_homogeneous_point_error(1000 + i, p, variance=0.1)
        assert rid != 0
        rids.append(rid)
        assert est.set_parameter_block(1000 + i, p + np.r_[0.2 * rng.uniform(-1, 1, 3), 0.0])
    est.optimize(50)
    s = est.summary()
    print("homogeneous point errors alone:", s)
    assert s["final_cost"] < 1e-10
    for i in range(100):
        assert np.max(np.abs(est.get_parameter_block(1000 + i) - pts[i])) < 1e-6
    # Map::removeResidualBlock / removeParameterBlock
    assert est.map_remove_residual_block(rids[0]) and not est.map_remove_residual_block(rids[0])
    assert est.map_remove_parameter_block(1000) and not est.parameter_block_exists(1000)
    assert est.residuals_of(1001) == [rids[1]]
    assert est.map_remove_parameter_block(1001)
    with pytest.raises(RuntimeError):   # the residual went with its block (Map.cpp:322-333)
        est.parameters_of(rids[1])


@pytest.mark.parametrize("constant_landmarks", [False, True])
def test_map_built_window_against_oracle_map(gpu_lib, constant_landmarks):
    """TestMap.cpp:60-150: one pose (with a weak PoseError), a constant extrinsics block, 300 landmarks -- CONSTANT as in the
    reference ("no point optimization", :93) or variable under a weak HomogeneousPointError each --, Cauchy-robustified
    reprojection residuals of an equidistant camera; some residuals and blocks removed again.  Cost, iteration count and fixed
    point against the oracle's Map (the CPU restatement of Map.cpp) built by the same calls."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    rng = np.random.default_rng(11)
    intr, dist = [350.0, 360.0, 378.0, 238.0], [-0.21, 0.14, 0.0006, 0.0003]   # PinholeCamera::createTestObject (PinholeCamera.hpp:276-280)
    T_WS = np.r_[rng.uniform(-3, 3, 3), 0, 0, 0, 1.0]
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    T_WS[3:] = q
    T_SC = np.r_[0.1, -0.05, 0.02, 0.0, 0.0, 0.0, 1.0]
    est, m, L = Estimator(0), orc.OracleMap(), orc.lib()
    est.add_camera(syn.DIST_EQUIDISTANT, intr, dist, 752, 480, [0, 0, 0, 0])
    T_init = T_WS.copy(); T_init[:3] += 0.05 * rng.normal(size=3)
    dq = np.r_[0.01 * rng.normal(size=3), 1.0]; dq /= np.linalg.norm(dq)
    x, y, z, w = T_WS[3:]; a, b, c, d = dq
    T_init[3:] = [w * a + x * d + y * c - z * b, w * b - x * c + y * d + z * a, w * c + x * b - y * a + z * d, w * d - x * a - y * b - z * c]
    assert est.map_add_parameter_block(1, est.BLOCK_POSE, T_init) and est.map_add_parameter_block(2, est.BLOCK_POSE, T_SC)
    assert est.set_parameter_block_constant(2)
    m.add_param(1, orc.BLOCK_POSE, T_init); m.add_param(2, orc.BLOCK_POSE, T_SC); m.set_constant(2)
    info6 = np.diag([1e-2] * 3 + [1e-1] * 3)
    assert est.map_add_pose_error(1, T_init, info6) != 0
    L.orc_map_add_pose_error(m.h, orc.dptr(orc.arr(T_init)), orc.dptr(orc.arr(info6)), 1)
    # rotation of T_WS and T_SC to place points in front of the camera
    def rot(qv):
        x, y, z, w = qv
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    Rws, Rsc = rot(T_WS[3:]), rot(T_SC[3:])
    rids, n = [], 300
    for i in range(n):
        pc = np.r_[rng.uniform(-1.5, 1.5, 2), 1.0] * (3.0 * (i % 10) + 2.0)
        pw = Rws @ (Rsc @ pc + T_SC[:3]) + T_WS[:3]
        # equidistant projection of the true point + pixel noise
        r = np.hypot(pc[0], pc[1]); th = np.arctan2(r, pc[2])
        thd = th * (1 + dist[0] * th ** 2 + dist[1] * th ** 4 + dist[2] * th ** 6 + dist[3] * th ** 8)
        s = thd / r if r > 1e-8 else 1.0
        uv = np.array([intr[0] * s * pc[0] + intr[2], intr[1] * s * pc[1] + intr[3]]) + rng.uniform(-1, 1, 2)
        hp = np.r_[pw + 0.05 * rng.normal(size=3), 1.0]
        assert est.map_add_parameter_block(10 + i, est.BLOCK_HOMOGENEOUS_POINT, hp)
        m.add_param(10 + i, orc.BLOCK_HPOINT, hp)
        rid = est.map_add_reprojection_error(1, 10 + i, 2, 0, uv, np.eye(2))
        ro = m.add_reproj(orc.DIST_EQUIDISTANT, intr, dist, uv, np.eye(2), orc.LOSS_CAUCHY, 1, 10 + i, 2)
        assert rid != 0
        if constant_landmarks:
            assert est.set_parameter_block_constant(10 + i) and est.is_parameter_block_constant(10 + i)
            m.set_constant(10 + i)
        else:
            est.add_homogeneous_point_error(10 + i, hp, variance=4.0)
            m.add_hpoint_error(hp, 4.0, 10 + i)
        rids.append((rid, ro))
        if i % 10 == 0:
            if i % 20 == 0:   # "randomly delete some just for fun to test" (TestMap.cpp:117-122)
                assert est.map_remove_parameter_block(10 + i)
                m.remove_param(10 + i)
            else:
                assert est.map_remove_residual_block(rid)
                m.remove_residual(ro)
    est.set_solver_options(1e-14, 1e-14, 1e-14)
    L.orc_map_set_tolerances(m.h, 1e-14, 1e-14, 1e-14)
    est.optimize(30)
    so = m.solve(30)
    s = est.summary()
    T, To = est.get_parameter_block(1), m.get_param(1)
    print("map window: gpu", s["final_cost"], s["iterations"], "oracle", so["final_cost"], so["iterations"], "dT", np.linalg.norm(T[:3] - To[:3]))
    assert s["iterations"] == so["iterations"]
    assert abs(s["final_cost"] - so["final_cost"]) < 1e-9 * so["final_cost"]
    assert np.linalg.norm(T[:3] - To[:3]) < 1e-8 and quat_close(T[3:], To[3:]) < 1e-8
    if constant_landmarks:   # the points did not move
        for i in (1, 7, 123):
            assert np.array_equal(est.get_parameter_block(10 + i), m.get_param(10 + i))
    # TestMap.cpp:140-144: converged to the true pose within the test's tolerances
    assert quat_close(T[3:], T_WS[3:]) * 2 < 1e-2 and np.linalg.norm(T[:3] - T_WS[:3]) < 1e-1


def test_imu_sonar_depth_errors_through_the_map_interface(gpu_lib):
    """Map::addResidualBlock (Map.cpp:341-376) for the factor kinds okvis::Estimator otherwise creates inside addStates --
    ImuError, SonarError, DepthError -- on blocks named by the caller (svin_ba_map_add_imu_error / _sonar_error / _depth_error):
    two poses + speed / bias blocks tied by an IMU factor, priors on the first pair, a depth and a sonar term on the second pose,
    reprojection residuals of 60 landmarks; cost, iteration count and the optimum against the oracle's Map built by the same calls."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=2, L=60, n_obs=None, seed=31, frame_dt=0.4)
    est, m, L = Estimator(0), orc.OracleMap(), orc.lib()
    cam0 = spec.cameras[0]
    for cam in spec.cameras:
        est.add_camera(cam["model"], cam["intr"], cam["dist"], cam["width"], cam["height"], [0, 0, 0, 0])
    POSE, SB, EXT, LM0 = (1, 3), (2, 4), (5, 6), 100
    T = [spec.T_WS_true[0].copy(), spec.T_WS_init[1].copy()]
    sb = [spec.sb_true[0].copy(), spec.sb_init[1].copy()]
    for k in range(2):
        assert est.map_add_parameter_block(POSE[k], est.BLOCK_POSE, T[k]) and est.map_add_parameter_block(SB[k], est.BLOCK_SPEED_AND_BIAS, sb[k])
        m.add_param(POSE[k], orc.BLOCK_POSE, T[k]); m.add_param(SB[k], orc.BLOCK_SPEEDBIAS, sb[k])
    for c, cam in enumerate(spec.cameras):
        assert est.map_add_parameter_block(EXT[c], est.BLOCK_POSE, cam["T_SC"]) and est.set_parameter_block_constant(EXT[c])
        m.add_param(EXT[c], orc.BLOCK_POSE, cam["T_SC"]); m.set_constant(EXT[c])
    info6, info9 = np.diag([1e6] * 3 + [1e6] * 3), np.diag([1e2] * 3 + [1e4] * 3 + [1e2] * 3)
    assert est.map_add_pose_error(POSE[0], T[0], info6) and est.map_add_speed_and_bias_error(SB[0], sb[0], info9)
    L.orc_map_add_pose_error(m.h, orc.dptr(orc.arr(T[0])), orc.dptr(orc.arr(info6)), POSE[0])
    L.orc_map_add_speedbias_error(m.h, orc.dptr(orc.arr(sb[0])), 1e-2, 1e-4, 1e-2, SB[0])
    # the IMU samples between the two frames, as addStates would be handed them
    t = spec.imu_t[:, 0].astype(float) - float(spec.imu_t[0, 0]) + 1e-9 * spec.imu_t[:, 1]
    f = spec.stamps[:, 0].astype(float) - float(spec.imu_t[0, 0]) + 1e-9 * spec.stamps[:, 1]
    sel = (t >= f[0] - 0.02) & (t <= f[1] + 0.02)
    ids4 = [POSE[0], SB[0], POSE[1], SB[1]]
    t0, t1 = (int(spec.stamps[0, 0]), int(spec.stamps[0, 1])), (int(spec.stamps[1, 0]), int(spec.stamps[1, 1]))
    rid_imu = est.map_add_imu_error(ids4, spec.imu_t[sel], spec.imu_meas[sel], spec.imu_params, t0, t1)
    assert rid_imu != 0
    m.add_imu(spec.imu_t[sel], spec.imu_meas[sel], orc.imu_params_vector(spec.imu_params), t0, t1, ids4)
    # depth and sonar on the second pose
    depth, first_depth = 1.3, 0.2
    assert est.map_add_depth_error(POSE[1], depth, 5.0, first_depth) != 0
    L.orc_map_add_depth_error(m.h, depth, 5.0, first_depth, POSE[1])
    patch = spec.T_WS_true[1][:3] + np.array([[0.9, 0.3, 0.1], [0.95, 0.28, 0.12], [0.88, 0.33, 0.08]])
    rng_m = float(np.linalg.norm(patch.mean(0) - spec.T_WS_true[1][:3])) + 0.01
    assert est.map_add_sonar_error(POSE[1], rng_m, 0.3, 1.0, patch) != 0
    Tid = np.r_[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    L.orc_map_add_sonar_error(m.h, orc.dptr(orc.arr(Tid)), rng_m, 0.3, 1.0, len(patch), orc.dptr(orc.arr(patch)), POSE[1])
    # landmarks and their reprojection residuals
    for l in range(spec.L):
        assert est.map_add_parameter_block(LM0 + l, est.BLOCK_HOMOGENEOUS_POINT, spec.lm_init[l])
        m.add_param(LM0 + l, orc.BLOCK_HPOINT, spec.lm_init[l])
    n_res = 0
    for i in range(spec.N):
        k, c, l = int(spec.obs_frame[i]), int(spec.obs_cam[i]), int(spec.obs_lm[i])
        w = 64.0 / spec.obs_size[i] ** 2
        assert est.map_add_reprojection_error(POSE[k], LM0 + l, EXT[c], c, spec.obs_uv[i], [w, 0, 0, w]) != 0
        cam = spec.cameras[c]
        m.add_reproj(cam["model"], cam["intr"], cam["dist"], spec.obs_uv[i], [w, 0, 0, w], orc.LOSS_CAUCHY, POSE[k], LM0 + l, EXT[c])
        n_res += 1
    assert n_res > 100
    est.set_solver_options(1e-14, 1e-14, 1e-14)
    L.orc_map_set_tolerances(m.h, 1e-14, 1e-14, 1e-14)
    est.optimize(40)
    s, so = est.summary(), m.solve(40)
    print("map-built IMU / sonar / depth graph: gpu", s, "oracle", so)
    assert s["iterations"] == so["iterations"]
    assert abs(s["initial_cost"] - so["initial_cost"]) <= 1e-9 * so["initial_cost"]
    assert abs(s["final_cost"] - so["final_cost"]) <= 1e-9 * max(so["final_cost"], 1e-12)
    for k in range(2):
        Tg, To = est.get_parameter_block(POSE[k]), m.get_param(POSE[k])
        assert np.linalg.norm(Tg[:3] - To[:3]) < 1e-8 and quat_close(Tg[3:], To[3:]) < 1e-8
        assert np.max(np.abs(est.get_parameter_block(SB[k]) - m.get_param(SB[k]))) < 1e-7
    # Map::removeResidualBlock on the IMU factor
    assert est.map_remove_residual_block(rid_imu) and not est.map_remove_residual_block(rid_imu)


@pytest.mark.parametrize("manifold,param", [(3, 2), (4, 3), (2, 4)])
def test_reduced_pose_manifolds_match_the_oracle_and_the_definition(gpu_lib, manifold, param):
    """Map::resetParameterization (Map.cpp:513-543) with Pose3d / Pose4d / Pose2d (PoseManifold.cpp:173-466) on the device: the six
    rows of the pose stay in the reduced system and k_lock_rows strikes out the held ones before the solve.  Same problem as
    tests/test_oracle_manifolds.py: the device must end where the oracle's reduced-manifold solve ends, and pass the same
    definition checks (stationary in the free directions only, held directions untouched)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    import test_oracle_manifolds as tom
    from svin_amd.estimator import Estimator
    g = np.load(os.path.join(GOLD, "tiny_window.npz"))
    T_start = tom.disturbed_start(g)
    est = Estimator(0)
    for c in range(2):
        est.add_camera(syn.DIST_RADTAN, g["intr"], g["dist"], 752, 480, [0.0, 0.0, 0.0, 0.0])
    info = 64.0 / float(g["size"]) ** 2 * np.eye(2)
    assert est.map_add_parameter_block(1, est.BLOCK_POSE, g["T0"]) and est.set_parameter_block_constant(1)
    assert est.map_add_parameter_block(2, est.BLOCK_POSE, T_start)
    for c in range(2):
        assert est.map_add_parameter_block(3 + c, est.BLOCK_POSE, g["T_SC"][c]) and est.set_parameter_block_constant(3 + c)
    nL = len(g["lm_init"])
    for l in range(nL):
        assert est.map_add_parameter_block(10 + l, est.BLOCK_HOMOGENEOUS_POINT, np.r_[g["lm_init"][l], 1.0])
        for f, pose in enumerate((1, 2)):
            for c in range(2):
                assert est.map_add_reprojection_error(pose, 10 + l, 3 + c, c, g["uv"][f, c, l], info) != 0
    assert est.parameterization(2) == est.POSE6D
    assert est.reset_parameterization(2, param) and est.parameterization(2) == param
    assert not est.reset_parameterization(999, param)                       # unknown block: the reference's false
    with pytest.raises(RuntimeError):
        est.reset_parameterization(10, param)                              # a landmark cannot take a pose manifold
    est.set_solver_options(1e-16, 1e-16, 1e-16)
    est.optimize(500)
    s = est.summary()
    T1 = est.get_parameter_block(2)
    lm = np.stack([est.get_parameter_block(10 + l)[:3] for l in range(nL)])
    Tn = T_start / np.r_[1, 1, 1, [np.linalg.norm(T_start[3:])] * 4]
    tom.check_solution(g, Tn, T1, lm, s["final_cost"], manifold)
    m = tom.build_oracle(g, T_start, manifold)
    so = m.solve(500)
    To = m.get_param(2)
    print("manifold", manifold, "gpu", s["final_cost"], s["iterations"], "oracle", so["final_cost"], so["iterations"])
    assert abs(s["final_cost"] - so["final_cost"]) < 1e-9 * so["final_cost"]
    assert np.linalg.norm(T1[:3] - To[:3]) < 1e-6 and quat_close(T1[3:], To[3:]) < 1e-6
    # back to six degrees of freedom: the same handle goes on to the unrestricted minimum
    assert est.reset_parameterization(2, est.POSE6D)
    est.optimize(500)
    assert est.summary()["final_cost"] < s["final_cost"] * (1 - 1e-6)


def test_host_cost_functions_match_the_builtin_error_terms(gpu_lib):
    """Map::addResidualBlock with a cost function the library has no kernel for (Map.cpp:341-376; svin_ba_map_add_host_residual): the
    host evaluates it before every evaluation launch.  Pinned against the library's own error terms: a PoseError and a
    SpeedAndBiasError added (a) as the built-in factors and (b) as Python cost functions that return the same residuals and
    minimal Jacobians (PoseError through its CPU twin svin_host_pose_error, SpeedAndBiasError from its definition,
    SpeedAndBiasError.cpp:83-113) must give the same solve -- plus a cost function no kernel exists for (the distance between two
    frames held to a value), checked by what it is supposed to achieve."""
    from svin_amd import estimator
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=6, L=300, n_obs=3000, seed=21)

    def build():
        est = Estimator(0)
        fids, lids = syn.feed(est, spec)
        sb = {}
        for bid in est.parameter_block_ids():
            d = est.describe_block(bid)
            if d is not None and d[1] == 2:
                sb[d[0]] = bid
        return est, fids, lids, sb

    rng = np.random.default_rng(3)
    a, fa, la, sba = build()
    b, fb, lb, sbb = build()
    assert fa == fb and sba == sbb
    pose_meas = a.get_T_WS(fa[3]) + np.r_[0.05, -0.03, 0.02, 0, 0, 0, 0]
    A6 = rng.standard_normal((6, 6))
    pose_info = A6 @ A6.T + 50.0 * np.eye(6)
    sb_meas = a.get_speed_and_bias(fa[2]) + 0.01 * rng.standard_normal(9)
    A9 = rng.standard_normal((9, 9))
    sb_info = A9 @ A9.T + 200.0 * np.eye(9)
    assert a.map_add_pose_error(fa[3], pose_meas, pose_info) != 0
    assert a.map_add_speed_and_bias_error(sba[fa[2]], sb_meas, sb_info) != 0
    W9 = np.linalg.cholesky(sb_info).T          # e = meas - x, r = L^T e (SpeedAndBiasError.cpp:60-67, :95-103)
    calls = {"pose": 0, "sb": 0}

    def pose_cost(ps):
        calls["pose"] += 1
        r, Jm, _ = estimator.host_pose_error(pose_meas, pose_info, ps[0])
        return r, [Jm]

    def sb_cost(ps):
        calls["sb"] += 1
        return W9 @ (sb_meas - ps[0]), [-W9]

    assert b.map_add_host_residual([fb[3]], [7], 6, pose_cost) != 0
    rid_sb = b.map_add_host_residual([sbb[fb[2]]], [9], 9, sb_cost)
    assert rid_sb != 0 and rid_sb in b.residuals_of(sbb[fb[2]])
    # refused: an unknown block, a landmark, a block twice, too many residuals
    assert b.map_add_host_residual([999999], [7], 6, pose_cost) == 0
    assert b.map_add_host_residual([lb[0]], [7], 3, pose_cost) == 0
    assert b.map_add_host_residual([fb[1], fb[1]], [7, 7], 3, pose_cost) == 0
    assert b.map_add_host_residual([fb[1]], [7], 16, pose_cost) == 0
    for e in (a, b):
        e.set_solver_options(1e-14, 1e-14, 1e-14)
        e.optimize(12)
    sa, sb_ = a.summary(), b.summary()
    print("built-in", sa["final_cost"], sa["iterations"], "host", sb_["final_cost"], sb_["iterations"], "callbacks", calls)
    assert calls["pose"] >= sb_["iterations"] + 1 and calls["sb"] == calls["pose"]
    assert sa["iterations"] == sb_["iterations"] and abs(sa["final_cost"] - sb_["final_cost"]) < 1e-10 * sa["final_cost"]
    for f1, f2 in zip(fa, fb):
        assert np.max(np.abs(a.get_T_WS(f1) - b.get_T_WS(f2))) < 1e-9
        assert np.max(np.abs(a.get_speed_and_bias(f1) - b.get_speed_and_bias(f2))) < 1e-8
    # a cost function of the caller's own: the distance between frames 1 and 4 held to 1.5 m (weight 1e3)
    c, fc, _, _ = build()
    target, w = 1.5, 1e3

    def distance_cost(ps):
        d = ps[1][:3] - ps[0][:3]
        n = np.linalg.norm(d)
        J0, J1 = np.zeros((1, 6)), np.zeros((1, 6))
        J0[0, :3], J1[0, :3] = -w * d / n, w * d / n
        return [w * (n - target)], [J0, J1]

    before = np.linalg.norm(c.get_T_WS(fc[4])[:3] - c.get_T_WS(fc[1])[:3])
    assert c.map_add_host_residual([fc[1], fc[4]], [7, 7], 1, distance_cost) != 0
    c.optimize(15)
    after = np.linalg.norm(c.get_T_WS(fc[4])[:3] - c.get_T_WS(fc[1])[:3])
    print("distance between frames 1 and 4:", before, "->", after)
    assert abs(before - target) > 0.05 and abs(after - target) < 5e-3
    # such a window does not marginalise (the linearisation of a host residual is not available to the M1 kernels): a clean error
    with pytest.raises(RuntimeError):
        c.apply_marginalization(2, 2)


def test_reduced_manifold_on_an_extrinsics_block_and_host_residual_with_a_constant_block(gpu_lib):
    """(a) Pose3d on a per-frame EXTRINSICS block of a stereo_rig_v2 window (variable extrinsics: the dense Schur form with extrinsics
    rows, the border / chain solvers): its position must stay bit for bit while its orientation and everything else move, and a second
    handle without the manifold must end elsewhere; (b) a host residual over a CONSTANT block and a variable one: the constant
    block's Jacobian is ignored (off = -1 in the record), the variable block still feels the term."""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=5, L=300, n_obs=3000, seed=17, rig="rig_v2")

    def build():
        est = Estimator(0)
        fids, lids = syn.feed(est, spec)
        ext = {}
        for bid in est.parameter_block_ids():
            d = est.describe_block(bid)
            if d is not None and d[1] == 1:
                ext[(d[0], d[2])] = bid
        return est, fids, ext

    a, fa, exa = build()
    b, fb, exb = build()
    key = (fa[2], 0)
    bid = exa[key]
    assert not a.is_parameter_block_constant(bid), "the rig must have variable extrinsics for this test"
    T0 = a.get_parameter_block(bid).copy()
    assert a.reset_parameterization(bid, a.POSE3D)
    a.optimize(8)
    b.optimize(8)
    Ta, Tb = a.get_parameter_block(bid), b.get_parameter_block(exb[key])
    print("extrinsics", T0, "->", Ta, "(6-DoF:", Tb, ")")
    assert np.array_equal(Ta[:3], T0[:3]) and not np.array_equal(Ta[3:], T0[3:])
    assert not np.array_equal(Tb[:3], T0[:3])
    assert a.summary()["final_cost"] < a.summary()["initial_cost"]
    # (b)
    c, fc, _ = build()
    assert c.set_parameter_block_constant(fc[0])
    w, target = 1e3, 0.9
    seen = []

    def distance_cost(ps):
        d = ps[1][:3] - ps[0][:3]
        n = np.linalg.norm(d)
        J0, J1 = np.zeros((1, 6)), np.zeros((1, 6))
        J0[0, :3], J1[0, :3] = 1e6 * np.ones(3), w * d / n     # (the constant block's Jacobian: garbage on purpose)
        seen.append(ps[0].copy())
        return [w * (n - target)], [J0, J1]

    P0 = c.get_T_WS(fc[0]).copy()
    assert c.map_add_host_residual([fc[0], fc[2]], [7, 7], 1, distance_cost) != 0
    c.optimize(15)
    assert np.array_equal(c.get_T_WS(fc[0]), P0) and all(np.array_equal(s, seen[0]) for s in seen)
    after = np.linalg.norm(c.get_T_WS(fc[2])[:3] - P0[:3])
    print("distance to the constant frame:", after)
    assert abs(after - target) < 5e-3
