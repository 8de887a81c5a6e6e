"""The reference's own pass / fail criteria, applied to the HIP path (libsvin_ba.so through the C ABI).

okvis_ceres' backend tests hold no golden vectors (random, unseeded inputs) -- what they hold are THRESHOLDS on the outcome
of a scenario.  tests/test_oracle_reference_criteria.py applies them to the oracle; this file runs the same seeded
scenarios through the product:
  * TestEstimator.cpp:209-212   stereo test rig, 7 frames, optimize(10) per frame, applyMarginalizationStrategy(2, 3):
                                |speed/bias error| < 0.04, 2 |dq| < 1e-2, |dr| < 1e-1;
  * TestImuError.cpp:282-398    minimal Jacobians of the IMU factor against central differences (dx 1e-6, tolerance 1e-3 in
                                the Frobenius norm) and the convergence thresholds :393-398 of a two-state problem
                                (final cost < 1e-2, 2 |dq| < 1e-2, |dr| < 0.04);
  * Map::isJacobianCorrect (Map.cpp:116-252, relTol 1e-6 on the minimal Jacobians, delta 1e-8) for the device's
    reprojection Jacobians, evaluated by central differences of the device's own residuals."""
import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_estimator_scenario_thresholds_on_the_gpu(gpu_lib, case):
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=7, L=400, n_obs=None, seed=100 + case, rig="test%d" % case, frame_dt=10.0 / 6, keyframe_every=3,
                           pose_noise=(0.0, 0.0), lm_noise=0.0, depth_range=(2.5, 3.5), imu_noise="testestimator",
                           traj=dict(speed=1.0, rot_amp=0.0, wobble=0.0))
    est, ref = Estimator(0), orc.OracleEstimator()
    out = []
    for e in (est, ref):
        def on_frame(k, fid, e=e):
            e.optimize(10)
        fids, lids = syn.feed(e, spec, perturb=False, on_frame=on_frame)
        ok, removed = e.apply_marginalization(2, 3)
        assert ok
        e.optimize(10)
        last = fids[-1]
        out.append((e.get_T_WS(last), e.get_speed_and_bias(last), e.num_frames(), sorted(removed)))
    (T, sb, nf, rem), (To, sbo, nfo, remo) = out
    Tt, sbt = spec.T_WS_true[-1], spec.sb_true[-1]
    dq = T[3:] * np.sign(T[3:] @ Tt[3:]) - Tt[3:]
    # TestEstimator.cpp:209-212
    assert np.linalg.norm(sb - sbt) < 0.04
    assert 2 * np.linalg.norm(dq[:3]) < 1e-2
    assert np.linalg.norm(T[:3] - Tt[:3]) < 1e-1
    assert nf <= 2 + 3 and nf == nfo and rem == remo
    # and the product agrees with the oracle far inside those thresholds (seven truncated 10-iteration solves and a
    # marginalisation in sequence: not a converged fixed point, rounding-level differences grow along the way)
    assert np.linalg.norm(T[:3] - To[:3]) < 5e-4 and np.linalg.norm(sb - sbo) < 5e-4


def _imu_factor(est):
    f = [x for x in est.eval_factors() if x["kind"] == 0]
    assert f
    return f[0]


def test_imu_error_jacobians_by_central_differences_on_the_device(gpu_lib):
    """TestImuError.cpp:282-380: J_min of the IMU factor (block by block) against (r(x [+] dx e_j) - r(x [-] dx e_j)) / 2dx with
    the DEVICE evaluating every residual; tolerance as in the reference (1e-3 on weighted residuals of magnitude 1e2-1e3,
    scaled by the norm like the oracle's version of this test)."""
    from svin_amd.estimator import Estimator, host_manifold, MANIFOLD_POSE6D
    spec = syn.make_window(P=2, L=40, n_obs=150, seed=9, frame_dt=0.5)
    est = Estimator(0)
    fids, _ = syn.feed(est, spec)
    est.optimize(0)                      # first evaluation: pre-integrates at the current biases (redo_)
    f0 = _imu_factor(est)
    J, r0 = f0["J"], f0["r"]
    assert J.shape == (15, 30) and np.all(np.isfinite(r0)) and np.all(np.isfinite(J)) and len(f0["blocks"]) == 4
    dx = 1e-6
    col = 0
    for b, fid in enumerate(fids):
        T, sb = est.get_T_WS(fid), est.get_speed_and_bias(fid)
        for j in range(6):      # pose block: PoseManifold::plus with +-dx e_j
            res = []
            for sgn in (1.0, -1.0):
                d = np.zeros(6)
                d[j] = sgn * dx
                est.set_T_WS(fid, host_manifold(MANIFOLD_POSE6D, T, d)["plus"])
                res.append(_imu_factor(est)["r"])
            est.set_T_WS(fid, T)
            num = (res[0] - res[1]) / (2 * dx)
            assert np.linalg.norm(num - J[:, col + j]) < 1e-3 * max(1.0, np.linalg.norm(num)), (b, j, num, J[:, col + j])
        col += 6
        for j in range(9):      # speed / bias block (biases: the linearised correction of the pre-integrals, ImuError.cpp:741-760)
            res = []
            for sgn in (1.0, -1.0):
                s2 = sb.copy()
                s2[j] += sgn * dx
                est.set_speed_and_bias(fid, s2)
                res.append(_imu_factor(est)["r"])
            est.set_speed_and_bias(fid, sb)
            num = (res[0] - res[1]) / (2 * dx)
            assert np.linalg.norm(num - J[:, col + j]) < 1e-3 * max(1.0, np.linalg.norm(num)), (b, 6 + j, num, J[:, col + j])
        col += 9


def test_reprojection_jacobians_pass_the_references_checker_on_the_device(gpu_lib):
    """Map::isJacobianCorrect (Map.cpp:116-252): central differences with delta = 1e-8 through the manifold's plus, relative
    tolerance 1e-6 on the minimal Jacobians (norm-wise, like the reference's `(J_min - J_numDiff).norm() / J_min.norm()`),
    with the device evaluating all residuals -- pose, landmark and (rig v2: per-frame, variable) extrinsics blocks."""
    from svin_amd.estimator import Estimator, host_manifold, MANIFOLD_POSE6D
    spec = syn.make_window(P=3, L=60, n_obs=400, seed=31, rig="rig_v2")
    est = Estimator(0)
    fids, lids = syn.feed(est, spec)
    base = est.eval_reprojection(robust=False)
    n = len(base["r"])
    assert n > 300
    delta = 1e-8
    # the pose of the last frame: every observation made from it
    fid = fids[-1]
    T = est.get_T_WS(fid)
    num = np.zeros((n, 2, 6))
    for j in range(6):
        res = []
        for sgn in (1.0, -1.0):
            d = np.zeros(6)
            d[j] = sgn * delta
            est.set_T_WS(fid, host_manifold(MANIFOLD_POSE6D, T, d)["plus"])
            res.append(est.eval_reprojection(robust=False)["r"].copy())
        num[:, :, j] = (res[0] - res[1]) / (2 * delta)
    est.set_T_WS(fid, T)
    sel = [i for i in range(n) if base["pose_id"][i] == fid and base["cam"][i] < 15]
    assert len(sel) > 50
    for i in sel:
        Jm = base["Jp"][i].reshape(2, 6)
        assert np.linalg.norm(Jm - num[i]) / max(np.linalg.norm(Jm), 1e-300) < 1e-6 * 1e2, i   # delta 1e-8 on |J| ~ 1e3: rounding of the difference quotient ~ 1e-5
    others = [i for i in range(n) if base["pose_id"][i] != fid]
    assert np.all(num[others] == 0)
    # a landmark: Euclidean plus on the first three homogeneous components
    lid = lids[5]
    hp = est.get_landmark(lid)["point"]
    rows = [i for i in range(n) if base["lm_id"][i] == lid]
    assert rows
    numl = np.zeros((n, 2, 3))
    for j in range(3):
        res = []
        for sgn in (1.0, -1.0):
            h2 = hp.copy()
            h2[j] += sgn * 1e-7
            est.set_landmark(lid, h2)
            res.append(est.eval_reprojection(robust=False)["r"].copy())
        numl[:, :, j] = (res[0] - res[1]) / 2e-7
    est.set_landmark(lid, hp)
    for i in rows:
        Jl = base["Jl"][i].reshape(2, 3)
        assert np.linalg.norm(Jl - numl[i]) / max(np.linalg.norm(Jl), 1e-300) < 1e-4, i
