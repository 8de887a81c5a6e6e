"""The HIP path against the committed golden fixtures DIRECTLY (tests/golden/*.npz: independent mpmath derivations,
tests/golden/make_golden*.py) -- no oracle in between.  Every case is staged through the C ABI the way the reference
would build it (addStates / addLandmark / addObservation / setters) and read back with the inspection hooks."""
import os

import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

IMU = dict(a_max=176.0, g_max=7.8, sigma_g_c=12.0e-4, sigma_a_c=8.0e-3, sigma_bg=0.03, sigma_ba=0.1, sigma_gw_c=4.0e-6,
           sigma_aw_c=4.0e-5, tau=3600.0, g=9.81, a0=[0.0, 0.0, 0.0])


def level_imu(t_end, rate=200, t0_sec=50):
    """a resting, level sensor: gyro 0, accelerometer (0, 0, g) -> initPoseFromImu gives the identity"""
    n = int(round(t_end * rate)) + 6
    ns = ((np.arange(n) - 2) * (1_000_000_000 // rate)).astype(np.int64) + 1_000_000_000
    t = np.stack([t0_sec + ns // 1_000_000_000, ns % 1_000_000_000], 1).astype(np.uint32)
    m = np.zeros((n, 6))
    m[:, 5] = IMU["g"]
    return t, m


def stamp(t, t0_sec=50):
    ns = int(round(t * 1e9)) + 1_000_000_000
    return (t0_sec + ns // 1_000_000_000, ns % 1_000_000_000)


def test_reprojection_matches_mpmath_fixture(gpu_lib):
    """R1-R3: residual and the three minimal Jacobians of 24 cases (4 distortion models, w != 1 landmarks)"""
    from svin_amd.estimator import Estimator
    g = np.load(os.path.join(GOLD, "error_terms.npz"))
    t, m = level_imu(0.0)
    worst = dict(r=0.0, Jp=0.0, Jl=0.0, Je=0.0)
    for i in range(len(g["reproj_model"])):
        model = int(g["reproj_model"][i])
        nd = {0: 0, 1: 4, 2: 4, 3: 8}[model]
        est = Estimator(0)
        for c in range(1):
            est.add_camera(model, g["reproj_intr"], g["reproj_dist"][i][:nd], 752, 480, [0.0, 0.0, 0.0, 0.0])
        est.add_imu(IMU)
        fid, lid = est.new_id(), est.new_id()
        assert est.add_states(fid, stamp(0.0), 400, g["reproj_T_SC"][i][None], t, m, True)
        assert est.set_T_WS(fid, g["reproj_T_WS"][i])
        assert est.set_camera_sensor_states(fid, 0, g["reproj_T_SC"][i])
        assert est.add_landmark(lid, g["reproj_hp"][i])
        assert est.add_observation(lid, fid, 0, 0, g["reproj_uv"][i], float(g["reproj_size"][i])) != 0
        ev = est.eval_reprojection(robust=False)
        assert len(ev["r"]) == 1
        for key, ref in (("r", g["reproj_r"][i]), ("Jp", g["reproj_Jp"][i]), ("Jl", g["reproj_Jl"][i]), ("Je", g["reproj_Je"][i])):
            worst[key] = max(worst[key], float(np.max(np.abs(ev[key][0] - ref)) / max(1.0, np.max(np.abs(ref)))))
    print("gpu vs mpmath reprojection", worst)
    assert max(worst.values()) < 1e-10, worst


def test_relative_pose_error_matches_mpmath_fixture(gpu_lib):
    """U3: two frames with per-frame extrinsics; the relative-extrinsics factor between them at the fixture's states"""
    from svin_amd.estimator import Estimator
    g = np.load(os.path.join(GOLD, "error_terms.npz"))
    dt = 0.5
    t, m = level_imu(dt)
    for i in range(len(g["relpose_T0"])):
        tv, rv = float(g["relpose_tv"][i]), float(g["relpose_rv"][i])
        est = Estimator(0)
        # variance of the factor = sigma_c^2 * dt (Estimator.cpp:391-398)
        est.add_camera(syn.DIST_NONE, [400.0, 400.0, 300.0, 200.0], [], 752, 480, [0.0, 0.0, np.sqrt(tv / dt), np.sqrt(rv / dt)])
        est.add_imu(IMU)
        f0, f1 = est.new_id(), 0
        assert est.add_states(f0, stamp(0.0), 400, g["relpose_T0"][i][None], t, m, True)
        f1 = est.new_id()
        assert est.add_states(f1, stamp(dt), 400, g["relpose_T1"][i][None], t, m, True)
        assert est.set_camera_sensor_states(f0, 0, g["relpose_T0"][i]) and est.set_camera_sensor_states(f1, 0, g["relpose_T1"][i])
        facs = [f for f in est.eval_factors() if f["kind"] == 3]
        assert len(facs) == 1
        f = facs[0]
        # the variance is rebuilt from sigma = sqrt(tv / dt): one rounding each way
        sc = max(1.0, np.max(np.abs(g["relpose_J0"][i])))
        assert np.max(np.abs(f["r"] - g["relpose_r"][i])) < 1e-10 * max(1.0, np.max(np.abs(g["relpose_r"][i])))
        assert np.max(np.abs(f["J"][:, :6] - g["relpose_J0"][i])) < 1e-10 * sc
        assert np.max(np.abs(f["J"][:, 6:] - g["relpose_J1"][i])) < 1e-10 * sc


def test_sonar_and_depth_match_mpmath_fixture(gpu_lib):
    """U4 / U5 through Estimator::addStates: the patch is gathered at the pose the frame is added with (landmarks inside
    the +-0.1 m box, the three just outside are not), the residual is then evaluated at the fixture's pose.  The sonar
    Jacobian is the reference's (SonarError.cpp:153-161), which is not the derivative of the residual."""
    from svin_amd.estimator import Estimator
    g = np.load(os.path.join(GOLD, "sonar_depth.npz"))
    t, m = level_imu(0.0)
    T_SC = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    for i in range(len(g["range"])):
        est = Estimator(0)
        est.add_camera(syn.DIST_NONE, [400.0, 400.0, 300.0, 200.0], [], 752, 480, [0.0, 0.0, 0.0, 0.0])
        est.add_imu(IMU)
        est.set_sonar_extrinsics(g["T_SSo"])
        k = int(g["npatch"][i])
        for p in np.vstack([g["patch"][i][:k], g["outside"][i]]):
            assert est.add_landmark(est.new_id(), np.r_[p, 1.0])
        # a landmark with w = 2: compared through its Euclidean image like the reference does (Estimator.cpp:292-294)
        fid = est.new_id()
        assert est.add_states(fid, stamp(0.0), 400, T_SC, t, m, True, sonar=[(float(g["range"][i]), float(g["heading"][i]))],
                              depth=[float(g["depth"][i])], first_depth=float(g["first_depth"][i]))
        assert np.max(np.abs(est.get_T_WS(fid) - g["T_add"][i])) < 1e-15
        assert est.set_T_WS(fid, g["T_eval"][i])
        facs = est.eval_factors()
        son = [f for f in facs if f["kind"] == 4]
        dep = [f for f in facs if f["kind"] == 5]
        assert len(son) == 1 and len(dep) == 1
        assert abs(son[0]["r"][0] - g["r"][i]) < 1e-12, (son[0]["r"], g["r"][i])
        assert np.max(np.abs(son[0]["J"][0] - g["J_ref"][i])) < 1e-12
        assert abs(dep[0]["r"][0] - g["depth_r"][i]) < 1e-12
        assert np.max(np.abs(dep[0]["J"][0] - g["depth_J"][i])) < 1e-14
    # no landmark inside the box -> no sonar factor (Estimator.cpp:305)
    est = Estimator(0)
    est.add_camera(syn.DIST_NONE, [400.0, 400.0, 300.0, 200.0], [], 752, 480, [0.0, 0.0, 0.0, 0.0])
    est.add_imu(IMU)
    est.set_sonar_extrinsics(g["T_SSo"])
    for p in g["outside"][0]:
        est.add_landmark(est.new_id(), np.r_[p, 1.0])
    fid = est.new_id()
    assert est.add_states(fid, stamp(0.0), 400, T_SC, t, m, True, sonar=[(float(g["range"][0]), float(g["heading"][0]))])
    assert [f for f in est.eval_factors() if f["kind"] == 4] == []


def test_imu_propagation_and_factor_match_mpmath_fixture(gpu_lib):
    """I1-I3 on the device against tests/golden/imu.npz (50-digit mpmath, make_golden_imu.py): k_imu_propagation's
    prediction / covariance / integrals, and the IMU factor's chi^2 = e^T P^-1 e in a two-frame window whose states are
    set to the fixture's (frame stamps = the fixture's t0 / t1, so first and last sample are interpolated; case 2 has a
    saturated gyroscope sample)"""
    from svin_amd.estimator import Estimator
    g = np.load(os.path.join(GOLD, "imu.npz"))
    names = ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c", "tau", "g")
    par = dict(zip(names, [float(v) for v in g["params"]]))
    par["a0"] = [0.0, 0.0, 0.0]
    worst = dict(p=0.0, q=0.0, v=0.0, cov=0.0, integ=0.0, chi2=0.0, e=0.0)
    for i in range(len(g["imu_n"])):
        n = int(g["imu_n"][i])
        it, im = g["imu_t"][i][:n], g["imu_m"][i][:n]
        t0, t1 = tuple(int(v) for v in g["t0"][i]), tuple(int(v) for v in g["t1"][i])
        est = Estimator(0)
        est.add_camera(1, [450.0, 450.0, 376.0, 240.0], [0.0, 0.0, 0.0, 0.0], 752, 480, [0.0, 0.0, 0.0, 0.0])
        est.add_imu(par)
        used, T, sb, cov, _, integ = est.imu_propagation(it, im, par, g["T0"][i], g["sb0"][i], t0, t1, True, True, want_integrals=True)
        assert used == int(g["used"][i])
        worst["p"] = max(worst["p"], float(np.max(np.abs(T[:3] - g["T_pred"][i][:3]))))
        worst["q"] = max(worst["q"], 2.0 * min(np.linalg.norm(T[3:] - g["T_pred"][i][3:]), np.linalg.norm(T[3:] + g["T_pred"][i][3:])))
        worst["v"] = max(worst["v"], float(np.max(np.abs(sb[:3] - g["v_pred"][i]))))
        worst["cov"] = max(worst["cov"], float(np.max(np.abs(cov - g["cov"][i])) / np.max(np.abs(g["cov"][i]))))
        worst["integ"] = max(worst["integ"], float(np.max(np.abs(integ - g["integrals"][i]))))
        # the factor, through the window the way the pipeline builds it
        T_SC = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
        f0, f1 = est.new_id(), est.new_id()
        assert est.add_states(f0, t0, 400, T_SC, it, im, True)
        assert est.set_T_WS(f0, g["T0"][i]) and est.set_speed_and_bias(f0, g["sb0"][i])
        assert est.add_states(f1, t1, 400, T_SC, it, im, False)
        assert est.set_T_WS(f1, g["T1"][i]) and est.set_speed_and_bias(f1, g["sb1"][i])
        imu = [f for f in est.eval_factors() if f["kind"] == 0]
        assert len(imu) == 1 and imu[0]["m"] == 15
        r = imu[0]["r"]
        worst["chi2"] = max(worst["chi2"], abs(float(r @ r) - float(g["chi2"][i])) / float(g["chi2"][i]))
        Lc = np.linalg.cholesky(np.linalg.inv(g["P_delta"][i]))
        worst["e"] = max(worst["e"], float(np.max(np.abs(np.linalg.solve(Lc.T, r) - g["e"][i])) / np.max(np.abs(g["e"][i]))))
        est.close() if hasattr(est, "close") else None
    print("gpu vs mpmath imu", worst)
    assert worst["p"] < 1e-13 and worst["q"] < 1e-13 and worst["v"] < 1e-13 and worst["integ"] < 1e-13, worst
    assert worst["cov"] < 1e-12 and worst["chi2"] < 1e-9 and worst["e"] < 1e-8, worst
