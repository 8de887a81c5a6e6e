"""Pins the oracle against the committed golden fixtures (tests/golden/*.npz, produced by the independent
mpmath / scipy derivation in tests/golden/make_golden.py -- no oracle or product code involved there)."""
import os

import numpy as np
import pytest

from oracle import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def terms():
    return np.load(os.path.join(GOLD, "error_terms.npz"))


def test_reprojection_residuals_and_minimal_jacobians(terms):
    g = terms
    m = orc.OracleMap()
    pid = 1
    worst = 0.0
    for i in range(len(g["reproj_model"])):
        model = int(g["reproj_model"][i])
        size = float(g["reproj_size"][i])
        info = 64.0 / (size * size)
        m.add_param(pid, orc.BLOCK_POSE, g["reproj_T_WS"][i])
        m.add_param(pid + 1, orc.BLOCK_HPOINT, g["reproj_hp"][i])
        m.add_param(pid + 2, orc.BLOCK_POSE, g["reproj_T_SC"][i])
        rid = m.add_reproj(model, g["reproj_intr"], g["reproj_dist"][i], g["reproj_uv"][i], [[info, 0], [0, info]],
                           orc.LOSS_NONE, pid, pid + 1, pid + 2)
        r, Js, Jm = m.eval(rid)
        for a, b in ((r, g["reproj_r"][i]), (Jm[0], g["reproj_Jp"][i]), (Jm[1], g["reproj_Jl"][i]), (Jm[2], g["reproj_Je"][i])):
            worst = max(worst, np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))
        # the ambient Jacobians are the minimal ones times the lift Jacobian (ReprojectionError.hpp:165-170)
        lift = np.zeros((6, 7))
        orc.lib().orc_manifold_lift_jacobian(orc.BLOCK_POSE, orc.dptr(orc.arr(g["reproj_T_WS"][i])), orc.dptr(lift))
        assert np.max(np.abs(Js[0] - Jm[0] @ lift)) < 1e-9
        pid += 3
    assert worst < 1e-10, worst


def test_pose_error(terms):
    g = terms
    m = orc.OracleMap()
    L = orc.lib()
    for i in range(len(g["pose_T"])):
        m.add_param(100 + i, orc.BLOCK_POSE, g["pose_T"][i])
        info = np.diag(g["pose_info"][i])
        rid = L.orc_map_add_pose_error(m.h, orc.dptr(orc.arr(g["pose_Tm"][i])), orc.dptr(orc.arr(info)), 100 + i)
        r, Js, Jm = m.eval(rid)
        assert np.max(np.abs(r - g["pose_r"][i])) < 1e-11
        assert np.max(np.abs(Jm[0] - g["pose_J"][i])) < 1e-10 * max(1.0, np.max(np.abs(g["pose_J"][i])))


def test_relative_pose_error(terms):
    g = terms
    m = orc.OracleMap()
    L = orc.lib()
    for i in range(len(g["relpose_T0"])):
        m.add_param(200 + 2 * i, orc.BLOCK_POSE, g["relpose_T0"][i])
        m.add_param(201 + 2 * i, orc.BLOCK_POSE, g["relpose_T1"][i])
        rid = L.orc_map_add_relpose_error(m.h, float(g["relpose_tv"][i]), float(g["relpose_rv"][i]), 200 + 2 * i, 201 + 2 * i)
        r, Js, Jm = m.eval(rid)
        sc = max(1.0, np.max(np.abs(g["relpose_J0"][i])))
        assert np.max(np.abs(r - g["relpose_r"][i])) < 1e-10 * max(1.0, np.max(np.abs(r)))
        assert np.max(np.abs(Jm[0] - g["relpose_J0"][i])) < 1e-10 * sc
        assert np.max(np.abs(Jm[1] - g["relpose_J1"][i])) < 1e-10 * sc


def test_tiny_window_fixed_point_matches_independent_minimiser():
    """scipy.optimize.least_squares (block Cauchy loss) and the oracle's Ceres-like dogleg solver must reach the
    same minimum of the same 2-pose / 12-landmark problem."""
    g = np.load(os.path.join(GOLD, "tiny_window.npz"))
    m = orc.OracleMap()
    L = orc.lib()
    size = float(g["size"])
    info = 64.0 / (size * size)
    m.add_param(1, orc.BLOCK_POSE, g["T0"])
    m.set_constant(1)
    m.add_param(2, orc.BLOCK_POSE, g["T1_init"])
    for c in range(2):
        m.add_param(3 + c, orc.BLOCK_POSE, g["T_SC"][c])
        m.set_constant(3 + c)
    nL = len(g["lm_init"])
    for l in range(nL):
        m.add_param(10 + l, orc.BLOCK_HPOINT, np.r_[g["lm_init"][l], 1.0])
        for f, pose in enumerate((1, 2)):
            for c in range(2):
                m.add_reproj(orc.DIST_RADTAN, g["intr"], g["dist"], g["uv"][f, c, l], [[info, 0], [0, info]], orc.LOSS_CAUCHY,
                             pose, 10 + l, 3 + c)
    L.orc_map_set_tolerances(m.h, 1e-16, 1e-16, 1e-16)
    s = m.solve(500)
    T1 = m.get_param(2)
    lm = np.stack([m.get_param(10 + l)[:3] for l in range(nL)])
    # same cost (the oracle reports sum 0.5*rho) and the same minimiser.  The quasi-Newton reference stops at a
    # gradient of ~1e-2 (finite-difference gradients), i.e. within ~3e-5 relative of the true minimum.
    print("oracle cost", s["final_cost"], "scipy cost", float(g["cost"]), "dT", np.linalg.norm(T1[:3] - g["T1_opt"][:3]),
          "dlm", np.max(np.abs(lm - g["lm_opt"])))
    assert s["final_cost"] <= float(g["cost"]) * (1 + 1e-9)
    assert abs(s["final_cost"] - float(g["cost"])) < 1e-4 * float(g["cost"])
    assert np.linalg.norm(T1[:3] - g["T1_opt"][:3]) < 2e-3
    assert min(np.linalg.norm(T1[3:] - g["T1_opt"][3:]), np.linalg.norm(T1[3:] + g["T1_opt"][3:])) < 1e-3
    assert np.max(np.abs(lm - g["lm_opt"])) < 3e-2


def test_sonar_and_depth_errors_match_independent_derivation():
    """SonarError (incl. the reference's Jacobian, which is not the derivative of its residual) and DepthError against
    tests/golden/sonar_depth.npz (mpmath restatement of the definitions, tests/golden/make_golden_r2.py)"""
    g = np.load(os.path.join(GOLD, "sonar_depth.npz"))
    m = orc.OracleMap()
    L = orc.lib()
    for i in range(len(g["range"])):
        k = int(g["npatch"][i])
        patch = orc.arr(g["patch"][i][:k])
        m.add_param(300 + i, orc.BLOCK_POSE, g["T_eval"][i])
        rid = L.orc_map_add_sonar_error(m.h, orc.dptr(orc.arr(g["T_SSo"])), float(g["range"][i]), float(g["heading"][i]), 1.0, k,
                                        orc.dptr(patch), 300 + i)
        r, Js, Jm = m.eval(rid)
        assert abs(r[0] - g["r"][i]) < 1e-13
        assert np.max(np.abs(Jm[0][0] - g["J_ref"][i])) < 1e-13
        assert np.max(np.abs(Js[0][0][:3] - g["J_ref"][i][:3])) < 1e-13 and np.all(Js[0][0][3:] == 0)
        rid = L.orc_map_add_depth_error(m.h, float(g["depth"][i]), 5.0, float(g["first_depth"][i]), 300 + i)
        r, Js, Jm = m.eval(rid)
        assert abs(r[0] - g["depth_r"][i]) < 1e-13
        assert np.max(np.abs(Jm[0][0] - g["depth_J"][i])) < 1e-15
    # the reference's sonar Jacobian has (nearly) the opposite sign of the residual's true derivative: recorded, not fixed
    assert np.max(np.abs(g["J_ref"] - g["J_true"])) > 1.0


def imu_case(g, i):
    n = int(g["imu_n"][i])
    par = dict(zip(("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c", "tau", "g"),
                   [float(v) for v in g["params"]]))
    par["a0"] = [0.0, 0.0, 0.0]
    return n, par, g["imu_t"][i][:n], g["imu_m"][i][:n], tuple(int(v) for v in g["t0"][i]), tuple(int(v) for v in g["t1"][i])


def quat_angle(a, b):
    return 2.0 * min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def test_imu_propagation_and_factor_match_independent_derivation():
    """I1-I3: the pre-integration loop (incl. interpolated first / last sample and a saturated gyroscope sample), the
    propagated covariance and the factor's chi^2 = e^T P^-1 e against tests/golden/imu.npz (50-digit mpmath restatement,
    tests/golden/make_golden_imu.py); the host twin of the product (svin_host_imu_propagation) against the same file"""
    from svin_amd import estimator
    g = np.load(os.path.join(GOLD, "imu.npz"))
    L = orc.lib()
    for i in range(len(g["imu_n"])):
        n, par, it, im, t0, t1 = imu_case(g, i)
        pv = orc.imu_params_vector(par)
        T, sb, cov, jac = orc.arr(g["T0"][i]).copy(), orc.arr(g["sb0"][i]).copy(), np.zeros((15, 15)), np.zeros((15, 15))
        used = L.orc_imu_propagation(n, orc.u32ptr(orc.arr(it, np.uint32)), orc.dptr(orc.arr(im)), orc.dptr(pv), orc.dptr(T), orc.dptr(sb),
                                     t0[0], t0[1], t1[0], t1[1], orc.dptr(cov), orc.dptr(jac))
        assert used == int(g["used"][i])
        assert np.max(np.abs(T[:3] - g["T_pred"][i][:3])) < 1e-13 and quat_angle(T[3:], g["T_pred"][i][3:]) < 1e-13
        assert np.max(np.abs(sb[:3] - g["v_pred"][i])) < 1e-13 and np.array_equal(sb[3:], g["sb0"][i][3:])
        assert np.max(np.abs(cov - g["cov"][i])) < 1e-12 * np.max(np.abs(g["cov"][i]))
        # the product's CPU twin
        nh, Th, sbh, covh, _, integ = estimator.host_imu_propagation(it, im, par, g["T0"][i], g["sb0"][i], t0, t1, True, True)
        assert nh == used
        assert np.max(np.abs(Th[:3] - g["T_pred"][i][:3])) < 1e-13 and quat_angle(Th[3:], g["T_pred"][i][3:]) < 1e-13
        assert np.max(np.abs(sbh[:3] - g["v_pred"][i])) < 1e-13
        assert np.max(np.abs(covh - g["cov"][i])) < 1e-12 * np.max(np.abs(g["cov"][i]))
        assert np.max(np.abs(integ - g["integrals"][i])) < 1e-13
        # the factor
        m = orc.OracleMap()
        m.add_param(1, orc.BLOCK_POSE, g["T0"][i])
        m.add_param(2, orc.BLOCK_SPEEDBIAS, g["sb0"][i])
        m.add_param(3, orc.BLOCK_POSE, g["T1"][i])
        m.add_param(4, orc.BLOCK_SPEEDBIAS, g["sb1"][i])
        rid = m.add_imu(it, im, pv, t0, t1, [1, 2, 3, 4])
        r, Js, Jm = m.eval(rid)
        assert abs(r @ r - g["chi2"][i]) < 1e-9 * g["chi2"][i]
        # weighted residual = S e with S^T S = P^-1: undo the weighting with the golden covariance's Cholesky factor
        Lc = np.linalg.cholesky(np.linalg.inv(g["P_delta"][i]))
        assert np.max(np.abs(np.linalg.solve(Lc.T, r) - g["e"][i])) < 1e-9 * np.max(np.abs(g["e"][i]))
