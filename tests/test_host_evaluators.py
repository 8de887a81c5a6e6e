"""the host-side twins (svin_host_imu_propagation, svin_host_reprojection_error) against the oracle and
the golden fixtures -- they run on the CPU, so these tests need no GPU."""
import os

import numpy as np

from oracle import orc
from svin_amd import estimator
from svin_amd import synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def test_host_imu_propagation_matches_oracle():
    for rig, dt_frames in (("euroc", 1), ("rig_v2", 2), ("test0", 1)):
        spec = syn.make_window(P=4, L=20, n_obs=100, seed=2, rig=rig)
        T0, sb0 = spec.T_WS_true[0].copy(), spec.sb_true[0].copy()
        sb0[3:] = [0.01, -0.02, 0.005, 0.05, -0.03, 0.02]
        t0, t1 = tuple(int(v) for v in spec.stamps[0]), tuple(int(v) for v in spec.stamps[dt_frames])
        n, T, sb, cov, jac, integ = estimator.host_imu_propagation(spec.imu_t, spec.imu_meas, spec.imu_params, T0, sb0, t0, t1, True, True)
        L = orc.lib()
        Tc, sbc, covc, jacc = T0.copy(), sb0.copy(), np.zeros((15, 15)), np.zeros((15, 15))
        it, im, par = orc.arr(spec.imu_t, np.uint32), orc.arr(spec.imu_meas), orc.imu_params_vector(spec.imu_params)
        nc = L.orc_imu_propagation(len(it), orc.u32ptr(it), orc.dptr(im), orc.dptr(par), orc.dptr(Tc), orc.dptr(sbc), t0[0], t0[1], t1[0],
                                   t1[1], orc.dptr(covc), orc.dptr(jacc))
        assert n == nc and n > 10
        assert np.max(np.abs(T - Tc)) < 1e-13 and np.max(np.abs(sb - sbc)) < 1e-13
        assert rel(cov, covc) < 1e-12 and rel(jac, jacc) < 1e-12
        # the integrals reproduce the prediction (second overload, ImuError.cpp:664-667)
        C0 = syn.quat_to_R(T0[3:] / np.linalg.norm(T0[3:]))
        gW = np.array([0.0, 0.0, spec.imu_params["g"]])
        dt = integ[6]
        assert np.max(np.abs(T0[:3] + sb0[:3] * dt + C0 @ integ[:3] - 0.5 * gW * dt * dt - T[:3])) < 1e-13
        assert np.max(np.abs(sb0[:3] + C0 @ integ[3:6] - gW * dt - sb[:3])) < 1e-13
    # a deque that ends before t_end: -1 like the reference (ImuError.cpp:279), states untouched
    n, T, sb, _, _, _ = estimator.host_imu_propagation(spec.imu_t[:20], spec.imu_meas[:20], spec.imu_params, T0, sb0, t0, t1)
    assert n == -1 and np.array_equal(T, T0) and np.array_equal(sb, sb0)


def test_host_reprojection_error_matches_golden_and_oracle():
    g = np.load(os.path.join(GOLD, "error_terms.npz"))
    nd = {0: 0, 1: 4, 2: 4, 3: 8}
    for i in range(len(g["reproj_model"])):
        model = int(g["reproj_model"][i])
        size = float(g["reproj_size"][i])
        info = np.eye(2) * 64.0 / (size * size)
        o = estimator.host_reprojection_error(model, g["reproj_intr"], g["reproj_dist"][i][:nd[model]], g["reproj_T_WS"][i], g["reproj_hp"][i],
                                              g["reproj_T_SC"][i], g["reproj_uv"][i], info)
        for key, ref in (("r", g["reproj_r"][i]), ("Jp", g["reproj_Jp"][i]), ("Jl", g["reproj_Jl"][i]), ("Je", g["reproj_Je"][i])):
            assert np.max(np.abs(o[key] - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref))), (i, key)
    # a general 2x2 information matrix and the ambient Jacobians against the oracle's ReprojectionError
    rng = np.random.default_rng(3)
    m = orc.OracleMap()
    for i in range(8):
        A = rng.normal(size=(2, 2))
        info = A @ A.T + 0.5 * np.eye(2)
        k = i % len(g["reproj_model"])
        model = int(g["reproj_model"][k])
        m.add_param(10 * i + 1, orc.BLOCK_POSE, g["reproj_T_WS"][k])
        m.add_param(10 * i + 2, orc.BLOCK_HPOINT, g["reproj_hp"][k])
        m.add_param(10 * i + 3, orc.BLOCK_POSE, g["reproj_T_SC"][k])
        rid = m.add_reproj(model, g["reproj_intr"], g["reproj_dist"][k], g["reproj_uv"][k], info, orc.LOSS_NONE, 10 * i + 1, 10 * i + 2, 10 * i + 3)
        r, Js, Jm = m.eval(rid)
        o = estimator.host_reprojection_error(model, g["reproj_intr"], g["reproj_dist"][k][:nd[model]], g["reproj_T_WS"][k], g["reproj_hp"][k],
                                              g["reproj_T_SC"][k], g["reproj_uv"][k], info)
        for a, b in ((o["r"], r), (o["Jp"], Jm[0]), (o["Jl"], Jm[1]), (o["Je"], Jm[2]), (o["J_pose"], Js[0]), (o["J_lm"], Js[1]), (o["J_ext"], Js[2])):
            assert np.max(np.abs(a - b)) <= 1e-10 * max(1.0, np.max(np.abs(b)))


def test_host_homogeneous_point_error_matches_oracle():
    """U6: svin_host_homogeneous_point_error against the oracle's HomogeneousPointError (variance form) and against the
    definition r = L^T (lm - meas), information = L L^T, for a full information matrix"""
    rng = np.random.default_rng(11)
    L = orc.lib()
    for k in range(8):
        hp = np.r_[rng.normal(size=3) * 3, 1.0]
        meas = np.r_[hp[:3] + rng.normal(size=3) * 0.1, 1.0]
        var = float(rng.uniform(0.01, 2.0))
        r, Jm, J = estimator.host_homogeneous_point_error(hp, meas, np.eye(3) / var)
        m = orc.OracleMap()
        m.add_param(1, orc.BLOCK_HPOINT, hp)
        rid = L.orc_map_add_hpoint_error(m.h, orc.dptr(orc.arr(meas)), var, 1)
        ro, Js, Jmo = m.eval(rid)
        assert np.max(np.abs(r - ro)) < 1e-15 and np.max(np.abs(Jm - Jmo[0])) < 1e-15
        assert np.max(np.abs(J - Js[0].reshape(3, 4))) < 1e-15
        A = rng.normal(size=(3, 3))
        info = A @ A.T + np.eye(3)
        r, Jm, J = estimator.host_homogeneous_point_error(hp, meas, info)
        Lc = np.linalg.cholesky(info)
        assert np.max(np.abs(r - Lc.T @ (hp[:3] - meas[:3]))) < 1e-13 and np.max(np.abs(Jm - Lc.T)) < 1e-13
        assert np.max(np.abs(J[:, :3] - Lc.T)) < 1e-13 and np.all(J[:, 3] == 0)
