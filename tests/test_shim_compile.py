"""The header-only C++ shim (integration/okvis/Estimator.hpp, integration/okvis/ceres/Map.hpp) compiles and links
against libsvin_ba.so.  This image has neither Eigen nor the okvis_common / okvis_cv headers, so the check is made
against minimal stand-ins that carry the reference's names and signatures (tests/csrc/mock_okvis/README.md)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I", os.path.join(ROOT, "integration"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "csrc", "mock_okvis")]


def build_shim_smoke(out):
    from svin_amd import estimator
    estimator.load_library()   # built?
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror"] + INC + [os.path.join(ROOT, "tests", "csrc", "shim_smoke.cpp"), "-o", out,
           "-L", os.path.join(ROOT, "svin_amd"), "-lsvin_ba", "-Wl,-rpath," + os.path.join(ROOT, "svin_amd"), "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return out


def test_shim_headers_are_self_contained(tmp_path):
    """each header on its own, every warning an error; a second translation unit proves there are no ODR-breaking definitions"""
    for hdr in ("okvis/Estimator.hpp", "okvis/ceres/Map.hpp", "okvis/ceres/HomogeneousPointError.hpp", "okvis/ceres/CeresTypes.hpp",
                "okvis/ceres/ErrorInterface.hpp", "okvis/ceres/ParameterBlock.hpp", "okvis/ceres/ParameterBlockSized.hpp",
                "okvis/ceres/PoseParameterBlock.hpp", "okvis/ceres/SpeedAndBiasParameterBlock.hpp", "okvis/ceres/HomogeneousPointParameterBlock.hpp",
                "okvis/ceres/ManifoldAdditionalInterfaces.hpp", "okvis/ceres/PoseManifold.hpp", "okvis/ceres/HomogeneousPointManifold.hpp",
                "okvis/ceres/PoseError.hpp"):
        src = tmp_path / "tu.cpp"
        pre = ""
        src.write_text(pre + "#include <%s>\n#include <%s>\nint main() { return 0; }\n" % (hdr, hdr))
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror"] + INC + [str(src)])


def test_shim_smoke_program_builds_and_links(tmp_path):
    """instantiates the templates (addObservation<GEOMETRY>) and resolves every svin_ba_* symbol the shim calls"""
    exe = build_shim_smoke(str(tmp_path / "shim_smoke"))
    assert os.path.exists(exe)
    # without a GPU the estimator constructor throws okvis::Estimator::Exception (no CPU fallback): the program aborts
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe, os.devnull], capture_output=True, text=True)
        assert p.returncode != 0 and "no HIP device" in p.stderr


def test_cpu_shim_classes_match_the_c_abi(tmp_path):
    """okvis::ceres::ImuError::propagation (both overloads) and ReprojectionError<G>::EvaluateWithMinimalJacobians of
    the shim, driven from C++ like ThreadedKFVio.cpp:599 / ProbabilisticStereoTriangulator.cpp:266-300, return the
    numbers of svin_host_imu_propagation / svin_host_reprojection_error called directly (bit for bit).  No GPU needed."""
    import numpy as np
    from svin_amd import estimator
    from svin_amd import synthetic as syn
    exe = str(tmp_path / "shim_host_eval")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror"] + INC + [os.path.join(ROOT, "tests", "csrc", "shim_host_eval.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "svin_amd"), "-lsvin_ba", "-Wl,-rpath," + os.path.join(ROOT, "svin_amd"), "-Wl,--allow-shlib-undefined"])
    spec = syn.make_window(P=3, L=20, n_obs=100, seed=2)
    p = spec.imu_params
    T0, sb0 = spec.T_WS_true[0].copy(), spec.sb_true[0].copy()
    sb0[3:] = [0.01, -0.02, 0.005, 0.05, -0.03, 0.02]
    t0, t1 = tuple(int(v) for v in spec.stamps[0]), tuple(int(v) for v in spec.stamps[1])
    g = np.load(os.path.join(ROOT, "tests", "golden", "error_terms.npz"))
    k = 7   # a radial-tangential case with a non-unit homogeneous scale
    info = np.array([[3.0, 0.4], [0.4, 2.0]])
    intr = list(g["reproj_intr"]) + list(g["reproj_dist"][k][:4])
    lines = [" ".join(repr(float(p[x])) for x in ("a_max", "g_max", "sigma_g_c", "sigma_a_c", "sigma_bg", "sigma_ba", "sigma_gw_c", "sigma_aw_c", "tau", "g"))
             + " " + " ".join(repr(float(v)) for v in p["a0"]), str(len(spec.imu_t))]
    for i in range(len(spec.imu_t)):
        lines.append("%d %d %s" % (spec.imu_t[i, 0], spec.imu_t[i, 1], " ".join(repr(float(v)) for v in spec.imu_meas[i])))
    lines.append(" ".join(repr(float(v)) for v in T0) + " " + " ".join(repr(float(v)) for v in sb0))
    lines.append("%d %d %d %d" % (t0 + t1))
    lines.append("RadialTangentialDistortion 752 480 %d %s" % (len(intr), " ".join(repr(float(v)) for v in intr)))
    for arr in (g["reproj_T_WS"][k], g["reproj_hp"][k], g["reproj_T_SC"][k], g["reproj_uv"][k], info.reshape(-1)):
        lines.append(" ".join(repr(float(v)) for v in arr))
    path = tmp_path / "in.txt"
    path.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, check=True).stdout.splitlines()
    n, T, sb, cov, jac, integ = estimator.host_imu_propagation(spec.imu_t, spec.imu_meas, p, T0, sb0, t0, t1, True, True)
    t = out[0].replace("|", " ").split()
    assert int(t[1]) == n > 10
    got = [float(v) for v in t[2:16]]
    want = list(T) + [sb[0], sb[1], sb[2], integ[0], integ[4], integ[6], sb[4]]
    assert got == want, (got, want)
    assert float(t[17]) == cov[0, 0] and float(t[19]) == cov[3, 9] and float(t[21]) == jac[0, 9] and float(t[23]) == jac[9, 0]
    t = out[1].split()
    assert int(t[1]) == n and float(t[2]) == T[0] and float(t[3]) == sb[2]
    o = estimator.host_reprojection_error(1, g["reproj_intr"], g["reproj_dist"][k][:4], g["reproj_T_WS"][k], g["reproj_hp"][k], g["reproj_T_SC"][k],
                                          g["reproj_uv"][k], info)
    t = out[2].replace("|", " ").split()
    assert int(t[1]) == 1
    got = [float(v) for v in t[2:]]
    want = [o["r"][0], o["r"][1], o["Jp"][0, 0], o["Jp"][1, 5], o["Jl"][1, 2], o["J_pose"][0, 6], o["J_lm"][0, 3], o["J_ext"][1, 6]]
    assert got == want, (got, want)
    t = out[3].split()
    assert int(t[1]) == 1 and float(t[2]) == o["Jl"][0, 0]


def _compile(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror"] + INC + [os.path.join(ROOT, "tests", "csrc", name + ".cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "svin_amd"), "-lsvin_ba", "-Wl,-rpath," + os.path.join(ROOT, "svin_amd"), "-Wl,--allow-shlib-undefined"])
    return exe


def test_reference_shaped_map_program_builds(tmp_path):
    """tests/csrc/shim_map_tests.cpp (TestHomogeneousPointError / TestMap re-created on okvis::ceres::Map as a graph builder)
    compiles with every warning an error and resolves the svin_ba_map_* symbols; it runs in tests/test_gpu_shim.py"""
    assert os.path.exists(_compile(tmp_path, "shim_map_tests"))


def test_no_shim_header_includes_ceres():
    """the shim set is Ceres-free -- no header under integration/ includes a ceres/ header"""
    import re
    for d, _, files in os.walk(os.path.join(ROOT, "integration")):
        for f in files:
            text = open(os.path.join(d, f)).read()
            assert not re.search(r'^\s*#\s*include\s*[<"]ceres/', text, re.M), f


def test_frontend_calls_on_ceres_free_shim_match_oracle(tmp_path):
    """tests/csrc/shim_frontend_calls.cpp makes the calls of ProbabilisticStereoTriangulator.cpp:87-99 / :266-300 and
    VioKeyframeWindowMatchingAlgorithm.cpp:453 against integration/okvis/ceres/{PoseError, PoseParameterBlock,
    HomogeneousPointParameterBlock, ReprojectionError, PoseManifold*, HomogeneousPointManifold}.hpp; what it prints is
    held against the oracle (PoseError, the manifolds, ReprojectionError) to 1e-12 and, for the manifolds the oracle does
    not restate (3d / 4d / 2d), against the reference's own criterion (ManifoldAdditionalInterfaces::verify)."""
    import numpy as np
    from oracle import orc
    exe = _compile(tmp_path, "shim_frontend_calls")
    rng = np.random.default_rng(5)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    Tab = np.concatenate([[0.11, -0.02, 0.015], q])
    A = rng.normal(size=(6, 6))
    info = A @ A.T + np.diag([50, 60, 70, 800, 900, 1000.0])
    q2 = q + 0.05 * rng.normal(size=4); q2 /= np.linalg.norm(q2)
    Tx = np.concatenate([[0.13, 0.01, -0.02], q2])
    delta = np.array([0.01, -0.02, 0.03, 0.02, -0.01, 0.015])
    g = np.load(os.path.join(ROOT, "tests", "golden", "error_terms.npz"))
    intr = list(g["reproj_intr"]) + list(g["reproj_dist"][7][:4])
    hPA = np.array([0.4, -0.3, 5.0, 1.0])
    uv = np.array([380.0, 250.0])
    lines = [" ".join(repr(float(v)) for v in a) for a in (Tab, info.reshape(-1), Tx, delta)]
    lines.append("RadialTangentialDistortion 752 480 %d %s" % (len(intr), " ".join(repr(float(v)) for v in intr)))
    lines += [" ".join(repr(float(v)) for v in a) for a in (hPA, uv)]
    path = tmp_path / "in.txt"
    path.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, check=True).stdout.splitlines()

    def vec(i, tag):
        assert out[i].startswith(tag + " "), (out[i][:20], tag)
        return np.array([float(v) for v in out[i][len(tag):].split()])

    L = orc.lib()
    m = orc.OracleMap()
    BLOCK_POSE = 0
    m.add_param(1, BLOCK_POSE, Tab)
    m.add_param(2, BLOCK_POSE, Tx)
    rid1 = L.orc_map_add_pose_error(m.h, orc.dptr(orc.arr(Tab)), orc.dptr(orc.arr(info.reshape(-1))), 1)
    rid2 = L.orc_map_add_pose_error(m.h, orc.dptr(orc.arr(Tab)), orc.dptr(orc.arr(info.reshape(-1))), 2)
    rid3 = L.orc_map_add_pose_error_var(m.h, orc.dptr(orc.arr(Tab)), 0.04, 0.0009, 2)
    tol = dict(rtol=1e-12, atol=1e-12)
    # ProbabilisticStereoTriangulator.cpp:87-99: evaluated at its own measurement
    assert out[0] == "pose_at_measurement 1"
    r, J, Jm = m.eval(rid1)
    np.testing.assert_allclose(vec(1, "r"), r, **tol)
    np.testing.assert_allclose(vec(2, "Jmin").reshape(6, 6), Jm[0], **tol)
    np.testing.assert_allclose(vec(3, "J").reshape(6, 7), J[0], **tol)
    H = vec(2, "Jmin").reshape(6, 6)
    np.testing.assert_allclose(H.T @ H, info, rtol=1e-10)   # J_min^T J_min = the information at the measurement (what the triangulator wants)
    # away from the measurement
    assert out[4].startswith("pose_away 1 dim 6 blocks 1 bdim 7 type PoseError id 7 fixed 0 t 3 4")
    r, J, Jm = m.eval(rid2)
    np.testing.assert_allclose(vec(5, "r"), r, **tol)
    np.testing.assert_allclose(vec(6, "Jmin").reshape(6, 6), Jm[0], **tol)
    np.testing.assert_allclose(vec(7, "J").reshape(6, 7), J[0], **tol)
    np.testing.assert_allclose(vec(8, "cov").reshape(6, 6, order="F"), np.linalg.inv(info), rtol=1e-10)   # mock Eigen: column-major
    t = out[9].split()
    assert t[0] == "pose_nojac" and t[1] == "1" and abs(float(t[2]) - r[5]) < 1e-12
    np.testing.assert_allclose(vec(10, "r_var"), m.eval(rid3)[0], **tol)
    # parameter-block operations == the oracle's pose manifold
    xp, dm, Jp, Jl = np.zeros(7), np.zeros(6), np.zeros(42), np.zeros(42)
    L.orc_manifold_plus(BLOCK_POSE, orc.dptr(Tx), orc.dptr(delta), orc.dptr(xp))
    L.orc_manifold_minus(BLOCK_POSE, orc.dptr(xp), orc.dptr(Tx), orc.dptr(dm))
    L.orc_manifold_plus_jacobian(BLOCK_POSE, orc.dptr(Tx), orc.dptr(Jp))
    L.orc_manifold_lift_jacobian(BLOCK_POSE, orc.dptr(Tx), orc.dptr(Jl))
    np.testing.assert_allclose(vec(11, "pb_plus"), xp, **tol)
    np.testing.assert_allclose(vec(12, "pb_minus"), dm, **tol)
    np.testing.assert_allclose(vec(13, "pb_Jplus"), Jp, **tol)
    np.testing.assert_allclose(vec(14, "pb_Jlift"), Jl, **tol)
    t = out[15].split()
    assert t[0] == "pb_estimate" and float(t[1]) == Tx[0] and float(t[2]) == Tx[3] and float(t[3]) == Tx[6] and t[5:] == ["dim", "7", "min", "6", "type", "PoseParameterBlock"]
    # manifolds: 6 lines each
    i = 16
    Jmn = np.zeros(42)
    L.orc_pose_minus_jacobian(orc.dptr(Tx), orc.dptr(Jmn))
    keep = {"m6": [0, 1, 2, 3, 4, 5], "m3": [3, 4, 5], "m4": [0, 1, 2, 5], "m2": [3, 4]}
    for tag in ("m6", "m3", "m4", "m2", "mh"):
        head = out[i].split()
        assert head[0] == tag and head[-2:] == ["verify", "1"], out[i]   # the reference's own acceptance criterion
        na, nt = int(head[2]), int(head[3])
        plus, minus = vec(i + 1, tag + " plus"), vec(i + 2, tag + " minus")
        Jplus, Jlift, Jminus = (vec(i + 3, tag + " Jplus").reshape(na, nt), vec(i + 4, tag + " Jlift").reshape(nt, na),
                                vec(i + 5, tag + " Jminus").reshape(nt, na))
        np.testing.assert_allclose(Jlift @ Jplus, np.eye(nt), atol=1e-12)
        if tag == "mh":
            np.testing.assert_allclose(plus, [0.3 + delta[0], -1.2 + delta[1], 4.0 + delta[2], 1.0], **tol)
            np.testing.assert_allclose(minus, delta[:3], atol=1e-15)
        else:
            k = keep[tag]
            d6 = np.zeros(6); d6[k] = delta[:nt]
            L.orc_manifold_plus(BLOCK_POSE, orc.dptr(Tx), orc.dptr(d6), orc.dptr(xp))
            L.orc_manifold_minus(BLOCK_POSE, orc.dptr(xp), orc.dptr(Tx), orc.dptr(dm))
            np.testing.assert_allclose(plus, xp, **tol)
            np.testing.assert_allclose(minus, dm[k], **tol)
            np.testing.assert_allclose(Jlift, Jl.reshape(6, 7)[k], **tol)
            np.testing.assert_allclose(Jminus, Jmn.reshape(6, 7)[k], **tol)
            if tag in ("m6", "m4"):
                np.testing.assert_allclose(Jplus, Jp.reshape(7, 6)[:, k], **tol)
        i += 6
    assert out[i] == "numdiff 1"
    t = out[i + 1].replace("|", " ").split()
    assert [float(v) for v in t[1:5]] == [0.3, -1.2, 4.0, 1.0] and t[5:7] == ["init", "1"] and float(t[7]) == 3.0 and float(t[8]) == 1.0 and t[9:] == ["init", "0", "dim", "4", "min", "3", "HomogeneousPointParameterBlock"]
    t = out[i + 2].split()
    assert abs(float(t[1]) - 1.8) < 1e-15 and abs(float(t[2]) - 0.3) < 1e-15 and t[3] == "SpeedAndBiasParameterBlock"
    # ProbabilisticStereoTriangulator.cpp:292-300: ReprojectionError fed from the parameter blocks (pose B = T_AB, identity extrinsics)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    m.add_param(10, BLOCK_POSE, Tab)
    m.add_param(11, 2, hPA)        # BLOCK_HPOINT
    m.add_param(12, BLOCK_POSE, ident)
    rid = m.add_reproj(1, g["reproj_intr"], g["reproj_dist"][7][:4], uv, np.eye(2) / (0.53 * 0.53), 0, 10, 11, 12)
    r, J, Jm = m.eval(rid)
    t = out[i + 3].split()
    assert t[:2] == ["reprojB", "1"]
    np.testing.assert_allclose([float(t[2]), float(t[3])], r, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(vec(i + 4, "J_TB_min").reshape(2, 6), Jm[0], rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(vec(i + 5, "J_hpB_min").reshape(2, 3), Jm[1], rtol=1e-11, atol=1e-10)
