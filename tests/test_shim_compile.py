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
    for hdr in ("okvis/Estimator.hpp", "okvis/ceres/Map.hpp", "okvis/ceres/HomogeneousPointError.hpp"):
        src = tmp_path / "tu.cpp"
        pre = "#include <mock_eigen.hpp>\n" if "HomogeneousPoint" in hdr else ""
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
