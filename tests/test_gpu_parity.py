"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (float path, stated per SURVEY.md 8(c)): residuals/Jacobians abs/rel 1e-10 (IMU factors 1e-6
relative because the 15x15 information matrix is inverted by different algorithms on the two sides),
reduced Hessian 1e-9 relative, converged poses <= 1e-4 relative (north_star), typically 1e-8.
"""
import os

import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu

DIAG = os.environ.get("SVIN_DIAG", "")


def log(*a):
    msg = " ".join(str(x) for x in a)
    print(msg)
    if DIAG:
        with open(DIAG, "a") as f:
            f.write(msg + "\n")


def make_pair(spec, **kw):
    from svin_amd.estimator import Estimator
    from oracle import orc
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    fg, lg = syn.feed(gpu, spec, **kw)
    fc, lc = syn.feed(cpu, spec, **kw)
    return gpu, cpu, fg, fc, lg, lc


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))) if a.size else 0.0


def pose_diff(Ta, Tb):
    dq = Ta[3:] - Tb[3:] * np.sign(Ta[3:] @ Tb[3:])
    return max(np.linalg.norm(Ta[:3] - Tb[:3]) / max(1.0, np.linalg.norm(Tb[:3])), np.linalg.norm(dq))


def swap_model(spec, model, dist):
    for c in spec.cameras:
        c["model"], c["dist"] = model, dist
    return spec


@pytest.mark.parametrize("model,dist", [(syn.DIST_RADTAN, None), (syn.DIST_EQUIDISTANT, [-0.21, 0.14, 0.0006, 0.0003]),
                                        (syn.DIST_RADTAN8, [-0.16, 0.15, 0.0003, 0.0002, 0.01, 0.02, -0.01, 0.005]),
                                        (syn.DIST_NONE, [])])
@pytest.mark.parametrize("robust", [False, True])
def test_reprojection_residuals_and_jacobians(gpu_lib, model, dist, robust):
    from oracle import orc
    spec = syn.make_window(P=4, L=120, n_obs=1200, seed=11)
    if dist is not None:
        swap_model(spec, model, dist)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    ev = gpu.eval_reprojection(robust=robust)
    n = len(ev["r"])
    assert n == spec.N
    m = cpu.map()
    worst = dict(r=0.0, Jp=0.0, Jl=0.0, Je=0.0)
    for i in range(n):
        rid = int(ev["res_id"][i])  # residual ids are handed out in insertion order on both sides
        r, Js, Jm = m.eval(rid)
        sc = 1.0
        if robust:  # Cauchy(1) corrector with rho'' < 0: scale by sqrt(rho')
            sc = np.sqrt(1.0 / (1.0 + r @ r))
        worst["r"] = max(worst["r"], np.max(np.abs(ev["r"][i] - sc * r)))
        worst["Jp"] = max(worst["Jp"], np.max(np.abs(ev["Jp"][i] - sc * Jm[0])) / max(1.0, np.max(np.abs(Jm[0]))))
        worst["Jl"] = max(worst["Jl"], np.max(np.abs(ev["Jl"][i] - sc * Jm[1])) / max(1.0, np.max(np.abs(Jm[1]))))
        worst["Je"] = max(worst["Je"], np.max(np.abs(ev["Je"][i] - sc * Jm[2])) / max(1.0, np.max(np.abs(Jm[2]))))
    log("reproj parity model", model, "robust", robust, worst)
    assert worst["r"] < 1e-9 and worst["Jp"] < 1e-10 and worst["Jl"] < 1e-10 and worst["Je"] < 1e-10


def oracle_imu_sensitivity(cpu, frames, rids):
    """How far do the weighted quantities of the IMU factors (chi^2 = r^T r, J^T J, J^T r: what the solver consumes) move when the
    ORACLE's own states move by a few units of the last place?  r and J are scaled by the Cholesky factor of the inverse of the
    propagated 15 x 15 covariance, whose condition number (1e8 ... 1e10 with the rigs' noise densities) multiplies every rounding
    error on the way: two correct double-precision implementations that invert P differently (Eigen's LLT solve in the reference,
    ImuError.cpp:562-564; a tile Cholesky and triangular solves on the device) agree to about that amplification, not to 1e-16.
    Returns per residual id the relative change of (chi^2, J^T J) and the change of J^T r in units of the column norms of J."""
    m = cpu.map()
    base = {rid: m.eval(rid) for rid in rids}
    saved = [(f, cpu.get_T_WS(f).copy(), cpu.get_speed_and_bias(f).copy()) for f in frames]
    rng = np.random.default_rng(12)
    for f, T, sb in saved:
        T2 = T.copy()
        T2[:3] *= 1.0 + 4e-16 * rng.normal(size=3)
        T2[3:] += 4e-16 * rng.normal(size=4)
        T2[3:] /= np.linalg.norm(T2[3:])
        assert cpu.set_T_WS(f, T2) and cpu.set_speed_and_bias(f, sb * (1.0 + 4e-16 * rng.normal(size=9)))
    out = {}
    for rid in rids:
        r, _, Jm = base[rid]
        r2, _, Jm2 = m.eval(rid)
        J, J2 = np.concatenate(Jm, axis=1), np.concatenate(Jm2, axis=1)
        cn = np.sqrt(np.maximum(np.sum(J * J, axis=0), 1e-300))
        out[rid] = (abs(float(r2 @ r2) - float(r @ r)) / max(float(r @ r), 1e-300), rel(J2.T @ J2, J.T @ J),
                    float(np.max(np.abs(J2.T @ r2 - J.T @ r) / cn)))
    for f, T, sb in saved:
        assert cpu.set_T_WS(f, T) and cpu.set_speed_and_bias(f, sb)
    return out


def check_small_factors(gpu, cpu, expect_kinds, n_sonar=None, oracle_frames=None):
    """every non-reprojection factor of the window: residual, stacked minimal Jacobian, J^T J, J^T r against the oracle.  The rows of
    an IMU factor are compared loosely (1e-6: they carry the square root of an ill-conditioned information matrix, and a Cholesky
    factor is only defined up to the rounding of the matrix it factors); what is HELD for them are the weighted invariants the
    solver consumes -- chi^2, J^T J, J^T r -- at 1e-10 or, where the oracle itself moves more than that under a last-place change
    of its states, at 50 x that measured sensitivity (oracle_imu_sensitivity; needs `oracle_frames`)."""
    m = cpu.map()
    facs = gpu.eval_factors()
    assert len(facs) > 0
    sens = oracle_imu_sensitivity(cpu, oracle_frames, [f["res_id"] for f in facs if f["kind"] == 0]) if oracle_frames is not None else {}
    kinds, count = set(), {}
    for f in facs:
        r, Js, Jm = m.eval(f["res_id"])
        J = np.concatenate(Jm, axis=1)
        kinds.add(f["kind"])
        count[f["kind"]] = count.get(f["kind"], 0) + 1
        tol = {0: 1e-6, 3: 1e-8}.get(f["kind"], 1e-10)  # IMU: different 15x15 inverse (see above); relpose: 1e8-scale weights
        dr = np.max(np.abs(f["r"] - r)) / max(1.0, np.max(np.abs(r)))
        dJ = np.max(np.abs(f["J"] - J)) / max(1.0, np.max(np.abs(J)))
        # weighting-independent invariants
        dH = rel(f["J"].T @ f["J"], J.T @ J)
        # gradient contribution J^T r, compared in units of the parameters' standard deviations (column norms of J):
        # a relative measure is meaningless where r is rounding noise (the relative-extrinsics factors start at r = 0
        # exactly, |r| ~ 1e-9 after one normalisation, times 1e8-scale weights), an absolute one is not
        cn = np.sqrt(np.maximum(np.sum(J * J, axis=0), 1e-300))
        dg = float(np.max(np.abs(f["J"].T @ f["r"] - J.T @ r) / cn))
        log("factor kind", f["kind"], "m", f["m"], "dr", dr, "dJ", dJ, "dJtJ", dH, "dJtr/sigma", dg)
        assert dr < tol and dJ < tol, (f["kind"], dr, dJ)
        assert dH < 1e-7
        assert dg < 1e-8 * max(1.0, float(np.max(np.abs(r)))), (f["kind"], dg)
        if f["kind"] == 0 and f["res_id"] in sens:
            s_chi, s_H, s_g = sens[f["res_id"]]
            dchi = abs(float(f["r"] @ f["r"]) - float(r @ r)) / max(float(r @ r), 1e-300)
            log("  IMU factor invariants: chi^2", dchi, "(oracle sensitivity %.1e)" % s_chi, "J^T J", dH, "(%.1e)" % s_H, "J^T r / sigma", dg, "(%.1e)" % s_g)
            assert dchi < max(1e-10, 50 * s_chi), (dchi, s_chi)
            assert dH < max(1e-10, 50 * s_H), (dH, s_H)
            assert dg < max(1e-10, 50 * s_g) * max(1.0, float(np.max(np.abs(r)))), (dg, s_g)
    assert expect_kinds.issubset(kinds), kinds
    if n_sonar is not None:
        assert count.get(4, 0) == n_sonar, count
    return count


def test_small_factors_parity(gpu_lib):
    """IMU (0), pose prior (1), speed/bias prior (2), relative extrinsics (3), SONAR (4) and DEPTH (5) factors of a
    rig-v2 window: every frame carries a sonar return whose visual patch exists (U4, Estimator.cpp:265-316)"""
    spec = syn.make_window(P=4, L=300, n_obs=2500, seed=5, rig="rig_v2", sonar=True, depth=True)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    count = check_small_factors(gpu, cpu, {0, 1, 2, 3, 4, 5}, n_sonar=spec.P, oracle_frames=fc)
    assert count[5] == spec.P
    # the same factors after the states have moved (the patch stays what it was at construction, SonarError.cpp:66);
    # both sides evaluate at the ORACLE's optimised states, so that the comparison stays one of the evaluation
    for e in (gpu, cpu):
        e.optimize(3)
    for a, b in zip(fg, fc):
        assert gpu.set_T_WS(a, cpu.get_T_WS(b)) and gpu.set_speed_and_bias(a, cpu.get_speed_and_bias(b))
        for c in (0, 1):
            assert gpu.set_camera_sensor_states(a, c, cpu.get_camera_sensor_states(b, c))
    # (no 1e-10 invariants here: each side re-integrates its IMU factors when ITS bias estimate has moved far enough from the
    # linearisation point of the pre-integration (ImuError.cpp:581-592), so after three iterations of two solvers the two factors
    # are first-order expansions around slightly different biases -- chi^2 differs by 2e-11, J^T r by 6e-10 sigma, measured --
    # a second-order effect of the reference's own algorithm, not rounding)
    check_small_factors(gpu, cpu, {0, 1, 2, 3, 4, 5}, n_sonar=spec.P)


def test_config3_full_size_sonar_depth(gpu_lib):
    """BASELINE config #3 at full size: rig v2 (per-frame extrinsics + relative-pose factors), 10 keyframes, 4 000
    landmarks, ~40 000 reprojection residuals, one depth and one sonar factor (8-30 point patch) per state"""
    spec = syn.make_window(P=10, L=4000, n_obs=40000, seed=20250629, rig="rig_v2", sonar=True, depth=True)
    assert spec.P == 10 and spec.L == 4000 and spec.N == 40000
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    count = check_small_factors(gpu, cpu, {0, 1, 2, 3, 4, 5}, n_sonar=10)
    assert count[5] == 10 and count[0] == 9 and count[3] == 18
    gpu.optimize(10)
    cpu.optimize(10)
    sg, sc = gpu.summary(), cpu.summary()
    log("config3 gpu", sg, "cpu", sc)
    assert sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
    assert sg["final_cost"] < sg["initial_cost"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    worst_e = max(pose_diff(gpu.get_camera_sensor_states(a, c), cpu.get_camera_sensor_states(b, c))
                  for a, b in zip(fg, fc) for c in (0, 1))
    err = max(np.linalg.norm(gpu.get_T_WS(a)[:3] - spec.T_WS_true[k, :3]) for k, a in enumerate(fg))
    log("config3 pose difference vs oracle", worst, "extrinsics", worst_e, "max position error vs truth", err)
    assert worst < 1e-4 and worst_e < 1e-4
    assert err < 0.1


@pytest.mark.parametrize("model,dist", [(syn.DIST_RADTAN, [-0.28, 0.07, 0.0002, 1.8e-05]), (syn.DIST_EQUIDISTANT, [-0.21, 0.14, 0.0006, 0.0003]),
                                        (syn.DIST_RADTAN8, [-0.16, 0.15, 0.0003, 0.0002, 0.01, 0.02, -0.01, 0.005]),
                                        (syn.DIST_NONE, [])])
def test_reprojection_edge_cases(gpu_lib, model, dist):
    """points the reference treats specially: closer than 0.2 m / behind the camera (zero Jacobians, residual kept,
    ReprojectionError.hpp:140-147), on the optical axis (equidistant limit r <= 1e-8, EquidistantDistortion.hpp:176-183),
    beyond rho = 9 (radtan8 gives up, RadialTangentialDistortion8.hpp:104,125), negative and tiny homogeneous scale"""
    from svin_amd.estimator import Estimator
    from oracle import orc
    imu = dict(syn.test_rig()[1])
    rate = 100
    ns = ((np.arange(8) - 2) * (1_000_000_000 // rate)).astype(np.int64) + 1_000_000_000
    t = np.stack([50 + ns // 1_000_000_000, ns % 1_000_000_000], 1).astype(np.uint32)
    m = np.zeros((8, 6))
    m[:, 5] = imu["g"]                       # level and at rest: the first pose is the identity
    T_SC = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    pts = [[0.0, 0.0, 5.0, 1.0],             # exactly on the optical axis
           [1e-9, -1e-9, 4.0, 1.0],          # inside the equidistant limit
           [0.3, -0.2, 0.15, 1.0],           # closer than 20 cm: invalid
           [0.3, 0.2, -2.0, 1.0],            # behind the camera: invalid
           [12.0, 1.0, 3.0, 1.0],            # rho = 16.1 > 9
           [2.9 * 2.0, 0.0, 2.0, 1.0],       # rho = 8.41: still accepted
           [-0.4, 0.6, -6.0, -2.0],          # negative homogeneous scale (the same point as (0.2, -0.3, 3))
           [0.5, 0.25, 3.0, 1e-9],           # |w| <= 1e-8: no validity test
           [0.4, -0.1, 2.5, 1.0]]            # an ordinary point
    ests = []
    for cls in (Estimator, orc.OracleEstimator):
        e = cls(0) if cls is Estimator else cls()
        e.add_camera(model, [350.0, 360.0, 378.0, 238.0], dist, 752, 480, [0.0, 0.0, 0.0, 0.0])
        e.add_imu(imu)
        lids = [e.new_id() for _ in pts]
        for lid, p in zip(lids, pts):
            assert e.add_landmark(lid, np.array(p))
        fid = e.new_id()
        assert e.add_states(fid, (51, 0), 400, T_SC, t, m, True)
        for k, lid in enumerate(lids):
            assert e.add_observation(lid, fid, 0, k, [300.0 + 7 * k, 200.0 - 5 * k], 6.0) != 0
        ests.append(e)
    gpu, cpu = ests
    assert np.max(np.abs(gpu.get_T_WS(fid) - cpu.get_T_WS(fid))) == 0.0
    ev = gpu.eval_reprojection(robust=False)
    mp_ = cpu.map()
    for i in range(len(pts)):
        r, Js, Jm = mp_.eval(int(ev["res_id"][i]))
        k = int(np.nonzero(np.array(lids) == int(ev["lm_id"][i]))[0][0])
        for a, b in ((ev["r"][i], r), (ev["Jp"][i], Jm[0]), (ev["Jl"][i], Jm[1]), (ev["Je"][i], Jm[2])):
            assert np.max(np.abs(a - b)) <= 1e-10 * max(1.0, np.max(np.abs(b))), (k, pts[k], a, b)
        if k in (2, 3):
            assert np.all(ev["Jp"][i] == 0) and np.all(ev["Jl"][i] == 0) and np.all(ev["Je"][i] == 0) and np.any(ev["r"][i] != 0)
        if k == 8:
            assert np.any(ev["Jp"][i] != 0)
    # the window still optimises: the invalid observations contribute cost but no curvature (V_l = 0: damping alone keeps
    # the landmark block positive definite, the step is zero).  One frame with one observation per landmark is a gauge-free
    # toy -- the minimiser is not unique, so only what is determined is compared: no failure, the same number of
    # accepted steps, a decreasing cost, and the invalid landmarks stay where they were on both sides.
    before = [gpu.get_landmark(lids[k])["point"].copy() for k in (2, 3)]
    gpu.optimize(5)
    cpu.optimize(5)
    sg, sc = gpu.summary(), cpu.summary()
    assert sg["termination"] != 3 and sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
    assert sg["final_cost"] < sg["initial_cost"] and abs(sg["initial_cost"] - sc["initial_cost"]) <= 1e-12 * sc["initial_cost"]
    for k, b in zip((2, 3), before):
        assert np.array_equal(gpu.get_landmark(lids[k])["point"], b) and np.array_equal(cpu.get_landmark(lids[k])["point"], b)


def map_blocks(est, ids):
    return [est.describe_block(int(b)) if hasattr(est, "describe_block") else None for b in ids]


def oracle_describe(cpu, bid):
    import ctypes as C
    f, k, ix = C.c_uint64(), C.c_int(), C.c_int()
    ok = cpu.L.orc_describe_block(cpu.h, int(bid), C.byref(f), C.byref(k), C.byref(ix))
    return (int(f.value), int(k.value), int(ix.value)) if ok else None


def reduced_permutation(gpu, cpu, fg, fc, lin_g, lin_c):
    """index map so that S_gpu[perm][:, perm] lines up with the oracle ordering"""
    fmap = {a: b for a, b in zip(fg, fc)}
    key_c = {}
    for bid, off in zip(lin_c["cam_ids"], lin_c["cam_off"]):
        key_c[oracle_describe(cpu, bid)] = int(off)
    perm = np.zeros(lin_c["d"], int)
    dims = {0: 6, 1: 6, 2: 9}
    for bid, off in zip(lin_g["block_ids"], lin_g["block_off"]):
        fr, kind, ix = gpu.describe_block(bid)
        oc = key_c[(fmap[fr], kind, ix)]
        for k in range(dims[kind]):
            perm[oc + k] = off + k
    return perm


@pytest.mark.parametrize("rig", ["euroc", "rig_v2"])
def test_reduced_system_parity(gpu_lib, rig):
    spec = syn.make_window(P=5, L=200, n_obs=2000, seed=21, rig=rig, depth=(rig == "rig_v2"))
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    lin_c = cpu.map().linearize(0.0)
    lin_g = gpu.linearize(0.0)
    assert lin_g["d"] == lin_c["d"]
    perm = reduced_permutation(gpu, cpu, fg, fc, lin_g, lin_c)
    S = lin_g["S"][np.ix_(perm, perm)]
    g = lin_g["g"][perm]
    # the 1e8 / 1e16 prior entries would hide everything else: compare the diagonally normalised system
    sd = np.sqrt(np.abs(np.diag(lin_c["S"])))
    Sn, Sc = S / np.outer(sd, sd), lin_c["S"] / np.outer(sd, sd)
    gn, gc = g / sd, lin_c["g"] / sd
    log(rig, "cost gpu/cpu", lin_g["cost"], lin_c["cost"], "dS(normalised)", rel(Sn, Sc), "dg(normalised)", rel(gn, gc),
        "asym", rel(S, S.T))
    assert abs(lin_g["cost"] - lin_c["cost"]) <= 1e-9 * lin_c["cost"]
    assert rel(Sn, Sc) < 1e-9
    assert rel(gn, gc) < 1e-9


@pytest.mark.parametrize("P,L,n_obs", [(5, 200, 2000), (10, 500, 5000), (14, 300, 2400), (12, 60, 1400), (3, 40, 200),
                                        (48, 900, 9000), (70, 1500, 12000)])  # last two: panel-pair kernel (dC = 288 / 420)
def test_dense_and_pairwise_schur_agree(gpu_lib, debug_option, P, L, n_obs):
    """The landmark-elimination kernels (Gram-matrix form on MFMA: one block for narrow windows, 96-row panel pairs for
    wide ones; pairwise blocks as the general fallback) must produce the same reduced system and the same optimisation
    result on a window both can handle."""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=P, L=L, n_obs=n_obs, seed=5, rig="euroc")
    out = {}
    for mode in ("dense", "pairwise"):
        if mode == "pairwise":
            debug_option("SVIN_SCHUR_PAIRWISE", 1)
        else:
            debug_option("SVIN_SCHUR_PAIRWISE", 0)
        est = Estimator(0)
        f, l = syn.feed(est, spec)
        lin = est.linearize(1e-8)
        est.set_solver_options(1e-12, 1e-12, 1e-12)
        est.optimize(30)
        out[mode] = (lin, [est.get_T_WS(i) for i in f], est.summary())
    (la, Ta, sa), (lb, Tb, sb) = out["dense"], out["pairwise"]
    sd = np.sqrt(np.abs(np.diag(lb["S"])))
    dS = rel(la["S"] / np.outer(sd, sd), lb["S"] / np.outer(sd, sd))
    dg = rel(la["g"] / sd, lb["g"] / sd)
    worst = max(pose_diff(a, b) for a, b in zip(Ta, Tb))
    log("dense vs pairwise P", P, "dS", dS, "dg", dg, "pose", worst, "iterations", sa["iterations"], sb["iterations"])
    assert dS < 1e-10 and dg < 1e-10
    assert sa["iterations"] == sb["iterations"]
    assert worst < 1e-8


def test_mailbox_and_memcpy_scalar_paths_agree(gpu_lib, debug_option):
    """Per-iteration scalars through the pinned-host mailbox or through memcpy + synchronise: identical runs."""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=6, L=300, n_obs=3000, seed=9, rig="euroc")
    res = []
    for no_mailbox in (False, True):
        if no_mailbox:
            debug_option("SVIN_NO_MAILBOX", 1)
        else:
            debug_option("SVIN_NO_MAILBOX", 0)
        est = Estimator(0)
        f, l = syn.feed(est, spec)
        est.optimize(15)
        res.append((est.summary(), [est.get_T_WS(i) for i in f]))
    assert res[0][0]["iterations"] == res[1][0]["iterations"]
    assert res[0][0]["final_cost"] == res[1][0]["final_cost"]
    assert max(np.max(np.abs(a - b)) for a, b in zip(res[0][1], res[1][1])) == 0.0


@pytest.mark.parametrize("rig,P,kw", [("euroc", 6, {}), ("rig_v2", 6, dict(sonar=True, depth=True)),
                                      ("rig_v2", 9, dict(depth=True))])  # d = 90 / 162 / 243 (LDS and global Cholesky)
def test_optimize_matches_oracle(gpu_lib, rig, P, kw):
    spec = syn.make_window(P=P, L=300, n_obs=3000, seed=33, rig=rig, **kw)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    for e in (gpu, cpu):
        e.set_solver_options(1e-12, 1e-12, 1e-12)
    gpu.optimize(40)
    cpu.optimize(40)
    sg, sc = gpu.summary(), cpu.summary()
    log(rig, "summary gpu", sg, "cpu", sc)
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    worst_sb = max(np.max(np.abs(gpu.get_speed_and_bias(a) - cpu.get_speed_and_bias(b))) for a, b in zip(fg, fc))
    worst_lm = max(np.max(np.abs(gpu.get_landmark(a)["point"] - cpu.get_landmark(b)["point"])) for a, b in zip(lg, lc))
    worst_q = max(abs(gpu.get_landmark(a)["quality"] - cpu.get_landmark(b)["quality"]) for a, b in zip(lg, lc))
    log(rig, "pose", worst, "sb", worst_sb, "lm", worst_lm, "quality", worst_q)
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert worst < 1e-4 and worst_sb < 1e-4 and worst_lm < 1e-3 and worst_q < 1e-6
    # and close to the ground truth (TestEstimator-style thresholds, TestEstimator.cpp:209-212)
    Tg = gpu.get_T_WS(fg[-1])
    assert np.linalg.norm(Tg[:3] - spec.T_WS_true[-1, :3]) < 1e-1


def test_imu_propagation_parity(gpu_lib):
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=3, L=20, n_obs=100, seed=2)
    gpu = Estimator(0)
    T0, sb0 = spec.T_WS_true[0].copy(), spec.sb_true[0].copy()
    sb0[3:] = [0.01, -0.02, 0.005, 0.05, -0.03, 0.02]
    t0, t1 = tuple(int(v) for v in spec.stamps[0]), tuple(int(v) for v in spec.stamps[1])
    n, T, sb, cov, jac = gpu.imu_propagation(spec.imu_t, spec.imu_meas, spec.imu_params, T0, sb0, t0, t1, True, True)
    L = orc.lib()
    Tc, sbc, covc, jacc = T0.copy(), sb0.copy(), np.zeros((15, 15)), np.zeros((15, 15))
    it, im, par = orc.arr(spec.imu_t, np.uint32), orc.arr(spec.imu_meas), orc.imu_params_vector(spec.imu_params)
    nc = L.orc_imu_propagation(len(it), orc.u32ptr(it), orc.dptr(im), orc.dptr(par), orc.dptr(Tc), orc.dptr(sbc), t0[0], t0[1],
                               t1[0], t1[1], orc.dptr(covc), orc.dptr(jacc))
    log("propagation steps", n, nc, "dT", np.max(np.abs(T - Tc)), "dsb", np.max(np.abs(sb - sbc)), "dcov", rel(cov, covc),
        "djac", rel(jac, jacc))
    assert n == nc
    assert np.max(np.abs(T - Tc)) < 1e-11 and np.max(np.abs(sb - sbc)) < 1e-11
    assert rel(cov, covc) < 1e-9 and rel(jac, jacc) < 1e-10


def run_sequence(est, spec, num_kf, num_imu, iters):
    removed_all = []

    def on_frame(k, fid):
        est.optimize(iters)
        ok, removed = est.apply_marginalization(num_kf, num_imu)
        assert ok
        removed_all.append(len(removed))
    f, l = syn.feed(est, spec, on_frame=on_frame)
    return f, l, removed_all


@pytest.mark.parametrize("rig", ["euroc", "test4", "rig_v2"])   # rig_v2: per-frame extrinsics, the first frame's fixed
def test_marginalization_sequence_parity(gpu_lib, rig):
    """E6 / M1-M4: optimise + applyMarginalizationStrategy every frame, compare priors and states."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=8, L=250, n_obs=2500, seed=44, rig=rig, keyframe_every=2, frame_dt=0.3)
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    for e in (gpu, cpu):
        e.set_solver_options(1e-12, 1e-12, 1e-12)
    fg, lg, rg = run_sequence(gpu, spec, 2, 3, 25)
    fc, lc, rc = run_sequence(cpu, spec, 2, 3, 25)
    log(rig, "removed landmarks per frame gpu", rg, "cpu", rc)
    assert rg == rc
    assert gpu.num_frames() == cpu.num_frames() and gpu.num_landmarks() == cpu.num_landmarks()
    assert fg == fc
    mg, mc = gpu.marg(), cpu.marg()
    assert (mg is None) == (mc is None)
    if mg is not None:
        # Rounding-level parity of M1-M3 is test_marginalization_one_shot's job (identical states in).  Here the prior
        # is the product of eight optimise + marginalise rounds in which GPU and oracle states drift apart by up to
        # ~1e-3 during the first, ill-conditioned frames (1e8^2 gauge prior next to unconstrained directions, 25
        # iterations without convergence) and contract again; the prior inherits that history.  Compared in units of
        # the parameters' standard deviations, one order tighter than the 1e-4 relative pose bar of the north star.
        # The prior is a function of the linearisation points, so it differs as the states do (logged); what is
        # asserted on it is consistency (J^T J reproduces H) and the order of magnitude, the bar proper is the final
        # window below.
        o = compare_priors(mg, mc, "sequence prior %s" % rig)
        assert o["selfH"] < 1e-9 and o["dH"] < 1e-2 and o["dJtJ"] < 1e-2
        assert o["db0"] < 0.05 * max(1.0, o["b0_scale"])
    gf, cf = gpu.frame_ids(), cpu.frame_ids()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(gf, cf))
    log(rig, "final window pose difference", worst)
    bar = 1e-4
    if rig == "rig_v2":
        # sigma_c_relative = 1e-8 puts 3e16 of information between consecutive extrinsics: the sequence amplifies
        # rounding.  The yardstick is the oracle against itself with its initial landmarks moved by 1e-13 m.
        spec2 = syn.make_window(P=8, L=250, n_obs=2500, seed=44, rig=rig, keyframe_every=2, frame_dt=0.3)
        spec2.lm_init[:, :3] += 1e-13 * np.random.default_rng(1).normal(size=(spec2.L, 3))
        cpu2 = orc.OracleEstimator()
        cpu2.set_solver_options(1e-12, 1e-12, 1e-12)
        run_sequence(cpu2, spec2, 2, 3, 25)
        sens = max(pose_diff(cpu2.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(cpu2.frame_ids(), cf))
        log(rig, "oracle vs oracle with landmarks moved by 1e-13 m:", sens)
        bar = max(1e-4, 30 * sens)
    assert worst < bar


def snapshot_states(est, n_cam=2):
    """every state of the window by id: poses, speed/bias, extrinsics, landmarks"""
    snap = dict(T={}, sb={}, ext={}, lm={})
    for f in est.frame_ids():
        snap["T"][f] = est.get_T_WS(f)
        snap["sb"][f] = est.get_speed_and_bias(f)
        snap["ext"][f] = [est.get_camera_sensor_states(f, c) for c in range(n_cam)]
    for l in est.landmark_ids():
        snap["lm"][l] = est.get_landmark(l)["point"]
    return snap


def inject_states(est, snap):
    assert est.frame_ids() == sorted(snap["T"]) and est.landmark_ids() == sorted(snap["lm"])
    for f, T in snap["T"].items():
        assert est.set_T_WS(f, T)
        if snap["sb"][f] is not None:
            assert est.set_speed_and_bias(f, snap["sb"][f])
        for c, Te in enumerate(snap["ext"][f]):
            assert est.set_camera_sensor_states(f, c, Te)
    for l, hp in snap["lm"].items():
        assert est.set_landmark(l, hp)


def compare_priors(mg, mc, tag):
    """H, b0, J^T J, J^T e0 and the numerical rank of two marginalisation priors (same frame ids on both sides), in the
    second one's ordering and in units of the parameters' standard deviations (entries span 1e-2 ... 1e16)"""
    assert mg is not None and mc is not None and mg["n"] == mc["n"]
    keyc = {(b["frame"], b["kind"], b["index"]): b for b in mc["blocks"]}
    perm = np.zeros(mg["n"], int)
    for b in mg["blocks"]:
        if b["frame"] is None:
            assert b["mdim"] == 0
            continue
        o = keyc[(b["frame"], b["kind"], b["index"])]
        assert o["mdim"] == b["mdim"]
        for k in range(b["mdim"]):
            perm[o["ordering"] + k] = b["ordering"] + k
    sd = np.sqrt(np.maximum(np.abs(np.diag(mc["H"])), 1e-300))

    def nrm(M):
        return M / np.outer(sd, sd)
    H, b0 = mg["H"][np.ix_(perm, perm)], mg["b0"][perm]
    Ht, bp = (mg["J"].T @ mg["J"])[np.ix_(perm, perm)], (mg["J"].T @ mg["e0"])[perm]
    Hc, b0c = mc["H"], mc["b0"]
    Htc, bpc = mc["J"].T @ mc["J"], mc["J"].T @ mc["e0"]
    out = dict(n=mg["n"], dH=float(np.max(np.abs(nrm(H) - nrm(Hc)))), db0=float(np.max(np.abs(b0 - b0c) / sd)),
               dJtJ=float(np.max(np.abs(nrm(Ht) - nrm(Htc)))), dJte0=float(np.max(np.abs(bp - bpc) / sd)),
               b0_scale=float(np.max(np.abs(b0c) / sd)),
               rank_g=int(np.sum(np.any(mg["J"] != 0, axis=1))), rank_c=int(np.sum(np.any(mc["J"] != 0, axis=1))),
               selfH=float(np.max(np.abs(nrm(Ht) - nrm(H)))))
    log(tag, out)
    return out


def one_shot_pass(est, spec, at, snaps=None, redo=False):
    """optimise + marginalise at the frames in `at`; with `snaps` the states recorded from another estimator are put in
    place right before each marginalisation.  Returns the recorded (snapshot, removed ids, prior) per marginalisation."""
    rec = []

    def cb(k, fid):
        if k not in at:
            return
        est.optimize(12)
        if snaps is not None:
            inject_states(est, snaps[len(rec)])
            if redo:
                est.invalidate_preintegration()
        snap = snapshot_states(est)
        ok, removed = est.apply_marginalization(2, 2)
        assert ok
        rec.append((snap, sorted(int(i) for i in removed), est.marg(), est.frame_ids(), est.num_landmarks(),
                    est.marg_pre() if hasattr(est, "marg_pre") else None))
    f, l = syn.feed(est, spec, on_frame=cb)
    return rec, f


@pytest.mark.parametrize("rig,kw", [("euroc", {}), ("test4", {}), ("rig_v2", dict(sonar=True, depth=True))])
def test_marginalization_one_shot(gpu_lib, rig, kw):
    """M1-M3 at rounding level: the SAME states on both sides (the oracle's optimised window is put
    into the GPU estimator), ONE applyMarginalizationStrategy, priors compared at 1e-9 of a standard deviation -- first
    without a previous prior (frame 5: four frames leave at once), then on top of that prior (frame 6).  No optimisation
    of its own between the injection and the comparison that could amplify anything.  rig_v2 carries every factor kind
    of the reference (sonar, depth, relative extrinsics with 1e16 information)."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=7, L=500, n_obs=4000, seed=52, rig=rig, keyframe_every=2, frame_dt=0.3, **kw)
    rec_0, _ = one_shot_pass(orc.OracleEstimator(), spec, at=(5, 6))
    snaps = [r[0] for r in rec_0]
    # both sides receive the recorded states through their setters (which normalise quaternions): bit-identical inputs
    rec_c, fc = one_shot_pass(orc.OracleEstimator(), spec, at=(5, 6), snaps=snaps)
    rec_g, fg = one_shot_pass(Estimator(0), spec, at=(5, 6), snaps=snaps)
    assert fg == fc and len(rec_c) == len(rec_g) == 2
    import mp_marg
    for i, (g, c) in enumerate(zip(rec_g, rec_c)):
        assert g[1] == c[1] and g[3] == c[3] and g[4] == c[4]          # removed landmarks, frames, landmark count
        assert len(c[1]) >= 1
        o = compare_priors(g[2], c[2], "one-shot marginalisation %s #%d (GPU vs oracle)" % (rig, i))
        # The two double-precision results differ by 1e-9 ... 1e-10 of a standard deviation -- which one is off?  Neither:
        # an independent 40-digit restatement of M2 / M3 (tests/mp_marg.py) applied to the oracle's system after M1 gives
        # the exact Schur complement, and BOTH sit at that distance from it (conditioning of a window whose first pose
        # carries a 1e8 prior and whose extrinsics chain carries 1e16), sometimes the one closer, sometimes the other.
        pm = c[5]
        exact = mp_marg.marginalize_mp(pm["H"], pm["b0"], pm["lm"], pm["dense"])
        ex = dict(n=g[2]["n"], H=exact["H"], b0=exact["b0"], blocks=c[2]["blocks"])
        sd = np.sqrt(np.abs(np.diag(exact["H"])))
        err = {}
        for name, m in (("gpu", g[2]), ("oracle", c[2])):
            keyc = {(b["frame"], b["kind"], b["index"]): b for b in c[2]["blocks"]}
            perm = np.zeros(m["n"], int)
            for b in m["blocks"]:
                if b["frame"] is not None:
                    for k in range(b["mdim"]):
                        perm[keyc[(b["frame"], b["kind"], b["index"])]["ordering"] + k] = b["ordering"] + k
            H, b0, J = m["H"][np.ix_(perm, perm)], m["b0"][perm], m["J"][:, perm]
            err[name] = dict(H=float(np.max(np.abs(H - exact["H"]) / np.outer(sd, sd))), b0=float(np.max(np.abs(b0 - exact["b0"]) / sd)),
                             JtJ=float(np.max(np.abs(J.T @ J - exact["JtJ"]) / np.outer(sd, sd))),
                             Jte0=float(np.max(np.abs(J.T @ m["e0"] - exact["Jte0"]) / sd)))
        log("one-shot marginalisation %s #%d distance to the exact (40-digit) result:" % (rig, i), err, "exact rank", exact["rank"],
            "smallest relative eigenvalues", exact["rel_eigs_small"][:4])
        bs = max(1.0, o["b0_scale"])
        for name in ("gpu", "oracle"):
            assert err[name]["H"] < 1e-8 and err[name]["JtJ"] < 1e-8, (name, err)
            assert err[name]["b0"] < 1e-8 * bs and err[name]["Jte0"] < 1e-8 * bs, (name, err)
        assert o["dH"] < 1e-8 and o["dJtJ"] < 1e-8 and o["selfH"] < 1e-9
        assert o["db0"] < 1e-8 * bs and o["dJte0"] < 1e-8 * bs
        # rank: eigenvalues that are zero in exact arithmetic come out as +-1e-15 ... 1e-14 of the largest on either side
        # and the reference's threshold (eps n lambda_max, MarginalizationError.cpp:741) sits right there, so a kept / dropped
        # decision on such a direction is a coin toss in ANY double implementation; it carries no weight (J^T J and
        # J^T e0 above agree).  Directions that are clearly non-zero must be kept by both.
        n = g[2]["n"]
        clear = int(np.sum(np.array(exact["rel_eigs_small"]) > 100 * n * 2.3e-16))
        assert o["rank_c"] >= n - (len(exact["rel_eigs_small"]) - clear) and o["rank_g"] >= n - (len(exact["rel_eigs_small"]) - clear)
        assert abs(o["rank_g"] - exact["rank"]) <= len(exact["rel_eigs_small"]) - clear


@pytest.mark.parametrize("rig", ["euroc", "rig_v2"])
def test_marginalization_m1_against_exact_chain(gpu_lib, debug_option, rig):
    """M1 AND M2 of the HIP path against a 40-digit chain that starts from the raw residual definitions (tests/mp_m1.py +
    tests/mp_marg.py; structure -- which residuals, ordering, which rows leave -- from the oracle's log, every number
    recomputed): five marginalisations in sequence on identical states (the oracle's snapshots are injected before each
    call).  Checked per call: the GPU's system after M1 (SVIN_MARG_KEEP_PRE) and its prior after M2, each next to the
    oracle's distance from the same exact result.  rig_v2 = the reference's shipped rig: per-frame extrinsics chained by
    RelativePoseErrors with sigma_c_relative = 1e-8 (3e16 of information); what the sequence tests' wide pose bars are NOT:
    per marginalisation the HIP path sits at the same rounding distance from the exact chain as on EuRoC (H 1e-12, b0 1e-9 of a
    standard deviation -- the b0 floor is the relative-extrinsics residual itself, 1e-16 rad of quaternion rounding against
    sigma = 1e-8 rad, in any double implementation); the 1e-3 .. 1e-4 the 13-frame sequences end apart is what the OPTIMISATIONS
    between the marginalisations make of such differences (test_marginalization_large_prior_per_frame_extrinsics measures that
    amplification against the oracle perturbed by 1e-13)."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    import mp_m1
    from test_marginalization_m1_exact import tiny_sequence_spec, tiny_sequence_spec_rig_v2, scaled
    debug_option("SVIN_MARG_KEEP_PRE", 1)
    spec = tiny_sequence_spec() if rig == "euroc" else tiny_sequence_spec_rig_v2()
    bar_b1 = 1e-10 if rig == "euroc" else 2e-9
    cpu, gpu = orc.OracleEstimator(), Estimator(0)
    rec = []

    def cb_cpu(k, fid):
        if k < 4:
            return
        cpu.optimize(6)
        snap = snapshot_states(cpu)
        ok, removed = cpu.apply_marginalization(2, 2)
        assert ok
        rec.append(dict(snap=snap, removed=sorted(int(i) for i in removed), log=cpu.marg_m1_log(), pre=cpu.marg_pre(), prior=cpu.marg()))
    syn.feed(cpu, spec, on_frame=cb_cpu)
    chain = mp_m1.ExactChain()
    step = [0]
    worst = dict(gpu_m1=0.0, cpu_m1=0.0, gpu_m2=0.0, cpu_m2=0.0)

    def cb_gpu(k, fid):
        if k < 4:
            return
        r = rec[step[0]]
        step[0] += 1
        gpu.optimize(6)
        inject_states(gpu, r["snap"])
        ok, removed = gpu.apply_marginalization(2, 2)
        assert ok and sorted(int(i) for i in removed) == r["removed"]
        H, b0 = chain.m1(r["log"])
        # the GPU's post-M1 system, rows permuted into the log's ordering by block id
        pg = gpu.marg_pre()
        n = H.shape[0]
        assert pg is not None and pg["H"].shape[0] == n
        perm = np.zeros(n, int)
        for b in r["log"]["blocks"]:
            if b["mdim"] > 0:
                o, m = pg["rows_of"][b["id"]]
                assert m == b["mdim"], (b, o, m)
                perm[b["ordering"]:b["ordering"] + m] = np.arange(o, o + m)
        Hg, bg = pg["H"][np.ix_(perm, perm)], pg["b0"][perm]
        dHg, dbg, bs = scaled(Hg, bg, H, b0)
        dHc, dbc, _ = scaled(r["pre"]["H"], r["pre"]["b0"], H, b0)
        log("m1 vs exact, frame", k, ": gpu H %.2e b0 %.2e | oracle H %.2e b0 %.2e" % (dHg, dbg, dHc, dbc))
        worst["gpu_m1"], worst["cpu_m1"] = max(worst["gpu_m1"], dHg, dbg / max(1.0, bs)), max(worst["cpu_m1"], dHc, dbc / max(1.0, bs))
        assert dHg < 1e-10 and dbg < bar_b1 * max(1.0, bs), (k, dHg, dbg, bs)
        ex = chain.m2(r["pre"]["lm"], r["pre"]["dense"])
        # priors after M2: the GPU's kept blocks against the chain's, by block id
        mg, mc = gpu.marg(), r["prior"]
        assert mg["n"] == mc["n"] == ex["H"].shape[0]
        keyc = {(b["frame"], b["kind"], b["index"]): b for b in mc["blocks"]}
        p2 = np.zeros(mg["n"], int)
        for b in mg["blocks"]:
            if b["frame"] is not None:
                for q in range(b["mdim"]):
                    p2[keyc[(b["frame"], b["kind"], b["index"])]["ordering"] + q] = b["ordering"] + q
        dHg2, dbg2, bs2 = scaled(mg["H"][np.ix_(p2, p2)], mg["b0"][p2], ex["H"], ex["b0"])
        dHc2, dbc2, _ = scaled(mc["H"], mc["b0"], ex["H"], ex["b0"])
        log("m2 vs exact, frame", k, ": gpu H %.2e b0 %.2e | oracle H %.2e b0 %.2e" % (dHg2, dbg2, dHc2, dbc2))
        worst["gpu_m2"], worst["cpu_m2"] = max(worst["gpu_m2"], dHg2, dbg2 / max(1.0, bs2)), max(worst["cpu_m2"], dHc2, dbc2 / max(1.0, bs2))
        assert dHg2 < 1e-8 and dbg2 < 1e-8 * max(1.0, bs2), (k, dHg2, dbg2, bs2)
    syn.feed(gpu, spec, on_frame=cb_gpu)
    assert step[0] == len(rec) == 5
    log(rig, "distance to the exact chain, worst of five marginalisations:", worst)
    # (the GPU carries its OWN previous prior from call to call, the chain the exact one: from the second call on its M1
    # distance includes what its previous M2 left -- the same holds for the oracle column next to it)


@pytest.mark.parametrize("rig,kw", [("euroc", {}), ("rig_v2", dict(sonar=True, depth=True))])
def test_marginalization_is_deterministic(gpu_lib, rig, kw):
    """M1 accumulates in a fixed order (no atomics): the same states in, the same prior out -- bit for bit"""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=7, L=500, n_obs=4000, seed=52, rig=rig, keyframe_every=2, frame_dt=0.3, **kw)
    rec_c, fc = one_shot_pass(orc.OracleEstimator(), spec, at=(5, 6))
    snaps = [r[0] for r in rec_c]
    a, _ = one_shot_pass(Estimator(0), spec, at=(5, 6), snaps=snaps, redo=True)
    b, _ = one_shot_pass(Estimator(0), spec, at=(5, 6), snaps=snaps, redo=True)
    for x, y in zip(a, b):
        for key in ("H", "b0", "J", "e0"):
            assert np.array_equal(x[2][key], y[2][key]), key


def test_config2_full_size_properties(gpu_lib):
    """BASELINE config #2 at full size: size-independent properties + oracle agreement."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window()  # 10 KF / 2000 landmarks / 20000 residuals
    assert spec.P == 10 and spec.L == 2000 and spec.N == 20000
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    fg, lg = syn.feed(gpu, spec)
    fc, lc = syn.feed(cpu, spec)
    gpu.optimize(10)
    cpu.optimize(10)
    sg, sc = gpu.summary(), cpu.summary()
    log("config2 gpu", sg, "cpu", sc)
    assert sg["final_cost"] < sg["initial_cost"]
    assert sg["iterations"] == sc["iterations"]
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    err = max(np.linalg.norm(gpu.get_T_WS(a)[:3] - spec.T_WS_true[k, :3]) for k, a in enumerate(fg))
    log("config2 pose difference vs oracle", worst, "max position error vs truth", err)
    assert worst < 1e-4
    assert err < 0.05
    # idempotence: once converged, another optimize() does not move the states
    gpu.set_solver_options(1e-10, 1e-12, 1e-10)
    gpu.optimize(200)
    assert gpu.summary()["termination"] == 0
    T_before = np.stack([gpu.get_T_WS(a) for a in fg])
    gpu.optimize(5)
    T_after = np.stack([gpu.get_T_WS(a) for a in fg])
    assert np.max(np.abs(T_before - T_after)) < 1e-6


def test_wide_window_global_paths(gpu_lib):
    """16 keyframes: dC = 96 (global-atomic Schur accumulation) and d = 240 (global-memory Cholesky)."""
    spec = syn.make_window(P=16, L=600, n_obs=9000, seed=61, frame_dt=0.25)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    gpu.optimize(8)
    cpu.optimize(8)
    sg, sc = gpu.summary(), cpu.summary()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    log("wide window gpu", sg, "cpu", sc, "pose diff", worst)
    assert sg["iterations"] == sc["iterations"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert worst < 1e-4


@pytest.mark.parametrize("P,rig", [(12, "euroc"), (13, "euroc"), (14, "euroc"), (15, "euroc"), (17, "euroc"), (18, "euroc"),
                                   (8, "rig_v2"), (10, "rig_v2")])
def test_left_looking_lds_solver_sizes(gpu_lib, P, rig):
    """Reduced systems of 12..17 tile rows (176 < d <= 272: 12-18 keyframes with fixed extrinsics, 8-10 with per-frame
    extrinsics) go through the one-workgroup left-looking LDS Cholesky (k_chol_solve_ll): every tile-row count, whole
    optimisation against the oracle."""
    spec = syn.make_window(P=P, L=500, n_obs=6000, seed=100 + P, rig=rig, frame_dt=0.25)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    gpu.optimize(8)
    cpu.optimize(8)
    sg, sc = gpu.summary(), cpu.summary()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    log("left-looking solver P", P, rig, "gpu", sg, "cpu", sc, "pose diff", worst)
    assert sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
    assert sg["final_cost"] < sg["initial_cost"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert worst < 1e-4


@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_lds_resident_solver_sizes(gpu_lib, P):
    """Reduced systems of 2..11 tile rows (d = 15 P <= 176) go through the LDS-resident barrier-free Cholesky
    (k_chol_solve_lds): its row ownership changes with the tile-row count (one row per owner wave for the last six rows, the rows
    above them on the wave next to the chain's), so every count is run as a whole optimisation against the oracle."""
    spec = syn.make_window(P=P, L=400, n_obs=4000, seed=200 + P, rig="euroc", frame_dt=0.25)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    gpu.optimize(8)
    cpu.optimize(8)
    sg, sc = gpu.summary(), cpu.summary()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    log("LDS-resident solver P", P, "gpu", sg, "cpu", sc, "pose diff", worst)
    assert sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
    assert sg["final_cost"] < sg["initial_cost"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert worst < 1e-4


def test_one_workgroup_solvers_repeat_themselves(gpu_lib):
    """The two one-workgroup dense solvers hand tiles between their waves through LDS counters without fences (DS operations of
    a wave execute in order).  A hand-over that is wrong once in a thousand launches would show as a run that ends differently:
    25 repeated optimisations for every tile-row count (2..18 keyframes), all must reproduce the first
    (tools/solver_stress.py; 600 repetitions = 77 000 solver launches were run when the scheme went in)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "solver_stress.py"), "25"], capture_output=True, text=True, timeout=600)
    log(r.stdout[-600:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_landmark_sharded_solve_emulated_two_ranks(gpu_lib):
    """the landmark-sharded solve (2 ranks emulated by 2 threads on one GPU, all-reduce through a
    barrier) must reproduce the single-GPU solve."""
    import threading
    from svin_amd.estimator import Estimator
    from svin_amd import distributed as sd
    spec = syn.make_window(P=6, L=300, n_obs=3000, seed=71)
    ref = Estimator(0)
    f_ref, l_ref = syn.feed(ref, spec)
    ref.optimize(10)
    world = 2
    ar = sd.ThreadAllReduce(world)
    ests, frames, errs = [], [], []
    for r in range(world):
        e = Estimator(0)
        f, l = syn.feed(e, sd.shard_spec(spec, r, world))
        e.set_distributed(r, world, ar.callback(r))
        ests.append(e)
        frames.append(f)

    def run(r):
        try:
            ests[r].optimize(10)
        except Exception as ex:  # pragma: no cover
            errs.append(ex)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    s_ref = ref.summary()
    for r in range(world):
        s = ests[r].summary()
        worst = max(pose_diff(ests[r].get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(frames[r], f_ref))
        log("sharded rank", r, "summary", s, "pose diff vs single-GPU", worst)
        assert s["iterations"] == s_ref["iterations"]
        assert abs(s["final_cost"] - s_ref["final_cost"]) <= 1e-9 * s_ref["final_cost"]
        assert worst < 1e-9


def test_native_rccl_path_single_rank(gpu_lib, debug_option):
    """The sharded solve with RCCL called natively on the solver's stream (ncclAllReduce in place, scalars published to
    the mailbox after the reduction, speculative build kept).  One GPU cannot hold two RCCL ranks, so the code path is
    driven with a ONE-rank communicator: every collective really runs (and is the identity), the rest of the path is
    what 8 ranks execute."""
    from svin_amd.estimator import Estimator, rccl_unique_id
    spec = syn.make_window(P=6, L=300, n_obs=3000, seed=71)
    ref = Estimator(0)
    f_ref, _ = syn.feed(ref, spec)
    ref.optimize(10)
    est = Estimator(0)
    f, _ = syn.feed(est, spec)
    est.set_distributed_rccl(0, 1, rccl_unique_id())
    debug_option("SVIN_FORCE_DISTRIBUTED", 1)
    est.optimize(10)
    debug_option("SVIN_FORCE_DISTRIBUTED", 0)
    s, s_ref = est.summary(), ref.summary()
    worst = max(pose_diff(est.get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(f, f_ref))
    log("native RCCL, one rank: summary", s, "reference", s_ref, "pose diff", worst)
    assert s["iterations"] == s_ref["iterations"] and s["successful"] == s_ref["successful"]
    assert abs(s["final_cost"] - s_ref["final_cost"]) <= 1e-9 * s_ref["final_cost"]
    assert worst < 1e-9
    # a wide window through the same path (panel-pair Schur kernel + multi-workgroup Cholesky)
    spec = syn.make_window(P=48, L=900, n_obs=9000, seed=5)
    ref = Estimator(0)
    f_ref, _ = syn.feed(ref, spec)
    ref.optimize(6)
    est = Estimator(0)
    f, _ = syn.feed(est, spec)
    est.set_distributed_rccl(0, 1, rccl_unique_id())
    debug_option("SVIN_FORCE_DISTRIBUTED", 1)
    est.optimize(6)
    debug_option("SVIN_FORCE_DISTRIBUTED", 0)
    worst = max(pose_diff(est.get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(f, f_ref))
    assert est.summary()["iterations"] == ref.summary()["iterations"] and worst < 1e-8, worst


def test_native_rccl_path_full_config4_size(gpu_lib, debug_option):
    """BASELINE configs[3] at its full size (64 KF / 50 000 landmarks / 500 000 residuals, d = 960) through the sharded
    code path with a one-rank RCCL communicator (packed lower-triangle message, [group B | gathered maxima] message, stop
    vote, scalars published after the reduction): same iterates as the plain single-GPU solve.  Then the time-limit
    callback in sharded mode: the ranks stop on the all-reduced vote (termination 2 = USER_SUCCESS), after the minimum
    number of iterations."""
    from svin_amd.estimator import Estimator, rccl_unique_id
    spec = syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25)
    ref = Estimator(0)
    f_ref, _ = syn.feed(ref, spec)
    ref.optimize(3)
    est = Estimator(0)
    f, _ = syn.feed(est, spec)
    est.set_distributed_rccl(0, 1, rccl_unique_id())
    debug_option("SVIN_FORCE_DISTRIBUTED", 1)
    est.optimize(3)
    s, s_ref = est.summary(), ref.summary()
    worst = max(pose_diff(est.get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(f, f_ref))
    log("native RCCL, one rank, config #4 full size: summary", s, "reference", s_ref, "pose diff", worst)
    assert s["iterations"] == s_ref["iterations"] == 3 and s["successful"] == s_ref["successful"]
    assert abs(s["final_cost"] - s_ref["final_cost"]) <= 1e-9 * s_ref["final_cost"] and worst < 1e-8
    us = est.bench_allreduce(960 * 961 // 2 + 3 * 960, 5)
    assert 0.0 < us < 1e5
    # time limit: a limit that is already over when the first iteration ends -> stop after min_iterations = 2
    assert est.set_time_limit(1e-6, 2)
    est.optimize(10)
    s = est.summary()
    debug_option("SVIN_FORCE_DISTRIBUTED", 0)
    assert s["termination"] == 2 and 2 <= s["iterations"] <= 3, s


def drop_underdetermined_landmarks(spec):
    """removes the observations of landmarks seen fewer than three times: with mu = 0 their 3x3 block is singular or
    nearly so, and a linearisation compared at 1e-9 needs every landmark block to be invertible without damping"""
    cnt = np.bincount(spec.obs_lm, minlength=spec.L)
    keep = cnt[spec.obs_lm] >= 3
    for name in ("obs_lm", "obs_frame", "obs_cam", "obs_uv", "obs_size"):
        setattr(spec, name, getattr(spec, name)[keep])
    return spec


def test_wide_window_panels_against_oracle(gpu_lib):
    """config-#4 shape at a size the oracle finishes in seconds: 48 keyframes (dC = 288: three 96-row panels, six panel
    pairs in k_schur_panels) and d = 720 unknowns through the one-launch tile Cholesky (k_big_chol_chain) -- held
    against the ORACLE (reduced system and optimised states), not against the product's other kernel"""
    spec = drop_underdetermined_landmarks(syn.make_window(P=48, L=1500, n_obs=15000, seed=31, frame_dt=0.25))
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    lin_c = cpu.map().linearize(0.0)
    lin_g = gpu.linearize(0.0)
    assert lin_g["d"] == lin_c["d"] == 48 * 15
    perm = reduced_permutation(gpu, cpu, fg, fc, lin_g, lin_c)
    S, g = lin_g["S"][np.ix_(perm, perm)], lin_g["g"][perm]
    sd = np.sqrt(np.abs(np.diag(lin_c["S"])))
    dS, dg = rel(S / np.outer(sd, sd), lin_c["S"] / np.outer(sd, sd)), rel(g / sd, lin_c["g"] / sd)
    log("wide window (P = 48) reduced system vs oracle: dS", dS, "dg", dg)
    assert dS < 1e-9 and dg < 1e-9
    gpu.optimize(6)
    cpu.optimize(6)
    sg, sc = gpu.summary(), cpu.summary()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    log("wide window (P = 48) gpu", sg, "cpu", sc, "pose diff", worst)
    assert sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert worst < 1e-4


def test_wide_window_with_per_frame_extrinsics(gpu_lib):
    """config #4 with online extrinsics calibration: 64 keyframes x (6 pose + 2 x 6 extrinsics + 9 speed/bias) =
    1 728 unknowns -- beyond the former d ~ 1 100 guard: pairwise Schur accumulation with global atomics,
    twenty-seven 64-column panels in the tile Cholesky"""
    spec = syn.make_window(P=64, L=2500, n_obs=25000, seed=37, rig="test4", frame_dt=0.25)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    lin_g = gpu.linearize(0.0)
    assert lin_g["d"] == 64 * 27 == 1728
    gpu.optimize(5)
    cpu.optimize(5, 8)
    sg, sc = gpu.summary(), cpu.summary()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    worst_e = max(pose_diff(gpu.get_camera_sensor_states(a, c), cpu.get_camera_sensor_states(b, c))
                  for a, b in zip(fg, fc) for c in (0, 1))
    log("64 KF, per-frame extrinsics: d", lin_g["d"], "gpu", sg, "cpu", sc, "pose diff", worst, "extrinsics", worst_e)
    assert sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert worst < 1e-4 and worst_e < 1e-4


def test_config4_full_size_single_gpu(gpu_lib):
    """BASELINE config #4 at full size on ONE GPU (64 keyframes / 50 000 landmarks / 500 000 residuals, d = 960):
    size-independent properties, and the oracle's first linearisation + one iteration beside it"""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25)
    assert spec.P == 64 and spec.L == 50000 and spec.N == 500000
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    fg, lg = syn.feed(gpu, spec)
    fc, lc = syn.feed(cpu, spec)
    gpu.optimize(2)
    cpu.optimize(2, 8)        # 8 threads: the summation order differs from one thread at rounding level only
    sg, sc = gpu.summary(), cpu.summary()
    log("config4 gpu", sg, "cpu", sc)
    assert sg["iterations"] == sc["iterations"] == 2
    assert abs(sg["initial_cost"] - sc["initial_cost"]) <= 1e-9 * sc["initial_cost"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    log("config4 pose difference vs oracle after 2 iterations", worst)
    assert worst < 1e-6
    gpu.optimize(10)
    s = gpu.summary()
    err = max(np.linalg.norm(gpu.get_T_WS(a)[:3] - spec.T_WS_true[k, :3]) for k, a in enumerate(fg))
    log("config4 after 12 iterations", s, "max position error vs truth", err)
    assert s["final_cost"] <= sg["final_cost"] and err < 0.1


def test_keyframe_hand_off_matches_oracle(gpu_lib):
    """the estimator-side content of the keyframe message for pose_graph (ThreadedKFVio.cpp:1147-1240),
    on a window that has been optimised and marginalised (landmarks and observations come and go)"""
    spec = syn.make_window(P=12, L=300, n_obs=3000, seed=19, keyframe_every=2)
    from svin_amd.estimator import Estimator
    from oracle import orc
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    checked, worst = [0], [0.0, 0.0]

    def per_frame(est):
        def cb(k, fid):
            est.optimize(10, 1, False)
            if k >= 6:
                est.apply_marginalization(4, 2)
        return cb
    fg, _ = syn.feed(gpu, spec, on_frame=per_frame(gpu))
    fc, _ = syn.feed(cpu, spec, on_frame=per_frame(cpu))
    assert fg == fc
    for fid in fg[-6:]:
        for cam in (0, 1):
            a, b = gpu.keyframe_points(fid, cam), cpu.keyframe_points(fid, cam)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
            assert len(a[4]) == len(b[4]) and all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
            if len(a[0]):   # the payload are solver outputs: same tolerance as the marginalisation-sequence tests
                worst[0] = max(worst[0], rel(a[1], b[1]))
                worst[1] = max(worst[1], float(np.max(np.abs(a[3] - b[3]))))
            checked[0] += len(a[0])
    log("keyframe hand-off: %d points compared, worst relative point difference %.2e, quality %.2e" % (checked[0], worst[0], worst[1]))
    assert checked[0] > 100 and worst[0] < 1e-3 and worst[1] < 1e-2


def test_rejected_steps_follow_the_oracle(gpu_lib):
    """a badly perturbed start makes the trust region reject steps: the accept / reject sequence (and with it the
    discard path of the speculative build, Window::solve) must follow the oracle's exactly"""
    found = False
    for pose_noise, lm_noise, seed in (((0.6, 0.15), 1.5, 71), ((1.0, 0.25), 2.5, 72), ((1.5, 0.4), 4.0, 73)):
        spec = syn.make_window(P=6, L=250, n_obs=2500, seed=seed, pose_noise=pose_noise, lm_noise=lm_noise)
        gpu, cpu, fg, fc, lg, lc = make_pair(spec)
        gpu.optimize(25)
        cpu.optimize(25)
        sg, sc = gpu.summary(), cpu.summary()
        log("noise", pose_noise, lm_noise, "gpu", sg["iterations"], sg["successful"], sg["final_cost"], "oracle",
            sc["iterations"], sc["successful"], sc["final_cost"])
        assert sg["iterations"] == sc["iterations"] and sg["successful"] == sc["successful"]
        # (unconverged after 25 iterations from these starts: the two cost trajectories drift apart at the 1e-6 level)
        assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-5 * sc["final_cost"]
        found |= sg["successful"] < sg["iterations"]
    assert found, "none of the starts produced a rejected step: raise the perturbation"


def test_marginalization_large_prior_per_frame_extrinsics(gpu_lib):
    """rig v2 with the reference's window (5 keyframes + 3 IMU frames): the prior grows past 96 unknowns, where G and
    Q no longer fit one LDS together (k_marg_final: the Cholesky-preconditioned solve by default, SVIN_MARG_EIG for the others)"""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=13, L=250, n_obs=3000, seed=45, rig="rig_v2", keyframe_every=2, frame_dt=0.3)
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    for e in (gpu, cpu):
        e.set_solver_options(1e-12, 1e-12, 1e-12)
    fg, lg, rg = run_sequence(gpu, spec, 5, 3, 25)   # converged frames: b0 sits next to 1e16 prior entries
    fc, lc, rc = run_sequence(cpu, spec, 5, 3, 25)
    assert rg == rc and gpu.num_frames() == cpu.num_frames() and gpu.num_landmarks() == cpu.num_landmarks()
    assert fg == fc
    mg, mc = gpu.marg(), cpu.marg()
    assert mg is not None and mg["n"] == mc["n"] and mg["n"] > 96, mg["n"]
    o = compare_priors(mg, mc, "rig v2 5+3 sequence prior")
    log("prior cost offset |e0|^2 gpu", float(mg["e0"] @ mg["e0"]), "oracle", float(mc["e0"] @ mc["e0"]))
    assert o["selfH"] < 1e-9 and o["dH"] < 1e-2 and o["dJtJ"] < 1e-2     # the priors differ as the linearisation points do (bound derived below)
    gf, cf = gpu.frame_ids(), cpu.frame_ids()
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(gf, cf))
    # How far apart may two CORRECT double-precision implementations end on this sequence?  The oracle against itself,
    # its initial landmarks moved by 1e-13 m (a ten-thousandth of the last bit of a pixel): the 3e16 information of the
    # relative-extrinsics factors next to weakly determined directions amplifies that by ~1e9 over the thirteen frames.
    # One-shot marginalisations from identical states agree to 1e-10 (test_marginalization_one_shot), so the distance
    # below is this sensitivity, not a solver error; the GPU must stay within the same order.
    spec2 = syn.make_window(P=13, L=250, n_obs=3000, seed=45, rig="rig_v2", keyframe_every=2, frame_dt=0.3)
    spec2.lm_init[:, :3] += 1e-13 * np.random.default_rng(1).normal(size=(spec2.L, 3))
    cpu2 = orc.OracleEstimator()
    cpu2.set_solver_options(1e-12, 1e-12, 1e-12)
    run_sequence(cpu2, spec2, 5, 3, 25)
    sens = max(pose_diff(cpu2.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(cpu2.frame_ids(), cf))
    log("rig v2 5+3: final window pose difference GPU vs oracle", worst, "; oracle vs oracle with landmarks moved by 1e-13 m", sens)
    assert worst < max(1e-4, 30 * sens) and worst < 5e-3
    # ... and the same yardstick for the PRIOR: the perturbed oracle's last prior against the oracle's
    o2 = compare_priors(cpu2.marg(), mc, "rig v2 5+3 sequence prior, oracle with landmarks moved by 1e-13 m vs oracle")
    assert o["dH"] < max(1e-9, 30 * o2["dH"]) and o["dJtJ"] < max(1e-9, 30 * o2["dJtJ"]), (o["dH"], o2["dH"], o["dJtJ"], o2["dJtJ"])


def test_marginalization_sequence_euroc_reference_window(gpu_lib):
    """BASELINE configs[0]: the EuRoC constants (config_fpga_p2_euroc.yaml: fixed extrinsics, its IMU noise densities) with
    the window the reference ships for it -- numKeyframes 5, numImuFrames 3 (:55-56) -- over thirteen frames, optimize(10)
    per frame like ThreadedKFVio (max_num_iterations 10), marginalisation every frame.  Same removed-landmark lists, same
    frames, priors consistent, final window within the north star's 1e-4 of the oracle."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=13, L=300, n_obs=3500, seed=46, rig="euroc", keyframe_every=2, frame_dt=0.3)
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    fg, lg, rg = run_sequence(gpu, spec, 5, 3, 10)
    fc, lc, rc = run_sequence(cpu, spec, 5, 3, 10)
    log("euroc 5+3: removed landmarks per frame gpu", rg, "cpu", rc)
    assert rg == rc and fg == fc and sum(rg) > 0
    assert gpu.num_frames() == cpu.num_frames() <= 8 and gpu.num_landmarks() == cpu.num_landmarks()
    assert gpu.frame_ids() == cpu.frame_ids()
    assert [gpu.is_keyframe(f) for f in gpu.frame_ids()] == [cpu.is_keyframe(f) for f in cpu.frame_ids()]
    mg, mc = gpu.marg(), cpu.marg()
    assert mg is not None and mc is not None and mg["n"] == mc["n"]
    o = compare_priors(mg, mc, "euroc 5+3 sequence prior")
    assert o["selfH"] < 1e-9 and o["dH"] < 1e-2 and o["dJtJ"] < 1e-2
    # the bound above is a ceiling; the yardstick is how far the ORACLE's prior moves when its initial landmarks move by 1e-13 m
    # (thirteen optimise + marginalise steps amplify the linearisation points' rounding): the GPU stays within 30 x that
    spec2 = syn.make_window(P=13, L=300, n_obs=3500, seed=46, rig="euroc", keyframe_every=2, frame_dt=0.3)
    spec2.lm_init[:, :3] += 1e-13 * np.random.default_rng(1).normal(size=(spec2.L, 3))
    cpu2 = orc.OracleEstimator()
    run_sequence(cpu2, spec2, 5, 3, 10)
    o2 = compare_priors(cpu2.marg(), mc, "euroc 5+3 sequence prior, oracle with landmarks moved by 1e-13 m vs oracle")
    assert o["dH"] < max(1e-9, 30 * o2["dH"]) and o["dJtJ"] < max(1e-9, 30 * o2["dJtJ"]), (o["dH"], o2["dH"], o["dJtJ"], o2["dJtJ"])
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(gpu.frame_ids(), cpu.frame_ids()))
    worst_sb = max(float(np.max(np.abs(gpu.get_speed_and_bias(a) - cpu.get_speed_and_bias(a)))) for a in gpu.frame_ids() if gpu.is_in_imu_window(a))
    log("euroc 5+3: final window pose difference", worst, "speed/bias", worst_sb)
    assert worst < 1e-4 and worst_sb < 1e-4


@pytest.mark.parametrize("rig,window,P", [("euroc", (2, 3), 8), ("rig_v2", (5, 3), 13)])
def test_prior_eigen_solver_fallback_agrees(gpu_lib, debug_option, rig, window, P):
    """M3's four routes: by default a Cholesky factor when it can certify that the rank rule drops nothing (k_marg_final_chol),
    else the direct eigen-solve (SVIN_MARG_EIG=direct runs it for every prior: tridiagonalisation + divide and conquer, up to
    128 unknowns), behind it the Cholesky-preconditioned one-sided Jacobi of rounds 2-4 (SVIN_MARG_EIG=cholesky runs it alone;
    also the solver of larger priors) and ITS fall-back, the one-sided Jacobi on A itself (SVIN_MARG_EIG=jacobi).  All must hand
    the optimiser the same prior: J^T J, J^T e0 of the last prior of a sliding window, and the window it leads to."""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=P, L=250, n_obs=2500 if rig == "euroc" else 3000, seed=44 if rig == "euroc" else 45, rig=rig,
                           keyframe_every=2, frame_dt=0.3)
    out = {}
    for mode in ("default", "direct", "cholesky", "jacobi"):
        if mode == "default":
            debug_option("SVIN_MARG_EIG", None)
        else:
            debug_option("SVIN_MARG_EIG", mode)
        est = Estimator(0)
        est.set_solver_options(1e-12, 1e-12, 1e-12)
        f, l, removed = run_sequence(est, spec, window[0], window[1], 25)
        m = est.marg()
        assert m is not None
        out[mode] = dict(m=m, removed=removed, poses=[est.get_T_WS(a) for a in est.frame_ids()])
    debug_option("SVIN_MARG_EIG", None)
    for other in ("direct", "cholesky", "jacobi"):
        o = compare_priors(out[other]["m"], out["default"]["m"], "eigen-solver %s vs default (direct), %s" % (other, rig))
        worst = max(pose_diff(a, b) for a, b in zip(out[other]["poses"], out["default"]["poses"]))
        log(rig, "pose difference %s vs default" % other, worst)
        assert out[other]["removed"] == out["default"]["removed"]
        assert o["selfH"] < 1e-9
        assert o["dH"] < 1e-6 and o["dJtJ"] < 1e-6
        assert worst < (1e-4 if rig == "euroc" else 5e-3)


@pytest.mark.parametrize("rig", ["euroc", "rig_v2"])
def test_homogeneous_point_error_in_the_window(gpu_lib, rig):
    """U6: HomogeneousPointError residuals on landmarks of the window (svin_ba_add_homogeneous_point_error; the oracle adds
    the same term through its Map).  They ride as pseudo-observations of their landmark -- Schur elimination, cost,
    back-substitution and landmark quality see them like any residual of the block.  Checked: residual / Jacobian of the
    pseudo-observations, cost and iterates against the oracle, a prior-only landmark (no observations) being pulled onto
    its measurement (TestHomogeneousPointError.cpp:97-99: final cost < 1e-10 for that term), the graph queries, removal,
    and that marginalisation refuses to touch such landmarks."""
    from oracle import orc
    spec = syn.make_window(P=4, L=120, n_obs=1000, seed=71, rig=rig)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    rng = np.random.default_rng(3)
    L = orc.lib()
    cmap = cpu.map()
    pri = []
    for k in range(0, 40, 3):
        pg, pc = gpu.get_landmark(lg[k])["point"], cpu.get_landmark(lc[k])["point"]
        assert np.array_equal(pg, pc)
        meas = np.r_[pg[:3] + rng.normal(size=3) * 0.05, 1.0]
        var = float(rng.uniform(0.001, 0.05))
        rid = gpu.add_homogeneous_point_error(lg[k], meas, variance=var)
        assert rid != 0
        assert L.orc_map_add_hpoint_error(cmap.h, orc.dptr(orc.arr(meas)), var, lc[k]) != 0
        pri.append((k, rid, meas, var))
    # a landmark nobody observes, held only by its prior (GPU side only)
    far = gpu.new_id()
    target, start = np.array([2.0, -1.0, 4.0, 1.0]), np.array([2.5, -0.5, 5.0, 1.0])
    assert gpu.add_landmark(far, start)
    rid_far = gpu.add_homogeneous_point_error(far, target, variance=0.01)
    assert rid_far != 0
    ids, kind = gpu.parameters_of(rid_far)
    assert ids == [far] and kind == 102 and rid_far in gpu.residuals_of(far)
    assert gpu.add_homogeneous_point_error(123456789, target, variance=0.01) == 0          # unknown landmark
    assert gpu.add_homogeneous_point_error(far, target, information=-np.eye(3)) == 0        # not positive definite
    # residuals / Jacobians of the pseudo-observations at the initial state
    ev = gpu.eval_reprojection(robust=True)
    assert int(np.sum(ev["cam"] == 15)) == 2 * (len(pri) + 1)
    for k, rid, meas, var in pri:
        rows = [i for i in range(len(ev["r"])) if ev["res_id"][i] == rid]
        assert len(rows) == 2 and all(ev["lm_id"][i] == lg[k] for i in rows)
        e = gpu.get_landmark(lg[k])["point"][:3] - meas[:3]
        r3 = np.r_[ev["r"][rows[0]], ev["r"][rows[1]][:1]]
        assert np.max(np.abs(r3 - e / np.sqrt(var))) < 1e-12 and ev["r"][rows[1]][1] == 0.0
        assert np.max(np.abs(ev["Jl"][rows[0]] - np.eye(3)[:2] / np.sqrt(var))) < 1e-12
        assert np.max(np.abs(ev["Jl"][rows[1]] - np.array([[0, 0, 1 / np.sqrt(var)], [0, 0, 0]]))) < 1e-12
        assert np.all(ev["Jp"][rows[0]] == 0) and np.all(ev["Je"][rows[1]] == 0)
    for e in (gpu, cpu):
        e.set_solver_options(1e-12, 1e-12, 1e-12)
    gpu.optimize(30)
    cpu.optimize(30)
    sg, sc = gpu.summary(), cpu.summary()
    log(rig, "landmark priors: gpu", sg, "cpu", sc)
    # the far landmark's term, 0.5 |r|^2 at the start, is part of the GPU's initial cost only
    far_cost0 = 0.5 * float(np.sum((start[:3] - target[:3]) ** 2)) / 0.01
    assert abs((sg["initial_cost"] - far_cost0) - sc["initial_cost"]) <= 1e-9 * sc["initial_cost"]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    worst_lm = max(np.max(np.abs(gpu.get_landmark(a)["point"] - cpu.get_landmark(b)["point"])) for a, b in zip(lg, lc))
    worst_q = max(abs(gpu.get_landmark(a)["quality"] - cpu.get_landmark(b)["quality"]) for a, b in zip(lg, lc))
    log(rig, "pose", worst, "lm", worst_lm, "quality", worst_q)
    assert worst < 1e-6 and worst_lm < 1e-5 and worst_q < 1e-6
    assert np.max(np.abs(gpu.get_landmark(far)["point"][:3] - target[:3])) < 1e-6     # pulled onto its measurement
    # marginalisation refuses while a prior-carrying landmark is observed from a leaving frame ...
    with pytest.raises(RuntimeError):
        gpu.apply_marginalization(2, 1)
    # ... and works again once the priors are gone
    for k, rid, meas, var in pri:
        assert gpu.remove_homogeneous_point_error(rid)
    assert gpu.remove_homogeneous_point_error(rid_far) and not gpu.remove_homogeneous_point_error(rid_far)
    ok, removed = gpu.apply_marginalization(2, 1)
    assert ok
