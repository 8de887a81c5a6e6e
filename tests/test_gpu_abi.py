"""C-ABI members of okvis::Estimator beyond the solve itself, each driven through
libsvin_ba.so and held against the oracle's Estimator (or, where the oracle has no counterpart, against an identity the
reference's own arithmetic implies)."""
import ctypes as C

import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def pose_diff(Ta, Tb):
    dq = Ta[3:] - Tb[3:] * np.sign(Ta[3:] @ Tb[3:])
    return max(np.linalg.norm(Ta[:3] - Tb[:3]) / max(1.0, np.linalg.norm(Tb[:3])), np.linalg.norm(dq))


def make_pair(spec, **kw):
    from svin_amd.estimator import Estimator
    from oracle import orc
    gpu, cpu = Estimator(0), orc.OracleEstimator()
    fg, lg = syn.feed(gpu, spec, **kw)
    fc, lc = syn.feed(cpu, spec, **kw)
    return gpu, cpu, fg, fc, lg, lc


def test_remove_observation_both_forms_match_oracle(gpu_lib):
    """E4 Estimator::removeObservation (Estimator.cpp:432-474): by (landmark, pose, camera, keypoint) and by residual id"""
    spec = syn.make_window(P=5, L=150, n_obs=1500, seed=3)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    ev = gpu.eval_reprojection()
    n0 = len(ev["r"])
    assert n0 == spec.N
    # 1) by key: every third observation of frame 2, camera 0 (keypoint indices are assigned per (frame, camera) in feed())
    idx = np.nonzero((spec.obs_frame == 2) & (spec.obs_cam == 0))[0]
    removed = 0
    for kp, i in enumerate(idx):
        if kp % 3:
            continue
        a = gpu.remove_observation(lg[int(spec.obs_lm[i])], fg[2], 0, kp)
        b = cpu.remove_observation(lc[int(spec.obs_lm[i])], fc[2], 0, kp)
        assert a and b
        removed += 1
    # a second removal of the same observation is the reference's `false`
    assert not gpu.remove_observation(lg[int(spec.obs_lm[idx[0]])], fg[2], 0, 0)
    assert not cpu.remove_observation(lc[int(spec.obs_lm[idx[0]])], fc[2], 0, 0)
    # 2) by residual id: ids are handed out in insertion order on both sides
    rids = [int(r) for r in ev["res_id"][(ev["pose_id"] == fg[3])][:25]]
    for rid in rids:
        assert gpu.remove_observation_by_id(rid)
        assert cpu.L.orc_remove_observation_by_id(cpu.h, rid) == 1
        removed += 1
    assert not gpu.remove_observation_by_id(rids[0])
    assert len(gpu.eval_reprojection()["r"]) == n0 - removed
    assert sum(1 for r in cpu.map().residual_ids() if cpu.map().residual_kind(r) == 0) == n0 - removed
    assert gpu.get_landmark(lg[int(spec.obs_lm[idx[0]])])["n_obs"] == cpu.get_landmark(lc[int(spec.obs_lm[idx[0]])])["n_obs"]
    for e in (gpu, cpu):
        e.set_solver_options(1e-12, 1e-12, 1e-12)
        e.optimize(30)
    assert gpu.summary()["iterations"] == cpu.summary()["iterations"]
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    assert worst < 1e-8, worst


def test_time_limit_stops_after_minimum_iterations(gpu_lib):
    """E7 setOptimizationTimeLimit + CeresIterationCallback (Estimator.cpp:932-951, CeresIterationCallback.hpp:73-81)"""
    spec = syn.make_window(P=5, L=150, n_obs=1500, seed=4)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    for e in (gpu, cpu):
        assert e.set_time_limit(0.0, 3)      # zero budget: stop as soon as the minimum is reached
        e.optimize(20)
    sg, sc = gpu.summary(), cpu.summary()
    assert sg["iterations"] == 3 and sc["iterations"] == 3
    assert sg["termination"] == 2 and sc["termination"] == 2
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(b)) for a, b in zip(fg, fc))
    assert worst < 1e-9, worst
    # a negative limit with a callback registered raises the minimum to max_num_iterations: the limit is off (:934-938)
    for e in (gpu, cpu):
        assert e.set_time_limit(-1.0, 0)
        e.optimize(6)
    assert gpu.summary()["iterations"] == cpu.summary()["iterations"]
    assert gpu.summary()["termination"] == cpu.summary()["termination"] != 2
    # and a generous budget never triggers
    for e in (gpu, cpu):
        e.set_time_limit(1e6, 1)
        e.optimize(4)
    assert gpu.summary()["termination"] != 2 and gpu.summary()["iterations"] == cpu.summary()["iterations"]


def test_state_queries_match_oracle(gpu_lib):
    """E8: frameIdByAge, currentKeyframeId, currentFrameId, isKeyframe, isInImuWindow, timestamp, setKeyframe,
    get/setCameraSensorStates, stateCount_ -- on a window that has been marginalised (IMU window shorter than the window)"""
    spec = syn.make_window(P=9, L=200, n_obs=2000, seed=6, rig="test4", keyframe_every=2, frame_dt=0.3)
    from svin_amd.estimator import Estimator
    from oracle import orc
    gpu, cpu = Estimator(0), orc.OracleEstimator()

    def cb(est):
        def f(k, fid):
            est.optimize(4)
            est.apply_marginalization(3, 2)
        return f
    fg, _ = syn.feed(gpu, spec, on_frame=cb(gpu))
    fc, _ = syn.feed(cpu, spec, on_frame=cb(cpu))
    assert fg == fc
    L = cpu.L
    assert gpu.frame_ids() == cpu.frame_ids()
    assert gpu.num_frames() == cpu.num_frames()
    assert gpu.current_keyframe_id() == L.orc_current_keyframe_id(cpu.h)
    assert gpu.current_frame_id() == L.orc_current_frame_id(cpu.h)
    for age in range(gpu.num_frames()):
        assert gpu.frame_id_by_age(age) == L.orc_frame_id_by_age(cpu.h, age)
    imu_flags = []
    for fid in gpu.frame_ids():
        assert gpu.is_keyframe(fid) == bool(L.orc_is_keyframe(cpu.h, fid))
        assert gpu.is_in_imu_window(fid) == bool(L.orc_is_in_imu_window(cpu.h, fid))
        imu_flags.append(gpu.is_in_imu_window(fid))
        k = fg.index(fid)
        assert gpu.timestamp(fid) == (int(spec.stamps[k, 0]), int(spec.stamps[k, 1]))
    assert any(imu_flags) and not all(imu_flags), imu_flags     # keyframes outside the IMU window lost their speed/bias
    assert gpu.timestamp(12345678) is None
    assert gpu.state_count() == spec.P                          # counts addStates calls, never decremented (Estimator.hpp:450)
    # setKeyframe (Estimator.hpp:444)
    last = gpu.current_frame_id()
    was = gpu.is_keyframe(last)
    assert gpu.set_keyframe(last, not was) and gpu.is_keyframe(last) == (not was)
    assert gpu.set_keyframe(last, was)
    assert not gpu.set_keyframe(999999, True)
    # camera sensor states: the online-calibration rig gives every frame its own extrinsics blocks
    for fid in gpu.frame_ids():
        for cam in (0, 1):
            a, b = gpu.get_camera_sensor_states(fid, cam), cpu.get_camera_sensor_states(fid, cam)
            assert pose_diff(a, b) < 1e-4     # nine optimise + marginalise rounds behind them (north-star tolerance)
    T = gpu.get_camera_sensor_states(last, 1).copy()
    T[:3] += [0.01, -0.02, 0.005]
    T[3:] *= 2.0                                                 # setEstimate stores a Transformation: normalised
    assert gpu.set_camera_sensor_states(last, 1, T)
    back = gpu.get_camera_sensor_states(last, 1)
    assert np.allclose(back[:3], T[:3]) and abs(np.linalg.norm(back[3:]) - 1.0) < 1e-15
    assert not gpu.set_camera_sensor_states(last, 5, T)
    assert gpu.get_camera_sensor_states(424242, 0) is None


def test_landmark_initialized_flag_and_bulk_getter(gpu_lib):
    """E8: isLandmarkInitialized / setLandmarkInitialized (Estimator.cpp:966-969, :1126-1129), getLandmarks (:974-990)"""
    spec = syn.make_window(P=4, L=60, n_obs=500, seed=8)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    lid = lg[5]
    assert gpu.is_landmark_added(lid) and not gpu.is_landmark_added(987654321)
    assert gpu.is_landmark_initialized(lid)          # HomogeneousPointParameterBlock(point, id, initialized = true)
    assert gpu.get_landmark(lid)["initialized"]
    assert gpu.set_landmark_initialized(lid, False)
    assert not gpu.is_landmark_initialized(lid) and not gpu.get_landmark(lid)["initialized"]
    assert gpu.is_landmark_initialized(lg[6])
    assert not gpu.set_landmark_initialized(987654321, True)
    with pytest.raises(RuntimeError):
        gpu.is_landmark_initialized(987654321)
    gpu.optimize(5)
    cpu.optimize(5)
    assert not gpu.is_landmark_initialized(lid)      # the flag is the caller's, the solver leaves it alone
    all_lm = gpu.get_landmarks()
    assert list(all_lm.keys()) == sorted(lg) and len(all_lm) == gpu.num_landmarks() == cpu.num_landmarks()
    for a, b in zip(lg, lc):
        o = cpu.get_landmark(b)
        assert np.max(np.abs(all_lm[a]["point"] - o["point"])) < 1e-6
        assert all_lm[a]["n_obs"] == o["n_obs"] and abs(all_lm[a]["quality"] - o["quality"]) < 1e-6
        assert abs(all_lm[a]["distance"] - o["distance"]) <= 1e-14 * o["distance"]   # set once by addLandmark (:423-427)
    assert all_lm[lid]["initialized"] is False and all_lm[lg[6]]["initialized"] is True


def test_init_pose_from_imu_matches_oracle(gpu_lib):
    """static Estimator::initPoseFromImu (Estimator.cpp:848-873)"""
    from svin_amd.estimator import Estimator
    from oracle import orc
    gpu = Estimator(0)
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = 20
        t = np.stack([np.full(n, 100, np.uint32), (np.arange(n) * 5_000_000).astype(np.uint32)], 1)
        g = rng.normal(size=3)
        g = 9.81 * g / np.linalg.norm(g)
        m = np.zeros((n, 6))
        m[:, 3:] = g + 0.05 * rng.normal(size=(n, 3))
        ok, T = gpu.init_pose_from_imu(t, m)
        Tc = np.zeros(7)
        tc, mc = orc.arr(t, np.uint32), orc.arr(m)
        okc = orc.lib().orc_init_pose_from_imu(n, orc.u32ptr(tc), orc.dptr(mc), orc.dptr(Tc))
        assert ok and okc == 1
        assert np.max(np.abs(T - Tc)) < 1e-14, (T, Tc)
        # the pose turns the mean accelerometer reading into world +z
        R = syn.quat_to_R(T[3:])
        a = R @ m[:, 3:].mean(0)
        assert abs(a[0]) < 1e-9 and abs(a[1]) < 1e-9 and a[2] > 0
    ok, T = gpu.init_pose_from_imu(np.zeros((0, 2), np.uint32), np.zeros((0, 6)))
    assert not ok


def test_imu_preintegral_map_follows_add_states(gpu_lib):
    """getImuPreIntegral / setImuPreIntegral (Estimator.cpp:1001-1014, :1081-1087) filled by addStates (:146-165) from the
    second ImuError::propagation overload (ImuError.cpp:479-697).  The integrals must reproduce the prediction that the
    same call returned: r1 = r0 + v0 dt + C0 * acc_doubleintegral - g dt^2 / 2, v1 = v0 + C0 * acc_integral - g dt."""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=4, L=40, n_obs=300, seed=9)
    gpu = Estimator(0)
    fg, lg = syn.feed(gpu, spec, perturb=False)
    assert gpu.get_imu_preintegral(fg[0]) is None     # the first frame is not propagated
    g = spec.imu_params["g"]
    for k in range(1, spec.P):
        adi, ai, dt = gpu.get_imu_preintegral(fg[k])
        assert abs(dt - 0.5) < 1e-12
        # feed(perturb=False) leaves frame k-1 untouched between its own add_states / setters and frame k's add_states
        T0, sb0 = gpu.get_T_WS(fg[k - 1]), gpu.get_speed_and_bias(fg[k - 1])
        t0, t1 = tuple(int(v) for v in spec.stamps[k - 1]), tuple(int(v) for v in spec.stamps[k])
        n, T1, sb1, _, _, integ = gpu.imu_propagation(spec.imu_t, spec.imu_meas, spec.imu_params, T0, sb0, t0, t1,
                                                      want_integrals=True)
        assert np.max(np.abs(integ[:3] - adi)) < 1e-12 and np.max(np.abs(integ[3:6] - ai)) < 1e-12 and integ[6] == dt
        C0 = syn.quat_to_R(T0[3:])
        gW = np.array([0.0, 0.0, g])
        assert np.max(np.abs(T0[:3] + sb0[:3] * dt + C0 @ adi - 0.5 * gW * dt * dt - T1[:3])) < 1e-12
        assert np.max(np.abs(sb0[:3] + C0 @ ai - gW * dt - sb1[:3])) < 1e-12
    # setImuPreIntegral is std::map::insert: an existing entry wins (Estimator.cpp:1086)
    adi, ai, dt = gpu.get_imu_preintegral(fg[1])
    gpu.set_imu_preintegral(fg[1], [1, 2, 3], [4, 5, 6], 7.0)
    assert np.array_equal(gpu.get_imu_preintegral(fg[1])[0], adi)
    gpu.set_imu_preintegral(777, [1, 2, 3], [4, 5, 6], 7.0)
    a, b, c = gpu.get_imu_preintegral(777)
    assert list(a) == [1, 2, 3] and list(b) == [4, 5, 6] and c == 7.0


def test_one_id_space_with_caller_chosen_ids(gpu_lib):
    """Upstream every id (frames, landmarks, the estimator's own extrinsics / speed-bias blocks) comes from one
    process-wide IdProvider.  Caller-chosen ids 1..N used to collide with the core's private counter and silently
    overwrite blocks; now the core draws from the host's provider (or above reserve_ids) and refuses collisions."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=5, L=80, n_obs=800, seed=10, rig="test4")   # per-frame extrinsics: 3 internal ids per frame

    class Counter:     # the host's IdProvider
        def __init__(self, start=0):
            self.v = start

        def new_id(self):
            self.v += 1
            return self.v

    class Hosted:      # estimator wrapper whose new_id() is the host's provider
        def __init__(self, est, prov):
            self.est, self.prov = est, prov

        def new_id(self):
            return self.prov.new_id()

        def __getattr__(self, n):
            return getattr(self.est, n)

    ref = Estimator(0)
    f_ref, _ = syn.feed(ref, spec)
    ref.optimize(8)
    # (a) provider callback
    prov = Counter()
    est = Estimator(0)
    est.set_id_provider(prov.new_id)
    f_a, l_a = syn.feed(Hosted(est, prov), spec)
    assert f_a == f_ref                        # one counter: the same sequence as the built-in one
    est.optimize(8)
    # (per-frame extrinsics: the pose blocks of the camera system are accumulated with LDS atomics -> run-to-run rounding)
    assert max(pose_diff(est.get_T_WS(a), ref.get_T_WS(b)) for a, b in zip(f_a, f_ref)) < 1e-9
    # (b) a provider that hands out an id that is already a landmark id: the host's error -- add_states refuses and leaves the
    # window (frames, state count, pre-integrals) as it was
    est = Estimator(0)
    bad = Counter()
    est.set_id_provider(bad.new_id)
    for cam in spec.cameras:
        est.add_camera(cam["model"], cam["intr"], cam["dist"], cam["width"], cam["height"], spec.extr_sigmas)
    est.add_imu(spec.imu_params)
    for l in range(spec.L):
        assert est.add_landmark(1 + l, spec.lm_init[l])
    T_SC = np.stack([c["T_SC"] for c in spec.cameras])
    fid = spec.L + 1
    sel = slice(0, 12)
    with pytest.raises(RuntimeError, match="id"):
        est.add_states(fid, (int(spec.stamps[0, 0]), int(spec.stamps[0, 1])), 400, T_SC, spec.imu_t[sel], spec.imu_meas[sel], True)
    assert est.num_frames() == 0 and est.state_count() == 0
    # (c) no provider, caller ids 1..N: the built-in counter skips every id that is in use (no reserve_ids needed; it only
    # saves the skipping)
    est = Estimator(0)
    for cam in spec.cameras:
        est.add_camera(cam["model"], cam["intr"], cam["dist"], cam["width"], cam["height"], spec.extr_sigmas)
    est.add_imu(spec.imu_params)
    for l in range(spec.L):
        assert est.add_landmark(1 + l, spec.lm_init[l])
    assert est.add_states(fid, (int(spec.stamps[0, 0]), int(spec.stamps[0, 1])), 400, T_SC, spec.imu_t[sel], spec.imu_meas[sel], True)
    assert est.num_frames() == 1 and est.state_count() == 1
    assert est.get_landmark(1)["point"][3] == spec.lm_init[0][3] and est.get_T_WS(fid) is not None
    assert est.new_id() == fid + 4             # 2 extrinsics + 1 speed/bias were drawn above the reservation
    # a frame id that is already a landmark id is the reference's `false` (Map::addParameterBlock refuses it)
    assert not est.add_states(3, (int(spec.stamps[1, 0]), int(spec.stamps[1, 1])), 400, T_SC, spec.imu_t[:120], spec.imu_meas[:120], True)
    assert not est.add_landmark(fid, spec.lm_init[0])
    del prov, orc


def test_add_states_false_returns_and_null_arguments(gpu_lib):
    """addStates returns false for: no IMU sample on the first frame (:110-113), <= 10 keypoints on the first frame
    (:116-122), an IMU deque that ends before the frame (:159-162), a reused frame id.  NULL pointers are argument
    errors, not crashes."""
    from svin_amd.estimator import Estimator
    from oracle import orc
    spec = syn.make_window(P=3, L=30, n_obs=200, seed=12)
    T_SC = np.stack([c["T_SC"] for c in spec.cameras])

    def fresh(cls):
        e = cls() if cls is orc.OracleEstimator else cls(0)
        for cam in spec.cameras:
            e.add_camera(cam["model"], cam["intr"], cam["dist"], cam["width"], cam["height"], spec.extr_sigmas)
        e.add_imu(spec.imu_params)
        return e
    st0, st1 = (int(spec.stamps[0, 0]), int(spec.stamps[0, 1])), (int(spec.stamps[1, 0]), int(spec.stamps[1, 1]))
    for cls in (Estimator, orc.OracleEstimator):
        e = fresh(cls)
        assert not e.add_states(100, st0, 400, T_SC, np.zeros((0, 2), np.uint32), np.zeros((0, 6)), True)
        assert not e.add_states(100, st0, 10, T_SC, spec.imu_t[:10], spec.imu_meas[:10], True)
        assert e.num_frames() == 0
        assert e.add_states(100, st0, 11, T_SC, spec.imu_t[:10], spec.imu_meas[:10], True)
        assert not e.add_states(101, st1, 400, T_SC, spec.imu_t[:50], spec.imu_meas[:50], True)   # deque ends at 0.24 s < 0.5 s
        assert e.num_frames() == 1
        assert e.add_states(101, st1, 400, T_SC, spec.imu_t[:120], spec.imu_meas[:120], True)
        assert e.num_frames() == 2
    gpu = fresh(Estimator)
    L, h = gpu.L, gpu.h
    assert L.svin_ba_add_states(h, 5, 1, 0, 400, None, 2, None, 5, 1, None, 0, None, 0, 0.0) == -1     # NULL imu with n_imu > 0
    s = (C.c_double * 7)()
    assert L.svin_ba_add_states(h, 5, 1, 0, 400, None, 2, None, 0, 1, None, 0, None, 0, 0.0) == -1     # NULL T_SC
    n = C.c_int()
    assert L.svin_ba_apply_marginalization_strategy(h, 2, 2, None, 16, C.byref(n)) == -1               # NULL buffer, cap > 0
    assert L.svin_ba_get_T_WS(None, 1, s) == -1
    assert L.svin_ba_init_pose_from_imu(None, 3, s) == -1
    # duplicate observation -> 0 like the reference's NULL (implementation/Estimator.hpp:55-57)
    gpu = fresh(Estimator)
    cpu = fresh(orc.OracleEstimator)
    for e in (gpu, cpu):
        assert e.add_states(100, st0, 400, T_SC, spec.imu_t[:10], spec.imu_meas[:10], True)
        assert e.add_landmark(7, spec.lm_init[0])
        a = e.add_observation(7, 100, 0, 3, [100.0, 120.0], 8.0)
        assert a != 0 and e.add_observation(7, 100, 0, 3, [100.0, 120.0], 8.0) == 0
        assert e.add_observation(7, 100, 1, 3, [100.0, 120.0], 8.0) != 0      # another camera is another observation
        assert e.add_observation(8, 100, 0, 4, [100.0, 120.0], 8.0) == 0      # unknown landmark


def test_map_graph_queries_match_oracle(gpu_lib):
    """okvis::ceres::Map surface: parameterBlockExists, residuals(id), parameters(residual),
    ParameterBlock::fixed(), setParameterBlockConstant / Variable (Map.cpp:495-620) -- answered from the core's graph,
    compared with the oracle's Map entry by entry, before and after a marginalisation."""
    spec = syn.make_window(P=6, L=300, n_obs=2500, seed=14, rig="rig_v2", sonar=True, depth=True, keyframe_every=2)
    from svin_amd.estimator import Estimator
    from oracle import orc
    gpu, cpu = Estimator(0), orc.OracleEstimator()

    def cb(est):
        def f(k, fid):
            if k == 5:
                est.optimize(5)
                est.apply_marginalization(2, 2)
        return f
    fg, lg = syn.feed(gpu, spec, on_frame=cb(gpu))
    fc, lc = syn.feed(cpu, spec, on_frame=cb(cpu))
    assert fg == fc and lg == lc
    m = cpu.map()
    kinds_seen = set()
    n_res = 0
    blocks = set()
    for fid in gpu.frame_ids():
        ids = [fid]
        # the extrinsics / speed-bias block ids of the frame, found through the factors that touch the pose
        for rid in gpu.residuals_of(fid):
            ps, kind = gpu.parameters_of(rid)
            kinds_seen.add(kind)
            assert ps == m.parameters_of(rid), (rid, kind, ps, m.parameters_of(rid))
            ids.extend(ps)
            n_res += 1
        for bid in set(ids):
            for rid in gpu.residuals_of(bid):      # relative-extrinsics factors only touch extrinsics blocks
                kinds_seen.add(gpu.parameters_of(rid)[1])
            blocks.add(bid)
            assert gpu.parameter_block_exists(bid)
            assert gpu.residuals_of(bid) == sorted(m.residuals_of(bid)), bid
            assert gpu.is_parameter_block_constant(bid) == m.is_constant(bid), bid
    assert {0, 3, 4, 5, 100, 101}.issubset(kinds_seen), kinds_seen      # imu, relative pose, sonar, depth, reprojection, prior
    assert n_res > 500 and len(blocks) > 100
    assert not gpu.parameter_block_exists(987654321)
    with pytest.raises(RuntimeError):
        gpu.residuals_of(987654321)
    # ErrorInterface sizes of a whole residual list in one call (svin_ba_residual_info): against the oracle's error terms
    some = sorted(gpu.residuals_of(gpu.frame_ids()[1]))[:400]
    for rid, (kind, mdim, dims) in zip(some, gpu.residual_info(some)):
        pars, k2 = gpu.parameters_of(rid)
        assert kind == k2
        if kind != 101:
            assert mdim == {100: 2, 102: 3, 0: 15, 1: 6, 2: 9, 3: 6, 4: 1, 5: 1}[kind] and len(dims) == len(pars)
            assert dims == [len(gpu.parameter_block(b)["values"]) for b in pars]
    assert gpu.residual_info([987654321])[0][0] == -1
    lid = gpu.landmark_ids()[3]
    assert gpu.residuals_of(lid) == sorted(m.residuals_of(lid)) and len(gpu.residuals_of(lid)) == gpu.get_landmark(lid)["n_obs"]
    # hold one landmark and one pose constant on both sides: they must not move, everything else follows the oracle
    assert gpu.set_parameter_block_constant(lid, True) and gpu.is_parameter_block_constant(lid)
    assert cpu.L.orc_map_set_constant(m.h, lid, 1)
    lm_before = gpu.get_landmark(lid)["point"].copy()
    hold = gpu.frame_ids()[-2]
    assert gpu.set_parameter_block_constant(hold, True) and gpu.is_parameter_block_constant(hold)
    assert cpu.L.orc_map_set_constant(m.h, hold, 1)
    before, before_c = gpu.get_T_WS(hold).copy(), cpu.get_T_WS(hold).copy()
    for e in (gpu, cpu):
        e.optimize(6)
    assert np.array_equal(gpu.get_T_WS(hold), before) and np.array_equal(cpu.get_T_WS(hold), before_c)
    assert np.array_equal(gpu.get_landmark(lid)["point"], lm_before)
    assert gpu.set_parameter_block_constant(lid, False) and not gpu.is_parameter_block_constant(lid)
    cpu.L.orc_map_set_constant(m.h, lid, 0)
    assert gpu.summary()["iterations"] == cpu.summary()["iterations"]
    worst = max(pose_diff(gpu.get_T_WS(a), cpu.get_T_WS(a)) for a in gpu.frame_ids())
    assert worst < 1e-4, worst
    assert gpu.set_parameter_block_constant(hold, False) and not gpu.is_parameter_block_constant(hold)
    gpu.optimize(2)
    assert not np.array_equal(gpu.get_T_WS(hold), before)


def test_host_evaluators_match_device(gpu_lib):
    """the CPU twins the shim's ImuError / ReprojectionError classes forward to, against the device
    versions of the same arithmetic (k_imu_propagation, k_eval_reproj)"""
    from svin_amd import estimator
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=4, L=150, n_obs=1200, seed=13)
    gpu = Estimator(0)
    fg, lg = syn.feed(gpu, spec)
    T0, sb0 = spec.T_WS_true[0].copy(), spec.sb_true[0].copy()
    sb0[3:] = [0.01, -0.02, 0.005, 0.05, -0.03, 0.02]
    t0, t1 = tuple(int(v) for v in spec.stamps[0]), tuple(int(v) for v in spec.stamps[2])
    n, T, sb, cov, jac, integ = gpu.imu_propagation(spec.imu_t, spec.imu_meas, spec.imu_params, T0, sb0, t0, t1, True, True, want_integrals=True)
    nh, Th, sbh, covh, jach, integh = estimator.host_imu_propagation(spec.imu_t, spec.imu_meas, spec.imu_params, T0, sb0, t0, t1, True, True)
    assert n == nh
    assert np.max(np.abs(T - Th)) < 1e-11 and np.max(np.abs(sb - sbh)) < 1e-11 and np.max(np.abs(integ - integh)) < 1e-11
    assert np.max(np.abs(cov - covh)) <= 1e-9 * np.max(np.abs(covh)) and np.max(np.abs(jac - jach)) <= 1e-10 * np.max(np.abs(jach))
    ev = gpu.eval_reprojection(robust=False)
    ids = {fid: k for k, fid in enumerate(fg)}
    worst = 0.0
    for i in range(0, len(ev["r"]), 7):
        k, c = ids[int(ev["pose_id"][i])], int(ev["cam"][i])
        cam = spec.cameras[c]
        hp = gpu.get_landmark(int(ev["lm_id"][i]))["point"]
        size = 8.0
        # the uv of this observation: found through its residual id order = insertion order per frame
        o = estimator.host_reprojection_error(cam["model"], cam["intr"], cam["dist"], gpu.get_T_WS(fg[k]), hp, gpu.get_camera_sensor_states(fg[k], c),
                                              [0.0, 0.0], np.eye(2) * 64.0 / size ** 2)
        # measurement-independent parts: the Jacobians; the residual differs by w * uv
        for key in ("Jp", "Jl", "Je"):
            worst = max(worst, float(np.max(np.abs(o[key] - ev[key][i])) / max(1.0, np.max(np.abs(ev[key][i])))))
    assert worst < 1e-12, worst


def test_parameter_block_snapshots_and_batched_landmark_readback(gpu_lib):
    """Map::parameterBlockPtr / id2parameterBlockMap as values (Map.hpp:166-188, svin_ba_get_parameter_block / _ids) against the
    oracle's Map parameter by parameter after an optimisation, and svin_ba_get_all_landmark_observations (the PointMap with
    its observation maps in one call, Estimator.cpp:974-990) against the per-landmark getters."""
    spec = syn.make_window(P=5, L=200, n_obs=1800, seed=21, rig="rig_v2", keyframe_every=2)
    gpu, cpu, fg, fc, lg, lc = make_pair(spec)
    for e in (gpu, cpu):
        e.optimize(4)
    m = cpu.map()
    ids = gpu.parameter_block_ids()
    assert ids == sorted(ids) and set(lg).issubset(ids) and set(fg).issubset(ids)
    n_by_type = {0: 0, 1: 0, 2: 0, 3: 0}
    for bid in ids:
        b = gpu.parameter_block(bid)
        n_by_type[b["type"]] += 1
        ref = m.get_param(bid)
        assert len(ref) == len(b["values"]) == {0: 7, 1: 7, 2: 9, 3: 4}[b["type"]]
        if b["type"] == 3:
            np.testing.assert_allclose(b["values"], ref, rtol=1e-6, atol=1e-7)
            assert b["initialized"] and not b["fixed"]
        else:
            assert np.abs(b["values"] - ref).max() < 1e-6, (bid, b["values"], ref)
            assert b["fixed"] == m.is_constant(bid)
    assert n_by_type[0] == len(fg) and n_by_type[2] == len(fg) and n_by_type[1] >= 2 and n_by_type[3] == len(lg)
    # poses carry their frame's time stamp and equal get_T_WS bit for bit
    for k, fid in enumerate(fg):
        b = gpu.parameter_block(fid)
        assert b["type"] == 0 and np.array_equal(b["values"], gpu.get_T_WS(fid)) and b["stamp"] == tuple(int(v) for v in spec.stamps[k])
    with pytest.raises(RuntimeError):
        gpu.parameter_block(123456789)
    # batched landmark read-back == the per-landmark calls
    allobs = gpu.all_landmark_observations()
    lms = gpu.get_landmarks()
    assert list(allobs.keys()) == list(lms.keys()) and len(allobs) == len(lg)
    total = 0
    for lid, (info, obs) in allobs.items():
        assert obs == gpu.landmark_observations(lid)
        assert np.array_equal(info["point"], lms[lid]["point"]) and info["n_obs"] == len(obs) == lms[lid]["n_obs"]
        assert info["quality"] == lms[lid]["quality"]
        total += len(obs)
    assert total == len(spec.obs) if hasattr(spec, "obs") else total > 1000


def test_wait_idle_after_an_enqueued_marginalisation(gpu_lib):
    """svin_ba_apply_marginalization_strategy returns once its device job is enqueued; svin_ba_wait_idle returns when it has run:
    the next optimize() then starts on an idle stream (its solve_time no longer holds the job), results unchanged"""
    import time
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=14, L=600, n_obs=6000, seed=23, keyframe_every=2, frame_dt=0.25)
    out = {}
    for spaced in (False, True):
        est, solve = Estimator(0), []

        def on_frame(k, fid):
            if spaced:
                est.wait_idle()
            est.optimize(4)
            solve.append(est.summary()["solve_time"])
            est.apply_marginalization(5, 3)
        frames, _ = syn.feed(est, spec, on_frame=on_frame)
        t0 = time.perf_counter()
        est.wait_idle()
        est.wait_idle()        # idempotent, immediate the second time
        assert time.perf_counter() - t0 < 1.0
        out[spaced] = (np.array([est.get_T_WS(f) for f in frames[-8:] if est.get_T_WS(f) is not None]), np.median(solve[5:]))
    assert np.array_equal(out[False][0], out[True][0])
    print("median solve_time back to back %.3f ms, frames spaced %.3f ms" % (1e3 * out[False][1], 1e3 * out[True][1]))
    assert out[True][1] <= out[False][1] * 1.05
