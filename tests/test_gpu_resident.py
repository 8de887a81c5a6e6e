"""The device-resident window (svin_amd/csrc/resident.hpp) against the host re-pack it replaces.

Two product estimators are driven with identical calls: one keeps its observation CSR on the device and patches it with
each frame's delta (pack mode 0), the other re-packs the whole graph on the host and uploads it (pack mode 1, the path of
rounds 1-3).  Integer / index work is held to bit-exactness (the CSR, its slot indices, the per-chunk pose order); so
are the uv / weight / landmark values, which are copies.  Because the tables are identical and every kernel downstream
is deterministic, the optimised states and the marginalisation priors must come out identical too.
Reference behaviour being replaced: okvis_ceres/src/Map.cpp:341-492 (add / remove residual blocks) and the landmark loop of
Estimator::applyMarginalizationStrategy (src/Estimator.cpp:671-766).
"""
import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def assert_same_csr(a, b, tag):
    assert a["resident"] and not b["resident"], tag
    assert (a["L"], a["N"]) == (b["L"], b["N"]), tag
    for k in ("lm_ptr", "obs_lm", "obs_idx", "obs_order"):
        assert np.array_equal(a[k], b[k]), "%s: %s differs" % (tag, k)
    for k in ("uv", "w", "lm"):
        assert np.array_equal(a[k], b[k]), "%s: %s differs (max %g)" % (tag, k, float(np.max(np.abs(a[k] - b[k]))))


class Driver:
    """feeds a WindowSpec frame by frame into several estimators at once, with extra graph edits in between"""

    def __init__(self, ests, spec, seed):
        self.ests, self.spec = ests, spec
        self.rng = np.random.default_rng(seed)
        for e in ests:
            for cam in spec.cameras:
                e.add_camera(cam["model"], cam["intr"], cam["dist"], cam["width"], cam["height"], spec.extr_sigmas)
            e.add_imu(spec.imu_params)
            if spec.T_SSo is not None:
                e.set_sonar_extrinsics(spec.T_SSo)
        self.T_SC = np.stack([c["T_SC"] for c in spec.cameras])
        self.lm_ids = {}       # landmark index -> id (the same in every estimator: one id sequence each)
        self.frame_ids = []
        self.live = []         # (landmark index, frame id, cam, kp, residual ids per estimator)
        imu_t, stamps = spec.imu_t, spec.stamps
        self.imu_sec = imu_t[:, 0].astype(np.float64) - float(imu_t[0, 0]) + 1e-9 * imu_t[:, 1]
        self.frm_sec = stamps[:, 0].astype(np.float64) - float(imu_t[0, 0]) + 1e-9 * stamps[:, 1]

    def all(self, fn):
        out = [fn(e) for e in self.ests]
        assert all(o == out[0] for o in out[1:]), out
        return out[0]

    def add_frame(self, k, lazy_landmarks=True, batched=True):
        spec = self.spec
        fid = self.all(lambda e: e.new_id())
        self.frame_ids.append(fid)
        margin = 2.5 / spec.imu_params["rate"]
        lo = self.frm_sec[k - 1] - margin if k > 0 else self.frm_sec[0] - margin
        sel = (self.imu_sec >= lo) & (self.imu_sec <= self.frm_sec[k] + margin)
        son = [spec.sonar[k]] if spec.sonar and spec.sonar[k] is not None else None
        dep = [spec.depth[k]] if spec.depth and spec.depth[k] is not None else None
        for e in self.ests:
            assert e.add_states(fid, (int(spec.stamps[k, 0]), int(spec.stamps[k, 1])), 400, self.T_SC, spec.imu_t[sel],
                                spec.imu_meas[sel], bool(spec.keyframe[k]), son, dep, spec.first_depth)
            if k > 0:
                e.set_T_WS(fid, spec.T_WS_init[k])
            e.set_speed_and_bias(fid, spec.sb_init[k])
        idx = np.nonzero(spec.obs_frame == k)[0]
        kp_next = {}
        rows = []
        for i in idx:
            l, c = int(spec.obs_lm[i]), int(spec.obs_cam[i])
            if l not in self.lm_ids:   # the frontend creates a landmark together with its first observations
                lid = self.all(lambda e: e.new_id())
                self.lm_ids[l] = lid
                for e in self.ests:
                    assert e.add_landmark(lid, spec.lm_init[l])
            kp = kp_next.get(c, 0)
            kp_next[c] = kp + 1
            rows.append((l, c, kp, i))
        if batched and rows:
            lids = np.array([self.lm_ids[r[0]] for r in rows], np.uint64)
            cams = np.array([r[1] for r in rows], np.uint64)
            kps = np.array([r[2] for r in rows], np.uint64)
            sel_i = np.array([r[3] for r in rows])
            rids = [e.add_observations(lids, np.full(len(rows), fid, np.uint64), cams, kps, spec.obs_uv[sel_i], spec.obs_size[sel_i])
                    for e in self.ests]
            for j, r in enumerate(rows):
                if all(x[j] != 0 for x in rids):
                    self.live.append((r[0], fid, r[1], r[2], [int(x[j]) for x in rids]))
        else:
            for (l, c, kp, i) in rows:
                rid = [e.add_observation(self.lm_ids[l], fid, c, kp, spec.obs_uv[i], float(spec.obs_size[i])) for e in self.ests]
                if all(x != 0 for x in rid):
                    self.live.append((l, fid, c, kp, rid))
                else:
                    assert all(x == 0 for x in rid)
        return fid

    def remove_some(self, n, newest_only=False):
        """the frontend's outlier rejection: remove observations again, by key and by residual id"""
        alive = set(self.ests[0].frame_ids())
        self.live = [r for r in self.live if r[1] in alive]
        pool = [r for r in self.live if (not newest_only or r[1] == self.frame_ids[-1])]
        for _ in range(min(n, len(pool))):
            r = pool.pop(int(self.rng.integers(len(pool))))
            self.live.remove(r)
            if self.rng.random() < 0.5:
                res = [e.remove_observation(self.lm_ids[r[0]], r[1], r[2], r[3]) for e in self.ests]
            else:
                res = [e.remove_observation_by_id(rid) for e, rid in zip(self.ests, r[4])]
            assert all(x == res[0] for x in res), res


def same_states(ests, tag, tol=0.0):
    a, b = ests
    fa, fb = a.frame_ids(), b.frame_ids()
    assert fa == fb, tag
    for f in fa:
        pairs = [(a.get_T_WS(f), b.get_T_WS(f)), (a.get_speed_and_bias(f), b.get_speed_and_bias(f))]
        pairs += [(a.get_camera_sensor_states(f, c), b.get_camera_sensor_states(f, c)) for c in range(2)]
        for va, vb in pairs:
            assert (va is None) == (vb is None), tag   # (a frame outside the IMU window has no speed / bias block any more)
            if va is not None:
                assert np.max(np.abs(va - vb)) <= tol, "%s: frame %d differs by %g" % (tag, f, float(np.max(np.abs(va - vb))))


@pytest.mark.parametrize("rig", ["euroc", "rig_v2", "rig_v2_sonar_depth"])   # rig_v2: per-frame extrinsics -> the per-chunk pose order is built on
def test_resident_csr_equals_host_rebuild_for_50_frames(gpu_lib, rig):           # the device; sonar: addStates reads the landmarks every frame (lazy fetch)
    """add / remove / set / optimise / marginalise interleaved for 50 frames: device CSR == host rebuild, every frame"""
    from svin_amd.estimator import Estimator
    kw = dict(sonar=True, depth=True) if rig == "rig_v2_sonar_depth" else {}
    spec = syn.make_window(P=50 if not kw else 30, L=1500, n_obs=22000, seed=31, rig=rig.split("_sonar")[0], keyframe_every=3, frame_dt=0.2, **kw)
    a, b = Estimator(0), Estimator(0)
    b.set_pack_mode(1)
    drv = Driver([a, b], spec, seed=5)
    exact = rig == "euroc"
    removed_total = 0
    for k in range(spec.P):
        drv.add_frame(k, batched=(k % 3 != 1))
        drv.remove_some(7, newest_only=True)        # rejected before they ever reached the device
        if k % 2 == 0:
            drv.remove_some(5)                       # older ones: tombstones in the device CSR
        if k % 5 == 2:                               # Estimator::setLandmark on landmarks of the window
            known = [l for l in drv.lm_ids if all(e.is_landmark_added(drv.lm_ids[l]) for e in drv.ests)]
            for l in known[:: max(1, len(known) // 4)][:4]:
                hp = a.get_landmark(drv.lm_ids[l])["point"] + np.array([1e-3, -2e-3, 1e-3, 0.0])
                for e in drv.ests:
                    assert e.set_landmark(drv.lm_ids[l], hp)
        ca, cb = a.debug_csr(), b.debug_csr()
        assert_same_csr(ca, cb, "%s frame %d" % (rig, k))
        for e in drv.ests:
            e.optimize(4)
        if exact:
            same_states(drv.ests, "%s frame %d after optimize" % (rig, k))
        else:
            # the Schur kernel of windows with variable extrinsics sums its pose / extrinsics blocks with LDS atomics: two runs
            # of the SAME estimator agree to rounding only, and sigma_c_relative = 1e-8 (3e16 of information between consecutive
            # extrinsics) amplifies that.  The follower is put onto the leader's states after every solve, so the tables of
            # the next frame are comparable bit by bit again.
            same_states(drv.ests, "%s frame %d after optimize" % (rig, k), tol=1e-6)
            for f in a.frame_ids():
                if a.get_T_WS(f) is not None:
                    b.set_T_WS(f, a.get_T_WS(f))
                if a.get_speed_and_bias(f) is not None:
                    b.set_speed_and_bias(f, a.get_speed_and_bias(f))
                for c in range(2):
                    if a.get_camera_sensor_states(f, c) is not None:
                        b.set_camera_sensor_states(f, c, a.get_camera_sensor_states(f, c))
            for i, v in a.get_landmarks().items():
                b.set_landmark(i, v["point"])
        if k % 7 == 3:   # a landmark read-back in the middle (the lazy fetch) must not disturb anything
            la, lb = a.get_landmarks(), b.get_landmarks()
            assert la.keys() == lb.keys()
            for i in la:
                assert np.array_equal(la[i]["point"], lb[i]["point"]), (k, i)
                assert la[i]["quality"] == lb[i]["quality"] or (not exact and abs(la[i]["quality"] - lb[i]["quality"]) < 1e-6), (k, i)
        ra = [e.apply_marginalization(4, 3) for e in drv.ests]
        assert ra[0][0] and ra[1][0] and list(ra[0][1]) == list(ra[1][1]), "%s frame %d: removed landmarks differ" % (rig, k)
        removed_total += len(ra[0][1])
    assert removed_total > (100 if not kw else 40)
    ma, mb = a.marg(), b.marg()
    assert (ma is None) == (mb is None)
    if ma is not None:
        assert [x["id"] for x in ma["blocks"]] == [x["id"] for x in mb["blocks"]]
        for key in ("H", "b0"):
            if exact:
                assert np.array_equal(ma[key], mb[key]), "prior %s differs by %g" % (key, float(np.max(np.abs(ma[key] - mb[key]))))
            else:
                assert np.max(np.abs(ma[key] - mb[key])) < 1e-6 * np.max(np.abs(mb[key]))
    la, lb = a.get_landmarks(), b.get_landmarks()
    assert la.keys() == lb.keys() and len(la) > 50
    for i in la:
        assert np.array_equal(la[i]["point"], lb[i]["point"])
        assert la[i]["quality"] == lb[i]["quality"] or (not exact and abs(la[i]["quality"] - lb[i]["quality"]) < 1e-6)


def test_resident_window_survives_switching_paths(gpu_lib):
    """resident -> inspection hooks that take the host path -> resident again; landmark priors switch the window to the host path"""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=10, L=300, n_obs=3000, seed=9, rig="euroc", keyframe_every=2, frame_dt=0.3)
    a, b = Estimator(0), Estimator(0)
    b.set_pack_mode(1)
    drv = Driver([a, b], spec, seed=2)
    for k in range(spec.P):
        drv.add_frame(k)
        if k == 4:
            a.set_pack_mode(1)      # one frame through the host path: the device copy is dropped ...
        if k == 5:
            a.set_pack_mode(0)      # ... and rebuilt from the graph
        if k not in (4,):
            assert_same_csr(a.debug_csr(), b.debug_csr(), "frame %d" % k)
        for e in drv.ests:
            e.optimize(3)
        same_states(drv.ests, "frame %d" % k)
        if k == 6:   # a HomogeneousPointError makes the window take the host path; removing it brings the resident one back
            lid = drv.lm_ids[sorted(drv.lm_ids)[0]]
            rid = [e.add_homogeneous_point_error(lid, np.array([1.0, 2.0, 3.0, 1.0]), np.eye(3)) for e in drv.ests]
            assert not a.debug_csr()["resident"]
            for e, r in zip(drv.ests, rid):
                e.optimize(2)
                assert e.remove_homogeneous_point_error(r)
            same_states(drv.ests, "frame %d with a landmark prior" % k)
        for e in drv.ests:
            e.apply_marginalization(3, 2)


def test_resident_window_handle_renumbering(gpu_lib, monkeypatch):
    """thousands of short-lived landmarks: the handle space is compacted (ids keep their order) and the window stays equal"""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=6, L=120, n_obs=900, seed=3, rig="euroc", keyframe_every=2, frame_dt=0.3)
    a, b = Estimator(0), Estimator(0)
    b.set_pack_mode(1)
    drv = Driver([a, b], spec, seed=4)
    for k in range(spec.P):
        drv.add_frame(k)
        # landmarks that come and go without ever being observed use up handles (applyMarginalizationStrategy erases them)
        for _ in range(2500):
            lid = drv.all(lambda e: e.new_id())
            for e in drv.ests:
                e.add_landmark(lid, np.array([1.0, 2.0, 3.0, 1.0]))
        assert_same_csr(a.debug_csr(), b.debug_csr(), "frame %d" % k)
        for e in drv.ests:
            e.optimize(3)
        same_states(drv.ests, "frame %d" % k)
        ra = [e.apply_marginalization(2, 2) for e in drv.ests]
        assert list(ra[0][1]) == list(ra[1][1])
