"""The marginalisation job is asynchronous: svin_ba_apply_marginalization_strategy returns once the host policy has handed the
launches of M1-M3 to the handle's enqueue thread (Window::enqueueAsync), and every entry point that needs the graph, the
stream or the prior joins it first (Window::quiesce).  That is a convention kept by hand in ~20 methods; this test fires the
C ABI's getters and mutators in random order IMMEDIATELY after the call, frame after frame, and holds every returned value and
the whole trajectory -- bit for bit -- against a second handle that gets the same calls behind svin_ba_wait_idle.
Reference behaviour: Estimator.cpp:495-814 is synchronous, so any interleaving must look as if the job had finished."""
import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def same(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    if isinstance(a, float) and isinstance(b, float):
        return a == b or (a != a and b != b)
    return a == b


def drive(spec, wait_first, seed, calls_per_frame):
    from svin_amd.estimator import Estimator
    est = Estimator(0)
    rng = np.random.default_rng(seed)
    log = []

    def rec(name, value):
        log.append((name, value))

    def prior_digest():
        m = est.marg()
        return None if m is None else dict(n=m["n"], H=m["H"], b0=m["b0"], J=m["J"], e0=m["e0"])

    def on_frame(k, fid):
        est.optimize(6)
        ok, removed = est.apply_marginalization(5, 3)
        rec("marginalise", (ok, sorted(int(r) for r in removed)))
        if wait_first:
            est.wait_idle()
        frames = est.frame_ids()                    # (itself one of the calls under test: first after the job was enqueued)
        rec("frame_ids", frames)
        for _ in range(calls_per_frame):
            f = int(frames[rng.integers(len(frames))])
            what = int(rng.integers(22))
            if what == 0:
                rec("T_WS", est.get_T_WS(f))
            elif what == 1:
                rec("speed_and_bias", est.get_speed_and_bias(f))
            elif what == 2:
                rec("extrinsics", est.get_camera_sensor_states(f, int(rng.integers(2))))
            elif what == 3:
                lids = est.landmark_ids()
                rec("num_landmarks", (len(lids), est.num_landmarks()))
                if lids:
                    rec("landmark", est.get_landmark(int(lids[rng.integers(len(lids))])))
            elif what == 4:
                rec("landmarks", est.get_landmarks())
            elif what == 5:
                rec("prior", prior_digest())
            elif what == 6:
                rec("keyframe", (est.is_keyframe(f), est.current_keyframe_id(), est.current_frame_id(), est.is_in_imu_window(f)))
            elif what == 7:
                lids = est.landmark_ids()
                if lids:
                    rec("observations", est.landmark_observations(int(lids[rng.integers(len(lids))])))
            elif what == 8:
                rec("all_observations", est.all_landmark_observations())
            elif what == 9:
                ids = est.parameter_block_ids()
                rec("blocks", ids)
                b = int(ids[rng.integers(len(ids))])
                rec("block", (est.parameter_block(b), est.residuals_of(b), est.is_parameter_block_constant(b)))
            elif what == 10:
                rec("keyframe_points", est.keyframe_points(f))
            elif what == 11:
                rec("summary", {k: v for k, v in est.summary().items() if "time" not in k})
            elif what == 12:
                rec("preintegral", (est.get_imu_preintegral(f), est.timestamp(f), est.state_count()))
            elif what == 13:   # mutators from here on: derived from this handle's own getters, so a stale read would propagate
                T = est.get_T_WS(f)
                T[:3] += 1e-4 * rng.normal(size=3)
                rec("set_T_WS", est.set_T_WS(f, T))
            elif what == 14:
                sb = est.get_speed_and_bias(f)
                if sb is not None:
                    rec("set_speed_and_bias", est.set_speed_and_bias(f, sb + 1e-5 * rng.normal(size=9)))
            elif what == 15:
                lids = est.landmark_ids()
                if lids:
                    lid = int(lids[rng.integers(len(lids))])
                    hp = np.array(est.get_landmark(lid)["point"])
                    hp[:3] += 1e-4 * rng.normal(size=3)
                    rec("set_landmark", est.set_landmark(lid, hp))
            elif what == 16:
                lids = est.landmark_ids()
                if lids:
                    lid = int(lids[rng.integers(len(lids))])
                    obs = est.landmark_observations(lid)          # [(frame, camera, keypoint, residual id)]
                    if len(obs) > 3:
                        fr, cam, kp, _ = obs[int(rng.integers(len(obs)))]
                        rec("remove_observation", est.remove_observation(lid, fr, cam, kp))
            elif what == 17:
                lids = est.landmark_ids()
                if lids:
                    lid = int(lids[rng.integers(len(lids))])
                    rec("initialized", (est.is_landmark_initialized(lid), est.set_landmark_initialized(lid, True)))
            elif what == 18:
                rec("eval_reprojection_sum", float(np.sum(est.eval_reprojection(False)["r"])))
            elif what == 19:
                lin = est.linearize(1e-4)
                rec("linearize", (lin["d"], lin["g"]))
            elif what == 20:
                rec("set_keyframe", (est.set_keyframe(f, est.is_keyframe(f)), est.frame_id_by_age(0)))
            else:
                rec("num", (est.num_frames(), est.num_landmarks()))
    fids, _ = syn.feed(est, spec, on_frame=on_frame)
    est.wait_idle()
    paths = est.path_counters()
    assert paths["host_pack_solves"] == 0 and paths["resident_solves"] >= spec.P, paths   # (nothing fell back to the host pack)
    final = dict(T=[est.get_T_WS(f) for f in est.frame_ids()], sb=[est.get_speed_and_bias(f) for f in est.frame_ids()],
                 lm=est.get_landmarks(), prior=prior_digest())
    return log, final


def test_every_entry_point_behind_an_enqueued_marginalisation_job(gpu_lib):
    spec = syn.make_window(P=200, L=9000, n_obs=100000, seed=71, rig="euroc", keyframe_every=2, frame_dt=0.25)
    log_a, fin_a = drive(spec, wait_first=False, seed=5, calls_per_frame=6)
    log_b, fin_b = drive(spec, wait_first=True, seed=5, calls_per_frame=6)
    assert len(log_a) == len(log_b) and len(log_a) > 1200
    kinds = {}
    for i, ((na, va), (nb, vb)) in enumerate(zip(log_a, log_b)):
        assert na == nb, (i, na, nb)
        assert same(va, vb), "call %d (%s) differs between the handle that waited for the job and the one that did not" % (i, na)
        kinds[na] = kinds.get(na, 0) + 1
    print("calls compared bit for bit:", kinds)
    assert same(fin_a, fin_b)
    assert len(kinds) >= 20
