"""Several windows through ONE launch sequence (svin_ba_solve_prepared_batch / svin_ba_optimize_batch; SURVEY 8(e):
"independent replicas processing different windows").  The reference optimises one window per call (Estimator.cpp:876-929), so the
yardstick is the product's own single-window path: every window of a batch must end BIT FOR BIT where it ends alone -- same
kernel bodies, same grids, same reduction orders -- with the same iteration and step counts."""
import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def states_of(est, fids, lids):
    T = np.stack([est.get_T_WS(f) for f in fids])
    sb = np.stack([v if v is not None else np.full(9, np.nan) for v in (est.get_speed_and_bias(f) for f in fids)])   # (old keyframes of a sliding window keep their pose only)
    lms = est.get_landmarks()
    lm = np.stack([np.r_[lms[l]["point"], lms[l]["quality"]] for l in lids])
    return T, sb, lm


def build(seed, **kw):
    from svin_amd.estimator import Estimator
    spec = syn.make_window(seed=seed, **kw)
    est = Estimator(0)
    fids, lids = syn.feed(est, spec)
    return est, fids, lids


@pytest.mark.parametrize("B", [16])
def test_batch_of_config2_windows_ends_bit_for_bit_where_each_ends_alone(gpu_lib, B):
    from svin_amd import estimator
    seeds = [20250629 + 7 * k for k in range(B)]
    alone = []
    for sd in seeds:
        est, fids, lids = build(sd)
        est.optimize(10)
        alone.append((states_of(est, fids, lids), est.summary()))
    # the path itself is reproducible run to run (otherwise "bit for bit" would be luck)
    est, fids, lids = build(seeds[0])
    est.optimize(10)
    again = states_of(est, fids, lids)
    assert all(np.array_equal(a, b) for a, b in zip(again, alone[0][0]))
    batch = [build(sd) for sd in seeds]
    n_batched = estimator.optimize_batch([b[0] for b in batch], 10)
    assert n_batched == B
    for k, (est, fids, lids) in enumerate(batch):
        got, s = states_of(est, fids, lids), est.summary()
        ref, s_ref = alone[k]
        assert s["iterations"] == s_ref["iterations"] and s["successful"] == s_ref["successful"] and s["termination"] == s_ref["termination"]
        assert s["initial_cost"] == s_ref["initial_cost"] and s["final_cost"] == s_ref["final_cost"], (k, s["final_cost"], s_ref["final_cost"])
        for a, b, name in zip(got, ref, ("poses", "speed / bias", "landmarks")):
            assert np.array_equal(a, b), "window %d: %s differ by %.3e" % (k, name, float(np.max(np.abs(a - b))))


def test_batch_with_rejected_steps_and_early_termination(gpu_lib):
    """windows of one geometry whose trust regions go different ways: badly perturbed starts (rejected steps: the round's
    k_step_retract launch), a nearly converged one (terminates early and sits out the remaining rounds)"""
    from svin_amd import estimator
    cfgs = [dict(seed=71, pose_noise=(0.6, 0.15), lm_noise=1.5), dict(seed=72, pose_noise=(1.0, 0.25), lm_noise=2.5),
            dict(seed=73, pose_noise=(1.5, 0.4), lm_noise=4.0), dict(seed=74, pose_noise=(1e-6, 1e-6), lm_noise=1e-6, pixel_noise=1e-3)]
    kw = dict(P=6, L=250, n_obs=2500)
    alone = []
    for c in cfgs:
        est, fids, lids = build(**dict(kw, **c))
        est.optimize(25)
        alone.append((states_of(est, fids, lids), est.summary()))
    its = [a[1]["iterations"] for a in alone]
    assert any(a[1]["successful"] < a[1]["iterations"] for a in alone), "no rejected step: raise the perturbation"
    assert min(its) < max(its), its
    batch = [build(**dict(kw, **c)) for c in cfgs]
    assert estimator.optimize_batch([b[0] for b in batch], 25) == len(cfgs)
    for k, (est, fids, lids) in enumerate(batch):
        s, s_ref = est.summary(), alone[k][1]
        assert (s["iterations"], s["successful"], s["termination"], s["final_cost"]) == (s_ref["iterations"], s_ref["successful"], s_ref["termination"], s_ref["final_cost"]), (k, s, s_ref)
        for a, b in zip(states_of(est, fids, lids), alone[k][0]):
            assert np.array_equal(a, b)


def test_mixed_geometries_fall_into_groups_and_singles(gpu_lib):
    """two windows of one geometry, two of another, one on its own (12 keyframes: a reduced system of 180 rows, which the border variant
    of the LDS-resident solver takes and the batched kernels do not): 4 batched, all five where they end alone"""
    from svin_amd import estimator
    specs = [dict(seed=1, P=6, L=250, n_obs=2500), dict(seed=2, P=6, L=250, n_obs=2500), dict(seed=3, P=8, L=400, n_obs=4000),
             dict(seed=4, P=8, L=400, n_obs=4000), dict(seed=5, P=12, L=400, n_obs=4000)]
    alone = []
    for c in specs:
        est, fids, lids = build(**c)
        est.optimize(6)
        alone.append(states_of(est, fids, lids))
    batch = [build(**c) for c in specs]
    assert estimator.optimize_batch([b[0] for b in batch], 6) == 4
    for k, (est, fids, lids) in enumerate(batch):
        for a, b in zip(states_of(est, fids, lids), alone[k]):
            assert np.array_equal(a, b), (k, float(np.max(np.abs(a - b))))
    # argument checks: a handle twice is an error, an empty batch is not
    with pytest.raises(RuntimeError):
        estimator.optimize_batch([batch[0][0], batch[0][0]], 2)
    assert estimator.optimize_batch([], 3) == 0


def test_batch_of_sliding_windows_with_marginalisation_priors(gpu_lib):
    """windows in SVIn's operating mode: fed frame by frame, optimised and marginalised after every frame -- each carries a
    marginalisation prior (the prior block of the batched evaluation, the prior's blocks of the batched build) and fixed-lag
    structure.  Three handles per seed take the identical history; before the last optimisation two of them go into the batch
    (partners of one geometry), the third is optimised alone: bit for bit."""
    from svin_amd import estimator
    from svin_amd.estimator import Estimator

    def history(seed):
        spec = syn.make_window(P=9, L=600, n_obs=6000, seed=seed, keyframe_every=2, frame_dt=0.3)
        est = Estimator(0)

        def on_frame(k, fid):
            if k < spec.P - 1:
                est.optimize(6)
                est.apply_marginalization(3, 2)
        fids, lids = syn.feed(est, spec, on_frame=on_frame)
        est.wait_idle()
        return est, est.frame_ids(), [l for l in lids if l in est.get_landmarks()]

    seeds = [11, 12, 13]
    alone = [history(sd) for sd in seeds]
    batch = [history(sd) for sd in seeds for _ in range(2)]   # two handles per seed: every window has a partner of its geometry
    for k, (b, fb, lb) in enumerate(batch):
        a, fa, la = alone[k // 2]
        assert a.marg() is not None and fa == fb and la == lb
        for x, y in zip(states_of(a, fa, la), states_of(b, fb, lb)):
            assert np.array_equal(x, y, equal_nan=True), "the histories of a seed differ before the batch"
    for a, _, _ in alone:
        a.optimize(8)
    assert estimator.optimize_batch([b[0] for b in batch], 8) == len(batch)
    for k, (b, fb, lb) in enumerate(batch):
        a, fa, la = alone[k // 2]
        sa, sb = a.summary(), b.summary()
        assert (sa["iterations"], sa["successful"], sa["final_cost"]) == (sb["iterations"], sb["successful"], sb["final_cost"]), (k, sa, sb)
        for x, y in zip(states_of(a, fa, la), states_of(b, fb, lb)):
            assert np.array_equal(x, y, equal_nan=True), k
