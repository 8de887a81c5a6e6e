"""Several handles on one GPU at the same time (SURVEY 8(b) threading row: one HIP stream per handle, a handle is used by one
caller at a time).  SVIn runs the estimator and pose_graph as two nodes side by side (okvis_ros/launch/svin_stereorig_v2.xml:17-34);
here a svin_ba handle works through a sliding window on one thread while a svin_pg handle optimises a loop-closure graph on
another, and several svin_ba handles solve windows concurrently.  Every result must be the one the handle produces alone."""
import threading

import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def run_sliding(spec):
    from svin_amd.estimator import Estimator
    est = Estimator(0)

    def on_frame(k, fid):
        est.optimize(6)
        est.apply_marginalization(3, 2)
    syn.feed(est, spec, on_frame=on_frame)
    frames = est.frame_ids()
    lms = est.get_landmarks()
    return ([est.get_T_WS(f) for f in frames], [est.get_speed_and_bias(f) for f in frames],
            {i: (v["point"].copy(), v["quality"]) for i, v in lms.items()}, est.summary()["final_cost"])


def run_posegraph(pspec, reps):
    from svin_amd import synthetic_pg as spg
    from svin_amd.posegraph import PoseGraph
    out = None
    for _ in range(reps):
        g = PoseGraph(0)
        g.set_partition(32, 0)
        earliest, cur = spg.feed(g, pspec)
        s = g.optimize(earliest, cur)
        out = (g.poses(), s["final_cost"], s["iterations"])
        g.close()
    return out


def same_sliding(a, b):
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert (x is None) == (y is None)
        if x is not None:
            assert np.array_equal(x, y)
    assert a[2].keys() == b[2].keys()
    for i in a[2]:
        assert np.array_equal(a[2][i][0], b[2][i][0]) and a[2][i][1] == b[2][i][1]
    assert a[3] == b[3]


def test_sliding_window_and_pose_graph_side_by_side(gpu_lib):
    from svin_amd import synthetic_pg as spg
    spec = syn.make_window(P=14, L=600, n_obs=6000, seed=12, rig="euroc", keyframe_every=2, frame_dt=0.25)
    pspec = spg.make_pose_graph(n=600, laps=4, loop_every=20, seed=3)
    ba_alone, pg_alone = run_sliding(spec), run_posegraph(pspec, 1)
    res, errs = {}, []

    def guard(key, fn, *a):
        try:
            res[key] = fn(*a)
        except Exception as ex:   # pragma: no cover
            errs.append((key, ex))
    th = [threading.Thread(target=guard, args=("ba", run_sliding, spec)), threading.Thread(target=guard, args=("pg", run_posegraph, pspec, 6))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    same_sliding(res["ba"], ba_alone)
    (Ta, Qa), ca, ia = res["pg"]
    (Tb, Qb), cb, ib = pg_alone
    assert ia == ib and ca == cb and np.array_equal(Ta, Tb) and np.array_equal(Qa, Qb)


def test_several_estimator_handles_at_once(gpu_lib):
    spec = syn.make_window(P=10, L=500, n_obs=5000, seed=21, rig="euroc", keyframe_every=2, frame_dt=0.25)
    alone = run_sliding(spec)
    res, errs = {}, []

    def guard(i):
        try:
            res[i] = run_sliding(spec)
        except Exception as ex:   # pragma: no cover
            errs.append((i, ex))
    th = [threading.Thread(target=guard, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for i in range(6):
        same_sliding(res[i], alone)
