"""Sliding-window smoke over the configurations the reference ships (exceptions / non-finite states are the failure
mode here; numerical parity lives in tests/): euroc, rig_v2 (per-frame extrinsics), rig_v2 + sonar + depth."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator

for name, kw in (("euroc", dict(rig="euroc")), ("rig_v2", dict(rig="rig_v2")), ("rig_v2 + sonar + depth", dict(rig="rig_v2", sonar=True, depth=True)),
                 ("euroc, every frame a keyframe", dict(rig="euroc", keyframe_every=1)), ("rig_v2, 3 keyframes", dict(rig="rig_v2", nkf=3))):
    nkf = kw.pop("nkf", 5)
    spec = syn.make_window(P=24, L=1500, n_obs=15000, seed=11, frame_dt=0.25, **({"keyframe_every": 2} | kw))
    est = Estimator(0)
    t_opt, t_marg = [], []
    def cb(k, fid):
        t0 = time.perf_counter(); est.optimize(10); t1 = time.perf_counter()
        est.apply_marginalization(nkf, 3); t2 = time.perf_counter()
        t_opt.append(t1 - t0); t_marg.append(t2 - t1)
    fids, _ = syn.feed(est, spec, on_frame=cb)
    T = np.array([est.get_T_WS(f) for f in est.frame_ids()])
    assert np.all(np.isfinite(T)), name
    print("%-32s frames in window %d, landmarks %d, optimize(10) median %.3f ms, marginalise %.3f ms, last pose error %.3f m" % (
        name, est.num_frames(), est.num_landmarks(), 1e3 * np.median(t_opt[4:]), 1e3 * np.median(t_marg[4:]),
        float(np.linalg.norm(T[-1, :3] - spec.T_WS_true[-1, :3]))), flush=True)
