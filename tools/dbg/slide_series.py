"""Per-frame series of the sliding window (every library call of a frame timed): back to back and with the handle idle before
optimize().   python tools/dbg/slide_series.py [euroc|rig_v2]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
rig = sys.argv[1] if len(sys.argv) > 1 else "rig_v2"
spec = syn.make_window(P=24, L=2400, n_obs=24000, seed=7, rig=rig, keyframe_every=2, frame_dt=0.25,
                       **({"sonar": True, "depth": True} if rig == "rig_v2" else {}))
for spaced in (False, True):
    est = Estimator(0); rows = []; timing = {}
    def on_frame(k, fid):
        if spaced: est.wait_idle()
        t0 = time.perf_counter(); est.optimize(10); t1 = time.perf_counter()
        est.apply_marginalization(5, 3); t2 = time.perf_counter()
        if spaced: est.wait_idle()
        rows.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
    syn.feed(est, spec, on_frame=on_frame, timing=timing)
    f = lambda a: " ".join("%.2f" % (1e3 * x) for x in a[4:])
    print("spaced" if spaced else "back to back")
    print("  add_states  :", f(timing["add_states_s"]))
    print("  set states  :", f(timing["set_states_s"]))
    print("  add_obs     :", f(timing["add_observations_s"]))
    print("  optimize    :", f([r[0] for r in rows]))
    print("  marg call   :", f([r[1] for r in rows]))
    print("  marg wait   :", f([r[2] for r in rows]))
    tot = np.array(timing["add_states_s"]) + np.array(timing["set_states_s"]) + np.array(timing["add_observations_s"]) + np.array([r[0] + r[1] for r in rows])
    print("  all calls of a frame: median %.2f ms, mean %.2f ms (steady state)" % (1e3 * np.median(tot[4:]), 1e3 * np.mean(tot[4:])))
