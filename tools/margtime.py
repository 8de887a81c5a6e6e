"""Times the per-frame operations of a running window (optimize + applyMarginalizationStrategy) on the GPU backend
and on the oracle: a 5-keyframe / 3-IMU-frame sliding window fed frame by frame (SVIn's operating mode)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator

def run(est, spec, iters, label):
    t_opt, t_marg, nrem = [], [], []
    def on_frame(k, fid):
        t0 = time.perf_counter(); est.optimize(iters); t1 = time.perf_counter()
        ok, removed = est.apply_marginalization(5, 3); t2 = time.perf_counter()
        t_opt.append(t1 - t0); t_marg.append(t2 - t1); nrem.append(len(removed))
    t0 = time.perf_counter()
    timing = {}
    syn.feed(est, spec, on_frame=on_frame, timing=timing)
    tot = time.perf_counter() - t0
    if timing:
        print("%s: add_observations %.3f ms per frame (%d observations, %.0f ns each)" % (label, 1e3 * np.median(timing['add_observations_s']), int(np.median(timing['add_observations_n'])), 1e9 * sum(timing['add_observations_s']) / sum(timing['add_observations_n'])))
    print("%s: frames %d  optimize(%d) median %.3f ms  marginalise median %.3f ms (max %.3f)  whole feed %.1f ms  landmarks removed/frame %s" %
          (label, len(t_opt), iters, 1e3 * np.median(t_opt[3:]), 1e3 * np.median(t_marg[3:]), 1e3 * max(t_marg[3:]), 1e3 * tot, nrem[-4:]))

if __name__ == "__main__":
    rig = "rig_v2" if "--rig-v2" in sys.argv else "euroc"   # rig_v2: per-frame extrinsics (sigma_c_relative > 0), sonar + depth off
    spec = syn.make_window(P=20, L=2000, n_obs=20000, seed=7, rig=rig, keyframe_every=2, frame_dt=0.25)
    run(Estimator(0), spec, 10, "gpu")
    if "--cpu" in sys.argv:
        from oracle import orc
        run(orc.OracleEstimator(), spec, 10, "oracle")
