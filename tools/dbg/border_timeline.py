"""in-kernel timeline of the LDS-resident solver's border variant on the stereo_rig_v2 sliding window (d = 198); needs a timing build:
tools/build_variant.sh choltiming -DSVIN_CHOL_TIMING -DSVIN_CHOL_TIMING_FINE; SVIN_BA_LIB=build/variants/choltiming.so SVIN_CHOL_TIMING=1 python tools/dbg/border_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec = syn.make_window(P=16, L=960, n_obs=9600, seed=3, rig="rig_v2", frame_dt=0.25, sonar=True, depth=True)
est = Estimator(0)
state = {}
def on_frame(k, fid):
    est.optimize(3)
    state["d"] = est.linearize(1e-4)["d"]
    if k + 1 < 16:
        est.apply_marginalization(5, 3)
syn.feed(est, spec, on_frame=on_frame)
print("d", state["d"], flush=True)
print(est.bench_kernel_times(20))
