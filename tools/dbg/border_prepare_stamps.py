"""stage stamps of k_chol_border_prepare (100 MHz ticks) on the stereo_rig_v2 window; needs tools/build_variant.sh bordertiming -DSVIN_BORDER_TIMING"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec = syn.make_window(P=16, L=960, n_obs=9600, seed=3, rig="rig_v2", frame_dt=0.25, sonar=True, depth=True)
est = Estimator(0)
def on_frame(k, fid):
    est.optimize(3)
    if k + 1 < 16:
        est.apply_marginalization(5, 3)
syn.feed(est, spec, on_frame=on_frame)
print("d", est.linearize(1e-4)["d"])
est.debug_reduced_solve(1e-4, fused=True)
off = 32 * 176 + 32 * 32 + 32 + 176 + 16
st = est.debug_peek_solver_scratch(off, 8)
names = ["start", "image + loads", "factor", "inverse", "Linv copy", "u", "q", "V + end"]
for n, a, b in zip(names[1:], st[:-1], st[1:]):
    print("%-14s %6.2f us" % (n, (b - a) / 100.0))
print("total %.2f us" % ((st[-1] - st[0]) / 100.0))
