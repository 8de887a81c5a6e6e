import os, sys, faulthandler
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
faulthandler.enable()
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spec = syn.make_window(P=6, L=200, n_obs=2000, seed=7, rig="euroc", keyframe_every=2, frame_dt=0.25)
est = Estimator(0)
est.set_pack_mode(mode)
def on_frame(k, fid):
    print("frame", k, "pack...", flush=True)
    c = est.debug_csr()
    print("  csr L %d N %d resident %s lm_ptr[-1] %d" % (c["L"], c["N"], c["resident"], c["lm_ptr"][-1] if c["L"] else -1), flush=True)
    est.optimize(3)
    print("  optimised: cost", est.summary()["final_cost"], flush=True)
    ok, rem = est.apply_marginalization(3, 2)
    print("  marginalised, removed", len(rem), flush=True)
syn.feed(est, spec, on_frame=on_frame)
print("done", flush=True)
