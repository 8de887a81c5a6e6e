import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
for rig in ("rig_v2", "euroc"):
    spec = syn.make_window(P=24, L=2400, n_obs=24000, seed=7, rig=rig, keyframe_every=2, frame_dt=0.25, sonar=rig == "rig_v2", depth=rig == "rig_v2")
    est = Estimator(0); ds = []
    def on_frame(k, fid):
        est.optimize(2); ds.append(est.linearize(1e-4)["d"]); est.apply_marginalization(5, 3)
    syn.feed(est, spec, on_frame=on_frame)
    print(rig, ds)
