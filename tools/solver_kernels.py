"""Workload for the MFMA / VALU counter passes: the solver kernels of the narrow window (config #2: k_schur_dense,
k_chol_solve_lds), of config #3 (per-frame extrinsics: dense Schur with 34 tile rows, multi-workgroup Cholesky) and of
the wide window (config #4 shape: k_schur_panels, the speed / bias chain elimination k_sb_*, k_big_chol_chain), and a few
frames of the stereo_rig_v2 sliding window (the marginalisation job: k_marg_dense, k_marg_final_dc = the prior's eigen-solve)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
for name, kw in (("config2", dict()), ("config3", dict(P=10, L=4000, n_obs=40000, rig="rig_v2", sonar=True, depth=True)),
                 ("config4", dict(P=64, L=12000, n_obs=120000, frame_dt=0.25))):
    spec = syn.make_window(seed=20250629, **kw)
    est = Estimator(0)
    syn.feed(est, spec)
    est.optimize(4)
    print(name, est.summary())
spec = syn.make_window(P=16, L=1600, n_obs=16000, seed=7, rig="rig_v2", keyframe_every=2, frame_dt=0.25, sonar=True, depth=True)
est = Estimator(0)


def on_frame(k, fid):
    est.optimize(10)
    est.apply_marginalization(5, 3)


syn.feed(est, spec, on_frame=on_frame)
est.wait_idle()
print("sliding rig_v2", est.summary(), est.path_counters())
