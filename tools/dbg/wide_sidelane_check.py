"""Wide windows of several shapes solved with and without the side lane (small factors + speed / bias chain beside the landmark
elimination, DeviceProblem::sideLane) and with / without the split block rows of k_schur_rows: the iterates must agree to rounding
(the side lane changes the ORDER in which the factors' atomic adds and the pose blocks' sums reach S, nothing else)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator

worst = 0.0
for P, L, N, seed in ((32, 12000, 120000, 1), (48, 20000, 200000, 2), (64, 30000, 300000, 3), (64, 50000, 500000, 4), (40, 8000, 64000, 5)):
    spec = syn.make_window(P=P, L=L, n_obs=N, seed=seed, frame_dt=0.25)
    out = []
    for opts in ({}, {"SVIN_NO_SB_EARLY": 1}, {"SVIN_NO_ROW_SPLIT": 1}):
        for k in ("SVIN_NO_SB_EARLY", "SVIN_NO_ROW_SPLIT"):
            Estimator.debug_set_option(k, opts.get(k, 0))
        est = Estimator(0)
        fids, _ = syn.feed(est, spec)
        for rep in range(3):      # (repeated: a missing dependency would show as run-to-run scatter)
            est.optimize(4)
        s = est.summary()
        out.append((s["final_cost"], np.stack([est.get_T_WS(f) for f in fids])))
    for k in ("SVIN_NO_SB_EARLY", "SVIN_NO_ROW_SPLIT"):
        Estimator.debug_set_option(k, 0)
    d = max(float(np.max(np.abs(out[0][1] - o[1]))) for o in out[1:])
    dc = max(abs(out[0][0] - o[0]) / out[0][0] for o in out[1:])
    worst = max(worst, d)
    print("P %d L %d N %d: final cost %.9e, max pose difference %.2e, relative cost difference %.2e" % (P, L, N, out[0][0], d, dc), flush=True)
print("worst pose difference", worst)
assert worst < 1e-8
