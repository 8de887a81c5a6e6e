"""Speed / bias chain elimination ahead of the blocked Cholesky (kernels.hip, k_sb_factor ...): the reduced-system solve of a
wide window with and without it (SVIN_NO_SB_ELIM=1 turns it off; the switch is read once per process, so run this twice).
    python tools/sb_elim_time.py [P] [L] [n_obs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svin_amd import synthetic as syn           # noqa: E402
from svin_amd.estimator import Estimator        # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
spec = syn.make_window(P=P, L=L, n_obs=N, seed=20250629, frame_dt=0.25)
est = Estimator(0)
frames, _ = syn.feed(est, spec)
est.optimize(1)
t0 = time.perf_counter()
est.optimize(10)
dt = time.perf_counter() - t0
s = est.summary()
ev, bd, sv = est.bench_kernel_times(20)
poses = np.array([est.get_T_WS(f) for f in frames])
print("elimination %s  P %d d %d: %d iterations, %.3f ms / iteration (solve_time %.3f), final cost %.9e" %
      ("off" if os.environ.get("SVIN_NO_SB_ELIM") else "on", P, 15 * P, s["iterations"], 1e3 * dt / max(s["iterations"], 1),
       1e3 * s["solve_time"] / max(s["iterations"], 1), s["final_cost"]))
print("   kernel times: eval %.1f us, build %.1f us, reduced solve %.1f us;  pose checksum %.12f" % (1e3 * ev, 1e3 * bd, 1e3 * sv, float(np.abs(poses).sum())))
