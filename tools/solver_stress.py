"""Repeats whole optimisations over every tile-row count of the two one-workgroup dense solvers (k_chol_solve_lds: 2..11 keyframes,
k_chol_solve_ll: 12..18) and checks that every run ends like the first one: same iteration count, final cost within 1e-9 relative.
The solvers synchronise their waves through LDS counters without fences; a hand-over that is wrong once in a thousand launches
would show up here as a run that differs (or as cholFail bit 2 / 4 -> a failed factorisation -> a different iteration count)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
t0 = time.time()
for P in list(range(2, 12)) + [12, 13, 14, 15, 17, 18]:
    spec = syn.make_window(P=P, L=400, n_obs=4000, seed=300 + P, rig="euroc", frame_dt=0.25)
    est = Estimator(0)
    fids, lids = syn.feed(est, spec)
    T0 = [est.get_T_WS(f) for f in fids]
    sb0 = [est.get_speed_and_bias(f) for f in fids]
    lm0 = [est.get_landmark(l)["point"] for l in lids]
    ref = None
    for r in range(reps):
        for f, T, sb in zip(fids, T0, sb0):
            est.set_T_WS(f, T); est.set_speed_and_bias(f, sb)
        for l, hp in zip(lids, lm0):
            est.set_landmark(l, hp)
        est.optimize(8)
        s = est.summary()
        key = (s["iterations"], s["successful"])
        if ref is None:
            ref = (key, s["final_cost"])
        elif key != ref[0] or abs(s["final_cost"] - ref[1]) > 1e-9 * abs(ref[1]):
            bad += 1
            print("P", P, "run", r, "differs:", key, s["final_cost"], "first run:", ref, flush=True)
    print("P %2d: %d runs, %d iterations each, final cost %.9e" % (P, reps, ref[0][0], ref[1]), flush=True)
print("solver stress: %d runs differ; %.1f s" % (bad, time.time() - t0))
sys.exit(1 if bad else 0)
