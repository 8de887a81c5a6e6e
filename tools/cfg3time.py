"""Timing of BASELINE config #3 (stereo+IMU+sonar+depth, per-frame extrinsics: d = 270, pairwise Schur path)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec = syn.make_window(P=10, L=4000, n_obs=40000, seed=3, rig="rig_v2", sonar=True, depth=True)
est = Estimator(0)
syn.feed(est, spec)
for rep in range(3):
    est.prepare()
    t0 = time.perf_counter(); est.solve_prepared(10); dt = time.perf_counter() - t0
    s = est.summary()
    print("config3 solve(10): %.2f ms, %d iterations -> %.1f it/s (%.3f ms/iteration) cost %.5e -> %.5e" %
          (1e3 * dt, s["iterations"], s["iterations"] / dt, 1e3 * dt / max(1, s["iterations"]), s["initial_cost"], s["final_cost"]), flush=True)
print("kernel ms (eval, build, solve):", est.bench_kernel_times(10))
