"""Single-GPU timing of a wide window (BASELINE config #4 shape: P=64, L=50000, N=500000) -- the landmark-sharded
multi-GPU mode runs this per rank on 1/world of the landmarks."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator

OLD = "--old" in sys.argv      # the round-5 tile form of the Schur complement (k_schur_panels) instead of the block-pair form
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
P, L, N = (int(a) for a in (argv[0:3] if len(argv) >= 3 else (64, 50000, 500000)))
# SVIN_WIDE_BENCH=1: the window of bench.py's config4 records (seed 20250629, frames 0.25 s apart: more co-visibility, denser S)
BENCH = os.environ.get("SVIN_WIDE_BENCH") == "1"
t0 = time.perf_counter()
spec = syn.make_window(P=P, L=L, n_obs=N, seed=20250629, frame_dt=0.25) if BENCH else syn.make_window(P=P, L=L, n_obs=N, seed=11, rig="euroc")
print("synthetic window built in %.1f s: P %d L %d N %d" % (time.perf_counter() - t0, spec.P, spec.L, spec.N), flush=True)
if OLD:
    Estimator.debug_set_option("SVIN_PANELS_OLD", 1)
est = Estimator(0)
t0 = time.perf_counter()
syn.feed(est, spec)
print("fed in %.1f s" % (time.perf_counter() - t0), flush=True)
for rep in range(3):
    est.prepare()
    t0 = time.perf_counter()
    est.solve_prepared(5)
    dt = time.perf_counter() - t0
    s = est.summary()
    print("solve(5): %.2f ms for %d iterations -> %.1f it/s (%.2f ms/iteration); cost %.6e -> %.6e, upload %.1f ms" %
          (1e3 * dt, s["iterations"], s["iterations"] / dt, 1e3 * dt / max(s["iterations"], 1), s["initial_cost"], s["final_cost"],
           1e3 * s["upload_time"]), flush=True)
print(est.bench_kernel_times(5))
