"""Runs sliding windows (optimize + applyMarginalizationStrategy per frame) over several seeds and rigs with the default
prior eigen-solver (Cholesky-preconditioned Jacobi) and with the plain one-sided Jacobi (SVIN_MARG_EIG=cholesky: the round-4 chain),
and reports rank and pose agreement of the two."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator


def run(spec, window, mode):
    Estimator.debug_set_option("SVIN_MARG_EIG", mode or None)   # (the library reads its environment once: options.hpp)
    est = Estimator(0)
    ranks = []
    def on_frame(k, fid):
        est.optimize(10)
        ok, removed = est.apply_marginalization(*window)
        m = est.marg()
        if m is not None:
            ranks.append((m["n"], int(np.sum(np.any(m["J"] != 0, axis=1)))))
    syn.feed(est, spec, on_frame=on_frame)
    poses = np.array([np.asarray(est.get_T_WS(f)).ravel() for f in est.frame_ids()])
    return ranks, poses


if __name__ == "__main__":
    worst = 0.0
    for rig, window in (("euroc", (5, 3)), ("test4", (5, 3)), ("rig_v2", (5, 3)), ("rig_v2", (3, 2))):
        for seed in range(1, 6):
            spec = syn.make_window(P=16, L=600, n_obs=6000, seed=seed, rig=rig, keyframe_every=2, frame_dt=0.25)
            ra, pa = run(spec, window, None)
            rb, pb = run(spec, window, "cholesky")
            d = float(np.max(np.abs(pa - pb)))
            worst = max(worst, d)
            same = ra == rb
            print("%-7s window %s seed %d: prior sizes %s rank sequences equal %s, max |pose difference| %.2e" %
                  (rig, window, seed, sorted(set(n for n, _ in ra)), same, d), flush=True)
            if not same:
                print("   default:", ra, "\n   cholesky (round-4 Jacobi chain):", rb)
    print("worst pose difference", worst)
