"""Generates tests/golden/prior_matrices.npz: Jacobi-scaled marginalisation priors (the matrices M3 diagonalises,
MarginalizationError.cpp:725-758) from the ORACLE running bench.py's sliding windows -- stereo_rig_v2 with sonar + depth
(n = 105 and 117: five 12-fold eigenvalue clusters from the extrinsics chain, spectrum 1e-8 .. 3.4), an early rig_v2 frame with a
numerical null space, and the EuRoC window (n = 45).  Run from the repository root: python tools/dump_prior_matrices.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svin_amd import synthetic as syn  # noqa: E402
from oracle import orc  # noqa: E402


def scaled(H):
    d = np.diag(H)
    p = np.where(d > 1e-9, np.sqrt(np.where(d > 0, d, 1)), 1e-3)
    return 0.5 * (H + H.T) / np.outer(p, p)


out = {}
for rig, keep in (("rig_v2", {8: "rig_v2_n69_null_space", 22: "rig_v2_n117", 23: "rig_v2_n105"}), ("euroc", {23: "euroc_n45"})):
    spec = syn.make_window(P=24, L=2400, n_obs=24000, seed=7, rig=rig, keyframe_every=2, frame_dt=0.25,
                           **({"sonar": True, "depth": True} if rig == "rig_v2" else {}))
    est = orc.OracleEstimator()

    def on_frame(k, fid):
        est.optimize(10)
        est.apply_marginalization(5, 3)
        if k in keep:
            out[keep[k]] = scaled(est.marg()["H"])
    syn.feed(est, spec, on_frame=on_frame)
for k, v in out.items():
    print(k, v.shape, "eigenvalues %.2e .. %.2e" % tuple(np.linalg.eigvalsh(v)[[0, -1]]))
np.savez_compressed(os.path.join("tests", "golden", "prior_matrices.npz"), **out)
