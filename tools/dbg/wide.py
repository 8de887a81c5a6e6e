import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
from oracle import orc
import test_gpu_parity as tp

for (P, L, n, dt) in ((16, 600, 9000, 0.25), (30, 900, 9000, 0.25), (44, 1200, 12000, 0.25), (48, 1500, 15000, 0.25)):
    for mode in ("default", "pairwise"):
        Estimator.debug_set_option("SVIN_SCHUR_PAIRWISE", 1 if mode == "pairwise" else 0)
        spec = tp.drop_underdetermined_landmarks(syn.make_window(P=P, L=L, n_obs=n, seed=31, frame_dt=dt))
        gpu, cpu, fg, fc, lg, lc = tp.make_pair(spec)
        lin_c = cpu.map().linearize(0.0); lin_g = gpu.linearize(0.0)
        perm = tp.reduced_permutation(gpu, cpu, fg, fc, lin_g, lin_c)
        S, g = lin_g["S"][np.ix_(perm, perm)], lin_g["g"][perm]
        sd = np.sqrt(np.abs(np.diag(lin_c["S"])))
        D = np.abs(S / np.outer(sd, sd) - lin_c["S"] / np.outer(sd, sd))
        i, j = np.unravel_index(np.argmax(D), D.shape)
        print(P, mode, "d", lin_g["d"], "dS", D.max(), "at", i, j, "S_g", S[i, j], "S_c", lin_c["S"][i, j], "sd", sd[i], sd[j],
              "dg", np.max(np.abs(g / sd - lin_c["g"] / sd)), "cost", lin_g["cost"], lin_c["cost"], "min sd", sd.min(), "asym", np.max(np.abs(S - S.T)))
        # block pattern of the error (15x15 frame blocks in oracle order)
        nb = lin_c["d"] // 15
        B = np.array([[D[15 * a:15 * a + 15, 15 * b:15 * b + 15].max() for b in range(nb)] for a in range(nb)])
        bad = np.argwhere(B > 1e-8)
        print("   bad frame-block pairs:", len(bad), bad[:12].tolist())
