"""Where does the GPU's marginalisation lose accuracy?  One-shot marginalisation from identical states (as in
tests/test_gpu_parity.py::test_marginalization_one_shot), three rigs.  With the GPU's own system after M1 (SVIN_MARG_KEEP_PRE=1)
the 40-digit arbiter (tests/mp_marg.py) gives
   e(M2+M3 | GPU)  = GPU prior  vs  arbiter applied to the GPU's post-M1 system      -- the GPU's elimination + eigen-solve alone
   e(M1 | GPU)     = arbiter(GPU post-M1)  vs  arbiter(oracle post-M1)               -- what M1 contributed (both exact afterwards)
next to the oracle's distance to arbiter(oracle post-M1).  Distances in units of the parameters' standard deviations."""
import os, sys
os.environ["SVIN_MARG_KEEP_PRE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mp_marg
import test_gpu_parity as T
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
from oracle import orc

for rig, kw in (("euroc", {}), ("test4", {}), ("rig_v2", dict(sonar=True, depth=True))):
    spec = syn.make_window(P=7, L=500, n_obs=4000, seed=52, rig=rig, keyframe_every=2, frame_dt=0.3, **kw)
    rec_0, _ = T.one_shot_pass(orc.OracleEstimator(), spec, at=(5, 6))
    snaps = [r[0] for r in rec_0]
    rec_c, fc = T.one_shot_pass(orc.OracleEstimator(), spec, at=(5, 6), snaps=snaps)
    rec_g, fg = T.one_shot_pass(Estimator(0), spec, at=(5, 6), snaps=snaps)
    for i, (g, c) in enumerate(zip(rec_g, rec_c)):
        pg, pc = g[5], c[5]
        ex_c = mp_marg.marginalize_mp(pc["H"], pc["b0"], pc["lm"], pc["dense"])
        ex_g = mp_marg.marginalize_mp(pg["H"], pg["b0"], pg["lm"], pg["dense"])
        # orderings: GPU prior (g[2]) and arbiter(GPU pre) are in the GPU's kept-block order; the oracle's in its own
        keyc = {(b["frame"], b["kind"], b["index"]): b for b in c[2]["blocks"]}
        perm = np.zeros(g[2]["n"], int)
        for b in g[2]["blocks"]:
            if b["frame"] is not None:
                for k in range(b["mdim"]):
                    perm[keyc[(b["frame"], b["kind"], b["index"])]["ordering"] + k] = b["ordering"] + k
        sd = np.sqrt(np.abs(np.diag(ex_c["H"])))

        def dist(Ha, ba, Hb, bb):
            return float(np.max(np.abs(Ha - Hb) / np.outer(sd, sd))), float(np.max(np.abs(ba - bb) / sd))
        Hg, bg = g[2]["H"][np.ix_(perm, perm)], g[2]["b0"][perm]
        Hxg, bxg = ex_g["H"][np.ix_(perm, perm)], ex_g["b0"][perm]
        Jg = g[2]["J"][:, perm]
        print("%-7s #%d n %3d  GPU total (H, b0) %.1e %.1e | GPU M2 alone %.1e %.1e | GPU M1 alone %.1e %.1e | oracle M2 %.1e %.1e | "
              "JtJ: GPU vs arbiter(GPU pre) %.1e, oracle vs arbiter %.1e" %
              ((rig, i, g[2]["n"]) + dist(Hg, bg, ex_c["H"], ex_c["b0"]) + dist(Hg, bg, Hxg, bxg) + dist(Hxg, bxg, ex_c["H"], ex_c["b0"]) +
               dist(c[2]["H"], c[2]["b0"], ex_c["H"], ex_c["b0"]) +
               (float(np.max(np.abs(Jg.T @ Jg - ex_g["JtJ"][np.ix_(perm, perm)]) / np.outer(sd, sd))),
                float(np.max(np.abs(c[2]["J"].T @ c[2]["J"] - ex_c["JtJ"]) / np.outer(sd, sd))))), flush=True)
