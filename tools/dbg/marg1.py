import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import test_gpu_parity as tp, mp_marg
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
from oracle import orc

for rig, kw in [("euroc", {}), ("test4", {}), ("rig_v2", dict(sonar=True, depth=True))]:
    spec = syn.make_window(P=7, L=500, n_obs=4000, seed=52, rig=rig, keyframe_every=2, frame_dt=0.3, **kw)
    rec_0, _ = tp.one_shot_pass(orc.OracleEstimator(), spec, at=(5, 6))
    snaps = [r[0] for r in rec_0]
    cpu = orc.OracleEstimator()
    pre = []
    def cb(k, fid):
        if k in (5, 6):
            cpu.optimize(12); tp.inject_states(cpu, snaps[len(pre)]); cpu.apply_marginalization(2, 2); pre.append((cpu.marg_pre(), cpu.marg()))
    syn.feed(cpu, spec, on_frame=cb)
    rec_g, fg = tp.one_shot_pass(Estimator(0), spec, at=(5, 6), snaps=snaps)
    for i, ((pm, mc), g) in enumerate(zip(pre, rec_g)):
        mg = g[2]
        r = mp_marg.marginalize_mp(pm["H"], pm["b0"], pm["lm"], pm["dense"])
        # permutation gpu -> oracle ordering
        keyc = {(b["frame"], b["kind"], b["index"]): b for b in mc["blocks"]}
        perm = np.zeros(mg["n"], int)
        for b in mg["blocks"]:
            if b["frame"] is None: continue
            o = keyc[(b["frame"], b["kind"], b["index"])]
            for k in range(b["mdim"]): perm[o["ordering"] + k] = b["ordering"] + k
        sd = np.sqrt(np.abs(np.diag(r["H"])))
        def n2(M): return M / np.outer(sd, sd)
        Hg = mg["H"][np.ix_(perm, perm)]; b0g = mg["b0"][perm]
        Jg = mg["J"][:, perm]
        print(rig, i, "n", mg["n"], "mp rank", r["rank"], "oracle rank", int(np.sum(np.any(mc["J"] != 0, axis=1))), "gpu rank", int(np.sum(np.any(mg["J"] != 0, axis=1))))
        print("   H   vs mp: gpu %.3e oracle %.3e   gpu vs oracle %.3e" % (np.max(np.abs(n2(Hg - r["H"]))), np.max(np.abs(n2(mc["H"] - r["H"]))), np.max(np.abs(n2(Hg - mc["H"])))))
        print("   b0  vs mp: gpu %.3e oracle %.3e" % (np.max(np.abs(b0g - r["b0"]) / sd), np.max(np.abs(mc["b0"] - r["b0"]) / sd)))
        print("   JtJ vs mp: gpu %.3e oracle %.3e" % (np.max(np.abs(n2(Jg.T @ Jg - r["JtJ"]))), np.max(np.abs(n2(mc["J"].T @ mc["J"] - r["JtJ"])))))
        print("   Jte0 vs mp: gpu %.3e oracle %.3e" % (np.max(np.abs(Jg.T @ mg["e0"] - r["Jte0"]) / sd), np.max(np.abs(mc["J"].T @ mc["e0"] - r["Jte0"]) / sd)))
        for name, H in (("gpu", Hg), ("oracle", mc["H"])):
            p = np.where(np.diag(H) > 1e-9, np.sqrt(np.abs(np.diag(H))), 1e-3)
            ev = np.linalg.eigvalsh(0.5 * (H + H.T) / np.outer(p, p))
            print("   numpy eigvalsh of the %s H (preconditioned), smallest / lmax:" % name, (ev[:5] / ev[-1]).tolist(), "threshold", 2.2e-16 * len(ev))
        print("   mp smallest:", r["rel_eigs_small"])
        print("   |e0|^2 gpu %.6e oracle %.6e" % (mg["e0"] @ mg["e0"], mc["e0"] @ mc["e0"]))
