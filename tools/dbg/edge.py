import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
from oracle import orc

def run(model, dist, keep, iters=5, verbose=False):
    imu = dict(syn.test_rig()[1]); rate = 100
    ns = ((np.arange(8) - 2) * (1_000_000_000 // rate)).astype(np.int64) + 1_000_000_000
    t = np.stack([50 + ns // 1_000_000_000, ns % 1_000_000_000], 1).astype(np.uint32)
    m = np.zeros((8, 6)); m[:, 5] = imu["g"]
    T_SC = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    pts = [[0.0, 0.0, 5.0, 1.0], [1e-9, -1e-9, 4.0, 1.0], [0.3, -0.2, 0.15, 1.0], [0.3, 0.2, -2.0, 1.0], [12.0, 1.0, 3.0, 1.0],
           [5.8, 0.0, 2.0, 1.0], [-0.4, 0.6, -6.0, -2.0], [0.5, 0.25, 3.0, 1e-9], [0.4, -0.1, 2.5, 1.0]]
    pts = [pts[k] for k in keep]
    out = []
    for cls in (Estimator, orc.OracleEstimator):
        e = cls(0) if cls is Estimator else cls()
        e.add_camera(model, [350.0, 360.0, 378.0, 238.0], dist, 752, 480, [0.0] * 4); e.add_imu(imu)
        lids = [e.new_id() for _ in pts]
        for lid, p in zip(lids, pts): e.add_landmark(lid, np.array(p))
        fid = e.new_id(); e.add_states(fid, (51, 0), 400, T_SC, t, m, True)
        for k, lid in enumerate(lids): e.add_observation(lid, fid, 0, k, [300.0 + 7 * k, 200.0 - 5 * k], 6.0)
        e.optimize(iters, 1, verbose)
        out.append((e.summary(), [e.get_landmark(l)["point"] for l in lids], e.get_T_WS(fid)))
    (sg, lg, Tg), (sc, lc, Tc) = out
    print("keep", keep, "iters", iters, "cost gpu %.9f cpu %.9f" % (sg["final_cost"], sc["final_cost"]), "it", sg["iterations"], sc["iterations"],
          "succ", sg["successful"], sc["successful"], "dT", np.max(np.abs(Tg - Tc)), "dlm", [float("%.2e" % np.max(np.abs(a - b))) for a, b in zip(lg, lc)])

all9 = list(range(9))
for it in (1, 2, 5):
    run(0, [], all9, it)
for drop in range(9):
    run(0, [], [k for k in all9 if k != drop], 5)
run(0, [], [8], 5)
run(0, [], [0, 8], 5)
run(0, [], all9, 2, True)
