import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from svin_amd import synthetic_pg as spg
from svin_amd.posegraph import PoseGraph
from oracle import orc
for (n, laps, le) in [(3000, 10, 2), (3000, 30, 3), (20000, 40, 25)]:
    spec = spg.make_pose_graph(n=n, laps=laps, loop_every=le, seed=3)
    for six in (False, True):
        g = PoseGraph(0, six_dof=six)
        e, c = spg.feed(g, spec)
        t0 = time.time(); s = g.optimize(e, c); wall = time.time() - t0
        part = g.partition()
        msg = "n %d loops %d %s: %d it, %.2f ms/it device, wall %.1f ms, root %d, pieces %d/%d" % (n, len(spec.loops), "6dof" if six else "4dof", s["iterations"], 1e3 * s["solve_seconds"] / max(1, s["iterations"]), 1e3 * wall, part["separator_unknowns"], part["pieces"], part["level2_pieces"])
        if n <= 3000:
            o = orc.OraclePoseGraph(six_dof=six, envelope=True)
            spg.feed(o, spec)
            so = o.optimize(e, c)
            dT = np.max(np.abs(g.poses()[0] - o.poses()[0]))
            msg += "; oracle it %d, max |dt| %.2e, cost %.6g vs %.6g" % (so["iterations"], dT, s["final_cost"], so["final_cost"])
        print(msg, flush=True)
