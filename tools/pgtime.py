"""Timing of the global pose-graph optimisation (BASELINE config #5 shape) on the GPU vs the oracle.
usage: python tools/pgtime.py [n] [laps] [loop_every] [reps] [--no-oracle]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svin_amd import synthetic_pg as spg  # noqa: E402
from svin_amd.posegraph import PoseGraph  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, laps, loop_every, reps = (int(a) for a in (args + ["5000", "20", "25", "5"][len(args):]))
t0 = time.time()
spec = spg.make_pose_graph(n=n, laps=laps, loop_every=loop_every, seed=7)
print("graph: %d keyframes, %d loops (%.1fs to generate)" % (n, len(spec.loops), time.time() - t0), flush=True)
for six in (False, True):
    for piece in ([int(os.environ["PG_PIECE"])] if "PG_PIECE" in os.environ else [0]):
        best = None
        for rep in range(reps):
            g = PoseGraph(0, six_dof=six)
            g.set_partition(piece, 128)
            if "PG_L2" in os.environ:
                g.set_levels(2 if int(os.environ["PG_L2"]) > 0 else 1, max(int(os.environ["PG_L2"]), 0))
            earliest, cur = spg.feed(g, spec)
            t1 = time.time()
            s = g.optimize(earliest, cur)
            wall = time.time() - t1
            t1 = time.time()
            g.optimize(earliest, cur)   # same object again: buffers are allocated, the problem is rebuilt from the SVIn poses
            wall2 = time.time() - t1
            if best is None or s["solve_seconds"] < best[0]["solve_seconds"]:
                best = (s, wall, g.partition(), wall2)
        s, wall, part, wall2 = best
        print("%s piece %d: %d iterations, device %.3f ms (%.3f ms / iteration), call wall %.3f ms (repeat call on the same handle %.3f ms), cost %.4g -> %.4g, %s"
              % ("6dof" if six else "4dof", piece, s["iterations"], 1e3 * s["solve_seconds"],
                 1e3 * s["solve_seconds"] / max(1, s["iterations"]), 1e3 * wall, 1e3 * wall2, s["initial_cost"], s["final_cost"], part), flush=True)
    if "--no-oracle" not in sys.argv:
        from oracle import orc
        c = orc.OraclePoseGraph(six_dof=six, envelope=True)
        earliest, cur = spg.feed(c, spec)
        t1 = time.time()
        sc = c.optimize(earliest, cur)
        dt = time.time() - t1
        print("  oracle: %d iterations, %.1f ms (%.2f ms / iteration)" % (sc["iterations"], 1e3 * dt, 1e3 * dt / max(1, sc["iterations"])), flush=True)
