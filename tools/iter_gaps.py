"""What the host in the trust-region loop costs: from a `rocprofv3 --kernel-trace` database of the headline
command, the five kernels of an iteration (k_schur_dense, k_reduce_slabs, k_chol_solve_lds, k_post_solve, k_eval_all) in time
order -- per iteration (one k_chol_solve_lds start to the next inside a solve): wall time, the sum of its kernels, and the idle
gaps between them.  usage: python tools/iter_gaps.py <results.db>"""
import sqlite3
import sys

import numpy as np

names = ("k_schur_dense", "k_reduce_slabs", "k_chol_solve_lds", "k_post_solve", "k_eval_all")
db = sqlite3.connect(sys.argv[1])
rows = [(n, s, e) for n, s, e in db.execute("select name, start, end from kernels order by start") if any(k in n for k in names)]
chol = [i for i, r in enumerate(rows) if "k_chol_solve_lds" in r[0]]
wall, busy, gaps, ngap = [], [], [], []
for a, b in zip(chol[:-1], chol[1:]):
    w = (rows[b][1] - rows[a][1]) / 1e3
    if w > 200.0 or b - a != 5:      # a new solve (upload, reset) or a rejected step in between
        continue
    seg = rows[a:b]
    k = sum(e - s for _, s, e in seg) / 1e3
    g = [(seg[i + 1][1] - seg[i][2]) / 1e3 for i in range(len(seg) - 1)] + [(rows[b][1] - seg[-1][2]) / 1e3]
    wall.append(w); busy.append(k); gaps.append(sum(max(x, 0.0) for x in g)); ngap.append(len(g))
wall, busy, gaps = np.array(wall), np.array(busy), np.array(gaps)
print("iterations analysed: %d (five launches each, inside a solve)" % len(wall))
print("per iteration: wall %.2f us (median %.2f), kernels %.2f us, idle between kernels %.2f us = %.1f %% of the wall time, %.2f us per boundary"
      % (wall.mean(), np.median(wall), busy.mean(), gaps.mean(), 100.0 * gaps.mean() / wall.mean(), gaps.mean() / 5.0))
print("a loop with no host in it and no kernel boundaries could gain at most %.1f %%: %.0f -> %.0f GN iterations/s"
      % (100.0 * gaps.mean() / wall.mean(), 1e6 / wall.mean(), 1e6 / busy.mean()))
