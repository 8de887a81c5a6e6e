"""Debug aid for the speed / bias chain elimination (kernels.hip, k_sb_factor ...): replays the cyclic reduction in numpy on the
system svin_ba_linearize returns and compares every intermediate the kernels leave in the solver scratch (records G / F_lo /
F_hi, Y, the reduced matrix, t) -- says which kernel went wrong first.   python tools/dbg/sb_elim_dbg.py [P]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svin_amd import synthetic as syn           # noqa: E402
from svin_amd.estimator import Estimator        # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mu = 1e-4
rig = sys.argv[2] if len(sys.argv) > 2 else "euroc"
spec = syn.make_window(P=P, L=800, n_obs=8000, seed=11, rig=rig, frame_dt=0.25)
est = Estimator(0)
syn.feed(est, spec)
lin = est.linearize(mu)
H = np.tril(lin["S"]) + np.tril(lin["S"], -1).T
g = lin["g"]
d = lin["d"]
n = P
assert n >= 16, "chains shorter than 16 blocks are not eliminated (planSbElimination): nothing to compare"
dK = d - 9 * n
y_dev = est.debug_reduced_solve(mu)
y_ref = np.linalg.solve(H, g)
print("d", d, "dK", dK, "n", n, "device vs host", np.abs(y_dev - y_ref).max() / np.abs(y_ref).max())

# scratch layout (planSbElimination)
def solver_class(dd):
    nT = (dd + 15) // 16
    if (nT * (nT + 1) // 2 * 16 * 17 + 3 * 16 * nT) * 8 + 48 * 4 <= 156 * 1024:
        return 0
    return 1 if 12 <= nT <= 17 else 2


dp = (dK + 63) // 64 * 64
nb = dp // 64
ldY = (dK + 1 + 15) // 16 * 16
rowsY = (9 * n + 3) // 4 * 4
compact = solver_class(dK) < 2
if compact:
    dpadK = (dK + 15) // 16 * 16
    off0 = 2 * dpadK * dpadK + dpadK
else:
    off0 = (dp + 64) * dp + dp + dp * 64 + ((nb + 3) * nb + 1) // 2 + 2
    off0 = (off0 + 1) & ~1
print("kept system: class", solver_class(dK), "(compact)" if compact else "(blocked)")
REC = 264
Lf = est.debug_peek_solver_scratch(off0, n * REC).reshape(n, REC)
Y = est.debug_peek_solver_scratch(off0 + n * REC, rowsY * ldY).reshape(rowsY, ldY)
tv = est.debug_peek_solver_scratch(off0 + n * REC + rowsY * ldY, rowsY)

# numpy replay (tools/chain_elim_replay.py: the same level schedule and per-block quantities as the kernels)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chain_elim_replay as cr   # noqa: E402
_, mid = cr.solve(H, g, dK, n)
G, Flo, Fhi = mid["G"], mid["Flo"], mid["Fhi"]
level = {0: 0}
for s_ in cr.levels(n):
    for b in cr.eliminated(n, s_):
        level[b] = s_
for b in range(n):
    eg = np.abs(Lf[b, 0:81].reshape(9, 9) - G[b]).max() / np.abs(G[b]).max()
    el = np.abs(Lf[b, 88:169].reshape(9, 9) - Flo[b]).max() / max(np.abs(Flo[b]).max(), 1e-300)
    eh = np.abs(Lf[b, 176:257].reshape(9, 9) - Fhi[b]).max() / max(np.abs(Fhi[b]).max(), 1e-300)
    if max(eg, el, eh) > 1e-9 or not np.isfinite(eg + el + eh):
        print("record of block %d (level s = %d): G %.2e F_lo %.2e F_hi %.2e" % (b, level[b], eg, el, eh))
print("records compared")
w = mid["Y"]
eY = np.abs(Y[:9 * n, :dK + 1] - w)
print("Y: max abs diff %.3e of %.3e; worst row %d col %d; pad rows zero: %s; nan count %d" %
      (np.nanmax(eY), np.abs(w).max(), *np.unravel_index(np.nanargmax(eY), eY.shape), bool(np.all(Y[9 * n:] == 0)), int(np.isnan(Y).sum())))
Mref = H[:dK, :dK] - w[:, :dK].T @ w[:, :dK]
if compact:
    M = est.debug_peek_solver_scratch(dpadK * dpadK, dpadK * dpadK).reshape(dpadK, dpadK)
    print("compact kept matrix: max rel diff %.3e; zero beyond dK: %s" % (np.abs(M[:dK, :dK] - Mref).max() / np.abs(Mref).max(),
          bool(np.all(M[dK:] == 0) and np.all(M[:, dK:] == 0))))
    gk_dev = est.debug_peek_solver_scratch(2 * dpadK * dpadK, dK)
    print("compact right-hand side: %.3e" % (np.abs(gk_dev - (g[:dK] - w[:, :dK].T @ w[:, dK])).max() / np.abs(g[:dK]).max()))
gk = g[:dK] - w[:, :dK].T @ w[:, dK]
xk = np.linalg.solve(Mref, gk)
print("kept part of the solution: device vs replay %.3e" % (np.abs(y_dev[:dK] - xk).max() / np.abs(xk).max()))
t = w[:, dK] - w[:, :dK] @ xk
print("t: %.3e" % (np.abs(tv[:9 * n] - t).max() / np.abs(t).max()))
