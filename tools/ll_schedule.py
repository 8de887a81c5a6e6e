"""Replays the schedule of k_chol_solve_ll (left-looking LDS Cholesky, svin_amd/csrc/kernels.hip): the slot map of the live
tiles never hands out a slot that still holds a live tile, and the factor / forward substitution are right, for 2..17 tile rows."""
# schedule simulation of the left-looking LDS Cholesky: slot map liveness + numerical check
import numpy as np, sys
def run(nT, seed=0):
    h = (nT + 1) // 2
    nslots = (nT - h) * h
    def slot(I, j):
        assert j < I < nT
        if I < h: return j * h + I            # rect slot (h + j, I)
        if j < h: return (I - h) * h + j
        return (j - h) * h + (I - h)          # rect slot (j, I - h)
    rng = np.random.default_rng(seed)
    d = 16 * nT
    A = rng.standard_normal((d, d + 20)); S = A @ A.T + d * np.eye(d)
    g = rng.standard_normal(d)
    T = lambda M, I, J: M[16*I:16*I+16, 16*J:16*J+16]
    lds = [None] * nslots      # (I, j, array)
    def put(I, j, X):
        s = slot(I, j); assert s < nslots
        lds[s] = (I, j, X.copy())
    def get(I, j):
        s = slot(I, j); e = lds[s]
        assert e is not None and e[0] == I and e[1] == j, ("stale", I, j, e and e[:2])
        return e[2]
    Lg = np.zeros((d, d))
    rhs = g.copy()
    # init: accD = S(0,0); column 0 tiles; column 1 init
    accD = T(S, 0, 0).copy()
    col = {I: T(S, 0, I).copy() for I in range(1, nT)}     # transposed tiles C(I,c)^T = S(c, I)
    H1 = None
    for k in range(nT):
        # ---- phase FD(k)
        if k >= 1:
            # F(k-1): final update of column k tiles with j = k-1
            for I in range(k + 1, nT):
                col[I] = col[I] - get(k, k - 1) @ get(I, k - 1).T
            x = get(k, k - 1)
            accD = H1 - x @ x.T
            # wave 4: rhs axpy for column k-1 + write-through
            for I in range(k, nT):
                rhs[16*I:16*I+16] -= get(I, k - 1) @ rhs[16*(k-1):16*k]
                T(Lg, I, k - 1)[:] = get(I, k - 1)
        # D(k): diag + lookahead of column k+1 (reads tiles (k+1, j<k), (I, j<k))
        Lkk = np.linalg.cholesky(accD)
        nxt = {}
        if k + 1 < nT:
            for I in range(k + 1, nT):
                t = T(S, k + 1, I).copy()
                for j in range(k):
                    t -= get(k + 1, j) @ get(I, j).T
                nxt[I] = t
            H1 = nxt.pop(k + 1)
        # ---- B1, phase P(k)
        for I in range(k + 1, nT):
            Xt = np.linalg.solve(Lkk, col[I])          # X^T = L^-1 T
            put(I, k, Xt.T)
        T(Lg, k, k)[:] = Lkk
        rhs[16*k:16*k+16] = np.linalg.solve(Lkk, rhs[16*k:16*k+16])
        col = nxt
        # ---- B2
    L = np.tril(Lg)
    err = np.abs(L @ L.T - S).max() / np.abs(S).max()
    y = np.linalg.solve(np.linalg.cholesky(S), g)
    print(nT, "slots", nslots, "recon err", err, "fwd err", np.abs(rhs - y).max())
for nT in range(2, 18): run(nT)
