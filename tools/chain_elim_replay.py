"""Host replay (numpy) of the speed / bias chain elimination of the wide-window solver (svin_amd/csrc/kernels.hip, K6'''):
cyclic reduction over the 9 x 9 blocks of a block-tridiagonal S_ss, forward sweep of [S_sk | g_s], Schur complement onto the kept
rows, backward sweep.  The kernels follow exactly this level schedule and these per-block quantities (G, F_lo, F_hi), so the
replay is what tools/dbg/sb_elim_dbg.py compares the device intermediates against and what tests/test_chain_elimination_host.py
checks against a dense solve on the CPU.  Test / debug infrastructure: nothing in the product imports it."""
import numpy as np

B = 9


def levels(n):
    """strides s = 1, 2, 4, ... < n; at stride s the blocks b = s (mod 2 s) are eliminated, their neighbours b - s, b + s survive"""
    s = 1
    while s < n:
        yield s
        s *= 2


def eliminated(n, s):
    return range(s, n, 2 * s)


def survivors(n, s):
    return range(0, n, 2 * s)


def factor(H, dK, n):
    """records of the chain's factor: G[b] = L_bb^-1, F_lo[b] = G S(b, b - s), F_hi[b] = G S(b + s, b)^T (S as level s sees it)"""
    blk = lambda b: slice(dK + B * b, dK + B * b + B)   # noqa: E731
    D = [H[blk(b), blk(b)].copy() for b in range(n)]
    C = [(H[blk(b), blk(b - 1)].copy() if b > 0 else np.zeros((B, B))) for b in range(n)]   # coupling with the lower active neighbour
    G = [None] * n
    Flo = [np.zeros((B, B)) for _ in range(n)]
    Fhi = [np.zeros((B, B)) for _ in range(n)]
    for s in levels(n):
        for b in eliminated(n, s):
            G[b] = np.linalg.inv(np.linalg.cholesky(D[b]))
            Flo[b] = G[b] @ C[b]
            if b + s < n:
                Fhi[b] = G[b] @ C[b + s].T
        newC = {}
        for m in survivors(n, s):
            if m >= s:
                D[m] = D[m] - Fhi[m - s].T @ Fhi[m - s]
            if m + s < n:
                D[m] = D[m] - Flo[m + s].T @ Flo[m + s]
            if m >= 2 * s:
                newC[m] = -Fhi[m - s].T @ Flo[m - s]
        for m, v in newC.items():
            C[m] = v
    G[0] = np.linalg.inv(np.linalg.cholesky(D[0]))
    return G, Flo, Fhi


def forward(W, G, Flo, Fhi):
    """Y = L^-1 W for the rows of the chain (W: 9 n x columns), level by level"""
    n = len(G)
    w = W.copy()
    blk = lambda b: slice(B * b, B * b + B)   # noqa: E731
    for s in levels(n):
        for b in eliminated(n, s):
            w[blk(b)] = G[b] @ w[blk(b)]
        for m in survivors(n, s):
            if m >= s:
                w[blk(m)] -= Fhi[m - s].T @ w[blk(m - s)]
            if m + s < n:
                w[blk(m)] -= Flo[m + s].T @ w[blk(m + s)]
    w[blk(0)] = G[0] @ w[blk(0)]
    return w


def backward(t, G, Flo, Fhi):
    """x = L^-T t, from the last eliminated block to the first"""
    n = len(G)
    x = t.copy()
    blk = lambda b: slice(B * b, B * b + B)   # noqa: E731
    x[blk(0)] = G[0].T @ x[blk(0)]
    for s in reversed(list(levels(n))):
        for b in eliminated(n, s):
            u = x[blk(b)] - Flo[b] @ x[blk(b - s)]
            if b + s < n:
                u = u - Fhi[b] @ x[blk(b + s)]
            x[blk(b)] = G[b].T @ u
    return x


def solve(H, g, dK, n):
    """the whole path: returns (x, dict of intermediates)"""
    G, Flo, Fhi = factor(H, dK, n)
    Y = forward(np.concatenate([H[dK:, :dK], g[dK:, None]], axis=1), G, Flo, Fhi)
    M = H[:dK, :dK] - Y[:, :dK].T @ Y[:, :dK]
    gk = g[:dK] - Y[:, :dK].T @ Y[:, dK]
    xk = np.linalg.solve(M, gk)
    t = Y[:, dK] - Y[:, :dK] @ xk
    xs = backward(t, G, Flo, Fhi)
    return np.concatenate([xk, xs]), dict(G=G, Flo=Flo, Fhi=Fhi, Y=Y, M=M, gk=gk, xk=xk, t=t)
