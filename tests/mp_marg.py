"""Independent high-precision restatement of MarginalizationError::marginalizeOut / updateErrorComputation (M2 / M3),
used to ARBITRATE between two double-precision implementations (the oracle and the HIP path): mpmath, 40 digits.
Written from the definitions (reference: okvis_ceres/src/MarginalizationError.cpp:557-667 and :725-758,
include/okvis/ceres/implementation/MarginalizationError.hpp:170-220), it shares no code with either.

  preconditioner   p_i = sqrt(H_ii) if H_ii > 1e-9 else 1e-3 ;  H <- P^-1 H P^-1 ,  b <- P^-1 b
  landmark part    per 3x3 block V: V^+ from its symmetric eigen-decomposition, eigenvalues <= eps * 3 * lambda_max dropped;
                   H_aa -= W V^+ W^T , b_a -= W V^+ b_b ; un-scale
  dense part       the same with ONE block V (symmetrised), threshold eps * n_m * lambda_max
  M3               H = P U S U^T P with eigenvalues <= eps * n * lambda_max dropped:  J^T J = P U S U^T P,
                   J^T e0 = -P U (S S^+) U^T P^-1 b0
eps is the DOUBLE-precision epsilon: the rank decisions are the reference's, the arithmetic behind them is exact.
"""
import mpmath as mp
import numpy as np

EPS = mp.mpf(2) ** -52


def _precondition(H, b):
    n = H.rows
    p = [mp.sqrt(H[i, i]) if H[i, i] > mp.mpf("1e-9") else mp.mpf("1e-3") for i in range(n)]
    Hs = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            Hs[i, j] = H[i, j] / (p[i] * p[j])
    bs = mp.matrix([b[i] / p[i] for i in range(n)])
    return p, Hs, bs


def _pinv_sym(V):
    n = V.rows
    Vs = (V + V.T) / 2
    E, Q = mp.eigsy(Vs)
    lmax = max(E)
    tol = EPS * n * lmax
    out = mp.matrix(n, n)
    dropped = 0
    for k in range(n):
        if E[k] > tol:
            for i in range(n):
                for j in range(n):
                    out[i, j] += Q[i, k] * Q[j, k] / E[k]
        else:
            dropped += 1
    return out, dropped, [E[k] / lmax for k in range(n)]


def _stage(H, b, ranges, per_block):
    n = H.rows
    p, Hs, bs = _precondition(H, b)
    marg = [r + k for r, m in ranges for k in range(m)]
    ms = set(marg)
    keep = [i for i in range(n) if i not in ms]
    na = len(keep)
    dH, db = mp.matrix(na, na), mp.matrix(na, 1)
    info = []
    blocks = [[r + k for k in range(m)] for r, m in ranges] if per_block else [marg]
    for blk in blocks:
        nm = len(blk)
        V = mp.matrix(nm, nm)
        for i in range(nm):
            for j in range(nm):
                V[i, j] = Hs[blk[i], blk[j]]
        Vp, dropped, rel = _pinv_sym(V)
        info.append((dropped, min(rel)))
        W = mp.matrix(na, nm)
        for i in range(na):
            for j in range(nm):
                W[i, j] = Hs[keep[i], blk[j]]
        WV = W * Vp
        dH += WV * W.T
        db += WV * mp.matrix([bs[i] for i in blk])
    Hn, bn = mp.matrix(na, na), mp.matrix(na, 1)
    for i in range(na):
        bn[i] = p[keep[i]] * (bs[keep[i]] - db[i])
        for j in range(na):
            Hn[i, j] = p[keep[i]] * (Hs[keep[i], keep[j]] - dH[i, j]) * p[keep[j]]
    return Hn, bn, keep, info


def marginalize_mp(H, b0, lm_ranges, dense_ranges, dps=40):
    """returns dict(H, b0 after M2; JtJ, Jte0, rank after M3; diagnostics) as float arrays"""
    mp.mp.dps = dps
    Hm = mp.matrix(H.tolist())
    bm = mp.matrix(b0.tolist())
    diag = {}
    if lm_ranges:
        n0 = Hm.rows
        Hm, bm, keep, info = _stage(Hm, bm, lm_ranges, per_block=True)
        diag["lm_dropped"] = sum(d for d, _ in info)
        # dense ranges were given in the old ordering: shift
        pos = {old: new for new, old in enumerate(keep)}
        dense_ranges = [(pos[r], m) for r, m in dense_ranges]
    if dense_ranges:
        Hm, bm, keep, info = _stage(Hm, bm, dense_ranges, per_block=False)
        diag["dense_dropped"], diag["dense_min_rel_eig"] = info[0][0], float(info[0][1])
    n = Hm.rows
    p, Hs, bs = _precondition(Hm, bm)
    E, Q = mp.eigsy((Hs + Hs.T) / 2)
    lmax = max(E)
    tol = EPS * n * lmax
    kept = [k for k in range(n) if E[k] > tol]
    diag["rank"] = len(kept)
    diag["rel_eigs_small"] = sorted(float(E[k] / lmax) for k in range(n))[:6]
    JtJ, Pm = mp.matrix(n, n), mp.matrix(n, n)
    for k in kept:
        for i in range(n):
            for j in range(n):
                JtJ[i, j] += p[i] * Q[i, k] * E[k] * Q[j, k] * p[j]
                Pm[i, j] += Q[i, k] * Q[j, k]
    Jte0 = mp.matrix(n, 1)
    for i in range(n):
        Jte0[i] = -p[i] * sum(Pm[i, j] * bs[j] for j in range(n))

    def arr(M, r, c):
        return np.array([[float(M[i, j]) for j in range(c)] for i in range(r)])
    return dict(H=arr(Hm, n, n), b0=arr(bm, n, 1)[:, 0], JtJ=arr(JtJ, n, n), Jte0=arr(Jte0, n, 1)[:, 0], **diag)
