"""numpy replay of svin_amd/csrc/symeig.hpp: the eigen-solver of the marginalisation prior (M3) -- Householder
tridiagonalisation, Cuppen's divide and conquer bottom-up from 1 x 1 leaves with dlaed2-style deflation, a fixed-weight
secular iteration with the fitted pole chosen by dominance and an Illinois safeguard, Gu / Eisenstat's recomputed weights,
back-transformation.  Same decisions and formulas as the kernel, scalar where the kernel is one lane (group) per item; used by
tests/test_sym_eig_dc_host.py (against LAPACK) and to study iteration counts (`python tools/sym_eig_dc_replay.py`)."""
import numpy as np

EPS = 2.220446049250313e-16


def tridiag(A):
    """dsytd2 (lower): d, e, the Householder vectors V[:, k] (v[k+1] = 1) and tau"""
    A = A.copy()
    n = A.shape[0]
    V, tau = np.zeros((n, n)), np.zeros(n)
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        alpha, s = x[0], float(np.sum(x[1:] ** 2))
        if s == 0.0:
            continue
        beta = -np.copysign(np.sqrt(alpha * alpha + s), alpha)
        tau[k] = (beta - alpha) / beta
        v = x / (alpha - beta)
        v[0] = 1.0
        V[k + 1:, k] = v
        B = A[k + 1:, k + 1:]
        p = tau[k] * (B @ v)
        w = p - 0.5 * tau[k] * (p @ v) * v
        B -= np.outer(v, w) + np.outer(w, v)
        A[k + 1, k] = A[k, k + 1] = beta
        A[k + 2:, k] = 0
        A[k, k + 2:] = 0
    return np.diag(A).copy(), (np.diag(A, -1).copy() if n > 1 else np.zeros(0)), V, tau


def backtransform(V, tau, Z):
    Z = Z.copy()
    for k in range(V.shape[0] - 3, -1, -1):
        v = V[:, k]
        Z -= tau[k] * np.outer(v, v @ Z)
    return Z


def quad_root_in(c, S, dI, R, dJ, lo, hi, tau):
    """step x with c + S / (dI - x) + R / (dJ - x) = 0 and lo < tau + x < hi; nan if there is none"""
    a, b, cc = c, -(c * (dI + dJ) + S + R), c * dI * dJ + S * dJ + R * dI
    if a == 0:
        x = cc / (-b) if b != 0 else np.nan
        return x if np.isfinite(x) and lo < tau + x < hi else np.nan
    disc = b * b - 4 * a * cc
    if not disc >= 0:
        return np.nan
    q = -0.5 * (b + np.copysign(np.sqrt(disc), b))
    for x in (q / a, (cc / q) if q != 0 else np.nan):
        if np.isfinite(x) and lo < tau + x < hi:
            return x
    return np.nan


def secular_root(i, K, dt, z2, rho, stat=None):
    """root i of 1 + rho sum z2_k / (dt_k - lam): (org, tau) with lam = dt[org] + tau"""
    if K == 1:
        return 0, rho * z2[0]
    last = i == K - 1
    with np.errstate(all="ignore"):
        if not last:
            I, J = i, i + 1
            gap = dt[J] - dt[I]
            mid = 0.5 * gap
            delta = dt - dt[I]
            msk = np.ones(K, bool)
            msk[[I, J]] = False
            rest = 1.0 + rho * np.sum(z2[msk] / (delta[msk] - mid))
            f = rest + rho * z2[I] / (-mid) + rho * z2[J] / (gap - mid)
            org, lo, hi = (I, 0.0, mid) if f > 0 else (J, -mid, 0.0)
            delta = dt - dt[org]
            x = quad_root_in(rest, rho * z2[I], delta[I], rho * z2[J], delta[J], lo, hi, 0.0)
            tau = x if np.isfinite(x) else 0.5 * (lo + hi)
        else:
            I = org = K - 1
            delta = dt - dt[org]
            lo, hi = 0.0, rho * np.sum(z2)
            mid = 0.5 * hi
            rest = 1.0 + rho * np.sum(z2[:K - 2] / (delta[:K - 2] - mid))
            x = quad_root_in(rest, rho * z2[K - 2], delta[K - 2], rho * z2[K - 1], 0.0, lo, hi, 0.0)
            tau = x if np.isfinite(x) else mid
        flo = fhi = None
        side = 0
        So = rho * z2[org]
        for it in range(48):
            den = delta - tau
            t = z2 / den
            t2 = t / den
            left = np.arange(K) <= I if not last else np.ones(K, bool)
            psi, phi = rho * np.sum(t[left]), rho * np.sum(t[~left])
            dpsi, dphi = rho * np.sum(t2[left]), rho * np.sum(t2[~left])
            f = 1.0 + psi + phi
            erretm = 8.0 * (abs(psi) + abs(phi)) + 2.0 + abs(tau) * (dpsi + dphi)
            if abs(f) <= EPS * erretm or f != f:
                break
            if f > 0:
                hi, fhi = tau, f
                if side == 1 and flo is not None:
                    flo *= 0.5
                side = 1
            else:
                lo, flo = tau, f
                if side == -1 and fhi is not None:
                    fhi *= 0.5
                side = -1
            if hi - lo <= 4 * EPS * max(abs(lo), abs(hi)):
                break
            contrib = t2.copy()
            contrib[org] = -1.0
            q = int(np.argmax(contrib))
            dO, dq = den[org], den[q]
            w, dw = psi + phi - So / dO, dpsi + dphi - So / (dO * dO)
            x = quad_root_in(1.0 + (w - dw * dq), So, dO, dw * dq * dq, dq, lo, hi, tau)
            new = tau + x
            if not (np.isfinite(x) and lo < new < hi):
                new = 0.5 * (lo + hi)
                if flo is not None and fhi is not None and fhi != flo:
                    rf = lo - flo * (hi - lo) / (fhi - flo)
                    if lo < rf < hi:
                        new = rf
            tau = new
    if stat is not None:
        stat.append(it)
    return org, tau


def merge(dL, zL, dR, zR, e_k, stat=None):
    """two solved halves (ascending eigenvalues dL, dR; zL = last row of Q1, zR = first row of Q2) torn at e_k"""
    m = len(dL) + len(dR)
    rho = 2.0 * abs(e_k)
    d = np.r_[dL, dR]
    z = np.r_[zL, (1.0 if e_k >= 0 else -1.0) * zR] * 0.70710678118654752440
    order = np.argsort(d, kind="stable")
    tol = 8 * EPS * max(np.abs(d).max(), np.abs(z).max())
    rots, defl, nd = [], [], []
    if rho * np.abs(z).max() <= tol:
        defl = list(order)
    else:
        pj = -1
        for idx in order:
            if rho * abs(z[idx]) <= tol:
                defl.append(idx)
                continue
            if pj < 0:
                pj = idx
                continue
            nj = idx
            s, c = z[pj], z[nj]
            tau = np.hypot(c, s)
            t = d[nj] - d[pj]
            c, s = c / tau, -s / tau
            if abs(t * c * s) <= tol:
                z[nj], z[pj] = tau, 0.0
                rots.append((pj, nj, c, s))
                d[pj], d[nj] = d[pj] * c * c + d[nj] * s * s, d[pj] * s * s + d[nj] * c * c
                defl.append(pj)
            else:
                nd.append(pj)
            pj = nj
        if pj >= 0:
            nd.append(pj)
    K = len(nd)
    dt, zt = d[nd], z[nd]
    roots = [secular_root(i, K, dt, zt * zt, rho, stat) for i in range(K)]
    zh = np.zeros(K)
    for k in range(K):
        prod = 1.0
        for j, (org, tau) in enumerate(roots):
            num = (dt[org] - dt[k]) + tau
            prod *= num if j == k else num / (dt[j] - dt[k])
        zh[k] = np.copysign(np.sqrt(abs(prod) / rho), zt[k])
    V = np.zeros((K, K))
    for j, (org, tau) in enumerate(roots):
        v = zh / ((dt - dt[org]) - tau)
        V[:, j] = v / np.linalg.norm(v)
    lam = np.array([dt[org] + tau for org, tau in roots])
    return dict(rots=rots, defl=defl, nd=nd, lam=lam, V=V, ddefl=d[defl] if defl else np.zeros(0), K=K)


def dc_eig(d, e, stat=None):
    """eigenvalues (ascending) and eigenvectors of the symmetric tridiagonal (d, e)"""
    n = len(d)
    lam = np.asarray(d, float).copy()
    for k in range(n - 1):
        lam[k] -= abs(e[k])
        lam[k + 1] -= abs(e[k])
    Q = np.eye(n)
    b = 1
    while b < n:
        for lo in range(0, n, 2 * b):
            mid, hi = lo + b, min(lo + 2 * b, n)
            if mid >= n:
                continue
            k = mid - 1
            plan = merge(lam[lo:mid], Q[k, lo:mid].copy(), lam[mid:hi], Q[k + 1, mid:hi].copy(), e[k], stat)
            Qb = Q[lo:hi, lo:hi].copy()
            for (p, q, c, s) in plan["rots"]:
                cp, cq = Qb[:, p].copy(), Qb[:, q].copy()
                Qb[:, p], Qb[:, q] = c * cp + s * cq, c * cq - s * cp
            newd = np.r_[plan["lam"], plan["ddefl"]]
            newQ = np.c_[Qb[:, plan["nd"]] @ plan["V"] if plan["K"] else np.zeros((hi - lo, 0)),
                         Qb[:, plan["defl"]] if len(plan["defl"]) else np.zeros((hi - lo, 0))]
            o = np.argsort(newd, kind="stable")
            lam[lo:hi], Q[lo:hi, lo:hi] = newd[o], newQ[:, o]
        b *= 2
    return lam, Q


def sym_eig(A, stat=None):
    """the whole pipeline on a symmetric matrix: eigenvalues ascending, eigenvectors as columns"""
    A = 0.5 * (A + A.T)
    d, e, V, tau = tridiag(A)
    lam, Z = dc_eig(d, e, stat)
    return lam, backtransform(V, tau, Z)


if __name__ == "__main__":
    import sys
    rng = np.random.default_rng(0)

    def randsym(ev):
        Qr, _ = np.linalg.qr(rng.normal(size=(len(ev), len(ev))))
        return (Qr * ev) @ Qr.T
    cases = {"graded 1e-15 .. 3, n = 128": randsym(np.logspace(-15, 0.5, 128)),
             "20 eigenvalues of multiplicity 5": randsym(np.repeat(np.arange(1.0, 21.0), 5))}
    if len(sys.argv) > 1:
        Z = np.load(sys.argv[1])
        cases.update({k: Z[k] for k in Z.files})
    for name, A in cases.items():
        st = []
        lam, X = sym_eig(A, st)
        n = A.shape[0]
        print("%-36s n %3d: secular iterations mean %.1f max %d; orthogonality %.1e, |A - X L X^T| %.1e" %
              (name, n, np.mean(st), max(st), np.abs(X.T @ X - np.eye(n)).max(), np.abs((X * lam) @ X.T - A).max()))
