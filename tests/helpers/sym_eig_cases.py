"""test matrices of the prior's eigen-solver: real Jacobi-scaled priors (tests/golden/prior_matrices.npz, made by
tools/dump_prior_matrices.py from the oracle's sliding windows) and the textbook hard cases of a symmetric eigen-solver"""
import os

import numpy as np


def cases():
    rng = np.random.default_rng(0)

    def randsym(ev):
        Q, _ = np.linalg.qr(rng.normal(size=(len(ev), len(ev))))
        return (Q * ev) @ Q.T
    out = {}
    Z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "prior_matrices.npz"))
    for k in Z.files:
        out[k] = Z[k]
    wil = np.diag(np.abs(np.arange(-10, 11)).astype(float)) + np.diag(np.ones(20), 1) + np.diag(np.ones(20), -1)
    out.update({
        "identity (everything deflates)": np.eye(40),
        "zero": np.zeros((30, 30)),
        "1 x 1": np.array([[3.0]]),
        "2 x 2": randsym(np.array([1.0, 2.0])),
        "3 x 3 with a double eigenvalue": randsym(np.array([1.0, 2.0, 2.0])),
        "null space of 6 + graded spectrum": randsym(np.r_[np.zeros(6), np.logspace(-8, 0, 54)]),
        "two 10-fold eigenvalues": randsym(np.r_[np.ones(10), 2 * np.ones(10), np.linspace(3, 4, 30)]),
        "20 eigenvalues of multiplicity 5": randsym(np.repeat(np.arange(1.0, 21.0), 5)),
        "Wilkinson W21+": wil,
        "cluster of width 1e-12": randsym(1 + 1e-12 * rng.normal(size=60)),
        "20 eigenvalues 1e-10 apart": randsym(np.r_[1 + 1e-10 * np.arange(20), np.linspace(2, 3, 40)]),
        "rank one": np.outer(np.ones(50), np.ones(50)),
        "indefinite, 30 eigenvalues 3e-9 apart": randsym(np.r_[-1e-3 * np.ones(5), 1 + 3e-9 * np.arange(30), np.linspace(2, 3, 40)]),
        "graded 1e-15 .. 3, n = 128": randsym(np.logspace(-15, 0.5, 128)),
        "random, n = 127": randsym(rng.normal(size=127)),
        "random, n = 17": randsym(rng.normal(size=17)),
        "already tridiagonal": np.diag(rng.normal(size=33)) + np.diag(rng.normal(size=32), 1) + np.diag(rng.normal(size=32), -1),
    })
    out["already tridiagonal"] = 0.5 * (out["already tridiagonal"] + out["already tridiagonal"].T)
    return out


def check(A, lam, X):
    """(orthogonality, reconstruction, eigenvalue error) relative to |A|"""
    n = A.shape[0]
    scale = max(np.abs(A).max(), 1e-300)
    ref = np.linalg.eigvalsh(A)
    return (float(np.abs(X.T @ X - np.eye(n)).max()), float(np.abs((X * lam) @ X.T - A).max() / scale),
            float(np.abs(np.sort(lam) - ref).max() / scale))
