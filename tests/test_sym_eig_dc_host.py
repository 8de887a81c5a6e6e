"""The algorithm of svin_amd/csrc/symeig.hpp (the eigen-solver of the marginalisation prior, M3:
MarginalizationError.cpp:725-758 calls Eigen::SelfAdjointEigenSolver) replayed in numpy -- tools/sym_eig_dc_replay.py:
Householder tridiagonalisation, divide and conquer from 1 x 1 leaves with deflation, the secular iteration, Gu / Eisenstat's
weights, back-transformation -- against LAPACK on real priors and on the hard cases (multiple and clustered eigenvalues,
null spaces, graded spectra).  Pins the algorithm on the CPU; tests/test_gpu_sym_eig.py holds the kernel to the same bars."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
import sym_eig_dc_replay as replay  # noqa: E402
import sym_eig_cases  # noqa: E402

CASES = sym_eig_cases.cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_replay_matches_lapack(name):
    A = CASES[name]
    n = A.shape[0]
    st = []
    lam, X = replay.sym_eig(A, st)
    orth, recon, dlam = sym_eig_cases.check(A, lam, X)
    print("%s: n %d, orthogonality %.1e, reconstruction %.1e, eigenvalues %.1e, secular iterations max %d" %
          (name, n, orth, recon, dlam, max(st) if st else 0))
    assert np.all(np.diff(lam) >= 0)
    assert orth < 40 * n * replay.EPS and recon < 40 * n * replay.EPS and dlam < 40 * n * replay.EPS
    assert not st or max(st) < 40


def test_secular_roots_interlace_and_are_accurate():
    """the secular solver on its own: roots strictly between the poles, f(root) at rounding level (mpmath)"""
    import mpmath as mp
    rng = np.random.default_rng(3)
    for trial in range(20):
        K = int(rng.integers(2, 40))
        dt = np.sort(rng.normal(size=K))
        if trial % 3 == 0:
            dt[K // 2] = dt[K // 2 - 1] + 1e-9          # two poles close together
        z2 = rng.uniform(1e-3, 1.0, size=K) ** 2
        if trial % 4 == 0:
            z2[rng.integers(0, K)] = 1e-20             # a pole with next to no weight
        rho = float(rng.uniform(0.1, 3.0))
        mp.mp.dps = 60
        for i in range(K):
            org, tau = replay.secular_root(i, K, dt, z2, rho)
            # the root lies strictly between its poles: measured from the nearer one (lam = dt[org] + tau may round onto it)
            assert org in (i, min(i + 1, K - 1)) and (tau > 0 if org == i else tau < 0)
            assert abs(tau) <= (dt[i + 1] - dt[i] if i + 1 < K else rho * z2.sum())
            f = 1 + rho * sum(mp.mpf(z2[k]) / ((mp.mpf(dt[k]) - mp.mpf(dt[org])) - mp.mpf(tau)) for k in range(K))
            scale = 1 + rho * sum(abs(mp.mpf(z2[k]) / ((mp.mpf(dt[k]) - mp.mpf(dt[org])) - mp.mpf(tau))) for k in range(K))
            assert abs(f) < 200 * replay.EPS * scale, (trial, i, float(f), float(scale))
