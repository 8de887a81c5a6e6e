"""The algorithm of the wide-window solver's speed / bias chain elimination (svin_amd/csrc/kernels.hip, K6'''), replayed on the
CPU (tools/chain_elim_replay.py: the level schedule and per-block quantities the kernels use) against a dense solve: chains of
2 ... 70 blocks, powers of two and not, with the scale spread of a real window (IMU information ~1e10 next to ~1e3).  The device
kernels are held against the same replay and against LAPACK in tests/test_gpu_reduced_solve.py / tools/dbg/sb_elim_dbg.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import chain_elim_replay as cr   # noqa: E402


def chain_system(n, dK, rng, spread=1.0):
    """SPD system [kept dK rows, dense | n chain blocks of 9 rows]: block-tridiagonal chain part, dense coupling to the kept rows"""
    d = dK + 9 * n
    H = np.zeros((d, d))
    A = rng.normal(size=(dK + 20, dK))
    H[:dK, :dK] = A.T @ A
    scale = 10.0 ** rng.uniform(0, spread, size=d)      # per-unknown scale: the blocks span `spread` decades
    for b in range(n):
        r = dK + 9 * b
        Bm = rng.normal(size=(12, 9))
        H[r:r + 9, r:r + 9] += Bm.T @ Bm + 3 * np.eye(9)
        if b > 0:
            Cb = 0.3 * rng.normal(size=(9, 9))
            H[r:r + 9, r - 9:r] = Cb
            H[r - 9:r, r:r + 9] = Cb.T
        W = 0.1 * rng.normal(size=(9, dK))
        H[r:r + 9, :dK] = W
        H[:dK, r:r + 9] = W.T
    H += 0.6 * d * np.eye(d)                              # diagonally dominant: SPD whatever the draws
    H = H * np.outer(scale, scale)
    return H, rng.normal(size=d) * scale


@pytest.mark.parametrize("n", [2, 3, 4, 7, 8, 13, 16, 29, 48, 64, 70])
def test_replay_equals_dense_solve(n):
    rng = np.random.default_rng(100 + n)
    H, g = chain_system(n, 30, rng)
    x, mid = cr.solve(H, g, 30, n)
    ref = np.linalg.solve(H, g)
    assert np.abs(x - ref).max() <= 1e-12 * np.abs(ref).max()
    # the Schur complement the kept rows see is the dense one
    S = H[:30, :30] - H[:30, 30:] @ np.linalg.solve(H[30:, 30:], H[30:, :30])
    assert np.abs(mid["M"] - S).max() <= 1e-11 * np.abs(S).max()


def test_levels_cover_every_block_once():
    for n in range(1, 80):
        seen = [b for s in cr.levels(n) for b in cr.eliminated(n, s)]
        assert sorted(seen + [0]) == list(range(n)), n
        for s in cr.levels(n):
            for b in cr.eliminated(n, s):
                assert b - s in cr.survivors(n, s) and (b + s >= n or b + s in cr.survivors(n, s))


def test_scale_spread_of_a_real_window():
    """five decades between the blocks' scales (1e10 in H): the factor's blocks are inverted explicitly, the solution still holds
    to 1e-9 of its own scale entry by entry"""
    rng = np.random.default_rng(5)
    H, g = chain_system(32, 48, rng, spread=5.0)
    x, _ = cr.solve(H, g, 48, 32)
    ref = np.linalg.solve(H, g)
    assert np.all(np.abs(x - ref) <= 1e-9 * np.abs(ref) + 1e-12 * np.abs(ref).max())
