"""CPU replay of k_chol_solve_ll's schedule (left-looking LDS Cholesky for 12..17 tile rows, svin_amd/csrc/kernels.hip) with
the kernel's own slot map (`llSlot`, exported as svin_debug_ll_slot): at no step may a slot be handed to a new tile while
the tile it holds is still read later, the slot count must be what the launch reserves, and the factor and the forward
substitution that come out of the replay must be right.  (The numerical path of the kernel itself is covered on the GPU by
tests/test_gpu_parity.py::test_left_looking_lds_solver_sizes.)"""
import ctypes as C

import numpy as np
import pytest

from svin_amd.estimator import load_library


def replay(nT, slot_of, n_slots, seed=0):
    rng = np.random.default_rng(seed)
    d = 16 * nT
    A = rng.standard_normal((d, d + 20))
    S = A @ A.T + d * np.eye(d)
    g = rng.standard_normal(d)
    T = lambda M, I, J: M[16 * I:16 * I + 16, 16 * J:16 * J + 16]
    lds = [None] * n_slots

    def put(I, j, X):
        s = slot_of(I, j)
        assert 0 <= s < n_slots, (I, j, s)
        lds[s] = (I, j, X.copy())

    def get(I, j):
        e = lds[slot_of(I, j)]
        assert e is not None and e[0] == I and e[1] == j, ("tile (%d, %d) was overwritten while live" % (I, j), e and e[:2])
        return e[2]

    Lg = np.zeros((d, d))
    rhs = g.copy()
    accD = T(S, 0, 0).copy()
    col = {I: T(S, 0, I).copy() for I in range(1, nT)}     # transposed tiles C(I, c)^T = S(c, I)
    H1 = None
    for k in range(nT):
        if k >= 1:   # phase F(k - 1): last update of block column k, forward substitution, write-through
            for I in range(k + 1, nT):
                col[I] = col[I] - get(k, k - 1) @ get(I, k - 1).T
            x = get(k, k - 1)
            accD = H1 - x @ x.T
            for I in range(k, nT):
                rhs[16 * I:16 * I + 16] -= get(I, k - 1) @ rhs[16 * (k - 1):16 * k]
                T(Lg, I, k - 1)[:] = get(I, k - 1)
        Lkk = np.linalg.cholesky(accD)   # phase D(k): diagonal tile beside the look-ahead of block column k + 1
        nxt = {}
        if k + 1 < nT:
            for I in range(k + 1, nT):
                t = T(S, k + 1, I).copy()
                for j in range(k):
                    t -= get(k + 1, j) @ get(I, j).T
                nxt[I] = t
            H1 = nxt.pop(k + 1)
        for I in range(k + 1, nT):       # phase P(k): panel solve, X goes to its slot
            put(I, k, np.linalg.solve(Lkk, col[I]).T)
        T(Lg, k, k)[:] = Lkk
        rhs[16 * k:16 * k + 16] = np.linalg.solve(Lkk, rhs[16 * k:16 * k + 16])
        col = nxt
    L = np.tril(Lg)
    return np.abs(L @ L.T - S).max() / np.abs(S).max(), np.abs(rhs - np.linalg.solve(np.linalg.cholesky(S), g)).max()


@pytest.mark.parametrize("nT", [2, 3, 5, 8, 12, 13, 14, 15, 16, 17])
def test_slot_map_never_overwrites_a_live_tile(nT):
    lib = load_library()
    lib.svin_debug_ll_slot.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.svin_debug_ll_slot.restype = C.c_int
    lib.svin_debug_ll_slots.argtypes = [C.c_int]
    lib.svin_debug_ll_slots.restype = C.c_int
    n_slots = lib.svin_debug_ll_slots(nT)
    h = (nT + 1) // 2
    assert n_slots == (nT - h) * h
    assert n_slots * 2048 + 2 * 2048 + 16 * 17 * 8 + 128 + 2 * 16 * nT * 8 <= 156 * 1024   # what the launch asks of LDS
    # the live set peaks at (nT - 1 - k)(k + 1) tiles: the map may not need more slots than that peak
    assert n_slots == max((nT - 1 - k) * (k + 1) for k in range(nT))
    recon, fwd = replay(nT, lambda I, j: lib.svin_debug_ll_slot(I, j, nT), n_slots, seed=nT)
    assert recon < 1e-13 and fwd < 1e-12
