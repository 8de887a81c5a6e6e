"""FlatMap64 (svin_amd/csrc/flat_map.hpp), the open-addressing map behind the window's residual-id -> landmark and landmark-id ->
handle look-ups, against std::unordered_map on the CPU: random set / erase / find streams over key sets chosen to collide
(sequential ids as okvis hands them out, ids that differ only in high bits, one long probe chain that wraps the table end)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fm") / "libfm.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "csrc", "flat_map_shim.cpp"), "-o", so])
    L = C.CDLL(so)
    L.fm_replay.restype = C.c_long
    L.fm_replay.argtypes = [C.POINTER(C.c_uint64), C.c_long, C.POINTER(C.c_uint64)]
    return L


def replay(L, ops):
    ops = np.ascontiguousarray(ops, dtype=np.uint64)
    size = C.c_uint64(0)
    bad = L.fm_replay(ops.ctypes.data_as(C.POINTER(C.c_uint64)), len(ops), C.byref(size))
    assert bad == -1, "first disagreement with std::unordered_map at operation %d: %s" % (bad, ops[min(bad, len(ops) - 1)])
    return int(size.value)


def stream(rng, keys, n, p_set=0.45, p_erase=0.35):
    kind = rng.choice(3, size=n, p=[p_set, p_erase, 1 - p_set - p_erase])
    return np.stack([kind, rng.choice(keys, size=n), rng.integers(0, 2**63, size=n)], axis=1)


@pytest.mark.parametrize("keys", ["sequential", "high_bits", "strided", "random"])
def test_random_stream(lib, keys):
    rng = np.random.default_rng(7)
    pool = {"sequential": np.arange(1, 6001, dtype=np.uint64),
            "high_bits": (np.arange(1, 3001, dtype=np.uint64) << np.uint64(44)) | np.uint64(5),
            "strided": np.arange(0, 4000 * 1024, 1024, dtype=np.uint64),
            "random": rng.integers(0, 2**64 - 1, size=5000, dtype=np.uint64)}[keys]
    replay(lib, stream(rng, pool, 200000))


def test_growth_and_drain(lib):
    """20 000 inserts (four rehashes from the initial 1024 slots), then every key erased in a shuffled order: the map ends empty
    and every find during the drain agrees"""
    rng = np.random.default_rng(3)
    keys = rng.permutation(np.arange(10, 20010, dtype=np.uint64))
    ins = np.stack([np.zeros_like(keys), keys, keys * np.uint64(3)], axis=1)
    assert replay(lib, ins) == 20000
    order = rng.permutation(keys)
    drain = np.empty((2 * len(order), 3), dtype=np.uint64)
    drain[0::2] = np.stack([np.ones_like(order), order, np.zeros_like(order)], axis=1)
    drain[1::2] = np.stack([np.full_like(order, 2), rng.permutation(order), np.zeros_like(order)], axis=1)
    assert replay(lib, np.concatenate([ins, drain])) == 0


def test_value_zero_and_overwrite(lib):
    """0 is a legal value (the first landmark handle); set on an existing key overwrites without growing"""
    ops = [[0, 42, 0], [2, 42, 0], [0, 42, 9], [2, 42, 0], [1, 42, 0], [2, 42, 0], [1, 42, 0]]
    assert replay(lib, np.array(ops, dtype=np.uint64)) == 0
