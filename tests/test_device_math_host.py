"""Host-side check of the product's DEVICE math header against the oracle (no GPU needed).

``svin_amd/csrc/dmath.hpp`` is the code the HIP kernels run per lane; ``SVIN_HD`` lets g++ compile the very
same functions for the host.  This test compiles a tiny shim (tests/csrc/dmath_host_shim.cpp) and compares
reprojection residuals / minimal Jacobians and the pose manifold operations with the CPU oracle for all four
distortion models.  (The GPU parity tests proper live in test_gpu_parity.py and go through the C ABI.)
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
pd = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def shim():
    out = os.path.join(tempfile.gettempdir(), "svin_dmath_host_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out,
                           os.path.join(HERE, "csrc", "dmath_host_shim.cpp")])
    return C.CDLL(out)


def d(a):
    return a.ctypes.data_as(pd)


def rand_pose(rng, tr=1.0, rot=0.5):
    a = rng.uniform(-rot, rot, 3)
    th = np.linalg.norm(a)
    return np.r_[rng.uniform(-tr, tr, 3), np.sin(th / 2) * a / th, np.cos(th / 2)]


def apply(T, p):
    x, y, z, w = T[3:]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return np.r_[R @ p[:3] + T[:3] * p[3], p[3]]


MODELS = {orc.DIST_NONE: [], orc.DIST_RADTAN: [-0.28, 0.07, 0.0002, 0.00002],
          orc.DIST_EQUIDISTANT: [-0.21, 0.14, 0.0006, 0.0003],
          orc.DIST_RADTAN8: [-0.16, 0.15, 0.0003, 0.0002, 0.01, 0.02, -0.01, 0.005]}


@pytest.mark.parametrize("model", sorted(MODELS))
def test_reprojection_device_math_matches_oracle(shim, model):
    rng = np.random.default_rng(100 + model)
    L = orc.lib()
    intr = [458.0, 457.0, 367.0, 248.0]
    dist = MODELS[model]
    cam = np.zeros(12)
    cam[:4] = intr
    cam[4:4 + len(dist)] = dist
    m = orc.OracleMap()
    pid = 1
    worst = 0.0
    for k in range(200):
        T_WS, T_SC = rand_pose(rng), rand_pose(rng, 0.2, 0.2)
        TW = np.zeros(7)
        L.orc_transformation_compose(d(T_WS), d(T_SC), d(TW))
        hw = 1.0 if k % 5 else rng.uniform(0.5, 2.0)          # homogeneous scale
        z = rng.uniform(0.05, 8.0) if k % 7 == 0 else rng.uniform(1.0, 8.0)  # some points closer than 0.2 m (invalid)
        pc = np.r_[rng.uniform(-0.5, 0.5, 2) * z, z, 1.0]
        hp = apply(TW, pc) * hw
        uv = np.array([300.0, 200.0]) + rng.normal(size=2) * 30
        size = rng.uniform(4, 12)
        w = np.sqrt(64.0 / (size * size))
        m.add_param(pid, orc.BLOCK_POSE, T_WS)
        m.add_param(pid + 1, orc.BLOCK_HPOINT, hp)
        m.add_param(pid + 2, orc.BLOCK_POSE, T_SC)
        rid = m.add_reproj(model, intr, dist, uv, [[w * w, 0], [0, w * w]], orc.LOSS_NONE, pid, pid + 1, pid + 2)
        r, Js, Jm = m.eval(rid)
        ro, Jp, Jl, Je = np.zeros(2), np.zeros(12), np.zeros(6), np.zeros(12)
        shim.hd_reproj(d(cam), model, d(T_WS), d(hp), d(T_SC), C.c_double(uv[0]), C.c_double(uv[1]), C.c_double(w), d(ro),
                       d(Jp), d(Jl), d(Je))
        for a, b in ((ro, r), (Jp.reshape(2, 6), Jm[0]), (Jl.reshape(2, 3), Jm[1]), (Je.reshape(2, 6), Jm[2])):
            worst = max(worst, np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))
        pid += 3
    assert worst < 1e-11, worst


def test_pose_manifold_device_math_matches_oracle(shim):
    rng = np.random.default_rng(7)
    L = orc.lib()
    for _ in range(100):
        x = rand_pose(rng)
        delta = np.r_[rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.2]
        xo, xc = np.zeros(7), np.zeros(7)
        shim.hd_pose_oplus(d(x), d(delta), d(xo))
        L.orc_manifold_plus(orc.BLOCK_POSE, d(x), d(delta), d(xc))
        assert np.max(np.abs(xo - xc)) < 1e-14
        dm, dc = np.zeros(6), np.zeros(6)
        shim.hd_pose_minus(d(xo), d(x), d(dm))
        L.orc_manifold_minus(orc.BLOCK_POSE, d(xc), d(x), d(dc))
        assert np.max(np.abs(dm - dc)) < 1e-14
