"""The eigen-solver of the marginalisation prior on the GPU (svin_amd/csrc/symeig.hpp through svin_ba_debug_sym_eig: one
workgroup, tridiagonalisation + divide and conquer) against LAPACK: real Jacobi-scaled priors of the stereo_rig_v2 and EuRoC
sliding windows and the hard cases of tests/helpers/sym_eig_cases.py.  FP64, absolute accuracy eps |A| like any backward-stable
symmetric solver (Eigen's, which the reference calls at MarginalizationError.cpp:732, included): orthogonality,
reconstruction and eigenvalues within 40 n eps of |A|."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import sym_eig_cases  # noqa: E402

pytestmark = pytest.mark.gpu
CASES = sym_eig_cases.cases()
EPS = 2.220446049250313e-16


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_eigen_solver_matches_lapack(gpu_lib, name):
    from svin_amd.estimator import Estimator
    A = CASES[name]
    n = A.shape[0]
    lam, X, ms = Estimator.debug_sym_eig(A)
    orth, recon, dlam = sym_eig_cases.check(A, lam, X)
    print("%s: n %d, %.0f us, orthogonality %.1e, reconstruction %.1e, eigenvalues %.1e" % (name, n, 1e3 * ms, orth, recon, dlam))
    assert np.all(np.diff(lam) >= 0)
    assert orth < 40 * n * EPS and recon < 40 * n * EPS and dlam < 40 * n * EPS


def test_device_eigen_solver_is_deterministic(gpu_lib):
    from svin_amd.estimator import Estimator
    A = CASES["rig_v2_n117"]
    a, b = Estimator.debug_sym_eig(A), Estimator.debug_sym_eig(A)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
