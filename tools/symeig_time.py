"""Stage timeline of the prior's eigen-solver (svin_amd/csrc/symeig.hpp) on the fixture priors.  Needs a timing build:
  tools/build_variant.sh symeigtiming -DSVIN_SYMEIG_TIMING && SVIN_BA_LIB=build/variants/symeigtiming.so python tools/symeig_time.py
The library prints the 100 MHz stamps of the last launch: start | tridiagonalisation | reflector spill + leaves | per merge level:
order, deflation, rotations, secular roots, z-hat, norms, output columns, products, write-back | ... | back-transformation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
import sym_eig_cases  # noqa: E402
from svin_amd.estimator import Estimator  # noqa: E402

C = sym_eig_cases.cases()
for name in (sys.argv[1:] or ["rig_v2_n117", "rig_v2_n105", "euroc_n45"]):
    lam, X, ms = Estimator.debug_sym_eig(C[name])
    print("%s: %.1f us" % (name, 1e3 * ms), sym_eig_cases.check(C[name], lam, X))
