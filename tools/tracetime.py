"""In-kernel timeline of the last launches of an optimize() call (variant build: tools/build_variant.sh trace -DSVIN_TRACE;
run with SVIN_BA_LIB=build/variants/trace.so).  Prints, per trace point, earliest and latest stamp in microseconds relative to
the earliest stamp of its group."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, '.')
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator, load_library

NAMES = {0: "post: block start", 1: "post: work done", 2: "post: last block in", 3: "post: sums reduced", 4: "post: dogleg coefficients",
         5: "post: blocks retracted", 6: "post: landmarks retracted", 7: "post: block end", 8: "post: landmark blocks done", 9: "post: factor blocks done", 10: "post: camera block done", 11: "post: block sums done", 12: "post: partials loaded",
         16: "eval: block start", 17: "eval: factor block done", 18: "eval: block work done", 19: "eval: last block in", 20: "eval: cost reduced", 21: "eval: sums ready, before mailbox", 22: "eval: after system fence",
         24: "schur: block start", 25: "schur: block end", 28: "reduce: start", 29: "reduce: end"}
spec = syn.make_window()
est = Estimator(0)
syn.feed(est, spec)
L = load_library()
L.svin_debug_trace.argtypes = [C.c_void_p, C.c_int]
for it in (0, 1, 10):
    L.svin_debug_trace(None, 1)
    est.optimize(it)
    out = np.zeros(128, np.uint64)
    L.svin_debug_trace(out.ctypes.data_as(C.c_void_p), 0)
    print("---- after optimize(%d)" % it)
    for grp in (range(0, 13), range(16, 23), range(24, 30)):
        t0 = min((int(out[2 * k]) for k in grp if k in NAMES and out[2 * k + 1] != 0), default=None)
        if t0 is None:
            continue
        for k in grp:
            if k in NAMES and out[2 * k + 1] != 0:
                print("%-28s first %8.2f us   last %8.2f us" % (NAMES[k], (int(out[2 * k]) - t0) / 100.0, (int(out[2 * k + 1]) - t0) / 100.0))
