import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec=syn.make_window(); est=Estimator(0); syn.feed(est,spec)
ms, by = est.bench_jacobian_eval(256, 20)
print("K1 batched: %.4f ms  %.1f GB/s" % (ms, by/ms/1e6))
