import os, sys
sys.path.insert(0, '/root/repo')
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec = syn.make_window(seed=20250629)
est = Estimator(0)
syn.feed(est, spec)
print(est.bench_kernel_times(5))
