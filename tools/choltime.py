import sys; sys.path.insert(0,'.')
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec=syn.make_window(); est=Estimator(0); syn.feed(est,spec)
print(est.bench_kernel_times(20))
