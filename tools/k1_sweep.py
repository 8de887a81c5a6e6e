"""K1 (reprojection residual + Jacobian evaluation) throughput against the size of the replica batch: is the HBM-roofline figure
of bench.py a property of the 1 GB batch or of the kernel?  Also a plain device-to-device copy of the same size (the guide's
6.29 TB/s copy ceiling was measured on some size, too)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
spec = syn.make_window(); est = Estimator(0); syn.feed(est, spec)
for copies in (64, 128, 256, 512, 1024, 2048):
    ms_each, ms_b2b, by = est.bench_jacobian_eval_b2b(copies, 10)
    n = int(by // 8 // 2)
    a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    cp = e0.elapsed_time(e1) / 10
    print("replicas %5d  %7.1f MB per launch: K1 per-launch events %.4f ms = %6.1f GB/s, back to back %.4f ms = %6.1f GB/s | copy of %7.1f MB read + as much written: %.4f ms = %6.1f GB/s"
          % (copies, by / 1e6, ms_each, by / ms_each / 1e6, ms_b2b, by / ms_b2b / 1e6, 8.0 * n / 1e6, cp, 16.0 * n / cp / 1e6), flush=True)
    del a, b
