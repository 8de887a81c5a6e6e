"""K1 (reprojection residual + Jacobian evaluation) on an HBM-resident batch of window replicas: the launch the roofline
object of bench.py describes.  Run under `rocprofv3 --kernel-trace --stats` for the kernel-trace average and under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for the HBM traffic (tools/pmc_summary.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from svin_amd import synthetic as syn
from svin_amd.estimator import Estimator
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = syn.make_window(); est = Estimator(0); syn.feed(est, spec)
ms_each, ms_b2b, by = est.bench_jacobian_eval_b2b(copies, 20)
print("K1 batched, %d replicas, %.1f MB algorithmic per launch: per-launch events %.4f ms = %.1f GB/s; back to back %.4f ms = %.1f GB/s"
      % (copies, by / 1e6, ms_each, by / ms_each / 1e6, ms_b2b, by / ms_b2b / 1e6))
