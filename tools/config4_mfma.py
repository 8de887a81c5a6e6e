"""The bench window of config #4 (64 KF / 50 000 landmarks / 500 000 residuals, seed 20250629, frames 0.25 s apart) for the counter
pass that measures the MFMA flops the wide-window Schur complement EXECUTES against the algorithmic count sum_l 3 (6 n_l)^2
(bench.py's config4_single_gpu.roofline): run under `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace`, then
`python tools/config4_mfma.py --summarise <results.db> <out.json> [kernel name, default k_schur_rows]`.  Round 6: the product
kernel is k_schur_rows (one v_mfma_f64_4x4x4_4b per quarter of an (8 x 8 padded) 6 x 6 block pair; the counter counts 512 flops
per unit for either MFMA shape); `k_schur_panels` with SVIN_PANELS_OLD=1 in the environment of the measured run is the round-5 form."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np


def spec():
    from svin_amd import synthetic as syn
    return syn.make_window(P=64, L=50000, n_obs=500000, seed=20250629, frame_dt=0.25)


if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ik, ic, iv = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    kname = sys.argv[4] if len(sys.argv) > 4 else "k_schur_rows"
    vals = [r[iv] for r in db.execute("select * from counters_collection") if kname in r[ik] and r[ic] == "SQ_INSTS_VALU_MFMA_MOPS_F64"]
    dur = [d for (d,) in db.execute("select end-start from kernels where name like '%%%s%%'" % kname)]
    sp = spec()
    n_l = np.bincount(sp.obs_lm, minlength=sp.L).astype(float)
    alg = float(np.sum(3.0 * (6.0 * n_l) ** 2))
    exe = 512.0 * float(np.mean(vals))
    out = dict(kernel=kname, workload="config #4 bench window (64 KF / 50000 landmarks / 500000 residuals, seed 20250629)",
               launches=len(vals), executed_mfma_flops_per_launch=exe, algorithmic_flops_per_launch=alg, executed_over_algorithmic=exe / alg,
               mean_launch_us_under_counters=float(np.mean(dur)) / 1e3,
               source="rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -- python tools/config4_mfma.py (MOPS x 512 flops; "
                      "tools/run_r06_profiles.sh)")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))
else:
    from svin_amd import synthetic as syn
    from svin_amd.estimator import Estimator
    if os.environ.get("SVIN_PANELS_OLD"):
        Estimator.debug_set_option("SVIN_PANELS_OLD", 1)
    est = Estimator(0)
    syn.feed(est, spec())
    est.optimize(3)
    print(est.summary())
