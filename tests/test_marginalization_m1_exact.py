"""M1 pinned independently: MarginalizationError::addResidualBlock restated at 40 digits from the raw residual
definitions (tests/mp_m1.py) and chained with the 40-digit M2 (tests/mp_marg.py) over a sliding window, so that the arbiter of
the marginalisation no longer starts from the oracle's own post-M1 system.  CPU part: the oracle against the exact chain.  GPU
part (tests/test_gpu_parity.py::test_marginalization_m1_against_exact_chain): the HIP path's post-M1 system and final prior
against the same chain."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from svin_amd import synthetic as syn   # noqa: E402


def tiny_sequence_spec():
    return syn.make_window(P=9, L=60, n_obs=700, seed=52, rig="euroc", keyframe_every=2, frame_dt=0.3)


def scaled(Ha, ba, Hb, bb):
    sd = np.sqrt(np.abs(np.diag(Hb)))
    sd[sd == 0] = 1.0
    return float(np.max(np.abs(Ha - Hb) / np.outer(sd, sd))), float(np.max(np.abs(ba - bb) / sd)), float(np.max(np.abs(bb) / sd))


def test_oracle_m1_matches_the_exact_restatement_over_a_sequence():
    from oracle import orc
    import mp_m1
    spec = tiny_sequence_spec()
    est = orc.OracleEstimator()
    chain = mp_m1.ExactChain()
    seen, worst = set(), [0.0, 0.0]

    def cb(k, fid):
        if k < 4:
            return
        est.optimize(6)
        ok, removed = est.apply_marginalization(2, 2)
        assert ok
        log, pre = est.marg_m1_log(), est.marg_pre()
        seen.update(e["kind"] for e in log["log"])
        H, b0 = chain.m1(log)
        dH, db, bs = scaled(pre["H"], pre["b0"], H, b0)
        # after M1: accumulation of J^T J in double against the exact sums -- rounding of well-conditioned products
        assert dH < 1e-11 and db < 1e-11 * max(1.0, bs), (k, dH, db, bs)
        ex = chain.m2(pre["lm"], pre["dense"])
        m = est.marg()
        dH2, db2, bs2 = scaled(m["H"], m["b0"], ex["H"], ex["b0"])
        worst[0], worst[1] = max(worst[0], dH2), max(worst[1], db2 / max(1.0, bs2))
        # after M2 the Schur complement of a 1e8-prior window: 1e-8 of a standard deviation (the bar of the one-shot test)
        assert dH2 < 1e-8 and db2 < 1e-8 * max(1.0, bs2), (k, dH2, db2, bs2)
    syn.feed(est, spec, on_frame=cb)
    # reprojection (with its Cauchy corrector), IMU and speed/bias-prior residuals all went through M1
    assert {0, 1, 3}.issubset(seen), seen
    print("oracle vs exact chain after M2, worst over the sequence: H %.2e b0 %.2e" % tuple(worst))
