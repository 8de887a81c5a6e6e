"""The round-6 forms of the wide-window iteration against each other: the side lane (small factors and the speed / bias chain's
factorisation beside the landmark elimination, DeviceProblem::sideLane), the split block rows of k_schur_rows' work list and the
split evaluation / fused step are ORDERINGS of the same sums -- a wide window solved with each of them switched off must end
where the default path ends, to rounding.  (Parity of the default path with the oracle: test_wide_window_* / test_config4_* in
tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(32, 12000, 120000, 1), (40, 8000, 64000, 5)])
def test_wide_window_orderings_agree(gpu_lib, debug_option, shape):
    from svin_amd.estimator import Estimator
    P, L, N, seed = shape
    spec = syn.make_window(P=P, L=L, n_obs=N, seed=seed, frame_dt=0.25)

    def run(**opts):
        for k, v in opts.items():
            debug_option(k, v)
        est = Estimator(0)
        fids, _ = syn.feed(est, spec)
        for _ in range(2):
            est.optimize(4)
        s = est.summary()
        for k in opts:
            debug_option(k, 0)
        return s["final_cost"], s["iterations"], np.stack([est.get_T_WS(f) for f in fids])

    ref = run()
    for opts in (dict(SVIN_NO_SB_EARLY=1), dict(SVIN_NO_ROW_SPLIT=1), dict(SVIN_NO_EVAL_SPLIT=1), dict(SVIN_PANELS_OLD=1)):
        c, it, T = run(**opts)
        assert it == ref[1], opts
        assert abs(c - ref[0]) < 1e-11 * ref[0], (opts, c, ref[0])
        assert np.max(np.abs(T - ref[2])) < 1e-9, (opts, float(np.max(np.abs(T - ref[2]))))
