"""The reduced-system solve on its own: (S + mu D) y = g as each device path computes it -- LDS-resident (d <= 176), left-looking
(d <= 272), blocked over many workgroups, and each of the three behind the speed / bias chain elimination (kernels.hip,
k_sb_factor / k_sb_forward / k_sb_load / k_sb_back: cyclic reduction over the 9x9 blocks the IMU factors chain together; the
kept rows then go to whichever dense solver their number selects) -- against a host
solve (numpy, LAPACK) of the very system svin_ba_linearize returns.  FP64 on systems whose entries span ~1e2 (landmark
information) to ~1e10 (IMU information): 1e-10 relative to |y| with mu = 1e-4; with mu = 1e-9 the conditioning of the system
itself separates two correct solvers by ~1e-8 (LAPACK against the device paths that were there before the elimination), so
the bound is 1e-6 there."""
import os

import numpy as np
import pytest

from svin_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture
def sb_elim_switch(gpu_lib):
    """the library reads SVIN_NO_SB_ELIM once per process; tests flip the switch through the debug entry point and leave it off"""
    from svin_amd.estimator import Estimator
    yield lambda off: Estimator.debug_set_switch("SVIN_NO_SB_ELIM", off)
    Estimator.debug_set_switch("SVIN_NO_SB_ELIM", False)


def host_solve(lin):
    S = np.tril(lin["S"]) + np.tril(lin["S"], -1).T
    return np.linalg.solve(S, lin["g"])


def window(P, L, n_obs, rig="euroc", seed=11):
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=P, L=L, n_obs=n_obs, seed=seed, rig=rig, frame_dt=0.25)
    est = Estimator(0)
    syn.feed(est, spec)
    return est


@pytest.mark.parametrize("P,L,n_obs,rig,path", [
    (10, 400, 4000, "euroc", "LDS-resident whole, d = 150 (no elimination)"),
    (12, 500, 5000, "euroc", "d = 180: LDS-resident whole, four border rows (no elimination)"),
    (14, 500, 5000, "euroc", "d = 210: chain of 14, kept 84 rows LDS-resident (was left-looking whole)"),
    (8, 500, 5000, "rig_v2", "stereo_rig_v2, d = 216: chain of 8, kept 144 rows LDS-resident (was left-looking whole)"),
    (10, 600, 6000, "rig_v2", "stereo_rig_v2, d = 270 (config #3's shape): chain of 10, kept 180 rows LDS-resident with four border rows"),
    (16, 600, 6000, "euroc", "d = 240: chain of 16, kept 96 rows LDS-resident (was left-looking)"),
    (18, 700, 7000, "euroc", "d = 270: chain of 18, kept 108 rows LDS-resident (was left-looking)"),
    (24, 800, 8000, "euroc", "d = 360: chain of 24, kept 144 rows LDS-resident (was blocked)"),
    (29, 800, 8000, "euroc", "d = 435: chain of 29 (not a power of two), kept 174 rows LDS-resident (was blocked)"),
    (32, 1000, 10000, "euroc", "d = 480: chain of 32, kept 192 rows: LDS-resident with a border of 16 rows (k_chol_border_prepare on the compact system)"),
    (33, 1000, 10000, "euroc", "d = 495: chain of 33, kept 198 rows: LDS-resident with a border of 22 rows"),
    (40, 1200, 12000, "euroc", "d = 600: chain of 40, kept 240 rows left-looking (was blocked)"),
    (48, 1500, 15000, "euroc", "d = 720: chain of 48, kept 288 rows blocked (padded to 320)"),
    (50, 1500, 15000, "euroc", "d = 750: chain of 50, kept 300 rows blocked (not a multiple of 16: padded to 320)"),
    (64, 2500, 25000, "euroc", "d = 960: chain of 64, kept 384 rows blocked"),
    (64, 2500, 25000, "test4", "per-frame extrinsics, d = 1728: chain of 64, kept 1152 rows blocked"),
])
def test_device_solve_equals_host_solve(gpu_lib, sb_elim_switch, P, L, n_obs, rig, path):
    est = window(P, L, n_obs, rig)
    for mu, tol in ((1e-4, 1e-10), (1e-9, 1e-6)):
        lin = est.linearize(mu)
        y_ref = host_solve(lin)
        scale = np.abs(y_ref).max()
        err = {}
        for off in (False, True):
            sb_elim_switch(off)
            # fused = metric and damping applied in the solver's load phase (k_sb_factor / k_sb_load or the dense solver's own
            # load): the form every iteration of optimize() runs
            for fused in (False, True):
                y = est.debug_reduced_solve(mu, fused=fused)
                assert y.shape == y_ref.shape
                err[off, fused] = np.abs(y - y_ref).max() / scale
        print("%s, mu %g: d %d, |y| %.3g, device vs host %.2e (fused %.2e; chain elimination off: %.2e, fused %.2e)" %
              (path, mu, lin["d"], scale, err[False, False], err[False, True], err[True, False], err[True, True]))
        assert max(err.values()) < tol


@pytest.mark.parametrize("keyframes,imu_frames,frames,rig,sizes", [
    (22, 2, 34, "euroc", (177,)),            # 25 poses + 3 speed / bias blocks: ONE row beyond the eleven tile rows
    (4, 3, 16, "rig_v2", (180,)),            # stereo_rig_v2 + sonar + depth: 8 poses + 16 extrinsics + 4 speed / bias blocks -- four rows beyond
    (23, 3, 36, "euroc", (186, 192, 198)),   # 10, 16, 22 rows beyond: the border block goes through k_chol_border_prepare
    (5, 3, 16, "rig_v2", (186, 198)),        # SVIn's own window (5 keyframes + 3 IMU frames): 198 in its steady state
])
def test_lds_solver_with_border_rows(gpu_lib, keyframes, imu_frames, frames, rig, sizes):
    """d = 177 .. 200 on the LDS-resident solver (k_chol_solve_lds<1> / <2>: the rows beyond 176 are eliminated while the tiles are
    loaded) on the systems of a sliding window that has been through marginalisations (prior + IMU chain + extrinsics chain in the
    reduced system): against LAPACK, and against the left-looking solver the switch SVIN_NO_LDS_BORDER falls back to."""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=frames, L=60 * frames, n_obs=600 * frames, seed=3, rig=rig, frame_dt=0.25,
                           sonar=rig == "rig_v2", depth=rig == "rig_v2")
    est = Estimator(0)
    seen, hits = set(), set()

    def compare():
        for mu, tol in ((1e-4, 1e-10), (1e-9, 1e-6)):
            lin = est.linearize(mu)
            y_ref = host_solve(lin)
            scale = np.abs(y_ref).max()
            err = {}
            for off in (False, True):
                Estimator.debug_set_switch("SVIN_NO_LDS_BORDER", off)
                for fused in (False, True):
                    err[off, fused] = np.abs(est.debug_reduced_solve(mu, fused=fused) - y_ref).max() / scale
            print("d %d, mu %g: border variant vs host %.2e (fused %.2e), left-looking %.2e (fused %.2e)" %
                  (lin["d"], mu, err[False, False], err[False, True], err[True, False], err[True, True]))
            assert max(err.values()) < tol

    def on_frame(k, fid):
        est.optimize(3)
        d = est.linearize(1e-4)["d"]
        seen.add(d)
        if d in sizes and d not in hits:
            compare()
            hits.add(d)
        est.apply_marginalization(keyframes, imu_frames)
    try:
        syn.feed(est, spec, on_frame=on_frame)
    finally:
        Estimator.debug_set_switch("SVIN_NO_LDS_BORDER", False)
    assert hits == set(sizes), "sizes %s wanted, reached %s (sizes seen: %s)" % (sizes, sorted(hits), sorted(seen))


def test_chain_elimination_is_used_and_can_be_switched_off(gpu_lib, sb_elim_switch):
    """the two blocked paths give different roundings of the same step (so the switch really selects code), and the solver
    converges to the same optimum either way"""
    from svin_amd.estimator import Estimator
    spec = syn.make_window(P=32, L=1000, n_obs=10000, seed=5, frame_dt=0.25)
    res = {}
    for off in (False, True):
        sb_elim_switch(off)
        est = Estimator(0)
        frames, _ = syn.feed(est, spec)
        y = est.debug_reduced_solve(1e-6)
        est.optimize(8)
        res[off] = (y, est.summary(), np.array([est.get_T_WS(f) for f in frames]))
    (y0, s0, p0), (y1, s1, p1) = res[False], res[True]
    assert np.any(y0 != y1) and np.abs(y0 - y1).max() < 1e-9 * np.abs(y1).max()
    assert s0["iterations"] == s1["iterations"] and s0["successful"] == s1["successful"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-9 * s1["final_cost"]
    assert np.abs(p0 - p1).max() < 1e-8
