"""GPU parity of the global pose-graph optimisation (include/svin_pg.h) against the CPU oracle
(oracle/orc_posegraph.cpp) on identical seeded synthetic loop-closure graphs, through the C ABI.

Tolerances: the two sides run the same Levenberg-Marquardt control flow on the same normal equations; only the
order of the floating-point sums differs (per-node accumulation and MFMA Cholesky on the GPU, per-edge accumulation
and scalar Cholesky in the oracle).  Costs agree to 1e-9 relative, iteration counts exactly, optimised poses to
1e-7 absolute (metres / quaternion coefficients) -- three orders inside north_star's 1e-4 relative bar.
"""
import numpy as np
import pytest

from svin_amd import synthetic_pg as spg

pytestmark = pytest.mark.gpu


def pair(six, spec, upto=None, max_iterations=0):
    from oracle import orc
    from svin_amd.posegraph import PoseGraph
    # the oracle's envelope Cholesky is the same arithmetic as its dense one restricted to the profile; it keeps the
    # larger graphs in seconds (tests/test_oracle_posegraph.py checks the two against each other)
    g = PoseGraph(0, six_dof=six, max_iterations=max_iterations)
    c = orc.OraclePoseGraph(six_dof=six, max_iterations=max_iterations, envelope=spec.n > 300)
    eg = spg.feed(g, spec, upto)
    ec = spg.feed(c, spec, upto)
    assert eg == ec
    return g, c, eg[0], eg[1]


def qdiff(Qa, Qb):
    s = np.sign(np.sum(Qa * Qb, axis=1, keepdims=True))
    return float(np.max(np.abs(Qa - s * Qb)))


def compare(g, c, sg, sc, tol_pose=1e-7):
    print("gpu", sg, "\noracle", sc)
    assert sg["iterations"] == sc["iterations"]
    assert sg["termination"] == sc["termination"]
    assert sg["successful"] == sc["successful"]
    assert abs(sg["initial_cost"] - sc["initial_cost"]) <= 1e-9 * max(1.0, sc["initial_cost"])
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-9 * max(1.0, sc["initial_cost"])
    Tg, Qg = g.poses()
    Tc, Qc = c.poses()
    dt, dq = float(np.max(np.abs(Tg - Tc))), qdiff(Qg, Qc)
    print("max |dt|", dt, "max |dq|", dq)
    assert dt < tol_pose and dq < tol_pose
    yg, rg, tg = g.drift()
    yc, rc, tc = c.drift()
    assert abs(yg - yc) < 1e-6 and np.max(np.abs(rg - rc)) < 1e-8 and np.max(np.abs(tg - tc)) < 1e-6


@pytest.mark.parametrize("six", [False, True])
@pytest.mark.parametrize("n,laps,loop_every", [(40, 2, 5), (160, 4, 8), (400, 4, 25)])
def test_optimize_matches_oracle(gpu_lib, six, n, laps, loop_every):
    # d = 4 * 39 = 156 (LDS-resident Cholesky) ... 6 * 399 = 2394 (multi-workgroup blocked Cholesky)
    spec = spg.make_pose_graph(n=n, laps=laps, loop_every=loop_every, seed=11 + n)
    g, c, earliest, cur = pair(six, spec)
    sg, sc = g.optimize(earliest, cur), c.optimize(earliest, cur)
    compare(g, c, sg, sc)
    assert sg["final_cost"] < 0.8 * sg["initial_cost"]


@pytest.mark.parametrize("six", [False, True])
def test_incremental_optimisation_and_drift(gpu_lib, six):
    """the optimisation thread's life: optimise at a loop, keyframes keep arriving (drift-corrected on arrival),
    optimise again from the SVIn poses (PoseGraph.cpp:127-132, :262-275, :356-375)"""
    spec = spg.make_pose_graph(n=300, laps=3, loop_every=20, seed=21)
    from oracle import orc
    from svin_amd.posegraph import PoseGraph
    g, c = PoseGraph(0, six_dof=six), orc.OraclePoseGraph(six_dof=six)
    earliest = None
    for k in range(spec.n):
        for pg in (g, c):
            pg.add_keyframe(k, 1, spec.t_svin[k], spec.q_svin[k], spec.loops.get(k))
        if k in spec.loops:
            earliest = spec.loops[k][0] if earliest is None else min(earliest, spec.loops[k][0])
            if k in (120, 200, 280):
                sg, sc = g.optimize(earliest, k), c.optimize(earliest, k)
                compare(g, c, sg, sc)
    Tg, _ = g.poses()
    # keyframes after the last optimised one follow the drift: P = r_drift * svin_P + t_drift
    _, r, t = g.drift()
    assert np.max(np.abs(Tg[281:] - (spec.t_svin[281:] @ r.T + t))) < 1e-9
    assert np.max(np.abs(Tg[281:] - spec.t_svin[281:])) > 1e-3


def test_multiple_sequences_and_constant_first_sequence(gpu_lib):
    """6-DoF: keyframes of sequence 0 stay constant (PoseGraph.cpp:449-452); sequential edges only inside a sequence"""
    spec = spg.make_pose_graph(n=200, laps=4, loop_every=10, seed=31)
    spec.sequence[:60] = 0
    spec.sequence[60:] = 1
    for six in (False, True):
        g, c, earliest, cur = pair(six, spec)
        sg, sc = g.optimize(earliest, cur), c.optimize(earliest, cur)
        compare(g, c, sg, sc)
        if six:
            Tg, _ = g.poses()
            lo = max(earliest, 0)
            assert np.max(np.abs(Tg[lo:60] - spec.t_svin[lo:60])) == 0.0


def test_no_loop_is_a_no_op(gpu_lib):
    spec = spg.make_pose_graph(n=50, laps=1, loop_every=1000, seed=2)
    assert not spec.loops
    g, c, earliest, cur = pair(False, spec)
    sg, sc = g.optimize(0, cur), c.optimize(0, cur)
    compare(g, c, sg, sc)
    Tg, _ = g.poses()
    assert np.max(np.abs(Tg - spec.t_svin)) < 1e-9


@pytest.mark.parametrize("six", [False, True])
@pytest.mark.parametrize("levels", [1, 2])
@pytest.mark.parametrize("n,laps,loop_every,piece", [(160, 4, 8, 8), (400, 4, 25, 16), (400, 4, 10, 64), (1200, 6, 20, 64),
                                                     (1200, 6, 40, 8)])
def test_piece_elimination_matches_oracle(gpu_lib, six, levels, n, laps, loop_every, piece):
    """the separator / piece solver (chain cut into pieces, banded Cholesky per piece, dense separator system) against
    the oracle's plain Cholesky of the same normal equations"""
    spec = spg.make_pose_graph(n=n, laps=laps, loop_every=loop_every, seed=5 + n)
    g, c, earliest, cur = pair(six, spec)
    g.set_partition(piece, 0)
    g.set_levels(levels, 8)   # level 2: the cut keyframes are eliminated too (pieces of 8), the root keeps the loop cover
    sg, sc = g.optimize(earliest, cur), c.optimize(earliest, cur)
    part = g.partition()
    print(part)
    assert part["pieces"] >= 2 and part["separators"] < part["free"]
    if levels == 1:
        assert part["level2_pieces"] == 0 and part["separator_unknowns"] == part["separators"] * (6 if six else 4)
    compare(g, c, sg, sc)


@pytest.mark.parametrize("six", [False, True])
def test_config5_graph_matches_oracle(gpu_lib, six):
    """BASELINE config #5 size: 5,000 keyframes, 190 loop closures (oracle: envelope Cholesky)"""
    from oracle import orc
    from svin_amd.posegraph import PoseGraph
    spec = spg.make_pose_graph(n=5000, laps=20, loop_every=25, seed=7)
    g, c = PoseGraph(0, six_dof=six), orc.OraclePoseGraph(six_dof=six, envelope=True)
    earliest, cur = spg.feed(g, spec)
    spg.feed(c, spec)
    sg, sc = g.optimize(earliest, cur), c.optimize(earliest, cur)
    print(g.partition())
    compare(g, c, sg, sc, tol_pose=1e-6)


@pytest.mark.parametrize("six", [False, True])
def test_outlier_loops_in_the_huber_region(gpu_lib, six):
    """a few grossly wrong loop measurements: their residuals sit far outside HuberLoss(0.1)'s quadratic region, so the
    corrector scaling sqrt(rho') is active on both sides (loss_function.cc / corrector.cc restated in the oracle)"""
    spec = spg.make_pose_graph(n=600, laps=4, loop_every=15, seed=17)
    rng = np.random.default_rng(1)
    bad = rng.choice(sorted(spec.loops), size=5, replace=False)
    for k in bad:
        li, rt, rq, ry = spec.loops[int(k)]
        spec.loops[int(k)] = (li, rt + rng.normal(0, 1.5, 3), rq, ry + 25.0)
    g, c, earliest, cur = pair(six, spec)
    sg, sc = g.optimize(earliest, cur), c.optimize(earliest, cur)
    compare(g, c, sg, sc)
    assert g.partition()["pieces"] >= 2


@pytest.mark.parametrize("six", [False, True])
def test_cost_and_minimum_match_the_mpmath_fixture(gpu_lib, six):
    """the HIP path against tests/golden/pg.npz directly (40-digit mpmath restatement of the reference's functors,
    make_golden_pg.py), no oracle in between: the cost at the SVIn poses (loop edges only; one in the Huber region, yaw
    differences across +-180 degrees) and, for 4 DoF, the converged solution against the fixture's minimum"""
    import os
    from svin_amd.posegraph import PoseGraph
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pg.npz"))
    loops = {int(k): (int(o), g["loop_t"][i], g["loop_q"][i], float(g["loop_yaw"][i])) for i, (k, o) in enumerate(zip(g["loop_cur"], g["loop_old"]))}
    pg = PoseGraph(0, six_dof=six, max_iterations=200)
    n = len(g["t_svin"])
    for k in range(n):
        pg.add_keyframe(k, 1, g["t_svin"][k], g["q_svin"][k], loops.get(k))
    s = pg.optimize(0, n - 1)
    ref = float(g["cost6_initial" if six else "cost4_initial"])
    print("gpu", s, "fixture initial", ref, "4-DoF minimum", float(g["cost4_min"]))
    assert abs(s["initial_cost"] - ref) < 1e-11 * ref
    if not six:
        T, _ = pg.poses()
        assert s["final_cost"] >= g["cost4_min"] * (1 - 1e-12)
        assert s["final_cost"] - g["cost4_min"] < 2e-5 * g["cost4_min"]
        assert np.max(np.abs(T - g["t4_min"])) < 2e-3
