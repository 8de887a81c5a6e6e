"""The host decisions of the trust-region loop (svin_amd/csrc/trust_region.hpp, driven by Window::solve) on the CPU.

Ceres 2.2 TrustRegionMinimizer / DoglegStrategy rules the reference's solve relies on (Estimator.cpp:878-890 sets DOGLEG,
everything else is Ceres' default): step acceptance at relative decrease > 1e-3, radius halved on a rejected step or below
0.25, grown to max(radius, 3 |step|) above 0.75, mu x10 per failed factorisation up to max_mu = 1, mu -> max(min_mu, 2 mu / 10)
after a successful step, five consecutive invalid steps = FAILURE, gradient / parameter / function tolerances, the
iteration callback's USER_SUCCESS.  And SURVEY 8(e): in the landmark-sharded mode two ranks that see the same ALL-REDUCED
numbers must take the same decisions -- the state machine reads nothing else."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACCEPTED, REJECTED, INVALID, TERMINATED = 0, 1, 2, 3


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("tr") / "libtr.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "csrc", "trust_region_shim.cpp"), "-o", so])
    L = C.CDLL(so)
    L.tr_create.restype = C.c_void_p
    L.tr_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_double]
    for n in ("tr_destroy", "tr_begin", "tr_retry", "tr_end", "tr_state"):
        getattr(L, n).argtypes = [C.c_void_p] + ([C.c_int] if n == "tr_begin" else [C.POINTER(C.c_double)] if n != "tr_destroy" else [])
    return L


class TR:
    def __init__(self, L, cost0, max_it=10, ftol=1e-6, gtol=1e-10, ptol=1e-8):
        self.L, self.h = L, L.tr_create(ftol, gtol, ptol, max_it, cost0)

    def begin(self, stop=False):
        return bool(self.L.tr_begin(self.h, 1 if stop else 0))

    @staticmethod
    def _v(cost, step2=1e-2, x2=1e2, grad=1.0, fail=0.0, jd2=0.0, jdr=0.0, dog=0.1):
        return (C.c_double * 8)(cost, step2, x2, grad, fail, jd2, jdr, dog)

    def retry(self, **kw):
        return bool(self.L.tr_retry(self.h, self._v(**kw)))

    def end(self, **kw):
        return self.L.tr_end(self.h, self._v(**kw))

    def state(self):
        o = (C.c_double * 10)()
        self.L.tr_state(self.h, o)
        keys = ("radius", "mu", "x_cost", "reuse", "initScale", "invalid", "iteration", "successful", "termination", "mu_after_accept")
        return dict(zip(keys, list(o)))


def model(decrease):
    """jdSq / jdDotR such that model_cost_change = -(jdDotR + jdSq / 2) = decrease"""
    return dict(jd2=2.0 * decrease, jdr=-2.0 * decrease)


def test_acceptance_and_radius_rules(lib):
    t = TR(lib, 100.0)
    assert t.state()["radius"] == 1e4 and t.state()["mu"] == 1e-8 and t.state()["initScale"] == 1
    # relative decrease 0.9 (> 0.75): accepted, radius = max(radius, 3 |step|)
    assert t.begin()
    assert not t.retry(cost=91.0)
    assert t.end(cost=91.0, dog=5e3, **model(10.0)) == ACCEPTED
    s = t.state()
    assert s["x_cost"] == 91.0 and s["radius"] == 1.5e4 and s["successful"] == 1 and s["reuse"] == 0 and s["initScale"] == 0
    assert s["mu"] == 1e-8    # max(min_mu, 2 mu / 10)
    # relative decrease 0.1 (< 0.25, > 1e-3): accepted, radius halved
    assert t.begin() and t.end(cost=90.0, **model(10.0)) == ACCEPTED
    assert t.state()["radius"] == 7.5e3
    # relative decrease 5e-4: rejected, radius halved, the linearisation is re-used
    assert t.begin() and t.end(cost=89.995, **model(10.0)) == REJECTED
    s = t.state()
    assert s["radius"] == 3.75e3 and s["reuse"] == 1 and s["x_cost"] == 90.0 and s["iteration"] == 3 and s["successful"] == 2
    # a re-used linearisation never retries its factorisation, whatever the flag says
    assert t.begin() and not t.retry(cost=80.0, fail=1.0)
    # cost went UP: rejected as well
    assert t.end(cost=95.0, **model(10.0)) == REJECTED
    # radius clamp
    big = TR(lib, 1.0)
    assert big.begin() and big.end(cost=0.1, dog=1e17, **model(0.9)) == ACCEPTED and big.state()["radius"] == 1e16


def test_mu_ladder_and_failure(lib):
    t = TR(lib, 10.0)
    assert t.begin()
    mus = []
    while t.retry(cost=10.0, fail=1.0):      # DoglegStrategy: mu x10 until it reaches max_mu = 1
        mus.append(t.state()["mu"])
    assert np.allclose(mus, [1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1]) and np.isclose(t.state()["mu"], 1.0)
    assert t.end(cost=10.0, **model(1.0)) == INVALID     # the step of a failed factorisation is invalid even if the model looks fine
    s = t.state()
    assert s["invalid"] == 1 and s["reuse"] == 0 and np.isclose(s["mu"], 10.0)
    # a non-positive model change is an invalid step too; five in a row = FAILURE (termination 3)
    for k in range(2, 5):
        assert t.begin() and t.end(cost=9.0, **model(-1.0)) == INVALID and t.state()["invalid"] == k
    assert t.begin() and t.end(cost=9.0, **model(-1.0)) == TERMINATED and t.state()["termination"] == 3
    # a valid step in between resets the count, and a successful step relaxes mu again
    u = TR(lib, 10.0)
    assert u.begin() and u.retry(cost=10.0, fail=1.0) and u.retry(cost=10.0, fail=1.0) and not u.retry(cost=10.0)
    assert np.isclose(u.state()["mu"], 1e-6) and np.isclose(u.state()["mu_after_accept"], 2e-7)
    assert u.end(cost=9.0, **model(1.0)) == ACCEPTED and np.isclose(u.state()["mu"], 2e-7) and u.state()["invalid"] == 0


def test_terminations(lib):
    # gradient tolerance: convergence, and the iteration is not counted (ceres checks it before the step is taken)
    t = TR(lib, 5.0)
    assert t.begin() and t.end(cost=5.0, grad=1e-11, **model(1.0)) == TERMINATED
    assert t.state()["termination"] == 0 and t.state()["iteration"] == 0
    # parameter tolerance: |step| <= pTol (|x| + pTol)
    t = TR(lib, 5.0)
    assert t.begin() and t.end(cost=4.0, step2=1e-20, x2=1.0, **model(1.0)) == TERMINATED and t.state()["termination"] == 0 and t.state()["iteration"] == 1
    # function tolerance: |cost change| <= fTol cost
    t = TR(lib, 5.0)
    assert t.begin() and t.end(cost=5.0 - 1e-7, **model(1.0)) == TERMINATED and t.state()["termination"] == 0
    # iteration limit
    t = TR(lib, 5.0, max_it=2)
    for _ in range(2):
        assert t.begin() and t.end(cost=t.state()["x_cost"] * 0.5, **model(t.state()["x_cost"] * 0.5)) == ACCEPTED
    assert not t.begin() and t.state()["termination"] == 1 and t.state()["iteration"] == 2
    # the iteration callback (Estimator::setOptimizationTimeLimit): USER_SUCCESS = 2, checked before the iteration limit
    t = TR(lib, 5.0, max_it=1)
    assert t.begin() and t.end(cost=2.0, **model(3.0)) == ACCEPTED
    assert not t.begin(stop=True) and t.state()["termination"] == 2
    # a collapsed radius ends the solve as converged
    t = TR(lib, 5.0, max_it=500)
    n = 0
    while t.begin():
        assert t.end(cost=6.0, **model(1.0)) == REJECTED
        n += 1
    assert t.state()["termination"] == 0 and t.state()["radius"] <= 1e-32 and n > 100


def test_two_ranks_take_the_same_decisions_from_the_same_reduced_numbers(lib):
    """the ranks of a sharded solve must stay in lock step without exchanging decisions.  The state machine is
    the ONLY place decisions are taken, and its input record holds exactly the all-reduced (or redundantly derived)
    fields; here two instances replay a long random trajectory from identical records and must agree bit for bit at
    every step.  (What differs between real ranks -- local partial sums, the device-side cholFail flag, their clocks --
    never reaches this interface: the time limit arrives as the all-reduced vote `stop`.)"""
    rng = np.random.default_rng(17)
    a, b = TR(lib, 50.0, max_it=400), TR(lib, 50.0, max_it=400)
    steps = 0
    outcomes = set()
    while True:
        stop = steps == 350
        ra, rb = a.begin(stop), b.begin(stop)
        assert ra == rb
        if not ra:
            break
        fail_rounds = int(rng.integers(0, 3)) if rng.random() < 0.2 else 0
        for k in range(fail_rounds + 1):
            rec = dict(cost=float(a.state()["x_cost"] * rng.uniform(0.5, 1.05)), fail=1.0 if k < fail_rounds else 0.0)
            qa, qb = a.retry(**rec), b.retry(**rec)
            assert qa == qb
            if not qa:
                break
        dec = float(a.state()["x_cost"] * rng.uniform(-0.05, 0.3))
        rec = dict(cost=float(a.state()["x_cost"] - dec * rng.uniform(0.0, 1.2)), step2=float(rng.uniform(1e-6, 1.0)), x2=100.0,
                   grad=float(rng.uniform(1e-3, 10.0)), dog=float(rng.uniform(0.01, 1e5)), **model(dec))
        oa, ob = a.end(**rec), b.end(**rec)
        assert oa == ob and a.state() == b.state()
        outcomes.add(oa)
        steps += 1
        if oa == TERMINATED:
            break
    assert steps > 20 and {ACCEPTED, REJECTED, INVALID}.issubset(outcomes)
