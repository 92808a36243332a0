"""CPU: pins the oracle (oracle/mpc.py).  The reference holds no golden vectors, so
the pins are known answers, an independent LP solver (HiGHS), the cross-formulation equalities the
reference itself asserts (test/runtests.jl:102-115,165-180) and the committed traces."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.optimize as so

from oracle import mpc
from oracle import qp as Q

REG = lambda: mpc.FixedRegularization(1e-8, -1e-8)
HERE = os.path.dirname(os.path.abspath(__file__))


def test_simple_lp_known_answer():
    """test/runtests.jl:24-55,165-180: objective 1.0 at (0.5, 0.5); |y| = 1; all formulations agree."""
    ref = mpc.solve(Q.simple_lp(), kkt_system="K2", regularization=mpc.NoRegularization())
    assert ref["status"] == mpc.SOLVE_SUCCEEDED
    assert abs(ref["objective"] - 1.0) < 1e-8 and np.allclose(ref["solution"], [0.5, 0.5], atol=1e-8)
    assert abs(abs(ref["multipliers"][0]) - 1.0) < 1e-6
    for kkt, reg in (("normal", mpc.FixedRegularization(1e-8, 0.0)), ("condensed", REG())):
        r = mpc.solve(Q.simple_lp(), kkt_system=kkt, regularization=reg)
        assert r["status"] == mpc.SOLVE_SUCCEEDED
        assert abs(r["objective"] - ref["objective"]) < 1e-6
        assert np.allclose(r["solution"], ref["solution"], atol=1e-6)
        assert np.allclose(r["multipliers"], ref["multipliers"], atol=1e-5)


def test_hs21_known_answer():
    for kkt in ("K2", "condensed"):
        r = mpc.solve(Q.hs21(), kkt_system=kkt, regularization=REG())
        assert r["status"] == mpc.SOLVE_SUCCEEDED
        assert abs(r["objective"] + 99.96) < 1e-7 and np.allclose(r["solution"], [2.0, 0.0], atol=1e-7)


def test_normal_kkt_rejects_qp():
    """src/KKT/normalkkt.jl:45-48."""
    with pytest.raises(ValueError):
        mpc.MPCSolver(Q.hs21(), kkt_system="normal")


@pytest.mark.parametrize("n,m", [(10, 0), (10, 5), (50, 10)])
@pytest.mark.parametrize("ncorr", [0, 5])
def test_k2_equals_condensed_dummy(n, m, ncorr):
    """test/runtests.jl:57-73,102-115 in spirit: same iteration count, objective, x, multipliers."""
    qp = Q.dummy_qp(n, m)
    a = mpc.solve(qp, kkt_system="K2", regularization=REG(), max_ncorr=ncorr)
    b = mpc.solve(qp, kkt_system="condensed", regularization=REG(), max_ncorr=ncorr)
    assert a["status"] == b["status"] == mpc.SOLVE_SUCCEEDED and a["iter"] == b["iter"]
    assert abs(a["objective"] - b["objective"]) < 1e-9
    assert np.allclose(a["solution"], b["solution"], atol=1e-8)
    assert np.allclose(a["multipliers"], b["multipliers"], atol=1e-6)
    # KKT conditions of the QP itself
    x, y = a["solution"], a["multipliers"]
    g = qp.H @ x + qp.q + qp.A.T @ y - a["multipliers_L"] + a["multipliers_U"]
    assert np.max(np.abs(g)) < 1e-6


def test_equality_rows_condensed():
    qp = Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7))
    a = mpc.solve(qp, kkt_system="K2", regularization=REG())
    b = mpc.solve(qp, kkt_system="condensed", regularization=REG())
    assert a["status"] == b["status"] == mpc.SOLVE_SUCCEEDED and a["iter"] == b["iter"]
    assert np.allclose(a["solution"], b["solution"], atol=1e-7)
    assert np.max(np.abs((qp.A @ a["solution"])[[0, 1, 2, 7]])) < 1e-7
    with pytest.raises(ValueError):  # delta_d = 0: equality rows are not representable
        mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, 0.0))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lp_against_highs(seed):
    qp = Q.synthetic_qp(seed, 30, 12, "lp")
    hi = so.linprog(qp.q, A_ub=np.vstack([qp.A, -qp.A]), b_ub=np.concatenate([qp.ucon, -qp.lcon]),
                    bounds=list(zip(qp.lvar, qp.uvar)), method="highs")
    assert hi.status == 0
    for kkt, reg in (("K2", REG()), ("normal", mpc.FixedRegularization(1e-8, 0.0)), ("condensed", REG())):
        r = mpc.solve(qp, kkt_system=kkt, regularization=reg)
        assert r["status"] == mpc.SOLVE_SUCCEEDED, kkt
        assert abs(r["objective"] - hi.fun) < 1e-6 * max(1, abs(hi.fun)), (kkt, r["objective"], hi.fun)


@pytest.mark.parametrize("rule", [mpc.AdaptiveStep(0.99), mpc.ConservativeStep(0.99),
                                  mpc.MehrotraAdaptiveStep(0.99)])
def test_step_rules(rule):
    """test/runtests.jl:80-92."""
    r = mpc.solve(Q.dummy_qp(10, 5), kkt_system="condensed", regularization=REG(), step_rule=rule)
    assert r["status"] == mpc.SOLVE_SUCCEEDED


@pytest.mark.parametrize("reg", [lambda: mpc.FixedRegularization(1e-8, -1e-9),
                                 lambda: mpc.AdaptiveRegularization(1e-8, -1e-9, 1e-9)])
def test_regularizations(reg):
    """test/runtests.jl:117-135."""
    qp = Q.dummy_qp(10, 5)
    ref = mpc.solve(qp, kkt_system="K2", regularization=mpc.NoRegularization())
    r = mpc.solve(qp, kkt_system="condensed", regularization=reg())
    assert r["status"] == mpc.SOLVE_SUCCEEDED
    assert abs(r["objective"] - ref["objective"]) < 1e-6
    assert np.allclose(r["solution"], ref["solution"], atol=1e-6)
    assert np.allclose(r["multipliers"], ref["multipliers"], atol=1e-6)


def test_kkt_solve_mul_consistency():
    """MadNLPTests.test_kkt_system in spirit (test/runtests.jl:149-163): K * solve(b) == b for the
    UNREDUCED system, for every formulation."""
    rng = np.random.default_rng(0)
    qp = Q.synthetic_qp(5, 25, 10)
    for kkt in ("K2", "condensed"):
        s = mpc.MPCSolver(qp, kkt_system=kkt, regularization=REG())
        s.initialize()
        s.update_regularization()
        s.factorize_regularized_system()
        b = rng.standard_normal(s.p.values.size)
        s.p.values[:] = b
        s.solve_system()
        assert s.last_residual_ratio < 1e-10


def test_aliasing_quirk_of_starting_point():
    """src/solver.jl:80-81: x_lr and x_ur are views of the same x, so the two primal shifts cancel
    on two-sided variables -- the start stays strictly interior."""
    s = mpc.MPCSolver(Q.synthetic_qp(9, 12, 5), kkt_system="condensed", regularization=REG())
    s.initialize()
    assert np.all(s.x > s.xl) and np.all(s.x < s.xu) and np.all(s.zl_r > 0) and np.all(s.zu_r > 0)


def test_generator_is_position_addressable():
    A = Q.gen_A(42, 7, 11)
    part = Q.gen_normal(Q.stream_key(42, Q.STREAM_A), np.arange(30, 50, dtype=np.uint64))
    assert np.array_equal(part, A.ravel()[30:50])
    H = Q.gen_H_wigner(42, 200)
    assert np.array_equal(H, H.T)
    ev = np.linalg.eigvalsh(H)
    assert 0.5 < ev[0] and ev[-1] < 5.6
    g = Q.gen_normal(Q.stream_key(1, 1), np.arange(200000, dtype=np.uint64))
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01


def test_golden_traces_are_reproduced():
    """The committed fixtures are exactly what tests/golden/make_golden.py produces today."""
    golden = json.load(open(os.path.join(HERE, "golden", "traces.json")))
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden

    assert set(golden) == set(make_golden.CASES)
    for name in golden:
        cur = make_golden.run(name)
        assert cur["iter"] == golden[name]["iter"] and cur["status"] == golden[name]["status"]
        for a, b in zip(cur["trace"], golden[name]["trace"]):
            for k in make_golden.KEYS:
                assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (name, a["k"], k)


def kkt_certificate(qp, r, tol):
    """Optimality conditions of  min c0 + q'x + x'Hx/2, lcon <= Ax <= ucon, lvar <= x <= uvar  checked from the
    returned primal-dual point alone (sign conventions of MadNLP: grad + A'y - zl + zu = 0, zl, zu >= 0)."""
    x, y, zl, zu = r["solution"], r["multipliers"], r["multipliers_L"], r["multipliers_U"]
    amax = lambda v: float(np.max(np.abs(v), initial=0.0))
    g = qp.H @ x + qp.q
    scale = max(1.0, np.max(np.abs(g)))
    assert amax(g + qp.A.T @ y - zl + zu) <= tol * scale  # stationarity
    ax = qp.A @ x
    assert np.all(x >= qp.lvar - tol) and np.all(x <= qp.uvar + tol)
    assert np.all(ax >= qp.lcon - tol) and np.all(ax <= qp.ucon + tol)
    assert np.all(zl >= -tol) and np.all(zu >= -tol)
    fin = lambda v: np.where(np.isfinite(v), v, 0.0)
    assert amax(zl * fin(x - qp.lvar)) <= tol * scale and amax(zu * fin(qp.uvar - x)) <= tol * scale
    # constraint multipliers: y_i > 0 only at the upper side, y_i < 0 only at the lower side
    assert amax(np.maximum(y, 0.0) * fin(qp.ucon - ax)) <= tol * scale
    assert amax(np.minimum(y, 0.0) * fin(ax - qp.lcon)) <= tol * scale


def test_box_qp_has_the_closed_form_solution():
    """Separable box QP: x* = clip(-q/h, l, u) -- no solver in the loop."""
    rng = np.random.default_rng(7)
    n = 40
    h, q = rng.uniform(0.5, 3.0, n), rng.standard_normal(n) * 2
    qp = Q.DenseQP(H=np.diag(h), q=q, A=np.zeros((0, n)), lvar=-np.ones(n), uvar=np.ones(n), lcon=np.zeros(0),
                   ucon=np.zeros(0), x0=np.zeros(n), name="box")
    xs = np.clip(-q / h, -1.0, 1.0)
    for kkt in ("K2", "condensed"):
        r = mpc.solve(qp, kkt_system=kkt, regularization=REG())
        assert r["status"] == mpc.SOLVE_SUCCEEDED
        assert np.max(np.abs(r["solution"] - xs)) < 1e-6
        assert abs(r["objective"] - (0.5 * xs @ (h * xs) + q @ xs)) < 1e-6  # tol = 1e-8 on scaled residuals
        kkt_certificate(qp, r, 1e-6)


@pytest.mark.parametrize("make", [lambda: Q.synthetic_qp(11, 30, 12), lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
                                  lambda: Q.hs21(), lambda: Q.synthetic_qp(12, 25, 10, "lp")])
def test_solutions_carry_an_optimality_certificate(make):
    """KKT conditions checked from the returned point alone, and the objective against scipy's
    independent trust-constr solver."""
    qp = make()
    r = mpc.solve(qp, kkt_system="condensed", regularization=REG())
    assert r["status"] == mpc.SOLVE_SUCCEEDED
    kkt_certificate(qp, r, 2e-6)
    cons = so.LinearConstraint(qp.A, qp.lcon, qp.ucon)
    res = so.minimize(lambda x: qp.c0 + qp.q @ x + 0.5 * x @ qp.H @ x, np.clip(qp.x0, qp.lvar, qp.uvar),
                      jac=lambda x: qp.q + qp.H @ x, hess=lambda x: qp.H, method="trust-constr",
                      bounds=so.Bounds(qp.lvar, qp.uvar), constraints=[cons],
                      options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
    # trust-constr stops at a barrier parameter of ~1e-5: it agrees to that accuracy and is never better than
    # the certified point
    assert abs(res.fun - r["objective"]) <= 1e-3 * max(1.0, abs(r["objective"]))
    assert res.fun >= r["objective"] - 1e-6 * max(1.0, abs(r["objective"]))


def test_fixed_variables_relax_bound_equals_elimination():
    """MadNLP.RelaxBound (what src/utils.jl:81 selects for condensed KKT systems): a fixed variable keeps both
    bounds, relaxed by bound_relax_factor; the optimum equals the one of the problem with the variable
    eliminated, to the size of the relaxation.  Without the option the oracle refuses (MakeParameter is not built)."""
    qp = Q.random_qp(5, 12, 7)
    fixed, vals = [3, 8], np.array([0.25, -0.4])
    qp.lvar[fixed] = qp.uvar[fixed] = vals
    with pytest.raises(NotImplementedError):
        mpc.solve(qp, kkt_system="condensed", regularization=REG())
    keep = [i for i in range(12) if i not in fixed]
    red = Q.DenseQP(H=qp.H[np.ix_(keep, keep)], q=qp.q[keep] + qp.H[np.ix_(keep, fixed)] @ vals, A=qp.A[:, keep],
                    lvar=qp.lvar[keep], uvar=qp.uvar[keep], lcon=qp.lcon - qp.A[:, fixed] @ vals,
                    ucon=qp.ucon - qp.A[:, fixed] @ vals, x0=np.zeros(10),
                    c0=qp.c0 + qp.q[fixed] @ vals + 0.5 * vals @ qp.H[np.ix_(fixed, fixed)] @ vals)
    ref = mpc.solve(red, kkt_system="condensed", regularization=REG())
    for kkt in ("K2", "condensed"):
        r = mpc.solve(qp, kkt_system=kkt, regularization=REG(), fixed_variable_treatment="relax_bound")
        assert r["status"] == ref["status"] == mpc.SOLVE_SUCCEEDED
        assert abs(r["objective"] - ref["objective"]) < 1e-6 and np.allclose(r["solution"][fixed], vals, atol=1e-7)
        assert np.max(np.abs(r["solution"][keep] - ref["solution"])) < 1e-5


@pytest.mark.parametrize("make", [lambda: Q.dummy_qp(10, 5), lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
                                  lambda: Q.random_qp(5, 130, 70), lambda: Q.simple_lp()])
def test_k25_equals_k2(make):
    """test/runtests.jl:95-115: ScaledSparseKKTSystem (K2.5) gives the same iteration count, objective, solution,
    constraints and multipliers as the default K2 system (atol 1e-6)."""
    qp = make()
    k2, k25 = mpc.solve(qp, kkt_system="K2"), mpc.solve(qp, kkt_system="K2.5")
    assert k2["status"] == k25["status"] == mpc.SOLVE_SUCCEEDED and k2["iter"] == k25["iter"]
    assert abs(k2["objective"] - k25["objective"]) <= 1e-6
    for key in ("solution", "constraints", "multipliers"):
        assert np.max(np.abs(k2[key] - k25[key])) <= 1e-6


def test_seed_9195_two_hosts_is_a_threshold_tie():
    """DESIGN.md section 4 / tests/parity.py: on soak seed 9195 (LP through the condensed form) the oracle does not agree
    with itself -- 12 iterations on the GPU box's host CPU, 13 in the build container.  Both traces are committed
    (tests/golden/make_seed9195.py, one run per host), so the claim is checkable here: the same path up to iteration
    11 to the accuracy the conditioning leaves -- the two LAPACK builds are 3.5e-7 apart at the START POINT already and
    6.2e-5 at iteration 10, on identical inputs with identical software: K = delta_w I + A' Theta A without a Hessian
    sits at the edge of fp64 -- and at iteration 12 one run is under the termination threshold while the other misses
    it by less than a factor 4: the definition of a threshold tie.  This host must reproduce one of the two."""
    import json
    import os

    from parity import TIE_FACTOR, worst_residual

    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    a = json.load(open(os.path.join(g, "seed9195_oracle_gpu_box_host.json")))
    b = json.load(open(os.path.join(g, "seed9195_oracle_build_container.json")))
    assert a["host"]["cpu"] != b["host"]["cpu"] and a["host"]["scipy"] == b["host"]["scipy"]  # same software, other CPU
    assert (a["status"], b["status"]) == (1, 1) and (a["iter"], b["iter"]) == (12, 13)
    for ta, tb in zip(a["trace"][:12], b["trace"][:12]):  # the same path until the deciding iteration
        for k in ("inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d"):
            assert abs(ta[k] - tb[k]) <= 1e-4 * max(1.0, abs(tb[k])), (ta["k"], k)
    wa, wb = worst_residual(a["trace"][12]), worst_residual(b["trace"][12])
    assert wa <= 1e-8 < wb <= TIE_FACTOR * 1e-8, (wa, wb)
    assert abs(a["objective"] - b["objective"]) <= 1e-7 * max(1.0, abs(b["objective"]))
    here = mpc.solve(Q.random_qp(9195, 186, 78, True), kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    assert here["status"] == 1 and here["iter"] in (12, 13)


def test_blocking_index_on_an_exact_tie_is_the_last_one():
    """src/kernels.jl:243-251: `mapreduce(..., (e1, e2) -> e1[1] < e2[1] ? e1 : e2, ...; init = (1.0, 0))` is a left fold
    that keeps the RIGHT element unless the left one is strictly smaller, so among exact ties the LAST index blocks
    (the oracle took the first).  The fold is written out literally here; the value is min(.., 1) and
    the index is only compared where the reference reads it (alpha < 1, src/kernels.jl:351-368)."""
    def reference_fold(vals):
        acc = (1.0, 0)
        for i, v in enumerate(vals, start=1):  # 1-based, as eachindex
            acc = acc if acc[0] < v else (v, i)
        return acc

    rng = np.random.default_rng(3)
    cases = [[0.5, 0.25, 0.7, 0.25, 0.9], [2.0, 3.0], [1.0, 1.0], [0.3], [np.inf, 0.1, 0.1, 0.1, np.inf], [0.25] * 7]
    cases += [list(rng.integers(1, 5, 40) / 8.0) for _ in range(20)]  # many exact ties
    for vals in cases:
        a, i = mpc.MPCSolver._argmin_last(np.array(vals, dtype=float))
        fa, fi = reference_fold(vals)
        assert a == fa
        if fa < 1.0:
            assert i + 1 == fi, (vals, i, fi)
            assert vals[i] == min(vals) and all(v > vals[i] for v in vals[i + 1:])
        else:
            assert i == -1


def test_oracle_on_the_edge_shapes():
    """tests/edge_cases.py on the CPU: the oracle reaches SOLVE_SUCCEEDED on every shape the GPU suites then hold the device
    against -- no constraint row, an empty bound list, one-sided rows, 1 x 1, everything mixed, and no bound at all, where
    the start point's 0 / 0 (src/solver.jl:93-94) is a NaN added to empty views, silently, as in Julia."""
    import warnings

    from edge_cases import EDGE_CASES, edge_qp

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for case in EDGE_CASES:
            qp = edge_qp(case)
            for form, reg in (("condensed", mpc.FixedRegularization(1e-8, -1e-8)), ("K2", None)):
                kw = dict(regularization=reg) if reg is not None else {}
                r = mpc.solve(qp, kkt_system=form, **kw)
                assert r["status"] == mpc.SOLVE_SUCCEEDED, (case, form, r["status"])
                assert np.all(np.isfinite(r["solution"])) and np.isfinite(r["objective"])
