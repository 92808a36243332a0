"""GPU parity of the whole path: KKT plugin (build / factorize / solve / mul / jtprod) and the full
Mehrotra predictor-corrector solve through the C ABI, against the CPU oracle and the golden traces.

Stated tolerance: per-iteration alpha_p, alpha_d, inf_pr, inf_du within 1e-9
(relative or absolute, whichever is looser) while mu >= 1e-4 and 1e-6 afterwards (conditioning ~ 1/mu);
identical iteration count; objective rel 1e-9; ||x_gpu - x_cpu||inf <= 1e-7.
"""
import json
import os

import numpy as np
import pytest
import torch

import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "traces.json")))


def to_device(qp, be):
    return M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon,
                                 qp.x0, qp.c0)


def solve_hip(qp, be, **opts):
    opts.setdefault("regularization", M.FixedRegularization(1e-8, -1e-8))
    s = M.MPCSolver(to_device(qp, be), be, **opts)
    r = s.solve()
    s.close()
    return r


DRIVERS = pytest.mark.parametrize("driver", ["python", "native"])


def close(a, b, tol):
    return abs(a - b) <= tol * max(1.0, abs(a), abs(b))


def compare_traces(tr, ref, name):
    assert len(tr) == len(ref), f"{name}: iteration count {len(tr) - 1} vs {len(ref) - 1}"
    for t, r in zip(tr, ref):
        tol = 1e-9 if min(t["mu"], r["mu"]) >= 1e-4 else 1e-6
        for key in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "inf_compl", "mu"):
            assert close(t[key], r[key], tol), f"{name}: iter {t['k']} {key}: {t[key]!r} vs {r[key]!r}"


CASES = {
    "hs21": lambda: Q.hs21(),
    "dummy_10_5": lambda: Q.dummy_qp(10, 5),
    "dummy_50_10": lambda: Q.dummy_qp(50, 10),
    "dummy_20_15_eq": lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
    "dummy_20_15_eq_gondzio": lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
    "synthetic_40_16": lambda: Q.synthetic_qp(20250614, 40, 16),
    "synthetic_40_16_gondzio": lambda: Q.synthetic_qp(20250614, 40, 16),
    "synthetic_lp_30_12": lambda: Q.synthetic_qp(20250615, 30, 12, "lp"),
}
NCORR = {"dummy_20_15_eq_gondzio": 5, "synthetic_40_16_gondzio": 3}


@DRIVERS
@pytest.mark.parametrize("name", list(CASES))
def test_golden_traces(hip, name, driver):
    g = GOLDEN[name]
    r = solve_hip(CASES[name](), hip, max_ncorr=NCORR.get(name, 0), driver=driver)
    assert r["status"] == g["status"] == M.SOLVE_SUCCEEDED
    assert r["iter"] == g["iter"]
    compare_traces(r["trace"], g["trace"], name)
    assert close(r["objective"], g["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - np.array(g["solution"]))) <= 1e-7


def test_one_call_solve(hip):
    """M.solve(qp, ...): MPCSolver + solve! + release in one call; result with status / objective / total_time."""
    r = M.solve(to_device(Q.hs21(), hip), hip, regularization=M.FixedRegularization(1e-8, -1e-8))
    assert r["status"] == M.SOLVE_SUCCEEDED and abs(r["objective"] + 99.96) < 1e-6 and r["total_time"] > 0
    r2 = M.solve(to_device(Q.hs21(), hip), regularization=M.FixedRegularization(1e-8, -1e-8))  # own backend
    assert r2["iter"] == r["iter"]


def test_known_answers(hip):
    r = solve_hip(Q.hs21(), hip)
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert abs(r["objective"] - (-99.96)) < 1e-6 and np.allclose(r["solution"], [2.0, 0.0], atol=1e-6)
    # simple_lp of test/runtests.jl:24-55 has an equality row: condensed form needs delta_d < 0
    r = solve_hip(Q.simple_lp(), hip)
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert abs(r["objective"] - 1.0) < 1e-6 and np.allclose(r["solution"], [0.5, 0.5], atol=1e-6)
    assert abs(abs(r["multipliers"][0]) - 1.0) < 1e-5
    with pytest.raises(ValueError):
        M.MPCSolver(to_device(Q.simple_lp(), hip), hip)  # default delta_d = 0 with an equality row


@DRIVERS
@pytest.mark.parametrize("n,m,ncorr", [(300, 120, 0), (300, 120, 3), (1500, 600, 0), (2500, 700, 3)])  # (the last: 20 blocks, one-pass A'x, Gondzio)
def test_synthetic_vs_oracle(hip, n, m, ncorr, driver):
    qp = Q.synthetic_qp(20250614 + n, n, m)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8),
                    max_ncorr=ncorr)
    r = solve_hip(qp, hip, max_ncorr=ncorr, driver=driver)
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
    compare_traces(r["trace"], ref["trace"], f"synthetic {n}x{m}")
    assert close(r["objective"], ref["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
    assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) <= 1e-6
    assert np.max(np.abs(r["constraints"] - ref["constraints"])) <= 1e-6  # stats.constraints (test/runtests.jl:18)


@DRIVERS
@pytest.mark.parametrize("rule", ["adaptive", "conservative", "mehrotra"])
def test_step_rules(hip, rule, driver):
    """test/runtests.jl:80-92: every step rule reaches SOLVE_SUCCEEDED; here also == oracle."""
    mk = {"adaptive": (M.AdaptiveStep(0.99), mpc.AdaptiveStep(0.99)),
          "conservative": (M.ConservativeStep(0.99), mpc.ConservativeStep(0.99)),
          "mehrotra": (M.MehrotraAdaptiveStep(0.99), mpc.MehrotraAdaptiveStep(0.99))}[rule]
    qp = Q.dummy_qp(10, 5)
    r = solve_hip(qp, hip, step_rule=mk[0], driver=driver)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8),
                    step_rule=mk[1])
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
    compare_traces(r["trace"], ref["trace"], rule)


@DRIVERS
@pytest.mark.parametrize("reg", ["fixed", "adaptive"])
def test_regularizations(hip, reg, driver):
    """test/runtests.jl:117-135: regularized runs match the reference solution to 1e-6."""
    qp = Q.dummy_qp(10, 5)
    sol_ref = mpc.solve(qp, kkt_system="K2", regularization=mpc.NoRegularization())
    r = solve_hip(qp, hip, regularization=(M.FixedRegularization(1e-8, -1e-9) if reg == "fixed"
                                           else M.AdaptiveRegularization(1e-8, -1e-9, 1e-9)),
                  driver=driver)
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert abs(r["objective"] - sol_ref["objective"]) < 1e-6
    assert np.max(np.abs(r["solution"] - sol_ref["solution"])) < 1e-6
    assert np.max(np.abs(r["multipliers"] - sol_ref["multipliers"])) < 1e-6
    assert np.max(np.abs(r["constraints"] - sol_ref["constraints"])) < 1e-6  # test/runtests.jl:131


@pytest.mark.parametrize("with_eq", [False, True])
def test_kkt_system_conformance(hip, with_eq):
    """Counterpart of MadNLPTests.test_kkt_system (test/runtests.jl:149-163): build, factorize,
    solve K x = b, then check mul!(., kkt, x) ~ b and jtprod! against the explicit Jacobian."""
    rng = np.random.default_rng(0)
    qp = Q.synthetic_qp(7, 90, 35)
    if with_eq:
        qp.lcon[[3, 10]] = qp.ucon[[3, 10]] = 0.25  # two equality rows: Theta = 1e8, cond(K) ~ 1e8
    s = M.MPCSolver(to_device(qp, hip), hip, regularization=M.FixedRegularization(1e-8, -1e-8))
    s.initialize()
    st, be = s.st, hip
    be.set_aug_diagonal_reg(st, 1e-8, -1e-8)
    s.kkt.factorize_wrapper()
    assert s.kkt.linear_solver.is_factorized()
    b = rng.standard_normal(st.ntot)
    st.p.copy_(torch.as_tensor(b))
    be.copy(st.p, st.d)
    s.kkt.solve(st.d)
    be.fill(0.0, st.w2)
    s.kkt.mul_solved(st.w2, st.d, 1.0, 0.0)  # the solve's own A dx instead of a second pass over A:
    be.fill(0.0, st.w1)
    s.kkt.mul(st.w1, st.d, 1.0, 0.0)
    assert torch.equal(st.w1, st.w2)         # bitwise the generic product
    be.fill(0.0, st.w2)
    s.kkt.mul_solved(st.w2, st.p, 1.0, 0.0)  # a vector that is NOT the last solution: the product is computed
    be.fill(0.0, st.w1)
    s.kkt.mul(st.w1, st.p, 1.0, 0.0)
    assert torch.equal(st.w1, st.w2)
    be.fill(0.0, st.w1)
    s.kkt.mul(st.w1, st.d, 1.0, 0.0)
    res = np.max(np.abs(st.w1.cpu().numpy() - b)) / max(1.0, np.max(np.abs(b)))
    # The same check through the oracle's condensed system (LAPACK dpotrf / dpotrs) on the same data is the floor of
    # this FORMULATION: an equality row enters as Theta = -1/delta_d = 1e8, cond(K) = 3.6e7, and the decondensation
    # dy = Theta (A dx - t) multiplies the error of dx by 1e8 -- LAPACK reaches 2.3e-7 here, 3e-12 without equality
    # rows.  The device path must stay within a small factor of that floor (it measured 3.5e-7); the formulation that
    # takes equality rows exactly reaches 1e-10 on this check (test_gpu_augmented.py::test_augmented_kkt_conformance).
    o = mpc.MPCSolver(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    o.initialize()
    o.del_w, o.del_c = 1e-8, -1e-8
    o.set_aug_diagonal_reg()
    o.kkt.build_and_factorize()
    o.d.values[:] = b
    o.kkt.solve(o.d)
    ow = mpc.KKTVec(o.n, o.m, o.nlb, o.nub, o.ind_lb, o.ind_ub)
    o.kkt.mul(ow, o.d, 1.0, 0.0)
    res_oracle = np.max(np.abs(ow.values - b)) / max(1.0, np.max(np.abs(b)))
    assert res_oracle < (1e-6 if with_eq else 1e-10), res_oracle
    assert res < max(4.0 * res_oracle, 1e-11), (res, res_oracle)
    # jtprod vs explicit Jacobian [A, -I_ineq]
    y = rng.standard_normal(st.m)
    yd = torch.as_tensor(y, device=be.device)
    s.kkt.jtprod(st.jacl, yd)
    Afull = np.zeros((st.m, st.n))
    Afull[:, : s.nx] = qp.A
    Afull[s.ind_ineq, s.nx + np.arange(s.ns)] = -1.0
    assert np.allclose(st.jacl.cpu().numpy(), Afull.T @ y, rtol=1e-13, atol=1e-13)
    s.kkt.close()


def test_factorization_failure_is_a_return_code(hip):
    """A non positive definite K must come back as info > 0 (never abort) so that the x100
    regularization retry of src/linear_solver.jl:11-15 can run."""
    qp = Q.synthetic_qp(3, 200, 80)
    s = M.MPCSolver(to_device(qp, hip), hip, regularization=M.FixedRegularization(1e-8, -1e-8))
    s.initialize()
    s.H.sub_(1e6 * torch.eye(200, dtype=torch.float64, device=hip.device))  # now strongly indefinite
    s.update_regularization()
    s.factorize_regularized_system()
    assert s.kkt.linear_solver.info > 0
    assert s.kkt.n_factorizations == 1 + 3  # start point + three trials
    assert s.del_w == pytest.approx(1e-8 * 100.0 ** 3)
    s.kkt.close()


@DRIVERS
def test_regularization_retry_succeeds(hip, driver):
    """src/linear_solver.jl:6-17 on the device: with a slightly indefinite H and a free variable, the
    factorisation fails at delta_w = 1e-8, succeeds at 1e-6; the retry count and the trace follow the oracle.
    max_iter is honoured (status 6) by both drivers."""
    qp, free = Q.synthetic_qp(8, 20, 8), 3
    qp.lvar[free], qp.uvar[free] = -np.inf, np.inf
    qp.H = np.diag(np.diag(qp.H))  # diagonal H: K_ii of the free variable is H_ii + delta_w + (A' Theta A)_ii
    qp.A[:, free] = 0.0
    qp.H[free, free] = -1e-7
    qp.q[free] = 0.0
    kw = dict(regularization=M.FixedRegularization(1e-8, -1e-8), max_iter=4, driver=driver)
    s = M.MPCSolver(to_device(qp, hip), hip, **kw)
    r = s.solve()
    ref = mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8), max_iter=4)
    assert r["status"] == ref["status"] == M.MAXIMUM_ITERATIONS_EXCEEDED and r["iter"] == ref["iter"] == 4
    assert r["n_factorizations"] == ref["n_factorizations"] > r["iter"] + 1  # retries happened
    compare_traces(r["trace"], ref["trace"], "retry")
    s.close()


def test_full_size_properties_n5k(hip):
    """BASELINE config 1 (n=5000, m=2000) is beyond what the oracle finishes in seconds:
    size-independent properties.  The condensed solve must satisfy the UNREDUCED KKT system
    (residual via mul!), K x = b round trip, and the solver must converge with primal/dual/
    complementarity infeasibilities below tol."""
    n, m = 5000, 2000
    dq = M.DeviceQP.synthetic(hip, 20250615, n, m)
    s = M.MPCSolver(dq, hip, regularization=M.FixedRegularization(1e-8, -1e-8), max_iter=100)
    r = s.solve()
    assert r["status"] == M.SOLVE_SUCCEEDED, (r["status"], r["iter"])
    t = r["trace"][-1]
    assert max(t["inf_pr"], t["inf_du"], t["inf_compl"]) <= 1e-8
    assert s.last_residual_ratio < 1e-7
    x = r["solution"]
    assert np.all(x >= -1e-7) and np.all(x <= 1 + 1e-7)
    Ax = (dq.A @ torch.as_tensor(x, device=hip.device)).cpu().numpy()
    assert np.all(Ax >= -1e-6) and np.all(Ax <= 1 + 1e-6)
    # optimality certificate from the returned primal-dual point alone (no oracle, no solver):
    # H x + q + A'y - zl + zu = 0, multipliers of the right sign, complementary to their constraints
    xd = torch.as_tensor(x, device=hip.device)
    y, zl, zu = (torch.as_tensor(r[k], device=hip.device) for k in ("multipliers", "multipliers_L", "multipliers_U"))
    g = dq.H @ xd + dq.q
    scale = max(1.0, float(g.abs().max()))
    assert float((g + dq.A.t() @ y - zl + zu).abs().max()) <= 1e-6 * scale
    assert float(zl.min()) >= -1e-8 and float(zu.min()) >= -1e-8
    assert float((zl * xd).abs().max()) <= 1e-6 * scale and float((zu * (1 - xd)).abs().max()) <= 1e-6 * scale
    Axd = torch.as_tensor(Ax, device=hip.device)
    assert float((y.clamp(min=0) * (1 - Axd)).abs().max()) <= 1e-6 * scale
    assert float((y.clamp(max=0) * Axd).abs().max()) <= 1e-6 * scale
    s.kkt.close()


@DRIVERS
@pytest.mark.parametrize("case", ["simple_lp", "lp_30_12", "lp_400_150"])
def test_normal_kkt_system(hip, case, driver):
    """The reference's own NormalKKTSystem formulation on the device (kkt_system="normal"), with the
    reference's default regularization (delta_d = 0) and equality rows: test/runtests.jl:165-180."""
    qp = {"simple_lp": Q.simple_lp, "lp_30_12": lambda: Q.synthetic_qp(20250615, 30, 12, "lp"),
          "lp_400_150": lambda: Q.synthetic_qp(5, 400, 150, "lp")}[case]()
    reg, oreg = M.FixedRegularization(1e-8, 0.0), mpc.FixedRegularization(1e-8, 0.0)
    r = solve_hip(qp, hip, kkt_system="normal", regularization=reg, driver=driver)
    ref = mpc.solve(qp, kkt_system="normal", regularization=oreg)
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
    compare_traces(r["trace"], ref["trace"], f"normal {case}")
    assert close(r["objective"], ref["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
    if case == "simple_lp":
        assert abs(r["objective"] - 1.0) < 1e-8 and np.allclose(r["solution"], [0.5, 0.5], atol=1e-8)
    # cross-formulation equality (test/runtests.jl:165-180): normal == condensed on the same LP
    rc = solve_hip(qp, hip)
    assert abs(rc["objective"] - r["objective"]) < 1e-6 and np.max(np.abs(rc["solution"] - r["solution"])) < 1e-6


def test_normal_kkt_conformance(hip):
    """MadNLPTests.test_kkt_system on NormalKKTSystem (test/runtests.jl:149-163): K * solve(b) == b."""
    rng = np.random.default_rng(1)
    qp = Q.synthetic_qp(11, 120, 45, "lp")
    qp.lcon[[2, 7]] = qp.ucon[[2, 7]] = 0.3
    s = M.MPCSolver(to_device(qp, hip), hip, kkt_system="normal")
    s.initialize()
    st = s.st
    hip.set_aug_diagonal_reg(st, 1e-8, 0.0)
    s.kkt.factorize_wrapper()
    assert s.kkt.linear_solver.is_factorized()
    b = rng.standard_normal(st.ntot)
    st.p.copy_(torch.as_tensor(b))
    hip.copy(st.p, st.d)
    s.kkt.solve(st.d)
    hip.fill(0.0, st.w1)
    s.kkt.mul(st.w1, st.d, 1.0, 0.0)
    assert np.max(np.abs(st.w1.cpu().numpy() - b)) / max(1.0, np.max(np.abs(b))) < 1e-9
    s.kkt.close()


def test_native_driver_is_bitwise_the_python_driver(hip):
    """csrc/mpc.hip issues the same kernels in the same order as solver.py: identical traces, bit for bit
    (Gondzio corrections and the Mehrotra step rule included), and a NaN surfaces as the same status."""
    for qp, kw in ((Q.synthetic_qp(77, 200, 80), dict(max_ncorr=3)),
                   (Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)), dict(step_rule=M.MehrotraAdaptiveStep(0.99))),
                   (Q.synthetic_qp(78, 120, 50, "lp"), dict(regularization=M.AdaptiveRegularization(1e-8, -1e-9, 1e-9)))):
        a = solve_hip(qp, hip, driver="python", **kw)
        b = solve_hip(qp, hip, driver="native", **kw)
        assert a["status"] == b["status"] == M.SOLVE_SUCCEEDED and a["iter"] == b["iter"]
        assert a["trace"] == b["trace"] and a["n_factorizations"] == b["n_factorizations"]
        assert np.array_equal(a["solution"], b["solution"]) and a["objective"] == b["objective"]
    for driver in ("python", "native"):  # NaN inside the loop -> SolveException -> ERROR_IN_STEP_COMPUTATION
        s = M.MPCSolver(to_device(Q.synthetic_qp(79, 60, 20), hip), hip, driver=driver,
                        regularization=M.FixedRegularization(1e-8, -1e-8))
        s.initialize()
        s.st.x[3] = float("nan")
        with pytest.raises(M.SolveException):
            s.mpc()
        s.close()


@DRIVERS
def test_equality_constrained_qp_without_any_bound(hip, driver):
    """No bound on any variable, every row an equality: both bound lists are empty (nlb = nub = 0, no slacks) -- the edge
    the fused per-variable passes (inverse lists of length n, all -1), the device-side decisions of the queued-ahead loop
    (step lengths from alpha_none_kernel, complementarity sums over nothing) and update_barrier!'s "any bound" rule all
    have to get right; and the start point's 0 / 0 (src/solver.jl:93-94: NaN shifts added to EMPTY views, where Python's
    float division would raise).  The condensed matrix of an all-equality problem is H + 1e8 A'A: the second iterate's dual
    residual is the conditioning noise of that solve, 4e-8 or 2e-10 depending on whether it is refined -- one iteration
    more or less.  So like is compared with like: the default library (AUTO refinement at this size) against the oracle
    with one refinement step, refine_steps = 0 against the plain oracle's count."""
    import parity

    qp = Q.synthetic_qp(123, 60, 20)
    qp.lvar[:], qp.uvar[:] = -np.inf, np.inf
    qp.ucon[:] = qp.lcon[:] = 0.25
    oreg = mpc.FixedRegularization(1e-8, -1e-8)
    ref = parity.oracle_execution(qp, "refine", regularization=oreg)
    r = solve_hip(qp, hip, driver=driver)
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
    compare_traces(r["trace"], ref["trace"], "no bounds")
    assert close(r["objective"], ref["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
    plain, r0 = mpc.solve(qp, kkt_system="condensed", regularization=oreg), solve_hip(qp, hip, driver=driver, refine_steps=0)
    assert r0["status"] == plain["status"] == M.SOLVE_SUCCEEDED and r0["iter"] == plain["iter"]
    assert np.max(np.abs(r0["solution"] - plain["solution"])) <= 1e-7


from edge_cases import edge_qp as _edge_qp  # noqa: E402


@DRIVERS
@pytest.mark.parametrize("case", ["no_constraints", "lower_bounds_only", "upper_bounds_only", "one_by_one", "mixed"])
def test_edge_shapes_against_the_oracle(hip, case, driver):
    """Shapes at the edges of the index lists (ind_lb / ind_ub / ind_ineq): no constraint row at all, an empty
    upper or lower list, one-sided rows, a 1 x 1 problem, and every kind of variable and row at once -- through both drivers,
    against the oracle with the refinement the default library applies at these sizes."""
    import parity

    qp = _edge_qp(case)
    oreg = mpc.FixedRegularization(1e-8, -1e-8)
    ref = parity.oracle_execution(qp, "refine", regularization=oreg)
    r = solve_hip(qp, hip, driver=driver)
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED, (r["status"], ref["status"])
    assert r["iter"] == ref["iter"], (case, r["iter"], ref["iter"])
    compare_traces(r["trace"], ref["trace"], case)
    assert close(r["objective"], ref["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7


def test_a_failed_verdict_behind_a_queued_update_leaves_the_iterates_where_they_were(hip):
    """csrc/mpc.hip, body_fused queued ahead: from the second pass on the update of the iterates is already queued when the
    host reads the corrector's block -- as kernels that do nothing unless the device's own verdict (mpc_decide_kernel) was
    "go on".  A NaN that appears in a LATER pass must therefore surface as in the waiting form: SolveException, with the
    iterates x, y, zl, zu and the bounds as they were when the pass began (src/linear_solver.jl:41-43 throws before
    src/solver.jl:332-335 moves anything).  (f, c and A'y are re-evaluated at that same x by the queued kernels: the same
    values for any state the loop itself produced.)"""
    import os

    def run(ahead):
        os.environ["MADQP_MPC_AHEAD"] = "1" if ahead else "0"
        try:
            s = M.MPCSolver(to_device(Q.synthetic_qp(79, 60, 20), hip), hip, driver="native",
                            regularization=M.FixedRegularization(1e-8, -1e-8))
            s.initialize()
        finally:
            os.environ.pop("MADQP_MPC_AHEAD", None)
        for _ in range(3):
            assert s.iteration_head() is None
            s.iteration_body()
        assert s.iteration_head() is None
        s.st.x[3] = float("nan")  # K and both right-hand sides carry it: the solves of the pass fail their verdict
        snap = {k: getattr(s.st, k).clone() for k in ("x", "y", "zl", "zu", "xl", "xu")}
        with pytest.raises(M.SolveException):
            s.iteration_body()
        torch.cuda.synchronize()
        fill = lambda t: torch.nan_to_num(t, nan=-12345.0)
        same = {k: bool(torch.equal(fill(getattr(s.st, k)), fill(v))) for k, v in snap.items()}
        stats = hip.mpc_ahead_stats(s._native)
        s.close()
        return same, stats

    for ahead in (False, True):
        same, stats = run(ahead)
        assert all(same.values()), (ahead, same)
        assert (stats[0] > 0) == ahead


def test_fused_per_variable_passes_are_bitwise_the_separate_kernels(hip):
    """csrc/kkt.hip, solve_pre_kernel / solve_post_kernel / resid_tail_kernel: reduce_rhs! + the condensation, its inverse +
    finish_aug_solve!, and the tail of _kktmul! each run as ONE kernel in which a variable gathers its terms through the
    inverse of the bound lists, in the order the scatters applied them -- same operations, same operands, same bits as the
    separate kernels (MADQP_KKT_FUSE=0), with bounds on slacks, variables in both lists, equality rows and Gondzio
    corrections in play."""
    import os

    def run(qp, fuse, **kw):
        os.environ["MADQP_KKT_FUSE"] = "1" if fuse else "0"
        try:
            s = M.MPCSolver(to_device(qp, hip), hip, driver="native", **kw)  # (read when the KKT object is created)
            r = s.solve()
            s.close()
        finally:
            os.environ.pop("MADQP_KKT_FUSE", None)
        return r

    both = Q.synthetic_qp(91, 150, 60)  # every variable in BOTH bound lists, half of the rows equalities
    both.lvar[:] = -1.0 - np.arange(both.lvar.size) % 3
    both.uvar[:] = 1.0 + np.arange(both.uvar.size) % 2
    both.lcon[::2] = both.ucon[::2]
    cases = ((Q.synthetic_qp(77, 200, 80), dict(max_ncorr=3)),
             (Q.synthetic_qp(78, 120, 50, "lp"), dict(regularization=M.AdaptiveRegularization(1e-8, -1e-9, 1e-9))),
             (Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
              dict(step_rule=M.MehrotraAdaptiveStep(0.99), regularization=M.FixedRegularization(1e-8, -1e-8))),
             (both, dict(max_ncorr=2, regularization=M.FixedRegularization(1e-8, -1e-8))),
             (Q.synthetic_qp(79, 140, 60, "lp"), dict(refine_steps=1)))
    for qp, kw in cases:
        a, b = run(qp, False, **kw), run(qp, True, **kw)
        assert a["status"] == b["status"] and a["iter"] == b["iter"] and a["trace"] == b["trace"]
        assert np.array_equal(a["solution"], b["solution"]) and a["objective"] == b["objective"]
        assert np.array_equal(a["multipliers"], b["multipliers"])
        assert np.array_equal(a["multipliers_L"], b["multipliers_L"]) and np.array_equal(a["multipliers_U"], b["multipliers_U"])


def test_fused_iteration_is_bitwise_the_sequential_one_with_two_readbacks(hip):
    """csrc/mpc.hip, body_fused: the reductions of an iteration are queued in the result block and read back twice
    (after the corrector, which also decides the first Gondzio trial; after the update + the next termination test;
    once more per further tried correction) instead of once per reduction -- sigma, mu, tau, a trial's mu_c and the
    first trial's step lengths are formed on the device from the block and read there by the kernels that need them; kernels, arithmetic and stream order are those of the sequential form,
    so traces and iterates agree bit for bit -- also when the first factorisation of an iteration fails (noticed one
    phase later now) and the x100 retry takes over."""
    import os

    def run(qp, fused, **kw):
        os.environ["MADQP_MPC_FUSED"] = "1" if fused else "0"
        kw = dict(dict(refine_steps=0), **kw)  # (the reference's solve_system! unless a case says otherwise: at these sizes
        try:                                   # the AUTO rule of options.py would add a refinement step)
            s = M.MPCSolver(to_device(qp, hip), hip, driver="native", **kw)  # the switch is read when the loop object
            r = s.solve()                                                    # is created (initialize)
            r["readbacks"] = hip.mpc_readbacks(s._native)
            r["ahead"] = hip.mpc_ahead_stats(s._native)
            s.close()
        finally:
            os.environ.pop("MADQP_MPC_FUSED", None)
        return r

    retry_qp, free = Q.synthetic_qp(8, 20, 8), 3
    retry_qp.lvar[free], retry_qp.uvar[free] = -np.inf, np.inf
    retry_qp.H = np.diag(np.diag(retry_qp.H))
    retry_qp.A[:, free] = 0.0
    retry_qp.H[free, free] = -1e-7
    retry_qp.q[free] = 0.0
    cases = ((Q.synthetic_qp(77, 200, 80), dict(regularization=M.FixedRegularization(1e-8, -1e-8)), True),
             (Q.synthetic_qp(78, 120, 50, "lp"), dict(regularization=M.AdaptiveRegularization(1e-8, -1e-9, 1e-9)), True),
             (Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
              dict(regularization=M.FixedRegularization(1e-8, -1e-8), step_rule=M.ConservativeStep(0.99)), True),
             (retry_qp, dict(regularization=M.FixedRegularization(1e-8, -1e-8), max_iter=4), False),
             (Q.synthetic_qp(77, 200, 80), dict(regularization=M.FixedRegularization(1e-8, -1e-8), max_ncorr=3), True),
             (Q.synthetic_qp(81, 150, 60, "lp"), dict(regularization=M.FixedRegularization(1e-8, -1e-8), max_ncorr=2), True),
             # round 5: the refinement steps are queued launches like the rest -- the fused form carries them too
             (Q.synthetic_qp(79, 140, 60, "lp"), dict(regularization=M.FixedRegularization(1e-8, -1e-8), refine_steps=1), True),
             (Q.synthetic_qp(77, 200, 80), dict(regularization=M.FixedRegularization(1e-8, -1e-8), max_ncorr=3, refine_steps=1), True),
             # the residual test of solve_system! switched on (src/linear_solver.jl:41): the device-side verdict compares too
             (Q.synthetic_qp(83, 90, 30), dict(regularization=M.FixedRegularization(1e-8, -1e-8), check_residual=True,
                                               tol_linear_solve=1e-6, max_ncorr=1), True))
    def run_waiting(qp, **kw):  # the fused form of round 4: nothing queued behind a read-back
        os.environ["MADQP_MPC_AHEAD"] = "0"
        try:
            return run(qp, True, **kw)
        finally:
            os.environ.pop("MADQP_MPC_AHEAD", None)

    for qp, kw, converges in cases:
        a, b, w = run(qp, False, **kw), run(qp, True, **kw), run_waiting(qp, **kw)
        assert a["status"] == b["status"] == w["status"] and a["iter"] == b["iter"] == w["iter"]
        assert (a["status"] == M.SOLVE_SUCCEEDED) == converges
        assert a["trace"] == b["trace"] == w["trace"] and a["n_factorizations"] == b["n_factorizations"] == w["n_factorizations"]
        assert np.array_equal(a["solution"], b["solution"]) and a["objective"] == b["objective"]
        assert np.array_equal(a["solution"], w["solution"]) and a["objective"] == w["objective"]
        # queued ahead (round 5): a pass finds its K assembled by the pass before it -- all but the first, and the last ones,
        # which start within 100 x the tolerance and queue nothing; at most one assembly is for nobody
        assert w["ahead"] == (0, 0)
        queued, used = b["ahead"]
        assert used <= queued <= b["iter"] and queued - used <= 1
        if converges and b["iter"] >= 6:
            assert used >= 2, (b["ahead"], b["iter"])
        if converges and not kw.get("max_ncorr"):  # one read-back for the first termination test, then two per iteration
            # (with or without refinement steps: they add launches, not read-backs)
            assert b["readbacks"] == 1 + 2 * b["iter"], (b["readbacks"], b["iter"])
        elif converges:  # plus one per tried Gondzio correction beyond the first, at most max_ncorr of them per iteration
            assert 1 + 2 * b["iter"] <= b["readbacks"] <= 1 + (1 + kw["max_ncorr"]) * b["iter"]


def test_batch_of_independent_qps(hip):
    """BASELINE configs[3] in miniature: a batch of independent QPs solved with several contexts /
    streams in flight gives, problem by problem, the oracle's result (status, iterations, solution)."""
    seeds = list(range(100, 112))

    def make(be, i):
        qp = Q.synthetic_qp(seeds[i], 64, 24)
        return to_device(qp, be)

    res = M.solve_batch(make, range(len(seeds)), streams=4, regularization=M.FixedRegularization(1e-8, -1e-8))
    assert sorted(res) == list(range(len(seeds)))
    for i, seed in enumerate(seeds):
        ref = mpc.solve(Q.synthetic_qp(seed, 64, 24), kkt_system="condensed",
                        regularization=mpc.FixedRegularization(1e-8, -1e-8))
        r = res[i]
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
    assert M.shard(range(10), 1, 4) == [1, 5, 9] and M.shard(range(10), 0, 1) == list(range(10))


def test_full_size_c_main_properties(hip):
    """... and, at this size only, the ORACLE itself: tests/golden/cmain_first_iteration.json holds the trace of
    oracle/mpc.py run at n_x = 50 000, m = 20 000 on a GPU box's host (`bench.py --cpu-full`, 88 s per iteration): the
    device iterate before and after the first iteration must match it to the stated per-iteration tolerance."""
    torch.cuda.empty_cache()
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cmain_first_iteration.json")))
    assert (golden["nx"], golden["m"], golden["seed"]) == (50000, 20000, 20250614 + 1)
    full_size_properties(hip, 50000, 20000, 20250614 + 1, golden=golden)


def full_size_properties(hip, nx, m, seed, golden=None):
    """BASELINE metric size (n_x = 50 000, m = 20 000; also configs[4], tests/test_gpu_configs.py): far beyond the oracle, so size-independent
    properties through the same C ABI: (a) the on-device generator reproduces randomly chosen tiles of
    A, H and q bit for bit; (b) the assembled condensed K equals H + Sigma + A' Theta A on sampled
    entries; (c) the factorisation succeeds and every solve_system! of two IPM iterations (condense,
    two triangular sweeps, decondense) satisfies the UNREDUCED KKT system (residual through mul!);
    the iteration makes progress."""
    dq = M.DeviceQP.synthetic(hip, seed, nx, m)
    rng = np.random.default_rng(0)
    # (a) generator tiles
    kA, kH = Q.stream_key(seed, Q.STREAM_A), Q.stream_key(seed, Q.STREAM_H)
    for _ in range(4):
        r0, c0 = int(rng.integers(0, m - 64)), int(rng.integers(0, nx - 64))
        idx = (np.arange(r0, r0 + 64, dtype=np.uint64)[:, None] * np.uint64(nx)
               + np.arange(c0, c0 + 64, dtype=np.uint64)[None, :])
        np.testing.assert_array_equal(dq.A[r0:r0 + 64, c0:c0 + 64].cpu().numpy(),
                                      Q.gen_normal(kA, idx.ravel()).reshape(64, 64))
        i0, j0 = int(rng.integers(0, nx - 64)), int(rng.integers(0, nx - 64))
        ii, jj = np.arange(i0, i0 + 64, dtype=np.uint64)[:, None], np.arange(j0, j0 + 64, dtype=np.uint64)[None, :]
        ref = Q.gen_normal(kH, (np.minimum(ii, jj) * np.uint64(nx) + np.maximum(ii, jj)).ravel()).reshape(64, 64)
        ref = ref * (1.0 / np.sqrt(nx))
        ref[ii == jj] = 3.0 + ref[ii == jj]
        np.testing.assert_array_equal(dq.H[i0:i0 + 64, j0:j0 + 64].cpu().numpy(), ref)
    np.testing.assert_array_equal(dq.q[:1000].cpu().numpy(), Q.gen_q(seed, nx)[:1000])

    s = M.MPCSolver(dq, hip, max_iter=300, step_rule=M.AdaptiveStep(0.995),
                    regularization=M.FixedRegularization(1e-8, -1e-8), mu_min=1e-12,
                    max_ncorr=golden["max_ncorr"] if golden else 0)
    s.initialize()
    assert s.last_residual_ratio < 1e-8
    st = s.st
    # (b) assembled K on sampled lower-triangle entries (state after the start-point factorisation is
    # gone: rebuild without factorising)
    hip.set_aug_diagonal_reg(st, 1e-8, -1e-8)
    s.kkt.build_kkt()
    Kp, ld = hip.kkt_matrix(s.kkt._h, nx)
    sig = st.pr_diag.cpu().numpy()
    theta = sig[nx:] / (1.0 - (-1e-8) * sig[nx:])
    for _ in range(6):
        j = int(rng.integers(0, nx - 8))
        i = int(rng.integers(j, nx - 8))
        got = hip.read_doubles(Kp + 8 * (i + j * ld), 8)  # K[i:i+8, j]
        Ai, Aj = dq.A[:, i:i + 8].cpu().numpy(), dq.A[:, j].cpu().numpy()
        ref = dq.H[i:i + 8, j].cpu().numpy() + (Ai * (theta * Aj)[:, None]).sum(axis=0)
        ref[np.arange(i, i + 8) == j] += sig[j]
        scale = np.abs(dq.H[i:i + 8, j].cpu().numpy()) + (np.abs(Ai) * (theta * np.abs(Aj))[:, None]).sum(axis=0)
        assert np.max(np.abs(got - ref) / scale) < 1e-13
    s.kkt.linear_solver.factorize()
    assert s.kkt.linear_solver.is_factorized()
    # (c) two iterations
    assert s.iteration_head() is None
    pr0 = s.inf_pr
    for _ in range(2):
        s.iteration_body()
        assert s.last_residual_ratio < 1e-7, s.last_residual_ratio
        assert s.iteration_head() is None
    assert s.inf_pr < pr0 and 0 < s.alpha_p <= 1 and 0 < s.alpha_d <= 1
    if golden:  # the oracle's own trace at this size (mu >= 1e-4: tolerance 1e-9)
        for t, g in zip(s.trace, golden["trace"]):
            for key in ("inf_pr", "inf_du", "inf_compl", "mu", "alpha_p", "alpha_d"):
                assert close(t[key], g[key], 1e-9), (t["k"], key, t[key], g[key])
    s.close()
    del s, dq
    torch.cuda.empty_cache()
