"""GPU: the lock-step batched driver (csrc/batch.hip, BASELINE configs[3]) against the CPU oracle, problem by
problem: same status, same iteration count, solution / objective / multipliers within the tolerance of
test_gpu_solver.py.  Problems of one batch converge at different iterations (status mask), one may fail
without disturbing the others, the order of the problems does not matter."""
import numpy as np
import pytest
import torch

import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q

pytestmark = pytest.mark.gpu
REG = M.FixedRegularization(1e-8, -1e-8)
OREG = mpc.FixedRegularization(1e-8, -1e-8)


def to_device(qp, be):
    return M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)


def close(a, b, tol):
    return abs(a - b) <= tol * max(1.0, abs(a), abs(b))


def check_against_oracle(qps, res, **okw):
    for i, (qp, r) in enumerate(zip(qps, res)):
        ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG, **okw)
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED, (i, r["status"], ref["status"])
        assert r["iter"] == ref["iter"], (i, r["iter"], ref["iter"])
        assert close(r["objective"], ref["objective"], 1e-9), (i, r["objective"], ref["objective"])
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
        assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) <= 1e-6
        t = ref["trace"][-1]
        assert close(r["inf_pr"], t["inf_pr"], 1e-6) and close(r["inf_du"], t["inf_du"], 1e-6)


@pytest.mark.parametrize("n,m,B,family", [(64, 24, 12, "wigner"), (40, 16, 6, "lp"), (300, 120, 5, "wigner"),
                                          (130, 1, 3, "wigner"),
                                          # H products: from the lower triangle up to n_x = 512 (all eight 64-column chunks
                                          # at 512, a partial last chunk at 300 above), the full-matrix passes beyond
                                          (512, 200, 3, "wigner"), (576, 64, 2, "wigner")])
def test_batched_vs_oracle(hip, n, m, B, family):
    qps = [Q.synthetic_qp(500 + 7 * i + n, n, m, family) for i in range(B)]
    s = M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip, regularization=REG)
    res = s.solve()
    s.close()
    assert len({r["iter"] for r in res}) > 1 or B < 4  # the mask is exercised: different iteration counts
    check_against_oracle(qps, res)


def test_batched_equality_rows_and_step_rule(hip):
    qps = []
    for i in range(6):
        qp = Q.synthetic_qp(900 + i, 48, 20)
        qp.lcon[[2, 7]] = qp.ucon[[2, 7]] = 0.25  # equality rows: Theta = -1/delta_d
        qps.append(qp)
    s = M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip, regularization=REG,
                           step_rule=M.ConservativeStep(0.99))
    res = s.solve(check_every=3)
    s.close()
    check_against_oracle(qps, res, step_rule=mpc.ConservativeStep(0.99))
    with pytest.raises(ValueError):  # default delta_d = 0 with equality rows
        M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip)


@pytest.mark.parametrize("variant", ["gondzio", "mehrotra_step", "adaptive_reg"])
def test_batched_options(hip, variant):
    """Gondzio corrections (per-problem number of accepted corrections), MehrotraAdaptiveStep (element
    reads at the blocking indices) and AdaptiveRegularization inside the lock-step engine."""
    qps = [Q.synthetic_qp(1200 + i, 72, 30) for i in range(7)]
    dq = [to_device(q, hip) for q in qps]
    kw, okw = dict(regularization=REG), dict(regularization=OREG)
    if variant == "gondzio":
        kw["max_ncorr"], okw["max_ncorr"] = 3, 3
    elif variant == "mehrotra_step":
        kw["step_rule"], okw["step_rule"] = M.MehrotraAdaptiveStep(0.99), mpc.MehrotraAdaptiveStep(0.99)
    else:
        kw["regularization"] = M.AdaptiveRegularization(1e-8, -1e-9, 1e-9)
        okw["regularization"] = mpc.AdaptiveRegularization(1e-8, -1e-9, 1e-9)
    s = M.BatchedMPCSolver(dq, hip, **kw)
    res = s.solve()
    s.close()
    for i, (qp, r) in enumerate(zip(qps, res)):
        ref = mpc.solve(qp, kkt_system="condensed", **okw)
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"], (i, r["iter"], ref["iter"])
        assert close(r["objective"], ref["objective"], 1e-9)
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7


def test_batched_is_the_per_problem_driver(hip):
    """Same library, two drivers: the batch and MPCSolver agree problem by problem; a problem that
    breaks down (indefinite H) ends with an error status and leaves the others untouched; a permuted
    batch gives the permuted results bit for bit."""
    qps = [Q.synthetic_qp(700 + i, 96, 40) for i in range(8)]
    bad = 3
    qps[bad].H = qps[bad].H - 1e6 * np.eye(96)
    dq = [to_device(q, hip) for q in qps]
    s = M.BatchedMPCSolver(dq, hip, regularization=REG)
    res = s.solve()
    s.close()
    assert res[bad]["status"] in (M.ERROR_IN_STEP_COMPUTATION, -1)
    for i, r in enumerate(res):
        if i == bad:
            continue
        one = M.MPCSolver(dq[i], hip, regularization=REG)
        ref = one.solve()
        one.close()
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-8
    perm = [5, 0, 7, 2, 1, 6, 4]
    s2 = M.BatchedMPCSolver([dq[i] for i in perm], hip, regularization=REG)
    res2 = s2.solve()
    s2.close()
    for k, i in enumerate(perm):
        assert res2[k]["iter"] == res[i]["iter"] and np.array_equal(res2[k]["solution"], res[i]["solution"])


def test_batched_rejects_mixed_patterns(hip):
    a, b = Q.synthetic_qp(1, 20, 8), Q.synthetic_qp(2, 20, 8)
    b.uvar[3] = np.inf
    with pytest.raises(ValueError):
        M.BatchedMPCSolver([to_device(a, hip), to_device(b, hip)], hip, regularization=REG)
    with pytest.raises(ValueError):
        M.BatchedMPCSolver([to_device(a, hip)], hip, regularization=REG, kkt_system="normal")


def test_batched_regularization_retry(hip):
    """src/linear_solver.jl:6-17 inside the lock-step engine: one problem of the batch has a slightly indefinite
    Hessian entry on a free variable, so its factorisation fails at delta_w = 1e-8 and succeeds after the x100 retry;
    it must follow the oracle's retry count (n_factorizations) and iterates while the other problems -- which sit the
    masked retry rounds out -- follow theirs.  The free variable is free in EVERY problem (one bound pattern per
    batch); only problem 2 has the negative curvature."""
    qps, free, bad = [], 3, 2
    for i in range(5):
        qp = Q.synthetic_qp(40 + i, 20, 8)
        qp.lvar[free], qp.uvar[free] = -np.inf, np.inf
        qp.H = np.diag(np.diag(qp.H))
        qp.A[:, free] = 0.0
        qp.q[free] = 0.0
        if i == bad:
            qp.H[free, free] = -1e-7
        qps.append(qp)
    s = M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip, regularization=REG, max_iter=4)
    res = s.solve()
    s.close()
    for i, (qp, r) in enumerate(zip(qps, res)):
        ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG, max_iter=4)
        assert r["status"] == ref["status"] and r["iter"] == ref["iter"], (i, r["status"], ref["status"], r["iter"], ref["iter"])
        assert r["n_factorizations"] == ref["n_factorizations"], (i, r["n_factorizations"], ref["n_factorizations"])
        t = ref["trace"][-1]
        assert close(r["inf_pr"], t["inf_pr"], 1e-6) and close(r["inf_du"], t["inf_du"], 1e-6), i
        assert close(r["mu"], t["mu"], 1e-6) and close(r["del_w"], t["del_w"], 1e-12), i
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-6, i
    assert res[bad]["n_factorizations"] > res[0]["n_factorizations"]  # only the bad problem paid for retries


def test_batched_normal_equations(hip):
    """The reference's own NormalKKTSystem (src/KKT/normalkkt.jl: A Sigma^-1 A', LP only, equality rows without dual
    regularization) inside the lock-step engine: problem by problem the oracle's normal-equations solve -- the case
    of test/runtests.jl:165-180 (simple_lp-like: equality rows, default FixedRegularization(1e-8, 0))."""
    qps = []
    for i in range(6):
        qp = Q.synthetic_qp(2100 + i, 40, 16, "lp")
        qp.lcon[[1, 6]] = qp.ucon[[1, 6]] = 0.3  # equality rows
        qps.append(qp)
    reg, oreg = M.FixedRegularization(1e-8, 0.0), mpc.FixedRegularization(1e-8, 0.0)
    s = M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip, kkt_system="normal", regularization=reg)
    res = s.solve()
    s.close()
    for i, (qp, r) in enumerate(zip(qps, res)):
        ref = mpc.solve(qp, kkt_system="normal", regularization=oreg)
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED, (i, r["status"], ref["status"])
        assert r["iter"] == ref["iter"], (i, r["iter"], ref["iter"])
        assert close(r["objective"], ref["objective"], 1e-9)
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
        assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) <= 1e-6
    one = Q.simple_lp()  # test/runtests.jl:24-55, as a batch of one
    s = M.BatchedMPCSolver([to_device(one, hip)], hip, kkt_system="normal", regularization=reg)
    r = s.solve()[0]
    s.close()
    assert r["status"] == M.SOLVE_SUCCEEDED and abs(r["objective"] - 1.0) < 1e-8 and np.allclose(r["solution"], [0.5, 0.5], atol=1e-8)
    with pytest.raises(ValueError):  # a QP: NormalKKTSystem supports only linear programs
        M.BatchedMPCSolver([to_device(Q.synthetic_qp(1, 20, 8), hip)], hip, kkt_system="normal", regularization=reg)


def _edge_batch(case, B=3):
    qps = []
    for i in range(B):
        qp = Q.synthetic_qp(700 + 11 * i, 40, 15)
        if case == "lower_bounds_only":  # nub = 0 on the variables, rows bounded below only
            qp.uvar[:] = np.inf
            qp.ucon[:] = np.inf
        elif case == "upper_bounds_only":
            qp.lvar[:] = -np.inf
            qp.lcon[:] = -np.inf
            qp.x0[:] = 0.5
        elif case == "mixed":  # free, lower-only, upper-only and boxed variables; equality, one-sided and ranged rows
            qp.lvar[0::4], qp.uvar[0::4] = -np.inf, np.inf
            qp.uvar[1::4] = np.inf
            qp.lvar[2::4] = -np.inf
            qp.ucon[0::3] = qp.lcon[0::3] = 0.1
            qp.ucon[1::3] = np.inf
        elif case == "no_bounds":  # both bound lists empty, every row an equality (the start point's 0 / 0, src/solver.jl:93-94)
            qp.lvar[:], qp.uvar[:] = -np.inf, np.inf
            qp.ucon[:] = qp.lcon[:] = 0.25
        qps.append(qp)
    return qps


@pytest.mark.parametrize("case", ["lower_bounds_only", "upper_bounds_only", "mixed", "no_bounds"])
def test_batched_edge_shapes(hip, case):
    """The edges of the index lists through the lock-step engine: an empty upper or lower list, one-sided rows, every kind of
    variable and row at once, and no bound at all (nlb = nub = 0: the workgroup programs' loops over the lists run zero
    times, the start point divides 0 by 0 and adds the NaN to nothing) -- problem by problem against the oracle."""
    qps = _edge_batch(case)
    s = M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip, regularization=REG)
    res = s.solve()
    s.close()
    check_against_oracle(qps, res)
