"""GPU: randomised bound patterns (free / one-sided / boxed variables; equality / >= / <= / ranged rows) through
every driver of the library against the CPU oracle.  The synthetic family of the benchmark has every bound
finite; here ind_lb != ind_ub, slack bounds are one-sided and some rows are equalities."""
import numpy as np
import pytest

import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q

pytestmark = pytest.mark.gpu
REG, OREG = M.FixedRegularization(1e-8, -1e-8), mpc.FixedRegularization(1e-8, -1e-8)
CASES = [(1, 5, 3, False), (2, 17, 9, False), (3, 40, 25, False), (4, 64, 10, True), (5, 130, 70, False),
         (6, 200, 90, True), (7, 33, 0, False), (8, 1, 1, False), (9, 90, 60, False), (10, 257, 120, False)]


def to_device(qp, be, sparse=False):
    return M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0,
                                 sparse=sparse)


def same(r, ref, what):
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED, (what, r["status"], ref["status"])
    assert r["iter"] == ref["iter"], (what, r["iter"], ref["iter"])
    assert abs(r["objective"] - ref["objective"]) <= 1e-8 * max(1.0, abs(ref["objective"])), what
    assert np.max(np.abs(r["solution"] - ref["solution"]), initial=0.0) <= 1e-6, what
    assert np.max(np.abs(r["multipliers"] - ref["multipliers"]), initial=0.0) <= 1e-5, what


@pytest.mark.parametrize("seed,n,m,lp", CASES)
def test_random_patterns_all_drivers(hip, seed, n, m, lp):
    qp = Q.random_qp(seed, n, m, lp)
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG)
    for driver in ("python", "native"):
        s = M.MPCSolver(to_device(qp, hip), hip, regularization=REG, driver=driver)
        same(s.solve(), ref, driver)
        s.close()
    if m and n > 1:  # the same problem with the Jacobian handed over as CSR
        s = M.MPCSolver(to_device(qp, hip, sparse=True), hip, regularization=REG)
        same(s.solve(), ref, "sparse front end")
        s.close()
    b = M.BatchedMPCSolver([to_device(qp, hip)], hip, regularization=REG)
    same(b.solve()[0], ref, "batched engine, B = 1")
    b.close()
    # MadNLP's default formulation (K2) with the reference's default regularization, dense and CSR Jacobian
    kref = mpc.solve(qp, kkt_system="K2")
    for sparse in ((False, True) if m and n > 1 else (False,)):
        s = M.MPCSolver(to_device(qp, hip, sparse=sparse), hip, kkt_system="augmented", driver="native")
        same(s.solve(), kref, f"augmented system, sparse={sparse}")
        s.close()
    if lp:  # the reference's own formulation (normal equations, delta_d = 0), dense and CSR Jacobian
        nref = mpc.solve(qp, kkt_system="normal", regularization=mpc.FixedRegularization(1e-8, 0.0))
        for sparse in (False, True):
            s = M.MPCSolver(to_device(qp, hip, sparse=sparse), hip, kkt_system="normal",
                            regularization=M.FixedRegularization(1e-8, 0.0))
            same(s.solve(), nref, f"normal equations, sparse={sparse}")
            s.close()


def test_random_patterns_batched(hip):
    """One bound pattern, different data: a batch of 9 against the oracle, problem by problem."""
    qps = [Q.random_qp(300 + i, 48, 30, pattern_seed=77) for i in range(9)]
    b = M.BatchedMPCSolver([to_device(q, hip) for q in qps], hip, regularization=REG)
    res = b.solve(check_every=2)
    b.close()
    for i, (qp, r) in enumerate(zip(qps, res)):
        same(r, mpc.solve(qp, kkt_system="condensed", regularization=OREG), f"batch member {i}")


def test_fixed_variables_relax_bound(hip):
    """Fixed variables (FX bounds of instance files): RelaxBound is the default of the condensed system, as in the
    reference (src/utils.jl:81); every driver follows the oracle; the normal equations refuse without the option."""
    qp = Q.random_qp(5, 40, 22)
    fixed = [3, 8, 31]
    qp.lvar[fixed] = qp.uvar[fixed] = np.array([0.25, -0.4, 0.0])
    ref = mpc.solve(qp, kkt_system="condensed", regularization=OREG, fixed_variable_treatment="relax_bound")
    for mk in (lambda: M.MPCSolver(to_device(qp, hip), hip, regularization=REG),
               lambda: M.MPCSolver(to_device(qp, hip), hip, regularization=REG, driver="native"),
               lambda: M.MPCSolver(to_device(qp, hip, sparse=True), hip, regularization=REG)):
        s = mk()
        r = s.solve()
        s.close()
        same(r, ref, "fixed variables")
        assert np.allclose(r["solution"][fixed], [0.25, -0.4, 0.0], atol=1e-7)
    b = M.BatchedMPCSolver([to_device(qp, hip)], hip, regularization=REG)
    same(b.solve()[0], ref, "fixed variables, batched")
    b.close()
    with pytest.raises(NotImplementedError):
        M.MPCSolver(to_device(qp, hip), hip, regularization=REG, fixed_variable_treatment="error")
    lp = Q.random_qp(6, 30, 12, lp=True)
    lp.lvar[2] = lp.uvar[2] = 0.1
    with pytest.raises(NotImplementedError):
        M.MPCSolver(to_device(lp, hip), hip, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0),
                    fixed_variable_treatment="error")
    s = M.MPCSolver(to_device(lp, hip), hip, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0),
                    fixed_variable_treatment="relax_bound")
    r = s.solve()
    s.close()
    same(r, mpc.solve(lp, kkt_system="normal", regularization=mpc.FixedRegularization(1e-8, 0.0),
                      fixed_variable_treatment="relax_bound"), "fixed variable, normal equations")


@pytest.mark.parametrize("eq", [(), (0, 1, 2, 7)])
def test_dummy_qp_with_fixed_variables(hip, eq):
    """The cases test/runtests.jl:71-75 names (DenseDummyQP n = 20, m = 15, fixed variables 1, 2, with and without
    equality rows 1, 2, 3, 8), through the condensed system with RelaxBound."""
    qp = Q.dummy_qp(20, 15, equality_cons=eq, fixed_variables=(0, 1))
    ref = mpc.solve(qp, kkt_system="K2", regularization=OREG, fixed_variable_treatment="relax_bound")
    for driver in ("python", "native"):
        s = M.MPCSolver(to_device(qp, hip), hip, regularization=REG, driver=driver)
        r = s.solve()
        s.close()
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
        assert abs(r["objective"] - ref["objective"]) < 1e-5 and np.max(np.abs(r["solution"] - ref["solution"])) < 1e-5
        assert np.max(np.abs(r["solution"][:2])) < 1e-7


@pytest.mark.parametrize("lp,ksys", [(True, "normal"), (False, "condensed"), (False, "augmented")])
def test_fixed_variables_make_parameter(hip, lp, ksys):
    """MadNLP.MakeParameter (src/utils.jl:81: the treatment of fixed variables for every KKT system that is not
    condensed, i.e. the default with the normal equations): fixed variables leave the problem.  Checked against the
    oracle on the hand-reduced model, and by the optimality conditions of the FULL model at the returned point;
    dense and CSR Jacobian, dense and diagonal Hessian."""
    import torch

    qp = Q.random_qp(31, 50, 24, lp)
    fixed = np.array([3, 8, 31, 49])
    qp.lvar[fixed] = qp.uvar[fixed] = np.array([0.25, -0.4, 0.0, 0.1])
    free = np.setdiff1d(np.arange(50), fixed)
    xf = qp.lvar[fixed]
    red = Q.DenseQP(H=qp.H[np.ix_(free, free)], q=qp.q[free] + qp.H[np.ix_(free, fixed)] @ xf, A=qp.A[:, free],
                    lvar=qp.lvar[free], uvar=qp.uvar[free], lcon=qp.lcon - qp.A[:, fixed] @ xf,
                    ucon=qp.ucon - qp.A[:, fixed] @ xf, x0=qp.x0[free],
                    c0=qp.c0 + qp.q[fixed] @ xf + 0.5 * xf @ qp.H[np.ix_(fixed, fixed)] @ xf)
    reg, oreg = (M.FixedRegularization(1e-8, 0.0), mpc.FixedRegularization(1e-8, 0.0)) if ksys != "condensed" else (REG, OREG)
    ref = mpc.solve(red, kkt_system={"normal": "normal", "condensed": "condensed", "augmented": "K2"}[ksys],
                    regularization=oreg)
    assert ref["status"] == M.SOLVE_SUCCEEDED
    kw = dict(kkt_system=ksys, regularization=reg, fixed_variable_treatment="make_parameter")
    if ksys == "normal":
        kw.pop("fixed_variable_treatment")  # the default there, as in the reference
    for sparse in (False, True):
        s = M.MPCSolver(to_device(qp, hip, sparse=sparse), hip, **kw)
        assert s.nx == 46
        r = s.solve()
        s.close()
        assert r["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
        assert abs(r["objective"] - ref["objective"]) <= 1e-8 * max(1.0, abs(ref["objective"]))
        x, y, zl, zu = r["solution"], r["multipliers"], r["multipliers_L"], r["multipliers_U"]
        assert len(x) == 50 and np.array_equal(x[fixed], xf) and np.max(np.abs(x[free] - ref["solution"])) <= 1e-6
        # optimality of the full model: stationarity with the parameters' reduced costs as their bound multipliers
        assert np.max(np.abs(qp.H @ x + qp.q + qp.A.T @ y - zl + zu)) <= 1e-6
        assert zl.min() >= 0 and zu.min() >= 0 and np.all(np.minimum(zl[fixed], zu[fixed]) == 0)
        assert np.max(np.abs(r["constraints"] - qp.A @ x)) <= 1e-6
    if not lp:  # diagonal Hessian as a vector
        qd = Q.random_qp(31, 50, 24, False)
        qd.H = np.diag(np.diag(qd.H))
        qd.lvar[fixed] = qd.uvar[fixed] = xf
        dq = to_device(qd, hip, sparse=True)
        dq.H = torch.as_tensor(np.diag(qd.H).copy(), device=hip.device)
        s = M.MPCSolver(dq, hip, kkt_system=ksys, regularization=reg, fixed_variable_treatment="make_parameter")
        r = s.solve()
        s.close()
        x, y = r["solution"], r["multipliers"]
        assert r["status"] == M.SOLVE_SUCCEEDED and np.array_equal(x[fixed], xf)
        assert np.max(np.abs(qd.H @ x + qd.q + qd.A.T @ y - r["multipliers_L"] + r["multipliers_U"])) <= 1e-6
        rb = M.MPCSolver(dq, hip, kkt_system=ksys, regularization=reg, fixed_variable_treatment="relax_bound")
        r2 = rb.solve()
        rb.close()
        assert abs(r2["objective"] - r["objective"]) <= 1e-6 * max(1.0, abs(r["objective"]))  # both treatments agree
