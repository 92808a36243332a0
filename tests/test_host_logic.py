"""CPU: the product's host logic (madqp_jl_amd/solver.py, kkt.py, options.py, qp.py) driven through the
numpy test double of the C ABI (tests/fake_backend.py), compared with the oracle.  Covers control flow
the GPU-less container can check: loop order, regularization retry, step rules, Gondzio, scaling."""
import numpy as np
import pytest
import torch

import madqp_jl_amd as M
from fake_backend import FakeBackend
from oracle import mpc
from oracle import qp as Q


def run(qp, be=None, **opts):
    opts.setdefault("regularization", M.FixedRegularization(1e-8, -1e-8))
    dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
    s = M.MPCSolver(dq, be or FakeBackend(), **opts)
    return s, s.solve()


def oracle(qp, **opts):
    opts.setdefault("regularization", mpc.FixedRegularization(1e-8, -1e-8))
    return mpc.solve(qp, kkt_system="condensed", **opts)


def assert_same_trace(r, ref, tol=1e-7):
    assert r["status"] == ref["status"] and r["iter"] == ref["iter"]
    for a, b in zip(r["trace"], ref["trace"]):
        for k in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "inf_compl", "mu"):
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(b[k])), (a["k"], k, a[k], b[k])


@pytest.mark.parametrize("make,ncorr", [
    (lambda: Q.hs21(), 0), (lambda: Q.simple_lp(), 0), (lambda: Q.dummy_qp(10, 5), 0),
    (lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)), 5),
    (lambda: Q.synthetic_qp(20250614, 60, 24), 0), (lambda: Q.synthetic_qp(20250614, 60, 24), 3),
    (lambda: Q.synthetic_qp(20250615, 30, 12, "lp"), 0),
])
def test_driver_matches_oracle(make, ncorr):
    qp = make()
    _, r = run(qp, max_ncorr=ncorr)
    ref = oracle(qp, max_ncorr=ncorr)
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert_same_trace(r, ref)
    assert abs(r["objective"] - ref["objective"]) <= 1e-9 * max(1, abs(ref["objective"]))
    assert np.max(np.abs(r["solution"] - ref["solution"])) < 1e-7


@pytest.mark.parametrize("rule", ["ConservativeStep", "AdaptiveStep", "MehrotraAdaptiveStep"])
def test_step_rules(rule):
    qp = Q.dummy_qp(10, 5)
    _, r = run(qp, step_rule=getattr(M, rule)(0.99))
    ref = oracle(qp, step_rule=getattr(mpc, rule)(0.99))
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert_same_trace(r, ref)


def test_adaptive_regularization_schedule():
    qp = Q.dummy_qp(10, 5)
    _, r = run(qp, regularization=M.AdaptiveRegularization(1e-8, -1e-9, 1e-9))
    ref = oracle(qp, regularization=mpc.AdaptiveRegularization(1e-8, -1e-9, 1e-9))
    assert_same_trace(r, ref)


def test_scaling_path():
    """MadNLP.set_scaling! (src/solver.jl:148-159): rows > 100 and gradient > 100 get scaled."""
    qp = Q.synthetic_qp(4, 30, 12)
    qp.A[3] *= 1000.0
    qp.ucon[3] *= 1000.0
    qp.q *= 400.0
    qp.H *= 400.0
    s, r = run(qp)
    ref = oracle(qp)
    assert s.obj_scale < 1.0 and float(s.con_scale[3]) < 1.0
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert_same_trace(r, ref)
    assert abs(r["objective"] - ref["objective"]) <= 1e-8 * max(1, abs(ref["objective"]))
    assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) < 1e-5


def test_regularization_retry():
    """src/linear_solver.jl:6-17: a failed factorization multiplies both regularizations by 100,
    at most 3 trials, and does not raise."""
    qp = Q.dummy_qp(10, 5)
    be = FakeBackend()
    dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0)
    s = M.MPCSolver(dq, be, regularization=M.FixedRegularization(1e-8, -1e-8))
    s.initialize()
    n0 = s.kkt.n_factorizations
    s.update_regularization()
    be.fail_factorizations = 2
    s.factorize_regularized_system()
    assert s.kkt.n_factorizations == n0 + 3 and s.kkt.linear_solver.is_factorized()
    assert s.del_w == pytest.approx(1e-8 * 1e4) and s.del_c == pytest.approx(-1e-8 * 1e4)
    s.update_regularization()
    be.fail_factorizations = 5
    s.factorize_regularized_system()  # all three trials fail: the loop just ends (no exception)
    assert not s.kkt.linear_solver.is_factorized()


def test_nan_residual_raises_solve_exception():
    """src/linear_solver.jl:41-43 -> ERROR_IN_STEP_COMPUTATION (src/solver.jl:381-383)."""
    qp = Q.dummy_qp(10, 5)
    dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0)
    s = M.MPCSolver(dq, FakeBackend(), regularization=M.FixedRegularization(1e-8, -1e-8))
    s.initialize()
    s.st.f[0] = float("nan")
    s.update_regularization()
    s.factorize_regularized_system()
    with pytest.raises(M.SolveException):
        s.affine_direction()


def test_equality_rows_need_negative_delta_d():
    qp = Q.simple_lp()
    dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0)
    with pytest.raises(ValueError):
        M.MPCSolver(dq, FakeBackend())
    with pytest.raises(TypeError):
        M.MPCSolver(dq, FakeBackend(), no_such_option=1)


def test_stream_keys_match_the_oracle_generator():
    for seed in (0, 1, 20250614, 2**63 + 5):
        for stream in (1, 2, 3):
            assert M.stream_key(seed, stream) == Q.stream_key(seed, stream)


def test_max_iter_status():
    qp = Q.dummy_qp(10, 5)
    _, r = run(qp, max_iter=2)
    assert r["status"] == M.MAXIMUM_ITERATIONS_EXCEEDED and r["iter"] == 2


@pytest.mark.parametrize("make", [lambda: Q.simple_lp(), lambda: Q.synthetic_qp(20250615, 30, 12, "lp")])
def test_normal_kkt_driver_matches_oracle(make):
    """kkt_system="normal": the reference's own NormalKKTSystem formulation (src/KKT/normalkkt.jl) with the
    reference's default regularization FixedRegularization(1e-8, 0.0) -- equality rows included."""
    qp = make()
    _, r = run(qp, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0))
    ref = mpc.solve(qp, kkt_system="normal", regularization=mpc.FixedRegularization(1e-8, 0.0))
    assert r["status"] == M.SOLVE_SUCCEEDED
    assert_same_trace(r, ref)
    assert np.max(np.abs(r["solution"] - ref["solution"])) < 1e-7


def test_normal_kkt_rejects_qp():
    """src/KKT/normalkkt.jl:45-48."""
    qp = Q.hs21()
    dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0)
    with pytest.raises(ValueError):
        M.MPCSolver(dq, FakeBackend(), kkt_system="normal")


def test_batch_sharding_is_a_partition():
    """QP b -> rank b mod N: every problem is owned by exactly one rank."""
    for world in (1, 2, 3, 8):
        owned = [M.shard(range(37), r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(37))
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def test_device_csr_container_on_cpu():
    """Host logic of the sparse front end (madqp_jl_amd/qp.py: DeviceCSR): CSR of A, CSR of A' through the stored
    permutation, row scaling that follows the values, row maxima, duplicate detection."""
    import torch

    from madqp_jl_amd.qp import DeviceCSR

    rng = np.random.default_rng(0)
    A = rng.standard_normal((9, 13)) * (rng.random((9, 13)) < 0.3)
    A[4] = 0.0
    A[:, 6] = 0.0  # an empty row and an empty column
    c = DeviceCSR.from_dense("cpu", A)
    assert c.nnz == np.count_nonzero(A) and np.array_equal(c.to_dense().numpy(), A)
    ptr, col = c.ptr.numpy(), c.col.numpy()
    assert all(np.all(np.diff(col[ptr[r]:ptr[r + 1]]) > 0) for r in range(9))  # ascending within a row
    assert np.array_equal(c.row_absmax().numpy(), np.abs(A).max(axis=1))
    At = np.zeros((13, 9))
    tp, tc, tv = c.t_ptr.numpy(), c.t_col.numpy(), c.t_val.numpy()
    for r in range(13):
        At[r, tc[tp[r]:tp[r + 1]]] = tv[tp[r]:tp[r + 1]]
    assert np.array_equal(At, A.T)
    scale = torch.arange(1, 10, dtype=torch.float64)
    s = c.scaled(scale)
    assert np.array_equal(s.to_dense().numpy(), A * np.arange(1, 10)[:, None]) and s.ptr is c.ptr
    assert np.array_equal(s.t_val.numpy(), (A * np.arange(1, 10)[:, None]).T[np.nonzero(A.T)])
    with pytest.raises(ValueError):
        DeviceCSR("cpu", 2, 2, [0, 0], [1, 1], [1.0, 2.0])


@pytest.mark.parametrize("sparse,diag_h", [(False, False), (True, False), (True, True)])
def test_make_parameter_elimination_on_cpu(sparse, diag_h):
    """DeviceQP.eliminate_fixed (MadNLP.MakeParameter, src/utils.jl:81) is plain tensor algebra: checked here on
    CPU tensors against the numpy reduction -- free part of H, q + H[:, fixed] x_fix, shifted row bounds, c0."""
    qp = Q.random_qp(17, 30, 12, False)
    if diag_h:
        qp.H = np.diag(np.diag(qp.H))
    fixed = np.array([0, 7, 29])
    xf = np.array([0.3, -0.2, 0.05])
    qp.lvar[fixed] = qp.uvar[fixed] = xf
    free = np.setdiff1d(np.arange(30), fixed)
    dq = M.DeviceQP.from_numpy(torch.device("cpu"), qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0,
                               sparse=sparse)
    if diag_h:
        dq.H = torch.as_tensor(np.diag(qp.H).copy())
    red, tfree, tfixed, txf, shift = dq.eliminate_fixed()
    assert np.array_equal(tfree.numpy(), free) and np.array_equal(tfixed.numpy(), fixed) and np.array_equal(txf.numpy(), xf)
    Hr = red.H.numpy() if not diag_h else np.diag(red.H.numpy())
    assert np.allclose(Hr, qp.H[np.ix_(free, free)], rtol=0, atol=0)
    assert np.allclose(red.q.numpy(), qp.q[free] + qp.H[np.ix_(free, fixed)] @ xf, rtol=1e-15, atol=1e-15)
    Ar = red.A.to_dense().numpy() if sparse else red.A.numpy()
    assert np.array_equal(Ar, qp.A[:, free])
    s = qp.A[:, fixed] @ xf
    assert np.allclose(shift.numpy(), s, atol=1e-15) and np.allclose(red.lcon.numpy(), qp.lcon - s, atol=1e-15)
    assert abs(red.c0 - (qp.c0 + qp.q[fixed] @ xf + 0.5 * xf @ qp.H[np.ix_(fixed, fixed)] @ xf)) < 1e-14
    assert red.nvar == 27 and red.ncon == 12 and np.array_equal(red.lvar.numpy(), qp.lvar[free])
    # nothing fixed: nothing to do
    qp2 = Q.random_qp(17, 30, 12, False)
    dq2 = M.DeviceQP.from_numpy(torch.device("cpu"), qp2.H, qp2.q, qp2.A, qp2.lvar, qp2.uvar, qp2.lcon, qp2.ucon, qp2.x0)
    if not np.any(qp2.lvar == qp2.uvar):
        assert dq2.eliminate_fixed() is None


@pytest.mark.parametrize("make", [Q.simple_lp, Q.hs21, lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)),
                                  lambda: Q.random_qp(23, 40, 22, False), lambda: Q.random_qp(24, 35, 20, True)])
def test_augmented_kkt_driver_matches_oracle(make):
    """kkt_system="augmented" (HIPAugmentedKKTSystem, the K2 form) through the host control flow with the
    reference's DEFAULT regularization (delta_d = 0) and equality rows: same trace as the oracle's K2 path."""
    qp = make()
    ref = mpc.solve(qp, kkt_system="K2")
    _, r = run(qp, kkt_system="augmented", regularization=M.FixedRegularization(1e-8, 0.0))
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
    for a, b in zip(r["trace"], ref["trace"]):
        tol = 1e-9 if min(a["mu"], b["mu"]) >= 1e-4 else 1e-6
        for k in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "mu"):
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(b[k])), (a["k"], k)
    assert np.max(np.abs(r["solution"] - ref["solution"])) < 1e-7
    assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) < 1e-6


def test_augmented_kkt_options():
    """Fixed variables default to RelaxBound for the augmented system (a condensed KKT type in the sense of
    src/utils.jl:81); the inertia reported is the K2 one."""
    qp = Q.dummy_qp(20, 15, equality_cons=(0, 1), fixed_variables=(0, 1))
    s, r = run(qp, kkt_system="augmented", regularization=M.FixedRegularization(1e-8, 0.0))
    ref = mpc.solve(qp, kkt_system="K2", fixed_variable_treatment="relax_bound")
    assert r["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"]
    assert s.kkt.is_inertia_correct(*s.kkt.linear_solver.inertia())
    assert s.kkt.linear_solver.inertia() == (20, 0, 15) and "L diag" in s.kkt.linear_solver.introduce()
    with pytest.raises(ValueError):
        run(qp, kkt_system="no-such-system")


def test_refine_steps_reduce_the_residual():
    """refine_steps (extension): d += K^-1 (p - K d) in solve_system, through the fake backend."""
    qp = Q.synthetic_qp(20250615, 30, 12, "lp")
    s0, r0 = run(qp, refine_steps=0)
    s1, r1 = run(qp, refine_steps=2)
    assert r0["status"] == r1["status"] == M.SOLVE_SUCCEEDED and abs(r0["iter"] - r1["iter"]) <= 1
    assert abs(r0["objective"] - r1["objective"]) <= 1e-8
    assert s1.last_residual_ratio <= 10 * s0.last_residual_ratio + 1e-14


def test_refinement_auto_rule_is_resolved_by_the_order_of_the_factorised_matrix(monkeypatch):
    """options.py (round 5): refine_steps = None is the AUTO rule -- one step of iterative refinement while the matrix
    that is factorised (n_x condensed, m normal equations, n_x + m augmented) has order <= 1024 (MADQP_REFINE_AUTO_MAX),
    none above; an explicit value is taken as it stands."""
    def make(qp, **opts):
        dq = M.DeviceQP.from_numpy("cpu", qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)
        return M.MPCSolver(dq, FakeBackend(), **opts)

    qp = Q.synthetic_qp(3, 30, 12)
    lp = Q.synthetic_qp(3, 30, 12, "lp")
    reg = M.FixedRegularization(1e-8, -1e-8)
    s = make(qp, regularization=reg)
    assert s.opt.refine_steps == 1
    assert make(qp, regularization=reg, refine_steps=0).opt.refine_steps == 0
    assert make(qp, regularization=reg, refine_steps=2).opt.refine_steps == 2
    monkeypatch.setenv("MADQP_REFINE_AUTO_MAX", "29")  # condensed: order 30 > 29; normal equations: order 12 <= 29
    assert make(qp, regularization=reg).opt.refine_steps == 0
    assert make(lp, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0)).opt.refine_steps == 1
    monkeypatch.setenv("MADQP_REFINE_AUTO_MAX", "41")  # augmented: order 30 + 12 = 42 > 41
    assert make(qp, kkt_system="augmented").opt.refine_steps == 0
    assert make(qp, regularization=reg).opt.refine_steps == 1
    # and the rule's result is what the oracle's refined run gives: same iteration count, same point
    monkeypatch.delenv("MADQP_REFINE_AUTO_MAX")
    r = make(qp, regularization=reg).solve()
    ref = mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8), refine_steps=1)
    assert r["iter"] == ref["iter"] and np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-9


def test_host_quotients_follow_ieee_like_the_reference():
    """The reference's loop divides host scalars without a guard -- the start point's 0 / 0 when a problem has no bound at
    all (src/solver.jl:93-94: the NaN is then added to EMPTY views), a step length against a zero direction component
    (src/kernels.jl:341-368) -- and Julia's Float64 division returns NaN / +-Inf where Python's raises: the host driver
    divides through this helper."""
    import math

    from madqp_jl_amd.solver import _fdiv

    assert math.isnan(_fdiv(0.0, 0.0)) and math.isnan(_fdiv(float("nan"), 0.0)) and math.isnan(_fdiv(0.0, -0.0))
    assert _fdiv(1.0, 0.0) == float("inf") and _fdiv(-2.5, 0.0) == -float("inf")
    assert _fdiv(1.0, -0.0) == -float("inf") and _fdiv(-1.0, -0.0) == float("inf")
    assert _fdiv(3.0, 2.0) == 1.5 and _fdiv(-1e-300, 1e300) == -1e-300 / 1e300
