"""GPU: the sparse-A front end (reference: coo_to_csr / assemble_normal_system!,
src/utils.jl:148-298, NormalKKTSystem constructor src/KKT/normalkkt.jl:51-101).  The Jacobian is CSR on
the device, the factorised matrix stays dense: assembled matrix, products and whole solves must agree
with the dense path of the same library and with the CPU oracle on the densified problem."""
import os

import numpy as np
import pytest
import torch

import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q

pytestmark = pytest.mark.gpu
REG = M.FixedRegularization(1e-8, -1e-8)


def to_device(qp, be, sparse):
    return M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0,
                                 sparse=sparse)


def close(a, b, tol):
    return abs(a - b) <= tol * max(1.0, abs(a), abs(b))


def lower(be, kkt, n):
    ptr, ld = be.kkt_matrix(kkt._h, n)
    return np.tril(be.read_doubles(ptr, ld * n).reshape(n, ld)[:, :n].T)


def test_csr_container(hip):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((7, 11)) * (rng.random((7, 11)) < 0.3)
    A[3] = 0.0  # an empty row
    c = M.DeviceCSR.from_dense(hip.device, A)
    assert c.nnz == np.count_nonzero(A) and np.array_equal(c.to_dense().cpu().numpy(), A)
    assert np.array_equal(c.row_absmax().cpu().numpy(), np.abs(A).max(axis=1))
    # CSR of A' through the stored permutation
    At = np.zeros((11, 7))
    tp, tc, tv = (t.cpu().numpy() for t in (c.t_ptr, c.t_col, c.t_val))
    for r in range(11):
        At[r, tc[tp[r]:tp[r + 1]]] = tv[tp[r]:tp[r + 1]]
    assert np.array_equal(At, A.T)
    s = c.scaled(torch.arange(1, 8, dtype=torch.float64, device=hip.device))
    assert np.array_equal(s.to_dense().cpu().numpy(), A * np.arange(1, 8)[:, None])
    with pytest.raises(ValueError):
        M.DeviceCSR(hip.device, 2, 2, [0, 0], [1, 1], [1.0, 2.0])


@pytest.mark.parametrize("ksys,family", [("condensed", "wigner"), ("condensed", "lp"), ("normal", "lp")])
def test_sparse_matches_dense_path(hip, ksys, family):
    """Same library, Jacobian held sparse vs dense: assembled matrix, jtprod!, mul!, solve!."""
    qp = Q.sparse_qp(21, 300, 130, 5, family, equality_cons=(3, 40) if ksys == "normal" else ())
    reg = M.FixedRegularization(1e-8, 0.0) if ksys == "normal" else REG
    solvers = []
    for sparse in (False, True):
        s = M.MPCSolver(to_device(qp, hip, sparse), hip, kkt_system=ksys, regularization=reg)
        s.initialize()
        hip.set_aug_diagonal_reg(s.st, 1e-8, reg.delta_d)
        s.kkt.factorize_wrapper()
        assert s.kkt.linear_solver.is_factorized()
        solvers.append(s)
    d, sp = solvers
    assert type(sp.kkt).__name__.startswith("HIPSparse")
    order = 130 if ksys == "normal" else 300
    Kd, Ks = lower(hip, d.kkt, order), lower(hip, sp.kkt, order)  # factors of the same matrix
    assert np.max(np.abs(Kd - Ks)) <= 1e-11 * np.max(np.abs(Kd))
    rng = np.random.default_rng(5)
    b = rng.standard_normal(d.st.ntot)
    outs = []
    for s in solvers:
        st = s.st
        st.p.copy_(torch.as_tensor(b))
        hip.copy(st.p, st.d)
        s.kkt.solve(st.d)
        hip.fill(0.0, st.w1)
        s.kkt.mul(st.w1, st.d, 1.0, 0.0)
        s.kkt.jtprod(st.jacl, st.y)
        outs.append((st.d.cpu().numpy().copy(), st.w1.cpu().numpy().copy(), st.jacl.cpu().numpy().copy()))
    for a, c in zip(outs[0], outs[1]):
        assert np.max(np.abs(a - c)) <= 1e-9 * max(1.0, np.max(np.abs(a)))
    assert np.max(np.abs(outs[1][1] - b)) / np.max(np.abs(b)) < 1e-8  # K * solve(b) == b
    for s in solvers:
        s.close()


@pytest.mark.parametrize("case", ["lp_normal", "lp_condensed", "qp_condensed", "lp_normal_native"])
def test_sparse_solves_vs_oracle(hip, case):
    if case.startswith("lp_normal"):
        qp = Q.sparse_qp(31, 160, 70, 4, "lp", equality_cons=(2, 9, 33))
        kw = dict(kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0))
        okw = dict(kkt_system="normal", regularization=mpc.FixedRegularization(1e-8, 0.0))
    elif case == "lp_condensed":
        qp = Q.sparse_qp(3, 100, 40, 6, "lp")  # (equality rows: Theta = 1e8 puts the condensed LP at 1e-8 accuracy)
        kw, okw = dict(regularization=REG), dict(kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    else:
        qp = Q.sparse_qp(3, 200, 60, 10)
        kw, okw = dict(regularization=REG), dict(kkt_system="condensed", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    if case.endswith("native"):
        kw["driver"] = "native"
    s = M.MPCSolver(to_device(qp, hip, True), hip, **kw)
    r = s.solve()
    s.close()
    ref = mpc.solve(qp, **okw)
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"], (r["iter"], ref["iter"])
    for t, g in zip(r["trace"], ref["trace"]):
        tol = 1e-9 if min(t["mu"], g["mu"]) >= 1e-4 else 1e-6
        for key in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "mu"):
            assert close(t[key], g[key], tol), (case, t["k"], key, t[key], g[key])
    assert close(r["objective"], ref["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7


def test_instance_file_to_gpu_pipeline(hip):
    """The reference's benchmark pipeline (scripts/benchmarks_cpu.jl:17-55) on the device: instance text ->
    read_qps -> scale_qp -> standard_form_qp -> sparse front end -> MPCSolver -> the nine recorded numbers.
    HS21: objective -99.96 at (2, 0); an LP in standard form through the reference's own normal equations."""
    from madqp_jl_amd import preprocess as P
    from tests.test_preprocess import HS21_QPS

    qp = P.read_qps(HS21_QPS)
    scaled, Dr, Dc = P.ruiz_scale(qp)
    s = M.MPCSolver(P.to_device(scaled, hip), hip, regularization=REG)
    r = s.solve()
    s.close()
    assert r["status"] == M.SOLVE_SUCCEEDED and abs(r["objective"] + 99.96) < 1e-6
    assert np.allclose(r["solution"] / Dc, [2.0, 0.0], atol=1e-6)
    row = P.benchmark_row(scaled, r, 1.0, 0.5)
    assert row[:6] == (2, 1, 2, 2, 1, r["iter"])
    # LP: random feasible LP -> standard form (all rows equalities, x + w = xu rows) -> normal equations
    lp = Q.random_qp(6, 40, 18, lp=True)
    import scipy.sparse as sp

    h = P.HostQP(lp.c0, lp.q, sp.csr_matrix((40, 40)), sp.csr_matrix(lp.A), lp.lvar, lp.uvar, lp.lcon, lp.ucon)
    sf = P.standard_form(h)
    assert np.array_equal(sf.lcon, sf.ucon)  # every row of the standard form is an equality
    s = M.MPCSolver(P.to_device(sf, hip), hip, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0))
    r = s.solve()
    s.close()
    ref = mpc.solve(lp, kkt_system="K2", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
    assert abs(r["objective"] - ref["objective"]) <= 1e-6 * max(1.0, abs(ref["objective"]))
    assert np.max(np.abs(r["solution"][:40] - ref["solution"])) < 1e-5


@pytest.mark.parametrize("seed,lp", [(0, False), (2, True)])
def test_presolve_scale_solve_postsolve_on_device(hip, seed, lp):
    """presolve_qp -> scale_qp -> MPCSolver (scripts/benchmarks_cpu.jl:27-45) with the device solver in the middle,
    then back: unscale, postsolve, and check the optimality conditions of the ORIGINAL model (multipliers of the
    rows the presolve removed included).  Augmented system: the model has equality rows, default regularization."""
    from madqp_jl_amd import preprocess as P
    from tests.test_preprocess import kkt_violation, planted_qp

    qp = planted_qp(seed, lp)
    ps = P.presolve(qp)
    assert ps.flag
    scaled, Dr, Dc = P.ruiz_scale(ps.qp)
    s = M.MPCSolver(P.to_device(scaled, hip), hip, kkt_system="augmented", tol=1e-9, driver="native")
    r = s.solve()
    s.close()
    assert r["status"] == M.SOLVE_SUCCEEDED
    # scale_qp: xs = Dc x, rows divided by Dr  =>  x = xs / Dc, y = ys / Dr, z = zs * Dc
    full = ps.postsolve(r["solution"] / Dc, r["multipliers"] / Dr, r["multipliers_L"] * Dc, r["multipliers_U"] * Dc)
    assert abs(full["objective"] - r["objective"]) <= 1e-8 * max(1.0, abs(r["objective"]))
    assert kkt_violation(qp, full) <= 2e-6
    assert len(full["x"]) == qp.nvar > ps.qp.nvar and len(full["y"]) == qp.ncon > ps.qp.ncon


@pytest.mark.parametrize("N,ksys", [(12, "normal"), (25, "normal"), (12, "condensed")])
def test_boundary_control_qp_diagonal_hessian(hip, N, ksys):
    """CONT-type QP (BASELINE configs[2] stand-in): sparse equality rows, diagonal Hessian kept as a vector.
    Normal equations A (H + Sigma)^-1 A' (the diagonal-H extension of the LP-only NormalKKTSystem) against the
    oracle's K2 system on the densified problem -- the equality the reference asserts for LPs
    (test/runtests.jl:165-180); condensed form with Theta = -1/delta_d on the equality rows."""
    from madqp_jl_amd import preprocess as P

    h = P.boundary_control_qp(N)
    dq = P.to_device(h, hip)
    assert dq.H.dim() == 1 and isinstance(dq.A, M.DeviceCSR)
    dense = Q.DenseQP(H=h.H.toarray(), q=h.c, A=h.A.toarray(), lvar=h.lvar, uvar=h.uvar, lcon=h.lcon, ucon=h.ucon,
                      x0=h.x0, c0=h.c0)
    if ksys == "normal":
        reg, oreg = M.FixedRegularization(1e-8, 0.0), mpc.FixedRegularization(1e-8, 0.0)
        ref = mpc.solve(dense, kkt_system="K2", regularization=oreg)
    else:
        reg, oreg = REG, mpc.FixedRegularization(1e-8, -1e-8)
        ref = mpc.solve(dense, kkt_system="condensed", regularization=oreg)
    for driver in ("python", "native"):
        s = M.MPCSolver(dq, hip, kkt_system=ksys, regularization=reg, driver=driver)
        r = s.solve()
        s.close()
        assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"], (r["iter"], ref["iter"])
        tol = 1e-9 if ksys == "normal" else 1e-6  # condensed: Theta = 1e8 on every row
        assert abs(r["objective"] - ref["objective"]) <= tol * max(1.0, abs(ref["objective"]))
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e3 * tol
        for t, g in zip(r["trace"], ref["trace"]):
            for key in ("alpha_p", "alpha_d", "mu"):
                assert close(t[key], g[key], 1e-6 if ksys == "normal" else 1e-4), (t["k"], key, t[key], g[key])


def test_benchmark_loop_over_instance_files(hip, tmp_path):
    """tools/run_benchmarks.py = the loop of scripts/benchmarks_cpu.jl:10-62 on instance files written here
    (HS21 from its published statement, two planted models, one LP that the presolve settles alone, one file that
    does not parse): nine numbers per solved instance, objective of the ORIGINAL model despite presolve + scaling."""
    import sys as _sys

    from madqp_jl_amd import preprocess as P
    from tests.test_preprocess import HS21_QPS, dense, planted_qp

    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_benchmarks

    d = tmp_path / "inst"
    d.mkdir()
    (d / "hs21.qps").write_text(HS21_QPS)
    models = {"hs21.qps": P.read_qps(HS21_QPS)}
    for seed, lp in ((0, False), (3, True)):
        qp = planted_qp(seed, lp)
        P.write_qps(qp, str(d / f"planted{seed}.mps.gz"))
        models[f"planted{seed}.mps.gz"] = qp
    import scipy.sparse as sp

    trivial = P.HostQP(1.0, np.array([2.0, -1.0]), sp.csr_matrix((2, 2)), sp.csr_matrix([[0.0, 1.0]]),
                       np.array([3.0, 0.0]), np.array([3.0, np.inf]), np.array([-np.inf]), np.array([4.0]))
    P.write_qps(trivial, str(d / "trivial.mps"))
    (d / "broken.mps").write_text("NAME x\nNOSUCHSECTION\n")
    out = str(tmp_path / "results.txt")
    names, res = run_benchmarks.run(str(d), out=out, backend=hip, verbose=False)
    assert names == sorted(["hs21.qps", "planted0.mps.gz", "planted3.mps.gz", "trivial.mps", "broken.mps"])
    rows = dict(zip(names, res))
    assert not rows["broken.mps"].any() and not rows["trivial.mps"].any()  # skipped: no row recorded
    for name, qp in models.items():
        row = rows[name]
        # the oracle does not take fixed variables: its reference is the presolved model (same objective value,
        # tests/test_preprocess.py checks that bookkeeping against the original model's optimality conditions)
        ref = mpc.solve(dense(P.presolve(qp).qp), kkt_system="K2", tol=1e-9)
        assert row[4] == M.SOLVE_SUCCEEDED and row[5] > 0 and row[7] > 0 and row[8] > 0
        assert abs(row[6] - ref["objective"]) <= 1e-6 * max(1.0, abs(ref["objective"])), name
        assert row[0] <= qp.nvar and row[1] <= qp.ncon  # presolved sizes
    assert abs(rows["hs21.qps"][6] + 99.96) < 1e-6
    lines = open(out).read().splitlines()
    assert len(lines) == 5 and all(len(l.split("\t")) == 10 for l in lines)


@pytest.mark.parametrize("case", ["lower_bounds_only", "upper_bounds_only", "one_by_one", "mixed", "no_bounds"])
def test_edge_shapes_with_a_sparse_jacobian(hip, case):
    """tests/edge_cases.py with the Jacobian held as CSR (the sparse front end of the condensed system: SpMV for A, A',
    the Gram matrix from the sparse rows) -- the fused per-variable passes serve this form too -- against the oracle with
    the refinement the default library applies at these sizes."""
    import parity
    from edge_cases import edge_qp

    qp = edge_qp(case)
    ref = parity.oracle_execution(qp, "refine", regularization=mpc.FixedRegularization(1e-8, -1e-8))
    s = M.MPCSolver(to_device(qp, hip, True), hip, regularization=REG, driver="native")
    assert type(s.qp.A).__name__ == "DeviceCSR"
    r = s.solve()
    s.close()
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED and r["iter"] == ref["iter"], (case, r["iter"], ref["iter"])
    for t, o in zip(r["trace"], ref["trace"]):
        for key in ("alpha_p", "alpha_d", "inf_pr", "inf_du", "inf_compl", "mu"):
            assert close(t[key], o[key], 1e-8), (case, t["k"], key, t[key], o[key])
    assert close(r["objective"], ref["objective"], 1e-9)
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
