"""GPU: the Julia boundary (julia/MadQPHIP.jl) replayed through the C ABI.

Julia cannot run in the build or GPU image.  tests/julia_replay.py issues, method by method, the ccall sequence of the
glue while the reference's loop order (src/solver.jl:6-125,127-182,254-345) drives it; here that replay is checked
against the oracle, and the set of ABI symbols the replay touched is checked against the set the glue binds."""
import os
import re

import numpy as np
import pytest
import scipy.linalg as sla
import torch

import julia_replay as JR
import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q
from test_gpu_solver import compare_traces

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rbe():
    be = JR.ReplayBackend(0)
    yield be
    be.close()


def to_device(qp, be):
    return M.DeviceQP.from_numpy(be.device, qp.H, qp.q, qp.A, qp.lvar, qp.uvar, qp.lcon, qp.ucon, qp.x0, qp.c0)


@pytest.mark.parametrize("symmetric", [0, 1])
def test_coo_map_matches_numpy(rbe, symmetric):
    """compress_jacobian! / compress_hessian!: COO values in pattern order -> dense, duplicates added in COO order."""
    rng = np.random.default_rng(5 + symmetric)
    rows, cols, ld = (37, 37, 41) if symmetric else (23, 58, 64)
    dense = rng.standard_normal((rows, cols)) * (rng.random((rows, cols)) < 0.3)
    if symmetric:
        dense = np.tril(dense)
    I, J, v = JR.coo_pattern(dense, rng, duplicates=9)
    h = JR.C.c_void_p()
    JR.ccall(rbe, "madqp_coo_map_create", rbe.ctx, len(I), I.ctypes.data_as(JR.C.c_void_p),
             J.ctypes.data_as(JR.C.c_void_p), rows, cols, symmetric, JR.C.byref(h))
    dst = torch.full((rows, ld), 7.0, dtype=torch.float64, device=rbe.device)  # stale content must be cleared
    JR.ccall(rbe, "madqp_coo_map_apply", h, JR.ptr(torch.as_tensor(v, device=rbe.device)), JR.ptr(dst), ld)
    out = dst.cpu().numpy()
    expect = np.zeros((rows, cols))
    np.add.at(expect, (I - 1, J - 1), v)  # duplicates: the same pairwise sums, so compare to a few ulp
    if symmetric:
        expect = expect + np.tril(expect, -1).T
    assert np.max(np.abs(out[:, :cols] - expect)) <= 4e-16 * max(1.0, np.max(np.abs(expect)))
    assert np.all(out[:, cols:] == 7.0)  # the padding beyond ncols is not the map's to touch
    rbe.lib.madqp_coo_map_destroy(h)
    with pytest.raises(M.MadQPError):  # an entry outside the matrix is a usage error, not a crash
        bad = np.array([rows + 1], dtype=np.int32)
        JR.ccall(rbe, "madqp_coo_map_create", rbe.ctx, 1, bad.ctypes.data_as(JR.C.c_void_p),
                 bad.ctypes.data_as(JR.C.c_void_p), rows, cols, 0, JR.C.byref(h))


def test_linear_solver_contract(rbe):
    """HIPCholeskySolver(aug_com; opt) keeps the matrix object; factorize! re-reads it; solve!(s, rhs) is in place
    (src/KKT/normalkkt.jl:99-101,196).  Also: build_kkt! works from the KKT's own fields before any
    set_aug_diagonal_reg! (src/solver.jl:16-21)."""
    rng = np.random.default_rng(11)
    qp = Q.synthetic_qp(3, 150, 40)
    ind_ineq = np.arange(40)
    jI, jJ, jv = JR.coo_pattern(qp.A, rng)
    hI, hJ, hv = JR.coo_pattern(np.tril(qp.H), rng)
    n = 150 + 40
    kkt = JR.ReplayKKTSystem(rbe, "condensed", 150, 40, ind_ineq, np.arange(n), np.arange(n), jI, jJ, hI, hJ)
    kkt.get_jacobian().copy_(torch.as_tensor(jv, device=rbe.device))
    kkt.compress_jacobian()
    kkt.get_hessian().copy_(torch.as_tensor(hv, device=rbe.device))
    kkt.compress_hessian()
    kkt.initialize()
    sig = rng.uniform(0.5, 2.0, n)
    kkt.pr_diag.copy_(torch.as_tensor(sig, device=rbe.device))
    kkt.du_diag.fill_(-1e-8)
    kkt.factorize_wrapper()  # the first plugin call of init_starting_point!
    assert kkt.linear_solver.is_factorized() and kkt.linear_solver.order == 150
    theta = sig[150:] / (1.0 + 1e-8 * sig[150:])
    K = qp.H + np.diag(sig[:150]) + (qp.A.T * theta) @ qp.A
    b = rng.standard_normal(150)
    x = torch.as_tensor(b, device=rbe.device).clone()
    assert kkt.linear_solver.solve(x) is x
    ref = sla.cho_solve(sla.cho_factor(K, lower=True), b)
    assert np.max(np.abs(x.cpu().numpy() - ref)) <= 1e-11 * np.max(np.abs(ref))
    kkt.pr_diag.mul_(3.0)  # the solver keeps a reference to the matrix object: a second factorize! sees the new values
    kkt.factorize_wrapper()
    theta = 3 * sig[150:] / (1.0 + 1e-8 * 3 * sig[150:])
    K = qp.H + np.diag(3 * sig[:150]) + (qp.A.T * theta) @ qp.A
    x = torch.as_tensor(b, device=rbe.device).clone()
    kkt.linear_solver.solve(x)
    assert np.max(np.abs(x.cpu().numpy() - np.linalg.solve(K, b))) <= 1e-11 * np.max(np.abs(b))
    kkt.close()


def test_distributed_linear_solver_contract(rbe):
    """HIPDistributedCholeskySolver: factorize! (madqp_dkkt_factorize, re-reads the matrix object) and solve!(s, rhs) in
    place on the replicated right-hand side (madqp_dist_solve), as src/KKT/normalkkt.jl:99-101,196 use a linear solver."""
    rng = np.random.default_rng(12)
    qp = Q.synthetic_qp(4, 300, 60)
    n = 300 + 60
    jI, jJ, jv = JR.coo_pattern(qp.A, rng)
    hI, hJ, hv = JR.coo_pattern(np.tril(qp.H), rng)
    kkt = JR.ReplayDistributedKKTSystem(rbe, 300, 60, np.arange(60), np.arange(n), np.arange(n), jI, jJ, hI, hJ, nb=128)
    kkt.get_jacobian().copy_(torch.as_tensor(jv, device=rbe.device))
    kkt.compress_jacobian()
    kkt.get_hessian().copy_(torch.as_tensor(hv, device=rbe.device))
    kkt.compress_hessian()
    kkt.initialize()
    sig = rng.uniform(0.5, 2.0, n)
    kkt.pr_diag.copy_(torch.as_tensor(sig, device=rbe.device))
    kkt.du_diag.fill_(-1e-8)
    kkt.factorize_wrapper()
    assert kkt.linear_solver.is_factorized()
    theta = sig[300:] / (1.0 + 1e-8 * sig[300:])
    K = qp.H + np.diag(sig[:300]) + (qp.A.T * theta) @ qp.A
    b = rng.standard_normal(300)
    x = torch.as_tensor(b, device=rbe.device).clone()
    assert kkt.linear_solver.solve(x) is x
    ref = sla.cho_solve(sla.cho_factor(K, lower=True), b)
    assert np.max(np.abs(x.cpu().numpy() - ref)) <= 1e-11 * np.max(np.abs(ref))
    kkt.close()


CASES = [
    ("simple_lp/normal", lambda: Q.simple_lp(), "normal", "normal", (1e-8, 0.0), 0),
    ("simple_lp/augmented", lambda: Q.simple_lp(), "augmented", "K2", (1e-8, 0.0), 0),
    ("hs21/condensed", lambda: Q.hs21(), "condensed", "condensed", (1e-8, -1e-8), 0),
    ("hs21/augmented", lambda: Q.hs21(), "augmented", "K2", (1e-8, 0.0), 0),
    ("dummy_10_5/condensed", lambda: Q.dummy_qp(10, 5), "condensed", "condensed", (1e-8, -1e-8), 0),
    ("dummy_20_15_eq/augmented/gondzio", lambda: Q.dummy_qp(20, 15, equality_cons=(0, 1, 2, 7)), "augmented", "K2",
     (1e-8, 0.0), 5),
    ("synthetic_40_16/condensed/gondzio", lambda: Q.synthetic_qp(20250614, 40, 16), "condensed", "condensed",
     (1e-8, -1e-8), 3),
    # free / one-sided / boxed variables, equality and ranged rows (ind_lb != ind_ub): through the form that takes
    # equality rows exactly -- the condensed form's Theta = 1e8 puts a 2e-7 noise floor under the traces
    # (test_gpu_solver.py::test_kkt_system_conformance), too coarse for the 1e-9 trace comparison used here
    ("random_130_70/augmented", lambda: Q.random_qp(5, 130, 70), "augmented", "K2", (1e-8, 0.0), 0),
    ("dummy_10_5/scaled_augmented", lambda: Q.dummy_qp(10, 5), "scaled_augmented", "K2.5", (1e-8, 0.0), 0),
    ("random_130_70/scaled_augmented/gondzio", lambda: Q.random_qp(5, 130, 70), "scaled_augmented", "K2.5", (1e-8, 0.0), 2),
    ("synthetic_lp_30_12/normal", lambda: Q.synthetic_qp(20250615, 30, 12, "lp"), "normal", "normal", (1e-8, 0.0), 0),
]


@pytest.mark.parametrize("name,make,form,oform,reg,ncorr", CASES, ids=[c[0] for c in CASES])
def test_madipm_loop_through_the_glue(rbe, name, make, form, oform, reg, ncorr):
    """MadIPM.solve!(MPCSolver(qp; kkt_system = MadQPHIP.HIP*KKTSystem, linear_solver = MadQPHIP.HIPCholeskySolver))."""
    qp = make()
    s = JR.ReplayMPCSolver(to_device(qp, rbe), rbe, kkt_system=form, regularization=M.FixedRegularization(*reg),
                           max_ncorr=ncorr)
    r = s.solve()
    s.close()
    ref = mpc.solve(qp, kkt_system=oform, regularization=mpc.FixedRegularization(*reg), max_ncorr=ncorr)
    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
    compare_traces(r["trace"], ref["trace"], name)
    assert abs(r["objective"] - ref["objective"]) <= 1e-9 * max(1.0, abs(ref["objective"]))
    assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
    assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) <= 1e-6


@pytest.mark.parametrize("P,Q,nb,n", [(1, 1, 128, 300), (2, 3, 128, 1000), (2, 4, 256, 1900), (3, 2, 128, 130)])
def test_cyclic_coo_maps_match_numpy(rbe, P, Q, nb, n):
    """compress_jacobian! / compress_hessian! of HIPDistributedKKTSystem: for EVERY rank (p, q) of the grid, the
    block-cyclic maps move the replicated COO callback values into exactly the pieces DistributedQP.from_dense cuts out
    of the dense matrices (A_I, A_J, the lower tiles of H with complete diagonal tiles), bit for bit."""
    import types

    from madqp_jl_amd.dist2d import DistributedQP

    rng = np.random.default_rng(P * 100 + Q)
    m = 70
    A = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.2)
    G = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.05)
    H = G + G.T + np.diag(rng.uniform(1, 2, n))
    jI, jJ, jv = JR.coo_pattern(A, rng, duplicates=7)
    hI, hJ, hv = JR.coo_pattern(np.tril(H), rng, duplicates=7)
    T = (n + nb - 1) // nb
    pad = lambda v: max(128, (v + 127) // 128 * 128)
    dev = rbe.device
    jv_d, hv_d = torch.as_tensor(jv, device=dev), torch.as_tensor(hv, device=dev)
    cv = lambda a: a.ctypes.data_as(JR.C.c_void_p)
    for p in range(P):
        for q in range(Q):
            mloc = sum(min(nb, n - I * nb) for I in range(p, T, P))
            nloc = sum(min(nb, n - J * nb) for J in range(q, T, Q))
            grid = types.SimpleNamespace(n=n, nb=nb, P=P, Q=Q, p=p, q=q, mloc=mloc, nloc=nloc, ld=pad(mloc), ncp=pad(nloc))
            ref = DistributedQP.from_dense(types.SimpleNamespace(device=dev), grid, H, np.zeros(n), A, np.zeros(n),
                                           np.ones(n), np.zeros(m), np.ones(m), np.zeros(n))
            for R, r, want, ldw in ((P, p, ref.A_I, grid.ld), (Q, q, ref.A_J, grid.ncp)):
                h = JR.C.c_void_p()
                JR.ccall(rbe, "madqp_coo_map_create_cols_cyclic", rbe.ctx, len(jI), cv(jI), cv(jJ), m, n, nb, R, r, JR.C.byref(h))
                got = torch.full_like(want, 0.0)
                JR.ccall(rbe, "madqp_coo_map_apply", h, JR.ptr(jv_d), JR.ptr(got), ldw)
                assert torch.allclose(got[:m], want[:m], rtol=0, atol=1e-15), (p, q, R)
                rbe.lib.madqp_coo_map_destroy(h)
            h = JR.C.c_void_p()
            JR.ccall(rbe, "madqp_coo_map_create_tiles_cyclic", rbe.ctx, len(hI), cv(hI), cv(hJ), n, nb, P, p, Q, q, JR.C.byref(h))
            got = torch.zeros_like(ref.H)
            JR.ccall(rbe, "madqp_coo_map_apply", h, JR.ptr(hv_d), JR.ptr(got), grid.ld)
            gi = (torch.arange(mloc) // nb * P + p)  # tile row of every local row
            gj = (torch.arange(nloc) // nb * Q + q)
            lower = (gi[None, :] >= gj[:, None]).to(dev)  # [local column, local row]: tiles on or below the diagonal
            want = ref.H[:nloc, :mloc] * lower
            assert torch.allclose(got[:nloc, :mloc], want, rtol=0, atol=1e-15), (p, q)
            rbe.lib.madqp_coo_map_destroy(h)


@pytest.mark.parametrize("name,make,ncorr,nb", [
    ("synthetic_300_120", lambda: Q.synthetic_qp(20250614, 300, 120), 0, 128),
    ("synthetic_300_120/gondzio", lambda: Q.synthetic_qp(77, 300, 120), 3, 256),
    ("random_130_70_lp", lambda: Q.synthetic_qp(20250615, 130, 50, "lp"), 0, 128),
])
def test_madipm_loop_through_the_distributed_glue(rbe, name, make, ncorr, nb):
    """MadIPM.solve!(MPCSolver(qp; kkt_system = MadQPHIP.HIPDistributedKKTSystem, linear_solver =
    MadQPHIP.HIPDistributedCholeskySolver)) on one rank of a 1 x 1 grid: create_kkt_system (grid, block-cyclic COO maps,
    madqp_dkkt_create), compress_*!, build_kkt!, factorize!, solve!, mul!, jtprod! -- against the oracle."""
    qp = make()
    reg = (1e-8, -1e-8)
    s = JR.ReplayMPCSolver(to_device(qp, rbe), rbe, kkt_system="condensed", regularization=M.FixedRegularization(*reg),
                           max_ncorr=ncorr, distributed_tile=nb)
    r = s.solve()
    assert type(s.kkt).__name__ == "ReplayDistributedKKTSystem"
    s.close()
    ref = mpc.solve(qp, kkt_system="condensed", regularization=mpc.FixedRegularization(*reg), max_ncorr=ncorr)
    from parity import assert_parity

    assert r["status"] == ref["status"] == M.SOLVE_SUCCEEDED
    # the stated bar, or 4 x the ensemble noise floor where the conditioning does not support it (tests/parity.py)
    assert_parity(r, ref, qp, name, regularization=mpc.FixedRegularization(*reg), max_ncorr=ncorr)


def test_every_symbol_the_glue_binds_was_replayed(rbe):
    """The ccall targets of julia/MadQPHIP.jl == the ABI symbols the replay touched (after one solve per form with
    Gondzio corrections, so that set_extra_correction! runs)."""
    for form, make, reg in (("condensed", lambda: Q.dummy_qp(10, 5), (1e-8, -1e-8)),
                            ("augmented", lambda: Q.dummy_qp(10, 5), (1e-8, 0.0)),
                            ("scaled_augmented", lambda: Q.dummy_qp(10, 5), (1e-8, 0.0)),
                            ("normal", lambda: Q.simple_lp(), (1e-8, 0.0))):
        s = JR.ReplayMPCSolver(to_device(make(), rbe), rbe, kkt_system=form,
                               regularization=M.FixedRegularization(*reg), max_ncorr=2)
        assert s.solve()["status"] == M.SOLVE_SUCCEEDED
        s.close()
    s = JR.ReplayMPCSolver(to_device(Q.dummy_qp(10, 5), rbe), rbe, kkt_system="condensed",
                           regularization=M.FixedRegularization(1e-8, -1e-8), max_ncorr=2, distributed_tile=128)
    assert s.solve()["status"] == M.SOLVE_SUCCEEDED  # HIPDistributedKKTSystem on one rank
    s.close()
    src = open(os.path.join(ROOT, "julia", "MadQPHIP.jl")).read()
    code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())  # comments name optional bindings
    bound = set(re.findall(r"(?::|@k )(madqp_[a-z0-9_]+)", code))
    lifecycle = {"madqp_ctx_create", "madqp_ctx_destroy", "madqp_last_error", "madqp_kkt_destroy",
                 "madqp_coo_map_destroy", "madqp_dkkt_destroy", "madqp_dist_destroy",  # HipBackend / close() here
                 "madqp_dist_unique_id"}  # several ranks only (`cfg.world > 1`): tests/test_gpu_dist2d.py drives it
    assert bound - lifecycle == JR.GLUE_ENTRY_POINTS - lifecycle, (
        sorted(bound - lifecycle - JR.GLUE_ENTRY_POINTS), sorted(JR.GLUE_ENTRY_POINTS - lifecycle - bound))
    assert bound <= set(M.EXPORTED_SYMBOLS)
