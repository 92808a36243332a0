"""GPU: the BASELINE.json configs at their FULL shapes (round 1 ran three of them only from tools/).

  configs[2]  CONT-300: the real Maros-Meszaros file is not available offline; its stand-in of the same class and
              shape (preprocess.boundary_control_qp(300): n_x = 91 200, m = 90 000, 450 000 non-zeros, diagonal H,
              all rows equalities -> 90 000 x 90 000 dense normal equations, 65 GB) is solved to the end
  configs[3]  batch of 1024 x (n_x = 512, m = 256): the 128-problem share of one of 8 GPUs through the lock-step
              batched engine, 8 sampled problems against the oracle, an optimality certificate for all 128
  configs[4]  n_x = 100 000, m = 40 000 on ONE MI355X (224 GB of the 288 GB)
  (configs[1], n_x = 5 000, and the metric config, n_x = 50 000, are tests/test_gpu_solver.py::test_full_size_*)

These sizes are far beyond the oracle, so they are checked through size-independent properties: generator tiles
bit for bit, sampled entries of the assembled K, the unreduced KKT residual of every solve (mul!), optimality
certificates computed from the returned primal-dual point alone.
"""
import numpy as np
import pytest
import torch

import madqp_jl_amd as M
from oracle import mpc
from oracle import qp as Q

pytestmark = pytest.mark.gpu
BENCH_OPTS = dict(max_iter=300, step_rule=M.AdaptiveStep(0.995), regularization=M.FixedRegularization(1e-8, -1e-8),
                  mu_min=1e-12)  # scripts/benchmarks_cpu.jl:35-44
ORACLE_OPTS = dict(max_iter=300, step_rule=mpc.AdaptiveStep(0.995), regularization=mpc.FixedRegularization(1e-8, -1e-8),
                   mu_min=1e-12)


def certificate(H, q, A, x, y, zl, zu, lvar, uvar, lcon, ucon, tol=1e-6):
    """Optimality conditions of  min x'Hx/2 + q'x  s.t. lcon <= Ax <= ucon, lvar <= x <= uvar  at (x, y, zl, zu),
    from the point alone: stationarity, primal feasibility, signs, complementarity (torch tensors on one device)."""
    g = q.clone() if H is None else H @ x + q
    scale = max(1.0, float(g.abs().max()))
    Ax = A @ x
    assert float((g + A.t() @ y - zl + zu).abs().max()) <= tol * scale, "stationarity"
    assert float((lvar - x).clamp(min=0).max()) <= tol and float((x - uvar).clamp(min=0).max()) <= tol
    assert float((lcon - Ax).clamp(min=0).max()) <= tol and float((Ax - ucon).clamp(min=0).max()) <= tol
    assert float(zl.min()) >= -1e-8 and float(zu.min()) >= -1e-8
    assert float((zl * (x - lvar)).abs().max()) <= tol * scale and float((zu * (uvar - x)).abs().max()) <= tol * scale
    # y > 0 pushes on the upper side of a row, y < 0 on the lower side (L = f + y'(Ax - s), zl - zu on the slacks)
    assert float((y.clamp(min=0) * (ucon - Ax)).abs().max()) <= tol * scale
    assert float((y.clamp(max=0) * (Ax - lcon)).abs().max()) <= tol * scale


def test_config_c4_batch_share_128_of_1024(hip):
    nx, m, seed, world = 512, 256, 20250614 + 3, 8
    mine = M.shard(range(1024), 0, world)  # problems 0, 8, 16, ...: the share of rank 0 of 8
    assert len(mine) == 128
    qps = [M.DeviceQP.synthetic(hip, seed + i, nx, m) for i in mine]
    s = M.BatchedMPCSolver(qps, hip, **BENCH_OPTS)
    res = s.solve(check_every=2)
    s.close()
    assert all(r["status"] == M.SOLVE_SUCCEEDED for r in res), [r["status"] for r in res]
    assert len({r["iter"] for r in res}) > 1  # problems leave the lock step at different iterations
    dev = hip.device
    for dq, r in zip(qps, res):
        t = lambda k: torch.as_tensor(r[k], device=dev)
        certificate(dq.H, dq.q, dq.A, t("solution"), t("multipliers"), t("multipliers_L"), t("multipliers_U"),
                    dq.lvar, dq.uvar, dq.lcon, dq.ucon)
        assert max(r["inf_pr"], r["inf_du"], r["inf_compl"]) <= 1e-8
    for k in (0, 17, 38, 64, 77, 101, 120, 127):  # sampled problems against the oracle on the same data
        ref = mpc.solve(Q.synthetic_qp(seed + mine[k], nx, m), kkt_system="condensed", **ORACLE_OPTS)
        r = res[k]
        assert r["status"] == ref["status"] and r["iter"] == ref["iter"], (k, r["iter"], ref["iter"])
        assert abs(r["objective"] - ref["objective"]) <= 1e-9 * max(1.0, abs(ref["objective"]))
        assert np.max(np.abs(r["solution"] - ref["solution"])) <= 1e-7
        assert np.max(np.abs(r["multipliers"] - ref["multipliers"])) <= 1e-6


def test_config_cont300_stand_in(hip):
    """boundary_control_qp(300) through the sparse front end and the diagonal-H normal equations: converges to
    tol 1e-8; the certificate is evaluated with the CSR Jacobian (sparse products on the device)."""
    from madqp_jl_amd import preprocess as P

    torch.cuda.empty_cache()
    h = P.boundary_control_qp(300)
    assert (h.nvar, h.ncon) == (91200, 90000)
    qp = P.to_device(h, hip)
    s = M.MPCSolver(qp, hip, kkt_system="normal", regularization=M.FixedRegularization(1e-8, 0.0), driver="native",
                    max_iter=100)
    r = s.solve()
    resid = s.last_residual_ratio
    s.close()
    assert r["status"] == M.SOLVE_SUCCEEDED, (r["status"], r["iter"])
    t = r["trace"][-1]
    assert max(t["inf_pr"], t["inf_du"], t["inf_compl"]) <= 1e-8 and resid < 1e-7
    dev = hip.device
    A = torch.sparse_csr_tensor(qp.A.ptr, qp.A.col, qp.A.val, size=(h.ncon, h.nvar))
    x, y, zl, zu = (torch.as_tensor(r[k], device=dev) for k in ("solution", "multipliers", "multipliers_L", "multipliers_U"))
    g = qp.H * x + qp.q  # diagonal Hessian
    scale = max(1.0, float(g.abs().max()))
    Ax = A @ x
    Aty = torch.zeros_like(x).index_add_(0, qp.A.col, qp.A.val * y[qp.A.row])
    assert float((g + Aty - zl + zu).abs().max()) <= 1e-6 * scale
    assert float((Ax - qp.lcon).abs().max()) <= 1e-6  # all rows are equalities (the discrete Laplace equation)
    assert float((qp.lvar - x).clamp(min=0).max()) <= 1e-7 and float((x - qp.uvar).clamp(min=0).max()) <= 1e-7
    assert float(zl.min()) >= -1e-8 and float(zu.min()) >= -1e-8
    assert float((zl * (x - qp.lvar)).abs().max()) <= 1e-6 * scale
    assert float((zu * (qp.uvar - x)).abs().max()) <= 1e-6 * scale
    # the discretisation is symmetric under i <-> j and the data are too: so is the state
    N = 300
    Y = x[: N * N].reshape(N, N)
    assert float((Y - Y.t()).abs().max()) <= 1e-4  # curvature h^2 ~ 1e-5 under a 1e-8 dual residual


def test_config_c5_100k_40k_on_one_gpu(hip):
    """The same properties as test_full_size_c_main_properties at n_x = 100 000, m = 40 000."""
    from test_gpu_solver import full_size_properties

    torch.cuda.empty_cache()
    full_size_properties(hip, 100000, 40000, 20250614 + 4)
