"""Lock-step solver for a batch of small, equally shaped QPs (BASELINE configs[3]; csrc/batch.hip).

All problems share (nx, m) and the pattern of finite bounds / equality rows.  The one-off set-up of
``MPCSolver.initialize`` (src/solver.jl:127-159: bounds, interior push, scaling) runs here as
elementwise torch ops over the stacked arrays; from ``madqp_batch_init`` on everything happens in
``libmadqp_hip.so``: a handful of launches per iteration for the whole batch, per-problem scalars on
the device, a finished problem masked out by its status word.  Across GPUs the batch is sharded by
``batch.shard`` (problem b -> rank b mod N, no communication).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import BATCH_SCALARS, CBatchData, ptr
from .options import IPMOptions
from .solver import _push_interior, get_index_constraints, native_options


class BatchedMPCSolver:
    """``qps``: list of :class:`DeviceQP` with identical shapes and bound patterns."""

    def __init__(self, qps, backend, **opts):
        if not qps:
            raise ValueError("empty batch")
        self.be, self.qps = backend, list(qps)
        self.opt = IPMOptions(**opts)
        if self.opt.kkt_system not in ("condensed", "normal") or self.opt.check_residual:
            raise ValueError("the batched driver supports the condensed KKT system and the normal equations, on one GPU")
        self.normal = self.opt.kkt_system == "normal"
        q0 = self.qps[0]
        self.B, self.nx, self.m = len(self.qps), q0.nvar, q0.ncon
        dev = backend.device
        st = lambda name: torch.stack([getattr(q, name) for q in self.qps]).contiguous()
        self.lvar, self.uvar, self.lcon, self.ucon = st("lvar"), st("uvar"), st("lcon"), st("ucon")
        host = lambda t: t.detach().cpu().numpy()
        ic = get_index_constraints(host(self.lvar[0]), host(self.uvar[0]), host(self.lcon[0]), host(self.ucon[0]),
                                   self.opt.fixed_variable_treatment or "relax_bound")
        same = lambda a: bool((torch.isfinite(a) == torch.isfinite(a[0])).all())
        if not (same(self.lvar) and same(self.uvar) and same(self.lcon) and same(self.ucon)
                and bool(((self.lcon == self.ucon) == (self.lcon[0] == self.ucon[0])).all())):
            raise ValueError("all problems of a batch must share the pattern of finite bounds and equality rows")
        if any(q.H is not None and q.H.dim() != 2 for q in self.qps) or any(not torch.is_tensor(q.A) for q in self.qps):
            raise ValueError("the batched driver takes dense H and dense A")
        if any((q.H is None) != (q0.H is None) for q in self.qps):
            raise ValueError("all problems of a batch must be QPs or all LPs")
        self.ind_ineq, self.ind_eq = ic["ind_ineq"], ic["ind_eq"]
        self.ns = len(self.ind_ineq)
        self.n = self.nx + self.ns
        self.ind_lb = torch.as_tensor(ic["ind_lb"], dtype=torch.int64, device=dev)
        self.ind_ub = torch.as_tensor(ic["ind_ub"], dtype=torch.int64, device=dev)
        self.nlb, self.nub = self.ind_lb.numel(), self.ind_ub.numel()
        reg = self.opt.regularization
        self._copt = native_options(self.opt)
        self._copt.kkt_form = 1 if self.normal else 0
        if self.normal and q0.H is not None:
            raise ValueError("The KKT system NormalKKTSystem supports only linear programs.")  # normalkkt.jl:45-48
        if not self.normal and len(self.ind_eq) and not (self._copt.regularization != 0 and reg.delta_d < 0.0):
            raise ValueError("the condensed KKT system needs dual regularization delta_d < 0 "
                             "when the problem has equality constraints")
        self.H = None if q0.H is None else st("H")
        self.A, self.q = st("A"), st("q")
        self.c0 = torch.as_tensor([q.c0 for q in self.qps], dtype=torch.float64, device=dev)
        self.x0, self.y0 = st("x0"), st("y0")
        self._h = None
        self.status = self.iters = self.scalars = None

    # ---- src/solver.jl:127-159, vectorised over the batch ----
    def initialize(self):
        opt, be, dev = self.opt, self.be, self.be.device
        B, nx, n, m = self.B, self.nx, self.n, self.m
        f64 = dict(dtype=torch.float64, device=dev)
        ineq = torch.as_tensor(self.ind_ineq, dtype=torch.int64, device=dev)
        one = torch.ones((), **f64)
        x = torch.zeros((B, n), **f64)
        x[:, :nx] = self.x0
        y = self.y0.clone()
        xl = torch.cat([self.lvar, self.lcon[:, ineq]], dim=1)
        xu = torch.cat([self.uvar, self.ucon[:, ineq]], dim=1)
        rhs = torch.where(self.lcon == self.ucon, self.lcon, torch.zeros_like(self.lcon))
        tol = opt.bound_relax_factor
        xl = torch.where(torch.isfinite(xl), xl - torch.maximum(one, xl.abs()) * tol, xl)
        xu = torch.where(torch.isfinite(xu), xu + torch.maximum(one, xu.abs()) * tol, xu)
        x = _push_interior(x, xl, xu, opt.bound_push, opt.bound_fac)
        H, A, q = self.H, self.A, self.q
        self.obj_scale = torch.ones(B, **f64)
        self.con_scale = torch.ones((B, m), **f64)
        if opt.scaling and (m or nx):  # MadNLP.set_scaling!(..., 100)
            if m and nx:
                self.con_scale = torch.minimum(one, 100.0 / A.abs().amax(dim=2))
            g = q.clone()
            if H is not None and nx:
                step = max(1, (1 << 27) // max(nx * nx, 1))  # bounded temporaries
                for b0 in range(0, B, step):
                    g[b0:b0 + step] += (H[b0:b0 + step] * x[b0:b0 + step, None, :nx]).sum(dim=2)
            gmax = g.abs().amax(dim=1) if nx else torch.zeros(B, **f64)
            self.obj_scale = torch.where(gmax > 0, torch.minimum(one, 100.0 / gmax), one)
            cs = self.con_scale
            y = y / cs
            rhs = rhs * cs
            x[:, nx:] *= cs[:, ineq]
            xl[:, nx:] *= cs[:, ineq]
            xu[:, nx:] *= cs[:, ineq]
            A = (cs[:, :, None] * A).contiguous()
            H = None if H is None else (self.obj_scale[:, None, None] * H).contiguous()
            q = self.obj_scale[:, None] * q
        self._H, self._A, self._q = H, A.contiguous(), q.contiguous()
        self._rhs, self._c0 = rhs.contiguous(), (self.obj_scale * self.c0).contiguous()
        self.x, self.xl, self.xu, self.y = x.contiguous(), xl.contiguous(), xu.contiguous(), y.contiguous()
        self.zl, self.zu = torch.zeros((B, n), **f64), torch.zeros((B, n), **f64)
        self.close()
        data = CBatchData(H=ptr(self._H), A=ptr(self._A), q=ptr(self._q), rhs=ptr(self._rhs), c0=ptr(self._c0),
                          x=ptr(self.x), xl=ptr(self.xl), xu=ptr(self.xu), zl=ptr(self.zl), zu=ptr(self.zu),
                          y=ptr(self.y))
        ineq_host = (C.c_int64 * max(1, self.ns))(*[int(i) for i in self.ind_ineq])
        h = C.c_void_p()
        be._ck(be.lib.madqp_batch_create(be.ctx, B, nx, m, self.ns, ineq_host, self.nlb, ptr(self.ind_lb),
                                         self.nub, ptr(self.ind_ub), C.byref(data), C.byref(self._copt),
                                         C.byref(h)))
        self._h = h
        be._ck(be.lib.madqp_batch_init(h, opt.mu_init, opt.bound_fac))

    def iterate(self, max_steps=None, check_every=1) -> int:
        """Advance every active problem by up to ``max_steps`` iterations; returns how many are still active."""
        n = C.c_int32()
        steps = self.opt.max_iter + 1 if max_steps is None else int(max_steps)
        self.be._ck(self.be.lib.madqp_batch_iterate(self._h, steps, int(check_every), C.byref(n)))
        return n.value

    def fetch(self):
        B = self.B
        status = (C.c_int32 * B)()
        iters = (C.c_int32 * B)()
        scal = (C.c_double * (B * len(BATCH_SCALARS)))()
        self.be._ck(self.be.lib.madqp_batch_results(self._h, status, iters, scal))
        self.status = np.frombuffer(status, dtype=np.int32).copy()
        self.iters = np.frombuffer(iters, dtype=np.int32).copy()
        self.scalars = np.frombuffer(scal, dtype=np.float64).reshape(B, len(BATCH_SCALARS)).copy()
        return self.status, self.iters, self.scalars

    def solve(self, check_every=1):
        """Returns one result dict per problem (same keys as :meth:`MPCSolver.result`, no trace)."""
        self.initialize()
        self.iterate(check_every=check_every)
        return self.results()

    def results(self):
        """One dict per problem.  The unscaling runs on the device over the whole batch and the four arrays come back in
        one copy each; the dicts hold row views (1024 problems: 2 ms instead of 8 for a per-problem numpy loop)."""
        status, iters, scal = self.fetch()
        nx, os_ = self.nx, self.obj_scale[:, None]
        h = lambda t: t.contiguous().cpu().numpy()
        x = h(self.x[:, :nx])
        y = h(self.y * self.con_scale / os_)
        zl, zu = h(self.zl[:, :nx] / os_), h(self.zu[:, :nx] / os_)
        obj = scal[:, BATCH_SCALARS.index("obj")] / h(self.obj_scale)
        col = {k: i for i, k in enumerate(BATCH_SCALARS)}
        c_pr, c_du, c_co, c_mu, c_dw, c_nf = (col[k] for k in ("inf_pr", "inf_du", "inf_compl", "mu", "del_w",
                                                                 "n_factorizations"))
        st, it, nf = status.tolist(), iters.tolist(), scal[:, c_nf].astype(np.int64).tolist()
        return [dict(status=st[b], iter=it[b], objective=obj[b], solution=x[b], multipliers=y[b],
                     multipliers_L=zl[b], multipliers_U=zu[b], inf_pr=scal[b, c_pr], inf_du=scal[b, c_du],
                     inf_compl=scal[b, c_co], mu=scal[b, c_mu], del_w=scal[b, c_dw], n_factorizations=nf[b])
                for b in range(self.B)]

    def close(self):
        if self._h is not None:
            self.be.lib.madqp_batch_destroy(self._h)
            self._h = None
