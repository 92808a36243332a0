"""Replay of julia/MadQPHIP.jl through the C ABI (test infrastructure).

Julia is absent from the build and GPU images, so the glue cannot run.  This module replays, call for call, what the
glue's methods do when MadIPM's own loop (src/solver.jl) drives them:

* ``ReplayKKTSystem`` / ``ReplayCholeskySolver``: one method per Julia method of the same name (``!`` dropped), each
  issuing exactly the ccall sequence of julia/MadQPHIP.jl -- COO callback buffers moved by ``madqp_coo_map_apply``,
  ``build_kkt!`` / ``solve!`` / ``mul!`` with the KKT's OWN state view (solver pointers NULL), the linear solver built
  from ``aug_com`` and answering ``solve!(ls, rhs)`` through ``madqp_chol_solve``.
* ``ReplayBackend``: the per-variable kernels.  The ones the glue overrides (``set_aug_diagonal_reg!`` ... ``get_fraction_
  to_boundary_step``) go to the library; everything the glue leaves to the reference's generic broadcasts over device
  arrays (axpy!, norms, the map!/mapreduce of init_starting_point!, the iterate update, adjust_boundary!, the model
  callbacks) is done here with torch ops on the same device tensors -- standing in for Julia's broadcasts -- and NOT
  with the library's fused kernels, so a missing override cannot hide behind them.

The loop order is madqp_jl_amd/solver.py's restatement of src/solver.jl:6-125,127-182,254-345 (itself tested against
the oracle); in particular ``factorize_wrapper!`` is the FIRST plugin call after ``initialize!`` (src/solver.jl:16-21),
before any ``set_aug_diagonal_reg!``.
"""
import ctypes as C
import os

import numpy as np
import torch

import madqp_jl_amd as M
from madqp_jl_amd._lib import CState, ptr

EPS = float(np.finfo(np.float64).eps)
GLUE_ENTRY_POINTS = set()  # every ABI symbol the replay (= the glue) touched, for the coverage assertion


def ccall(be, name, *args):
    """``check(ctx, ccall((name, libmadqp), Int32, ...))`` of the glue."""
    GLUE_ENTRY_POINTS.add(name)
    rc = getattr(be.lib, name)(*args)
    if rc < 0:
        be._ck(rc)
    if rc > 0:
        raise M.SolveException()
    return rc


class ReplayCholeskySolver:
    """HIPCholeskySolver of the glue: ``HIPCholeskySolver(aug_com; opt)``, ``factorize!``, ``solve!(s, rhs)``."""

    def __init__(self, be, aug_com_handle):  # ctor: keeps the matrix object, asks for its madqp_chol
        self.be, self.aug_com = be, aug_com_handle
        ch, order = C.c_void_p(), C.c_int64()
        ccall(be, "madqp_kkt_chol", aug_com_handle, C.byref(ch), C.byref(order))
        self.chol, self.order, self.info = ch, order.value, 0

    def factorize(self):  # MadNLP.factorize!
        info = C.c_int32()
        ccall(self.be, "madqp_kkt_factorize", self.aug_com, C.byref(info))
        self.info = info.value
        return self

    def is_factorized(self):  # MadIPM.is_factorized
        return self.info == 0

    def solve(self, rhs):  # MadNLP.solve!(s, rhs), in place (src/KKT/normalkkt.jl:196)
        ccall(self.be, "madqp_chol_solve", self.chol, ptr(rhs))
        return rhs


class ReplayKKTSystem:
    """HIPKKTSystem{..., F} of the glue, F in ("condensed", "augmented", "normal")."""

    def __init__(self, be, form, nx, m, ind_ineq, ind_lb, ind_ub, jac_I, jac_J, hess_I, hess_J):
        """``_create`` of the glue (create_kkt_system): patterns are 1-based int32 host arrays, as MadNLP keeps them."""
        self.be, self.form, self.nx, self.m = be, form, int(nx), int(m)
        dev, f64 = be.device, torch.float64
        self.ns = len(ind_ineq)
        self.n = self.nx + self.ns
        z = lambda k: torch.zeros(max(int(k), 1), dtype=f64, device=dev)[: int(k)]
        nnzj, nnzh = len(jac_I), len(hess_I)
        if form == "normal" and nnzh > 0:
            raise ValueError("The KKT system NormalKKTSystem supports only linear programs.")
        jI, jJ = np.ascontiguousarray(jac_I, dtype=np.int32), np.ascontiguousarray(jac_J, dtype=np.int32)
        hI, hJ = np.ascontiguousarray(hess_I, dtype=np.int32), np.ascontiguousarray(hess_J, dtype=np.int32)
        if form == "normal":
            self.jac_map = self._coo_map(jJ, jI, self.nx, self.m, 0)
            self.At, self.lda = z(self.nx * self.m), max(self.m, 1)
        else:
            self.jac_map = self._coo_map(jI, jJ, self.m, self.nx, 0)
            self.At, self.lda = z(self.nx * self.m), max(self.nx, 1)
        self.hess_map = self._coo_map(hI, hJ, self.nx, self.nx, 1) if nnzh else None
        self.H = z(self.nx * self.nx) if nnzh else None
        self.jac, self.hess = z(nnzj), z(nnzh)
        ineq0 = (C.c_int64 * max(self.ns, 1))(*[int(i) for i in ind_ineq])
        h = C.c_void_p()
        ldh = max(self.nx, 1)
        if form == "condensed":
            ccall(be, "madqp_kkt_create", be.ctx, self.nx, self.m, self.ns, ineq0, ptr(self.H), ldh, ptr(self.At),
                  self.lda, C.byref(h))
        elif form in ("augmented", "scaled_augmented"):
            ccall(be, "madqp_kkt_create_augmented" if form == "augmented" else "madqp_kkt_create_scaled_augmented",
                  be.ctx, self.nx, self.m, self.ns, ineq0, ptr(self.H), ldh, ptr(self.At), self.lda, C.byref(h))
        else:
            ccall(be, "madqp_kkt_create_normal", be.ctx, self.nx, self.m, self.ns, ineq0, ptr(self.At), self.lda,
                  C.byref(h))
        self.handle = h
        # the glue's AUTO refinement request (ENV["MADQP_KKT_REFINE"], default -1): MadIPM's solve_system! calls solve! once
        ccall(be, "madqp_kkt_set_refine", h, int(os.environ.get("MADQP_KKT_REFINE", "-1")))
        self.linear_solver = ReplayCholeskySolver(be, h)  # linear_solver(aug_com; opt = opt_linear_solver)
        n, m = self.n, self.m
        self.ind_lb0 = torch.as_tensor(ind_lb, dtype=torch.int64, device=dev).contiguous()
        self.ind_ub0 = torch.as_tensor(ind_ub, dtype=torch.int64, device=dev).contiguous()
        nlb, nub = self.ind_lb0.numel(), self.ind_ub0.numel()
        self.reg, self.pr_diag, self.du_diag = z(n), z(n), z(m)
        self.l_diag, self.u_diag, self.l_lower, self.u_lower = z(nlb), z(nub), z(nlb), z(nub)
        cs = CState()  # the KKT's own view: solver-level pointers stay NULL
        cs.n, cs.m, cs.nlb, cs.nub = n, m, nlb, nub
        cs.ind_lb, cs.ind_ub = ptr(self.ind_lb0), ptr(self.ind_ub0)
        for k in ("reg", "pr_diag", "du_diag", "l_diag", "l_lower", "u_diag", "u_lower"):
            setattr(cs, k, ptr(getattr(self, k)))
        self.cstate = cs
        self.n_factorizations = 0

    def _coo_map(self, I, J, nrows, ncols, symmetric):
        h = C.c_void_p()
        ccall(self.be, "madqp_coo_map_create", self.be.ctx, len(I), I.ctypes.data_as(C.c_void_p),
              J.ctypes.data_as(C.c_void_p), nrows, ncols, symmetric, C.byref(h))
        return h

    def close(self):
        if self.handle is not None:
            self.be.lib.madqp_kkt_destroy(self.handle)
            self.be.lib.madqp_coo_map_destroy(self.jac_map)
            if self.hess_map is not None:
                self.be.lib.madqp_coo_map_destroy(self.hess_map)
            self.handle = None

    def num_variables(self):
        return self.n

    def get_jacobian(self):  # the nnzj buffer the SparseCallback fills
        return self.jac

    def get_hessian(self):
        return self.hess

    def initialize(self):  # MadNLP.initialize!(kkt)
        ccall(self.be, "madqp_kkt_initialize", self.handle, C.byref(self.cstate))

    def set_aug_diagonal_reg(self, del_w, del_c):  # MadIPM.set_aug_diagonal_reg!(kkt, solver): state(solver)
        ccall(self.be, "madqp_kkt_set_aug_diagonal_reg", self.handle, C.byref(self.solver_state.cstruct), del_w, del_c)

    def compress_jacobian(self):
        ccall(self.be, "madqp_coo_map_apply", self.jac_map, ptr(self.jac), ptr(self.At), self.lda)

    def compress_hessian(self):
        if self.hess_map is not None:
            ccall(self.be, "madqp_coo_map_apply", self.hess_map, ptr(self.hess), ptr(self.H), max(self.nx, 1))

    def jtprod(self, out, y):
        ccall(self.be, "madqp_kkt_jtprod", self.handle, ptr(out), ptr(y))

    def build_kkt(self):
        ccall(self.be, "madqp_kkt_build", self.handle, C.byref(self.cstate))

    def factorize_wrapper(self):  # MadNLP.factorize_wrapper! = build_kkt!(kkt); factorize!(kkt.linear_solver)
        self.build_kkt()
        self.linear_solver.factorize()
        self.n_factorizations += 1

    def solve(self, w):
        ccall(self.be, "madqp_kkt_solve", self.handle, C.byref(self.cstate), ptr(w))
        return w

    def mul(self, w, v, alpha=1.0, beta=0.0):
        ccall(self.be, "madqp_kkt_mul", self.handle, C.byref(self.cstate), ptr(w), ptr(v), alpha, beta)
        return w


class ReplayDistributedCholeskySolver:
    """HIPDistributedCholeskySolver of the glue: factorize! -> madqp_dkkt_factorize, solve!(s, rhs) -> madqp_dist_solve."""

    def __init__(self, be, aug_com):
        self.be, self.aug_com, self.info = be, aug_com, 0

    def factorize(self):
        info = C.c_int32()
        ccall(self.be, "madqp_dkkt_factorize", self.aug_com["dkkt"], C.byref(info))
        self.info = info.value
        return self

    def is_factorized(self):
        return self.info == 0

    def solve(self, rhs):
        ccall(self.be, "madqp_dist_solve", self.aug_com["dist"], ptr(rhs))
        return rhs


class ReplayDistributedKKTSystem:
    """HIPDistributedKKTSystem of the glue (create_kkt_system and every method), for ONE rank of a P x Q grid; with
    world == 1 no RCCL id is drawn (`cfg.world > 1 && ...` in the glue)."""

    def __init__(self, be, nx, m, ind_ineq, ind_lb, ind_ub, jac_I, jac_J, hess_I, hess_J, rank=0, world=1, P=1, Q=1,
                 nb=128, comm_ops=None):
        self.be, self.nx, self.m = be, int(nx), int(m)
        dev, f64 = be.device, torch.float64
        self.ns = len(ind_ineq)
        self.n = self.nx + self.ns
        z = lambda k: torch.zeros(max(int(k), 1), dtype=f64, device=dev)[: int(k)]
        d = C.c_void_p()
        ccall(be, "madqp_dist_create", be.ctx, rank, world, P, Q, self.nx, nb, None, comm_ops, C.byref(d))
        lay = (C.c_int64 * 8)()
        ccall(be, "madqp_dist_layout", d, lay)
        p, q, _, _, self.mloc, self.nloc, self.ld, self.ncp = list(lay)
        m16 = max((self.m + 15) // 16 * 16, 1)
        self.A_I = torch.zeros((m16, self.ld), dtype=f64, device=dev)
        self.A_J = torch.zeros((m16, self.ncp), dtype=f64, device=dev)
        nnzj, nnzh = len(jac_I), len(hess_I)
        self.Hloc = torch.zeros((self.ncp, self.ld), dtype=f64, device=dev) if nnzh else None
        jI, jJ = np.ascontiguousarray(jac_I, dtype=np.int32), np.ascontiguousarray(jac_J, dtype=np.int32)
        hI, hJ = np.ascontiguousarray(hess_I, dtype=np.int32), np.ascontiguousarray(hess_J, dtype=np.int32)
        self._keep = (jI, jJ, hI, hJ)

        def cols_map(R, r):
            h = C.c_void_p()
            ccall(be, "madqp_coo_map_create_cols_cyclic", be.ctx, nnzj, jI.ctypes.data_as(C.c_void_p),
                  jJ.ctypes.data_as(C.c_void_p), self.m, self.nx, nb, R, r, C.byref(h))
            return h

        self.jacI_map, self.jacJ_map = cols_map(P, p), cols_map(Q, q)
        self.hess_map = None
        if nnzh:
            h = C.c_void_p()
            ccall(be, "madqp_coo_map_create_tiles_cyclic", be.ctx, nnzh, hI.ctypes.data_as(C.c_void_p),
                  hJ.ctypes.data_as(C.c_void_p), self.nx, nb, P, p, Q, q, C.byref(h))
            self.hess_map = h
        self.jac, self.hess = z(nnzj), z(nnzh)
        ineq0 = (C.c_int64 * max(self.ns, 1))(*[int(i) for i in ind_ineq])
        k = C.c_void_p()
        ccall(be, "madqp_dkkt_create", d, self.nx, self.m, self.ns, ineq0, ptr(self.Hloc), self.ld, ptr(self.A_I),
              self.ld, ptr(self.A_J), self.ncp, C.byref(k))
        self.aug_com = {"dist": d, "dkkt": k}
        self.linear_solver = ReplayDistributedCholeskySolver(be, self.aug_com)
        n, m = self.n, self.m
        self.ind_lb0 = torch.as_tensor(ind_lb, dtype=torch.int64, device=dev).contiguous()
        self.ind_ub0 = torch.as_tensor(ind_ub, dtype=torch.int64, device=dev).contiguous()
        nlb, nub = self.ind_lb0.numel(), self.ind_ub0.numel()
        self.reg, self.pr_diag, self.du_diag = z(n), z(n), z(m)
        self.l_diag, self.u_diag, self.l_lower, self.u_lower = z(nlb), z(nub), z(nlb), z(nub)
        cs = CState()
        cs.n, cs.m, cs.nlb, cs.nub = n, m, nlb, nub
        cs.ind_lb, cs.ind_ub = ptr(self.ind_lb0), ptr(self.ind_ub0)
        for f in ("reg", "pr_diag", "du_diag", "l_diag", "l_lower", "u_diag", "u_lower"):
            setattr(cs, f, ptr(getattr(self, f)))
        self.cstate = cs
        self.n_factorizations = 0

    def close(self):
        if self.aug_com is not None:
            self.be.lib.madqp_dkkt_destroy(self.aug_com["dkkt"])
            self.be.lib.madqp_dist_destroy(self.aug_com["dist"])
            for h in (self.jacI_map, self.jacJ_map, self.hess_map):
                if h is not None:
                    self.be.lib.madqp_coo_map_destroy(h)
            self.aug_com = None

    def num_variables(self):
        return self.n

    def get_jacobian(self):
        return self.jac

    def get_hessian(self):
        return self.hess

    def initialize(self):  # fill! of the glue (torch stands in for Julia's broadcasts)
        for t, v in ((self.reg, 1.0), (self.pr_diag, 1.0), (self.du_diag, 0.0), (self.l_lower, 0.0), (self.u_lower, 0.0),
                     (self.l_diag, 1.0), (self.u_diag, 1.0)):
            t.fill_(v)

    def set_aug_diagonal_reg(self, del_w, del_c):
        ccall(self.be, "madqp_set_aug_diagonal_reg", self.be.ctx, C.byref(self.solver_state.cstruct), del_w, del_c)

    def compress_jacobian(self):
        for mp, dst, ld in ((self.jacI_map, self.A_I, self.ld), (self.jacJ_map, self.A_J, self.ncp)):
            ccall(self.be, "madqp_coo_map_apply", mp, ptr(self.jac), ptr(dst), ld)

    def compress_hessian(self):
        if self.hess_map is not None:
            ccall(self.be, "madqp_coo_map_apply", self.hess_map, ptr(self.hess), ptr(self.Hloc), self.ld)

    def jtprod(self, out, y):
        ccall(self.be, "madqp_dkkt_jtprod", self.aug_com["dkkt"], ptr(out), ptr(y))

    def build_kkt(self):
        ccall(self.be, "madqp_dkkt_build", self.aug_com["dkkt"], C.byref(self.cstate))

    def factorize_wrapper(self):
        self.build_kkt()
        self.linear_solver.factorize()
        self.n_factorizations += 1

    def solve(self, w):
        ccall(self.be, "madqp_dkkt_solve", self.aug_com["dkkt"], C.byref(self.cstate), ptr(w))
        return w

    def mul(self, w, v, alpha=1.0, beta=0.0):
        ccall(self.be, "madqp_dkkt_mul", self.aug_com["dkkt"], C.byref(self.cstate), ptr(w), ptr(v), alpha, beta)
        return w


class ReplayState(M.State):
    """MPCSolver's vectors, with the KKT diagonals ALIASED to the fields of the KKT object (in Julia the solver
    reads ``solver.kkt.reg`` etc. -- there is one copy, owned by the KKT system)."""

    def adopt(self, kkt):
        for k in ("reg", "pr_diag", "du_diag", "l_diag", "l_lower", "u_diag", "u_lower"):
            setattr(self, k, getattr(kkt, k))
        self.ind_lb, self.ind_ub = kkt.ind_lb0, kkt.ind_ub0
        self._c = None  # state(solver) of the glue is built after the KKT system exists
        kkt.solver_state = self


class ReplayBackend(M.HipBackend):
    """Kernels of src/kernels.jl: library calls where the glue overrides a method, torch broadcasts elsewhere."""

    def new_state(self, n, m, ind_lb, ind_ub):
        return ReplayState(n, m, ind_lb, ind_ub, self.device)

    # ---- overridden by the glue: the inherited HipBackend methods call the library; record them ----
    def _bound(name):  # noqa: N805
        def call(self, *a, **k):
            GLUE_ENTRY_POINTS.add("madqp_" + name)
            return getattr(M.HipBackend, name)(self, *a, **k)
        return call

    for _n in ("set_initial_primal_rhs", "set_initial_dual_rhs", "set_predictive_rhs",
               "set_correction_rhs", "get_correction", "set_extra_correction", "get_complementarity_measure",
               "get_affine_complementarity_measure", "get_alpha_max"):
        locals()[_n] = _bound(_n)
    del _n, _bound

    # ---- NOT overridden by the glue: generic broadcasts over device arrays (torch stands in for Julia) ----
    def copy(self, src, dst):
        dst.copy_(src)

    def fill(self, value, dst):
        dst.fill_(value)

    def axpy(self, alpha, x, y):
        y.add_(x, alpha=alpha)

    def norm_inf(self, a):
        return float(a.abs().max()) if a.numel() else 0.0

    def norm_inf3(self, a, b, c):
        nrm = lambda t: float("nan") if bool(torch.isnan(t).any()) else self.norm_inf(t)
        return nrm(a), nrm(b), nrm(c)

    def gemv(self, trans, rows, cols, alpha, A, lda, x, beta, y):
        Am = A.reshape(-1)[: rows * lda].reshape(rows, lda)[:, :cols]
        r = (Am.t() @ x[:rows]) if trans else (Am @ x[:cols])
        (y[:cols] if trans else y[:rows]).mul_(beta).add_(r, alpha=alpha)

    def get_inf(self, st):  # MadNLP.get_inf_pr / get_inf_du / get_inf_compl as called at src/solver.jl:264-272
        ilb, iub = st.ind_lb, st.ind_ub
        nrm = lambda t: float(t.abs().max()) if t.numel() else 0.0
        compl = max(nrm((st.x[ilb] - st.xl[ilb]) * st.zl[ilb]), nrm((st.xu[iub] - st.x[iub]) * st.zu[iub]))
        return nrm(st.c), nrm(st.f - st.zl + st.zu + st.jacl), compl

    def update_iterates(self, st, alpha_p, alpha_d):  # src/solver.jl:332-335
        st.x.add_(st.primal(st.d), alpha=alpha_p)
        st.y.add_(st.dual(st.d), alpha=alpha_d)
        st.zl[st.ind_lb] += alpha_d * st.dual_lb(st.d)
        st.zu[st.ind_ub] += alpha_d * st.dual_ub(st.d)

    def adjust_boundary(self, st, mu):  # MadNLP.adjust_boundary!
        c1, c2 = EPS * mu, EPS ** 0.75
        ilb, iub = st.ind_lb, st.ind_ub
        x, xl, xu = st.x, st.xl, st.xu
        one = torch.ones((), dtype=x.dtype, device=x.device)
        xl[ilb] = torch.where(x[ilb] - xl[ilb] < c1, xl[ilb] - c2 * torch.maximum(one, x[ilb].abs()), xl[ilb])
        xu[iub] = torch.where(xu[iub] - x[iub] < c1, xu[iub] + c2 * torch.maximum(one, x[iub].abs()), xu[iub])

    # init_starting_point! (src/solver.jl:37-123): map! / mapreduce over the solver's vectors
    def sp_init_duals(self, st):
        res, l, u = st.jacl, st.xl, st.xu
        fl, fu = torch.isfinite(l), torch.isfinite(u)
        st.zl.copy_(torch.where(fl & fu, 0.5 * res, torch.where(fl, res, st.zl)))
        st.zu.copy_(torch.where(fl & fu, -0.5 * res, torch.where(fu, -res, st.zu)))

    def sp_mins(self, st):
        ilb, iub = st.ind_lb, st.ind_ub
        mn = lambda t: min(0.0, float(t.min())) if t.numel() else 0.0
        return [mn(st.x[ilb] - st.xl[ilb]), mn(st.xu[iub] - st.x[iub]), mn(st.zl[ilb]), mn(st.zu[iub])]

    def sp_shift(self, st, dx, dz):
        ilb, iub = st.ind_lb, st.ind_ub
        st.x[ilb] += dx  # x_lr and x_ur are views of the same x
        st.x[iub] -= dx
        st.zl[ilb] += dz
        st.zu[iub] += dz

    def sp_sums(self, st):
        ilb, iub = st.ind_lb, st.ind_ub
        f = lambda t: float(t) if t.numel() else 0.0
        return [f(st.x[ilb] @ st.zl[ilb]), f(st.xl[ilb] @ st.zl[ilb]), f(st.xu[iub] @ st.zu[iub]),
                f(st.x[iub] @ st.zu[iub]), f(st.zl[ilb].sum()), f(st.zu[iub].sum()),
                f((st.x[ilb] - st.xl[ilb]).sum()), f((st.xu[iub] - st.x[iub]).sum())]

    def sp_project(self, st, kappa):
        x, l, u = st.x, st.xl, st.xu
        one = torch.ones((), dtype=x.dtype, device=x.device)
        pl = torch.minimum(kappa * torch.maximum(one, l), kappa * (u - l))
        pu = torch.minimum(kappa * torch.maximum(one, u), kappa * (u - l))
        x.copy_(torch.where(x < l, l + pl, torch.where(u < x, u - pu, x)))

    def sp_check(self, st):
        ilb, iub = st.ind_lb, st.ind_ub
        return bool((st.zl[ilb] > 0).all() and (st.zu[iub] > 0).all() and (st.x[ilb] > st.xl[ilb]).all()
                    and (st.x[iub] < st.xu[iub]).all())


def coo_pattern(A, rng, duplicates=3):
    """A sparsity pattern as a model would report it: the non-zeros of ``A`` in shuffled order, 1-based int32, with a
    few entries split in two (duplicates add up, as in MadNLP's COO matrices).  Returns (I, J, values)."""
    i, j = np.nonzero(A)
    v = A[i, j].astype(np.float64)
    if len(i) and duplicates:
        pick = rng.choice(len(i), size=min(duplicates, len(i)), replace=False)
        part = v[pick] * 0.5  # halves add up exactly: the dense operand equals A bit for bit
        v[pick] -= part
        i, j, v = np.concatenate([i, i[pick]]), np.concatenate([j, j[pick]]), np.concatenate([v, part])
    order = rng.permutation(len(i))
    return (i[order] + 1).astype(np.int32), (j[order] + 1).astype(np.int32), v[order]


class ReplayMPCSolver(M.MPCSolver):
    """MadIPM.MPCSolver with ``kkt_system = MadQPHIP.HIP*KKTSystem, linear_solver = MadQPHIP.HIPCholeskySolver``:
    the loop of solver.py, the plugin objects of this module, the model behind a SparseCallback-like COO interface."""

    def __init__(self, qp, be, seed=0, distributed_tile=None, **opts):
        opts["driver"] = "python"
        super().__init__(qp, be, **opts)
        self._rng = np.random.default_rng(seed)
        self._dist_nb = distributed_tile  # not None: kkt_system = MadQPHIP.HIPDistributedKKTSystem (one rank, tile nb)

    def _create_kkt_system(self):
        A = self.A.detach().cpu().numpy()
        H = None if self.H is None else self.H.detach().cpu().numpy()
        self._jI, self._jJ, jv = coo_pattern(A, self._rng)
        if H is not None:
            hI, hJ, hv = coo_pattern(np.tril(H), self._rng)  # MadNLP's Hessians are lower triangular
        else:
            hI, hJ, hv = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)
        st = self.st
        if self._dist_nb:
            kkt = ReplayDistributedKKTSystem(self.be, self.nx, self.m, self.ind_ineq, st.ind_lb.cpu().numpy(),
                                             st.ind_ub.cpu().numpy(), self._jI, self._jJ, hI, hJ, nb=self._dist_nb)
        else:
            kkt = ReplayKKTSystem(self.be, self.opt.kkt_system, self.nx, self.m, self.ind_ineq,
                                  st.ind_lb.cpu().numpy(), st.ind_ub.cpu().numpy(), self._jI, self._jJ, hI, hJ)
        st.adopt(kkt)
        # eval_jac_wrapper! / eval_lag_hess_wrapper! (src/solver.jl:167,170): the callback fills the buffers, then
        # compress_*! moves them into the dense operands (QP: constant, evaluated once)
        kkt.get_jacobian().copy_(torch.as_tensor(jv, device=self.be.device))
        kkt.compress_jacobian()
        if len(hv):
            kkt.get_hessian().copy_(torch.as_tensor(hv, device=self.be.device))
            kkt.compress_hessian()
        kkt.eval_model = self._eval_model
        return kkt

    def _eval_model(self, q, rhs, c0):
        """obj / grad! / cons! of the model (NLPModels callbacks in Julia, scripts/qp_gpu.jl:29-40)."""
        st, nx = self.st, self.nx
        x = st.x[:nx]
        Hx = torch.zeros_like(x) if self.H is None else self.H @ x
        st.f[:nx] = Hx + q
        st.f[nx:] = 0.0
        c = (self.A @ x) if self.m else st.c
        if self.m:
            c[torch.as_tensor(self.ind_ineq, device=c.device)] -= st.x[nx:]
            st.c.copy_(c - rhs)
        return float(c0 + q @ x + 0.5 * (x @ Hx))
