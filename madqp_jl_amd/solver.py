"""Host driver of the Mehrotra predictor-corrector loop on the MI355X path.

This is the host side of the drop-in: it keeps the reference's control flow
(``src/solver.jl``, ``src/linear_solver.jl``) and calls, one for one, the fused
HIP kernels that replace ``src/kernels.jl`` and the KKT / linear-solver plugin
(``madqp_jl_amd/kkt.py``).  Only scalars cross the PCIe bus inside the loop.
One-off set-up (bounds, push into the interior, scaling; ``src/solver.jl:127-159``)
uses elementwise torch ops on the device.

In Julia the same sequence is MadIPM's own ``mpc!`` with the methods of
``julia/MadQPHIP.jl`` dispatched on ``HIPCondensedKKTSystem``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ._lib import CMpcInfo, CMpcOptions
from .kkt import (HIPAugmentedKKTSystem, HIPCondensedKKTSystem, HIPNormalKKTSystem, HIPScaledAugmentedKKTSystem,
                  HIPSparseAugmentedKKTSystem, HIPSparseCondensedKKTSystem, HIPSparseNormalKKTSystem)
from .options import (AdaptiveRegularization, AdaptiveStep, ConservativeStep, FixedRegularization,
                      IPMOptions, MehrotraAdaptiveStep, NoRegularization)
from .qp import DeviceCSR, DeviceQP


def _fdiv(a: float, b: float) -> float:
    """IEEE division of two host scalars, as Julia's ``/`` on Float64 (x / 0 = +-Inf, 0 / 0 = NaN): the reference's loop
    forms such quotients without a guard -- the start point's 0 / 0 when a problem has no bound at all
    (src/solver.jl:93-94), a step length against a zero direction component (src/kernels.jl:341-368) -- and goes on;
    Python's float division raises instead."""
    if b != 0.0:
        return a / b
    if a == 0.0 or a != a:
        return float("nan")
    return math.copysign(float("inf"), a) * math.copysign(1.0, b)

SOLVE_SUCCEEDED = 1
MAXIMUM_ITERATIONS_EXCEEDED = 6
ERROR_IN_STEP_COMPUTATION = -3
INTERNAL_ERROR = -1


class SolveException(Exception):
    """MadNLP.SolveException (src/linear_solver.jl:41-43)."""


def get_index_constraints(lvar, uvar, lcon, ucon, fixed_variable_treatment="error"):
    """MadNLP.get_index_constraints as called at src/structure.jl:95-102 (host, numpy); EnforceEquality.
    Fixed variables (lvar == uvar): "relax_bound" = MadNLP.RelaxBound (they stay variables with both bounds, which
    initialize relaxes by bound_relax_factor); anything else refuses them -- MakeParameter removes them before this
    point (DeviceQP.eliminate_fixed), so none are left when that treatment is active."""
    ind_eq = np.flatnonzero(lcon == ucon)
    ind_ineq = np.flatnonzero(lcon != ucon)
    xl = np.concatenate([lvar, lcon[ind_ineq]])
    xu = np.concatenate([uvar, ucon[ind_ineq]])
    if np.any(xl == xu) and fixed_variable_treatment != "relax_bound":
        raise NotImplementedError("fixed variables: pass fixed_variable_treatment='relax_bound' or 'make_parameter'")
    return dict(ind_eq=ind_eq, ind_ineq=ind_ineq, ind_lb=np.flatnonzero(xl != -np.inf),
                ind_ub=np.flatnonzero(xu != np.inf))


def _push_interior(x, l, u, bound_push, bound_fac):
    """MadNLP._initialize_variables! (via src/solver.jl:131-142), elementwise on device."""
    one = torch.ones((), dtype=x.dtype, device=x.device)
    fl, fu = torch.isfinite(l), torch.isfinite(u)
    both, lo, up = fl & fu, fl & ~fu, ~fl & fu
    pl = torch.minimum(bound_push * torch.maximum(one, l.abs()), bound_fac * (u - l))
    pu = torch.minimum(bound_push * torch.maximum(one, u.abs()), bound_fac * (u - l))
    out = torch.where(both, torch.minimum(u - pu, torch.maximum(l + pl, x)), x)
    out = torch.where(lo, torch.maximum(l + bound_push * torch.maximum(one, l.abs()), x), out)
    out = torch.where(up, torch.minimum(u - bound_push * torch.maximum(one, u.abs()), x), out)
    return out


def native_options(opt) -> CMpcOptions:
    """``madqp_mpc_options`` (include/madqp.h) of an :class:`IPMOptions`."""
    rule, reg = opt.step_rule, opt.regularization
    c = CMpcOptions(tol=opt.tol, max_iter=opt.max_iter, max_ncorr=opt.max_ncorr, mu_min=opt.mu_min,
                    check_residual=int(bool(opt.check_residual)), tol_linear_solve=opt.tol_linear_solve,
                    refine_steps=int(opt.refine_steps or 0))  # (None: AUTO, resolved by MPCSolver; the batched engine does not refine)
    if isinstance(rule, ConservativeStep):
        c.step_rule, c.step_param = 0, rule.tau
    elif isinstance(rule, AdaptiveStep):
        c.step_rule, c.step_param = 1, rule.tau_min
    elif isinstance(rule, MehrotraAdaptiveStep):
        c.step_rule, c.step_param = 2, rule.gamma_f
    else:
        raise TypeError(rule)
    if isinstance(reg, NoRegularization):
        c.regularization = 0
    elif isinstance(reg, FixedRegularization):
        c.regularization, c.delta_p, c.delta_d = 1, reg.delta_p, reg.delta_d
    else:
        c.regularization, c.delta_p, c.delta_d, c.delta_min = 2, reg.delta_p, reg.delta_d, reg.delta_min
    return c


class MPCSolver:
    """``MPCSolver(nlp; opts...)`` (src/structure.jl:77-176) for a :class:`DeviceQP`."""

    def __init__(self, qp: DeviceQP, backend, **opts):
        self.be = backend
        self.opt = IPMOptions(**opts)
        host = lambda t: t.detach().cpu().numpy()
        # src/utils.jl:81: RelaxBound for condensed KKT systems, MakeParameter otherwise
        fvt = self.opt.fixed_variable_treatment or (
            "relax_bound" if self.opt.kkt_system in ("condensed", "augmented", "scaled_augmented") else "make_parameter")
        if fvt not in ("relax_bound", "make_parameter", "error"):
            raise ValueError(f"unknown fixed_variable_treatment {fvt!r}")
        self.full_qp, self._fixed = qp, None
        if fvt == "make_parameter":
            self._fixed = qp.eliminate_fixed()
            if self._fixed is not None:
                qp = self._fixed[0]
                if qp.nvar == 0:
                    raise ValueError("every variable is fixed: nothing to optimise")
        self.qp = qp
        ic = get_index_constraints(host(qp.lvar), host(qp.uvar), host(qp.lcon), host(qp.ucon), fvt)
        self.ind_ineq, self.ind_eq = ic["ind_ineq"], ic["ind_eq"]
        self.nx, self.ns = qp.nvar, len(self.ind_ineq)
        self.n, self.m = self.nx + self.ns, qp.ncon
        self.st = backend.new_state(self.n, self.m, ic["ind_lb"], ic["ind_ub"])
        self.nlb, self.nub = self.st.nlb, self.st.nub
        reg = self.opt.regularization
        if self.opt.kkt_system not in ("condensed", "normal", "augmented", "scaled_augmented"):
            raise ValueError(f"unknown kkt_system {self.opt.kkt_system!r}")
        diag_h = qp.H is not None and qp.H.dim() == 1  # H = diag(vector): sparse front end only
        if self.opt.kkt_system == "scaled_augmented" and (diag_h or isinstance(qp.A, DeviceCSR)):
            raise ValueError("the scaled augmented (K2.5) system takes a dense Jacobian and a dense (or no) Hessian")
        if diag_h and not isinstance(qp.A, DeviceCSR) and self.opt.kkt_system != "augmented":
            raise ValueError("a diagonal Hessian (1-D tensor) needs the sparse front end (A as DeviceCSR) "
                             "or kkt_system='augmented'")
        if self.opt.kkt_system in ("augmented", "scaled_augmented") and hasattr(qp, "grid"):
            raise ValueError("the augmented KKT system is factorised on one GPU")
        if self.opt.kkt_system == "normal" and qp.H is not None and not diag_h:
            raise ValueError("The KKT system NormalKKTSystem supports only linear programs.")
        if self.opt.kkt_system == "condensed" and len(self.ind_eq) and (
                isinstance(reg, NoRegularization) or reg.delta_d >= 0.0):
            raise ValueError("the condensed KKT system needs dual regularization delta_d < 0 "
                             "when the problem has equality constraints")
        if self.opt.refine_steps is None:  # AUTO (options.py): by the order of the matrix that is factorised
            order = {"condensed": self.nx, "normal": self.m}.get(self.opt.kkt_system, self.nx + self.m)
            self.opt.refine_steps = IPMOptions.refine_auto(order)
        self.obj_scale, self.con_scale = 1.0, None
        self.H, self.A, self.q = qp.H, qp.A, qp.q  # replaced by scaled copies if scaling != 1
        self.kkt = None
        self.obj_val = 0.0
        self.inf_pr = self.inf_du = self.inf_compl = 0.0
        self.norm_b = self.norm_c = 0.0
        self.mu = 0.0
        self.alpha_p = self.alpha_d = 0.0
        self.del_w = self.del_c = 0.0
        self.k = 0
        self.status = None
        self.trace = []
        self.last_residual_ratio = 0.0
        self.dnorm = 0.0
        if self.opt.driver not in ("python", "native"):
            raise ValueError(f"unknown driver {self.opt.driver!r}")
        if hasattr(qp, "grid") and self.opt.driver == "native":
            raise ValueError("the native driver factorizes on one GPU; a DistributedQP needs driver='python'")
        self._native = None  # madqp_mpc handle (driver="native")
        self._info = CMpcInfo()
        self._fact_closed = 0  # factorizations of KKT objects that initialize() has already released

    @property
    def n_factorizations_total(self) -> int:
        """Factorizations since this solver was constructed: monotone over re-initialisations (``initialize``
        releases and recreates the KKT object, whose own counter restarts at 0).  counters of
        scripts/benchmarks_cpu.jl:52-55 are per solve; a benchmark that re-initialises inside its timed region
        reads this one."""
        return self._fact_closed + (self.kkt.n_factorizations if self.kkt is not None else 0)

    # ---- driver="native": the loop body runs in csrc/mpc.hip, one foreign call per iteration ----
    def _native_open(self):
        self._native_close()
        st = self.st
        self._native = self.be.mpc_create(self.kkt._h, st, st.w1, st.w2, self.q, st.rhs,
                                          self.obj_scale * self.qp.c0, self.norm_b, self.norm_c,
                                          native_options(self.opt))
        self.be.mpc_set_scalars(self._native, self.mu, self.del_w, self.del_c, self.obj_val, self.k)

    def _native_close(self):
        if self._native is not None:
            self.be.mpc_destroy(self._native)
            self._native = None

    def _native_pull(self):
        i = self._info
        self.k, self.obj_val, self.mu, self.dnorm = i.k, i.obj, i.mu, i.dnorm
        self.inf_pr, self.inf_du, self.inf_compl = i.inf_pr, i.inf_du, i.inf_compl
        self.del_w, self.del_c, self.alpha_p, self.alpha_d = i.del_w, i.del_c, i.alpha_p, i.alpha_d
        self.last_residual_ratio = i.residual_ratio
        self.kkt.n_factorizations = self._n_fact0 + i.n_factorizations
        self.kkt.linear_solver.info = i.factor_info

    # ---- src/kernels.jl:380-417 (host scalars) ----
    def init_regularization(self):
        reg = self.opt.regularization
        self.del_w = 1.0
        self.del_c = 0.0 if isinstance(reg, NoRegularization) else reg.delta_d

    def update_regularization(self):
        reg = self.opt.regularization
        if isinstance(reg, NoRegularization):
            self.del_w, self.del_c = 0.0, 0.0
        elif isinstance(reg, FixedRegularization):
            self.del_w, self.del_c = reg.delta_p, reg.delta_d
        else:
            reg.delta_p = max(reg.delta_p / 10.0, reg.delta_min)
            reg.delta_d = min(reg.delta_d / 10.0, -reg.delta_min)
            self.del_w, self.del_c = reg.delta_p, reg.delta_d

    # ---- src/linear_solver.jl ----
    def factorize_regularized_system(self):  # :6-17
        for _ in range(3):
            self.kkt.set_aug_diagonal_reg(self.del_w, self.del_c)
            self.kkt.factorize_wrapper()
            if self.kkt.linear_solver.is_factorized():
                break
            self.del_w *= 100.0
            self.del_c *= 100.0

    def solve_system(self):  # :19-45
        st, be = self.st, self.be
        be.copy(st.p, st.d)
        self.kkt.solve(st.d)
        be.copy(st.p, st.w1)
        getattr(self.kkt, "mul_solved", self.kkt.mul)(st.w1, st.d, -1.0, 1.0)  # d is the solve's result, untouched
        for _ in range(self.opt.refine_steps):  # extension, off by default: d += K^-1 (p - K d)
            self.kkt.solve(st.w1)
            be.axpy(1.0, st.w1, st.d)
            be.copy(st.p, st.w1)
            self.kkt.mul(st.w1, st.d, -1.0, 1.0)
        norm_w, norm_p, norm_d = be.norm_inf3(st.w1, st.p, st.d)
        ratio = norm_w / max(1.0, norm_p)
        self.last_residual_ratio = ratio
        if math.isnan(ratio) or (self.opt.check_residual and ratio > self.opt.tol_linear_solve):
            raise SolveException()

    # ---- src/kernels.jl step rules (host scalars around one fused reduction) ----
    def get_fraction_to_boundary_step(self, tau):  # kernels.jl:290-305
        a, _ = self.be.get_alpha_max(self.st, tau)
        return min(a[0], a[1]), min(a[2], a[3])

    def update_barrier(self, mu_affine):  # kernels.jl:226-236 (field-order quirk: SURVEY 0c)
        has_inequalities = (self.nlb + self.nub) > 0
        mu_curr = self.be.get_complementarity_measure(self.st)
        if has_inequalities:
            t = _fdiv(mu_affine, mu_curr)
            sigma = min(max(t * t * t, 1e-6), 10.0)  # t^3 as Julia's literal power forms it (and csrc/mpc.hip, mpc_mu_kernel)
        else:
            sigma = 1.0
        self.mu = max(self.opt.mu_min, sigma * mu_curr)
        return mu_curr

    def update_step(self):  # kernels.jl:307-374
        rule = self.opt.step_rule
        if isinstance(rule, ConservativeStep):
            self.alpha_p, self.alpha_d = self.get_fraction_to_boundary_step(rule.tau)
        elif isinstance(rule, AdaptiveStep):
            tau = max(1 - self.mu, rule.tau_min)
            self.alpha_p, self.alpha_d = self.get_fraction_to_boundary_step(tau)
        elif isinstance(rule, MehrotraAdaptiveStep):
            self._mehrotra_adaptive_step(rule)
        else:
            raise TypeError(rule)

    def _mehrotra_adaptive_step(self, rule):  # kernels.jl:325-374
        """The reference reads ~20 single elements at the blocking indices (`CUDA.@allowscalar`, :349-369).  Here the
        (at most four) blocking rows are gathered on the device and cross the bus in ONE read-back."""
        st = self.st
        gamma_a = 1.0 / (1.0 - rule.gamma_f)
        (axl, axu, azl, azu), (i_xl, i_xu, i_zl, i_zu) = self.be.get_alpha_max(st, 1.0)
        max_ap, max_ad = min(axl, axu), min(azl, azu)
        mu_full = self.be.get_affine_complementarity_measure(st, max_ap, max_ad) / gamma_a
        dx, dzl, dzu = st.primal(st.d), st.dual_lb(st.d), st.dual_ub(st.d)
        # one row per blocking index: (x, bound, z, dx, dz) at that index
        picks = []  # (key, list index i, lower?)
        if max_ap < 1.0:
            picks.append(("p", i_xl, True) if axl <= axu else ("p", i_xu, False))
        if max_ad < 1.0:
            picks.append(("d", i_zl, True) if azl <= azu else ("d", i_zu, False))
        vals = {}
        if picks:
            rows = []
            for _, i, lower in picks:
                j = (st.ind_lb if lower else st.ind_ub)[i]
                rows.append(torch.stack([st.x[j], (st.xl if lower else st.xu)[j], (st.zl if lower else st.zu)[j], dx[j],
                                         (dzl if lower else dzu)[i]]))
            host = torch.stack(rows).cpu().numpy()  # the single device-to-host copy
            vals = {k: host[r] for r, (k, _, _) in enumerate(picks)}
        alpha_p = alpha_d = 1.0
        if max_ap < 1.0:
            x, bnd, z, dxi, dzi = vals["p"]
            if axl <= axu:
                tmp = _fdiv(mu_full, z + max_ad * dzi)
                alpha_p = _fdiv(x - bnd - tmp, -dxi)
            else:
                tmp = _fdiv(mu_full, z + max_ad * dzi)
                alpha_p = _fdiv(bnd - x - tmp, dxi)
        if max_ad < 1.0:
            x, bnd, z, dxi, dzi = vals["d"]
            if azl <= azu:
                tmp = _fdiv(mu_full, x + max_ap * dxi - bnd)
                alpha_d = _fdiv(-(z - tmp), dzi)
            else:
                tmp = _fdiv(mu_full, bnd - x - max_ap * dxi)
                alpha_d = _fdiv(-(z - tmp), dzi)
        self.alpha_p = max(alpha_p, rule.gamma_f * max_ap)
        self.alpha_d = max(alpha_d, rule.gamma_f * max_ad)

    # ---- model callbacks ----
    def eval_model(self):
        """eval_f / eval_cons / eval_grad_f wrappers (src/solver.jl:166-169, 338-340)."""
        self.obj_val = self.kkt.eval_model(self.q, self.st.rhs, self.obj_scale * self.qp.c0)

    # ---- src/solver.jl ----
    def init_starting_point(self):  # :6-125
        st, be, kkt = self.st, self.be, self.kkt
        be.fill(self.del_w, st.reg)  # :16-18
        be.fill(self.del_w, st.pr_diag)
        be.fill(self.del_c, st.du_diag)
        kkt.factorize_wrapper()  # :21
        be.set_initial_primal_rhs(st)  # :25
        self.solve_system()
        be.axpy(1.0, st.primal(st.d), st.x)  # :28
        be.set_initial_dual_rhs(st)  # :31
        self.solve_system()
        be.copy(st.dual(st.d), st.y)  # :33
        kkt.jtprod(st.jacl, st.y)  # :37 (jacl is the reference's scratch `res`)
        be.axpy(1.0, st.f, st.jacl)  # :39
        be.sp_init_duals(st)  # :41-66
        m = be.sp_mins(st)  # :68-78
        delta_x = max(0.0, -1.5 * m[0], -1.5 * m[1])
        delta_s = max(0.0, -1.5 * m[2], -1.5 * m[3])
        be.sp_shift(st, delta_x, 1.0 + delta_s)  # :80-83
        s = be.sp_sums(st)  # :85-94
        mu = 0.0
        if self.nlb > 0:
            mu += s[0] - s[1]
        if self.nub > 0:
            mu += s[2] - s[3]
        # (IEEE division as Julia's: with no bound at all both are 0 / 0 = NaN, added to EMPTY views at :96-99 -- the
        # reference goes on; Python's float division would raise)
        delta_x2 = _fdiv(mu, 2 * (s[4] + s[5]))
        delta_s2 = _fdiv(mu, 2 * (s[6] + s[7]))
        be.sp_shift(st, delta_x2, delta_s2)  # :96-99
        be.sp_project(st, self.opt.bound_fac)  # :101-118
        if not be.sp_check(st):  # :120-123
            raise AssertionError("starting point is not strictly interior")

    def _create_kkt_system(self):
        """``MadNLP.create_kkt_system(opt.kkt_system, cb, ind_cons, opt.linear_solver)`` (src/structure.jl:115-121) on
        the scaled model data; a seam for tests that swap the plugin types."""
        opt, be, st, nx = self.opt, self.be, self.st, self.nx
        if hasattr(self.qp, "grid"):  # one QP over a P x Q grid of ranks: madqp_dkkt_*
            from .dist2d import HIPDistributedCondensedKKTSystem2D

            if opt.kkt_system != "condensed":
                raise ValueError("the P x Q distributed path factorises the condensed KKT system")
            return HIPDistributedCondensedKKTSystem2D(be, st, nx, self.ind_ineq, self.qp.grid, self.H, self.A_I, self.A_J)
        if isinstance(self.A, DeviceCSR):  # sparse front end: dense K / Cholesky, CSR products
            cls = {"normal": HIPSparseNormalKKTSystem, "augmented": HIPSparseAugmentedKKTSystem,
                   "condensed": HIPSparseCondensedKKTSystem}[opt.kkt_system]
            return cls(be, st, nx, self.ind_ineq, self.H, self.A)
        if opt.kkt_system == "augmented":
            return HIPAugmentedKKTSystem(be, st, nx, self.ind_ineq, self.H, self.A)
        if opt.kkt_system == "scaled_augmented":
            return HIPScaledAugmentedKKTSystem(be, st, nx, self.ind_ineq, self.H, self.A)
        if opt.kkt_system == "normal":
            self.At = self.A.t().contiguous()  # (nx, m): the layout the normal-equations GEMM consumes
            return HIPNormalKKTSystem(be, st, nx, self.ind_ineq, self.H, self.At)
        return HIPCondensedKKTSystem(be, st, nx, self.ind_ineq, self.H, self.A)

    def initialize(self):  # :127-182
        qp, opt, st, be = self.qp, self.opt, self.st, self.be
        nx, dev = self.nx, st.device
        ineq = torch.as_tensor(self.ind_ineq, dtype=torch.int64, device=dev)
        # MadNLP.initialize!(cb, ...) (:131-142)
        st.x[:nx] = qp.x0
        st.x[nx:] = 0.0
        st.y.copy_(qp.y0)
        st.xl[:nx], st.xu[:nx] = qp.lvar, qp.uvar
        st.xl[nx:], st.xu[nx:] = qp.lcon[ineq], qp.ucon[ineq]
        st.rhs.copy_(torch.where(qp.lcon == qp.ucon, qp.lcon, torch.zeros_like(qp.lcon)))
        tol = opt.bound_relax_factor
        one = torch.ones((), dtype=torch.float64, device=dev)
        st.xl.copy_(torch.where(torch.isfinite(st.xl), st.xl - torch.maximum(one, st.xl.abs()) * tol, st.xl))
        st.xu.copy_(torch.where(torch.isfinite(st.xu), st.xu + torch.maximum(one, st.xu.abs()) * tol, st.xu))
        st.x.copy_(_push_interior(st.x, st.xl, st.xu, opt.bound_push, opt.bound_fac))
        be.fill(0.0, st.jacl)  # :144
        self.H, self.A, self.q = qp.H, qp.A, qp.q
        spread = hasattr(qp, "grid")  # dist2d.DistributedQP: the matrices live on a P x Q grid of ranks
        if spread:
            self.A_I, self.A_J = qp.A_I, qp.A_J
        if opt.scaling and (self.m or nx):  # MadNLP.set_scaling!(..., 100) (:148-159)
            con_scale = torch.ones(self.m, dtype=torch.float64, device=dev)
            sparse = isinstance(qp.A, DeviceCSR)
            if self.m and nx:
                rowmax = qp.row_absmax() if spread else (
                    qp.A.row_absmax() if sparse else torch.linalg.vector_norm(qp.A, ord=float("inf"), dim=1))
                con_scale = torch.minimum(one, 100.0 / rowmax)
            g = st.f[:nx]  # scratch: gradient at the pushed start
            g.copy_(qp.q)
            if qp.H is not None and nx:
                if spread:
                    g.add_(qp.hess_times(st.x[:nx]))
                elif qp.H.dim() == 1:
                    g.add_(qp.H * st.x[:nx])
                else:
                    be.gemv(0, nx, nx, 1.0, qp.H, nx, st.x, 1.0, g)
            gmax = be.norm_inf(g) if nx else 0.0
            self.obj_scale = min(1.0, 100.0 / gmax) if gmax > 0 else 1.0
            self.con_scale = con_scale
            if self.m and bool((con_scale != 1.0).any()):
                cs = con_scale[ineq]
                st.y.div_(con_scale)
                st.rhs.mul_(con_scale)
                st.x[nx:] *= cs
                st.xl[nx:] *= cs
                st.xu[nx:] *= cs
                if spread:
                    self.A_I, self.A_J = qp.A_I.clone(), qp.A_J.clone()
                    self.A_I[: self.m] *= con_scale[:, None]
                    self.A_J[: self.m] *= con_scale[:, None]
                else:
                    self.A = qp.A.scaled(con_scale) if sparse else (con_scale[:, None] * qp.A).contiguous()
            if self.obj_scale != 1.0:
                self.H = None if qp.H is None else (self.obj_scale * qp.H).contiguous()
                self.q = self.obj_scale * qp.q
        if self.kkt is not None:
            self._fact_closed += self.kkt.n_factorizations
            self.kkt.close()
            self.kkt = None
        self.kkt = self._create_kkt_system()
        self.kkt.initialize()  # :162
        self.init_regularization()  # :163
        self.eval_model()  # :166-169
        self.norm_b = be.norm_inf(st.rhs)  # :173
        self.norm_c = be.norm_inf(st.f)  # :174
        self.init_starting_point()  # :177
        self.mu = opt.mu_init  # :179
        self.k = 0
        self.trace = []
        if opt.driver == "native":
            self._n_fact0 = self.kkt.n_factorizations
            self._native_open()

    def affine_direction(self):  # :188-192
        self.be.set_predictive_rhs(self.st)
        self.solve_system()

    def mehrotra_correction_direction(self):  # :194-198
        self.be.set_correction_rhs(self.st, self.mu)
        self.solve_system()

    def gondzio_correction_direction(self, mu_curr, max_ncorr):  # :200-251
        st, be = self.st, self.be
        delta, bmin, bmax, tau = 0.1, 0.1, 10.0, 0.995
        alpha_p, alpha_d = self.get_fraction_to_boundary_step(tau)
        for _ in range(max_ncorr):
            ta_p, ta_d = min(alpha_p + delta, 1.0), min(alpha_d + delta, 1.0)
            ga = be.get_affine_complementarity_measure(st, ta_p, ta_d)
            mu = (ga / mu_curr) ** 2 * ga
            be.set_extra_correction(st, ta_p, ta_d, bmin, bmax, mu)
            be.set_correction_rhs(st, mu)
            be.copy(st.d, st.w2)
            self.solve_system()
            ha_p, ha_d = self.get_fraction_to_boundary_step(tau)
            if ha_p < 1.005 * alpha_p or ha_d < 1.005 * alpha_d:
                be.copy(st.w2, st.d)
                break
            alpha_p, alpha_d = ha_p, ha_d
        return alpha_p, alpha_d

    def record(self):
        self.trace.append(dict(
            k=self.k, obj=self.obj_val / self.obj_scale, inf_pr=self.inf_pr, inf_du=self.inf_du,
            inf_compl=self.inf_compl, mu=self.mu, dnorm=0.0 if self.k == 0 else self.dnorm,
            del_w=self.del_w, alpha_d=self.alpha_d, alpha_p=self.alpha_p))
        if self.opt.print_level > 0:
            t = self.trace[-1]
            print("%4d % .7e %6.2e %6.2e %5.1f %6.2e %6.2e %6.2e" % (
                t["k"], t["obj"], t["inf_pr"], t["inf_du"], math.log10(t["mu"]) if t["mu"] > 0 else 0,
                t["dnorm"], t["alpha_d"], t["alpha_p"]), flush=True)

    def iteration_head(self):
        """src/solver.jl:259-283: residuals and the termination test."""
        if self._native is not None:
            status = self.be.mpc_head(self._native, self._info)
            self._native_pull()
            self.record()
            return status or None
        st = self.st
        self.kkt.jtprod(st.jacl, st.y)  # :259
        nc, nd, ncompl = self.be.get_inf(st)
        self.inf_pr = nc / max(1.0, self.norm_b)  # :264
        self.inf_du = nd / max(1.0, self.norm_c)  # :265-271
        self.inf_compl = ncompl / max(1.0, self.norm_c)  # :272
        self.record()
        if max(self.inf_pr, self.inf_du, self.inf_compl) <= self.opt.tol:  # :279
            return SOLVE_SUCCEEDED
        if self.k >= self.opt.max_iter:
            return MAXIMUM_ITERATIONS_EXCEEDED
        return None

    def iteration_body(self):
        """src/solver.jl:288-343: one predictor-corrector step (the timed unit of bench.py)."""
        if self._native is not None:
            rc = self.be.mpc_body(self._native, self._info)
            if rc != 0:
                raise SolveException()
            self._native_pull()
            return
        st, be = self.st, self.be
        self.update_regularization()  # :288
        self.factorize_regularized_system()  # :289
        self.affine_direction()  # :294
        a_aff_p, a_aff_d = self.get_fraction_to_boundary_step(1.0)  # :295
        mu_affine = be.get_affine_complementarity_measure(st, a_aff_p, a_aff_d)  # :296
        be.get_correction(st)  # :297
        mu_curr = self.update_barrier(mu_affine)  # :302
        self.mehrotra_correction_direction()  # :307
        if self.opt.max_ncorr > 0:  # :316-324
            self.gondzio_correction_direction(mu_curr, self.opt.max_ncorr)
        self.update_step()  # :329
        self.dnorm = be.norm_inf(st.primal(st.d))  # print_iter, src/structure.jl:190
        be.update_iterates(st, self.alpha_p, self.alpha_d)  # :332-335
        self.eval_model()  # :338-340
        be.adjust_boundary(st, self.mu)  # :342
        self.k += 1

    def mpc(self):  # :254-345
        while True:
            status = self.iteration_head()
            if status is not None:
                return status
            self.iteration_body()

    def solve(self):  # :347-403
        import time

        t0 = time.perf_counter()
        try:
            self.initialize()
            self.status = self.mpc()
        except SolveException:
            self.status = ERROR_IN_STEP_COMPUTATION
            if self.opt.rethrow_error:
                raise
        except AssertionError:
            self.status = INTERNAL_ERROR
            if self.opt.rethrow_error:
                raise
        finally:
            self.total_time = time.perf_counter() - t0  # counters.total_time (scripts/benchmarks_cpu.jl:54)
        return self.result()

    def close(self):
        """Release the library objects (native driver, KKT system)."""
        self._native_close()
        if self.kkt is not None:
            self.kkt.close()

    def result(self):
        """MadNLP.update!(stats, solver): unscaled solution / objective / multipliers."""
        st, nx = self.st, self.nx
        h = lambda t: t.detach().cpu().numpy().copy()
        cs = h(self.con_scale) if self.con_scale is not None else np.ones(self.m)
        x = h(st.x[:nx])
        # stats.constraints = A x of the unscaled model, from c = (A x - s - rhs) of the scaled one
        cons = h(st.c) + h(st.rhs)
        cons[self.ind_ineq] += h(st.x[nx:])
        cons, y = cons / cs, h(st.y) * cs / self.obj_scale
        zl, zu = h(st.zl[:nx]) / self.obj_scale, h(st.zu[:nx]) / self.obj_scale
        if self._fixed is not None:  # MakeParameter: put the parameters back; their multipliers = reduced costs
            _, free, fixed, xf, shift = self._fixed
            free, fixed, fq = h(free), h(fixed), self.full_qp
            n_full = fq.nvar
            xs, zls, zus = np.zeros(n_full), np.zeros(n_full), np.zeros(n_full)
            xs[free], xs[fixed], zls[free], zus[free] = x, h(xf), zl, zu
            xd, yd = torch.as_tensor(xs, device=st.y.device), torch.as_tensor(y, device=st.y.device)
            g = fq.q.clone()
            if fq.H is not None:
                g += fq.H * xd if fq.H.dim() == 1 else fq.H @ xd
            if isinstance(fq.A, DeviceCSR):
                g.index_add_(0, fq.A.col, fq.A.val * yd[fq.A.row])
            elif fq.ncon:
                g += fq.A.t() @ yd
            r = h(g)[fixed]
            zls[fixed], zus[fixed] = np.maximum(r, 0.0), np.maximum(-r, 0.0)
            x, zl, zu, cons = xs, zls, zus, cons + h(shift)
        return dict(
            status=self.status, iter=self.k, objective=self.obj_val / self.obj_scale, solution=x,
            constraints=cons,
            multipliers=y,
            multipliers_L=zl,
            multipliers_U=zu,
            trace=self.trace, n_factorizations=self.kkt.n_factorizations if self.kkt else 0,
            total_time=getattr(self, "total_time", 0.0),
            primal_feas=self.inf_pr, dual_feas=self.inf_du,  # stats.primal_feas / dual_feas of MadNLP.update!
        )


def solve(qp: DeviceQP, backend=None, **opts):
    """Convenience (no counterpart in the reference, whose callers write ``MPCSolver(qp; ...)`` then ``solve!``,
    test/runtests.jl:13-14): construct, ``solve!`` (src/solver.jl:347-403), release; returns the result
    dictionary of :meth:`MPCSolver.result`."""
    from .backend import HipBackend

    own = backend is None
    be = HipBackend(qp.q.device.index or 0) if own else backend
    solver = MPCSolver(qp, be, **opts)
    try:
        return solver.solve()
    finally:
        solver.close()
        if own:
            be.close()
