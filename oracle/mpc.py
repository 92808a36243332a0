"""Numpy restatement of the reference's Mehrotra predictor-corrector path.

Oracle (test infrastructure) -- see ``oracle/__init__.py``.  PARITY UNPINNED
(no reference golden vectors exist); pinned by known answers only.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  Items marked *MadNLP-recall* restate the published
behaviour of the un-vendored MadNLP.jl 0.8.x (``Project.toml:17``) at the
reference call site named next to them.  All index vectors are 0-based.

Three dense KKT formulations are provided, mirroring the cross-formulation
tests of the reference (``test/runtests.jl:102-115,165-180``):

``K2``         augmented system ``[H+S, A'; A, dc I]`` (MadNLP's default
               ``SparseKKTSystem``; symmetric indefinite solve)
``K2.5``       the same system symmetrically scaled (``ScaledSparseKKTSystem``,
               src/kernels.jl:149-165, test/runtests.jl:95-115)
``normal``     ``A S^-1 A'`` -- ``src/KKT/normalkkt.jl`` verbatim (LP only)
``condensed``  ``H + S_x + A_x' Th A_x`` -- the generalisation the HIP path
               implements; Cholesky
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

from .qp import DenseQP

EPS = np.finfo(np.float64).eps

SOLVE_SUCCEEDED = 1
MAXIMUM_ITERATIONS_EXCEEDED = 6
ERROR_IN_STEP_COMPUTATION = -3
INTERNAL_ERROR = -1


class SolveException(Exception):
    """MadNLP.SolveException raised at src/linear_solver.jl:41-43."""


# --------------------------------------------------------------------------
# options -- src/utils.jl:17-48, 69-103
# --------------------------------------------------------------------------
class ConservativeStep:  # src/utils.jl:19-21
    def __init__(self, tau=0.995):
        self.tau = tau


class AdaptiveStep:  # src/utils.jl:23-25
    def __init__(self, tau_min=0.99):
        self.tau_min = tau_min


class MehrotraAdaptiveStep:  # src/utils.jl:27-29
    def __init__(self, gamma_f=0.99):
        self.gamma_f = gamma_f


class NoRegularization:  # src/utils.jl:37
    pass


class FixedRegularization:  # src/utils.jl:39-42
    def __init__(self, delta_p, delta_d):
        self.delta_p, self.delta_d = delta_p, delta_d


class AdaptiveRegularization:  # src/utils.jl:44-48
    def __init__(self, delta_p, delta_d, delta_min):
        self.delta_p, self.delta_d, self.delta_min = delta_p, delta_d, delta_min


class IPMOptions:
    """Defaults of src/utils.jl:69-103 (tol from the preset at :110)."""

    def __init__(self, **kw):
        self.tol = 1e-8
        self.max_iter = 3000
        self.scaling = True
        self.bound_push = 1e-2
        self.bound_fac = 1e-2
        self.bound_relax_factor = 1e-8
        self.regularization = FixedRegularization(1e-8, 0.0)
        self.step_rule = AdaptiveStep(0.99)
        self.max_ncorr = 0
        self.mu_init = 1e-1
        self.mu_min = 1e-11
        self.tol_linear_solve = 1e-8
        self.check_residual = False
        self.kkt_system = "K2"
        self.fixed_variable_treatment = "error"  # or "relax_bound" (MadNLP.RelaxBound)
        # NOT in the reference (default 0 = its solve_system!): steps of iterative refinement d += K^-1 (p - K d) with
        # the residual src/linear_solver.jl:29-31 forms anyway.  Mirrors the product's `refine_steps` extension, and
        # gives the tests a second, equally valid CPU run of every problem: the distance between the two runs is what
        # the conditioning of the problem lets any two accurate implementations agree to (tests/parity.py).
        self.refine_steps = 0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError(f"unknown option {k}")
            setattr(self, k, v)


# --------------------------------------------------------------------------
# MadNLP-recall pieces
# --------------------------------------------------------------------------
def get_index_constraints(lvar, uvar, lcon, ucon, fixed_variable_treatment="error"):
    """MadNLP.get_index_constraints (called src/structure.jl:95-102).

    EnforceEquality (src/utils.jl:82).  Fixed variables: "relax_bound" = MadNLP.RelaxBound (MadNLP-recall: they
    stay variables with both bounds, relaxed by bound_relax_factor in initialize like every bound -- the
    treatment src/utils.jl:81 selects for condensed KKT systems); MakeParameter (elimination) is out of the
    oracle's scope.
    """
    ind_eq = np.flatnonzero(lcon == ucon)
    ind_ineq = np.flatnonzero(lcon != ucon)
    xl = np.concatenate([lvar, lcon[ind_ineq]])
    xu = np.concatenate([uvar, ucon[ind_ineq]])
    if np.any(xl == xu) and fixed_variable_treatment != "relax_bound":
        raise NotImplementedError("fixed variables (MakeParameter) are out of the oracle's scope")
    ind_lb = np.flatnonzero(xl != -np.inf)
    ind_ub = np.flatnonzero(xu != np.inf)
    return dict(ind_eq=ind_eq, ind_ineq=ind_ineq, ind_lb=ind_lb, ind_ub=ind_ub)


def _initialize_variable(x, l, u, bound_push, bound_fac):
    """MadNLP._initialize_variables! (MadNLP-recall; via src/solver.jl:131-142)."""
    out = x.copy()
    both = np.isfinite(l) & np.isfinite(u)
    lo = np.isfinite(l) & ~np.isfinite(u)
    up = ~np.isfinite(l) & np.isfinite(u)
    with np.errstate(invalid="ignore"):
        pl = np.minimum(bound_push * np.maximum(1.0, np.abs(l)), bound_fac * (u - l))
        pu = np.minimum(bound_push * np.maximum(1.0, np.abs(u)), bound_fac * (u - l))
        out[both] = np.minimum(u - pu, np.maximum(l + pl, x))[both]
        out[lo] = np.maximum(l + bound_push * np.maximum(1.0, np.abs(l)), x)[lo]
        out[up] = np.minimum(u - bound_push * np.maximum(1.0, np.abs(u)), x)[up]
    return out


class KKTVec:
    """MadNLP.UnreducedKKTVector: one contiguous ``values`` = [x(n) | y(m) | zl(nlb) | zu(nub)]."""

    def __init__(self, n, m, nlb, nub, ind_lb, ind_ub):
        self.n, self.m, self.nlb, self.nub = n, m, nlb, nub
        self.ind_lb, self.ind_ub = ind_lb, ind_ub
        self.values = np.zeros(n + m + nlb + nub)

    @property
    def xp(self):
        return self.values[: self.n]

    @property
    def y(self):
        return self.values[self.n : self.n + self.m]

    @property
    def zl(self):
        return self.values[self.n + self.m : self.n + self.m + self.nlb]

    @property
    def zu(self):
        return self.values[self.n + self.m + self.nlb :]


def reduce_rhs(w: KKTVec, l_diag, u_diag):
    """MadNLP.reduce_rhs! (called src/KKT/normalkkt.jl:183)."""
    w.xp[w.ind_lb] -= w.zl / l_diag
    w.xp[w.ind_ub] -= w.zu / u_diag


def finish_aug_solve(w: KKTVec, l_lower, u_lower, l_diag, u_diag):
    """MadNLP.finish_aug_solve! (called src/KKT/normalkkt.jl:203)."""
    w.zl[:] = (-w.zl + l_lower * w.xp[w.ind_lb]) / l_diag
    w.zu[:] = (w.zu - u_lower * w.xp[w.ind_ub]) / u_diag


def kktmul(w: KKTVec, v: KKTVec, reg, du_diag, l_lower, u_lower, l_diag, u_diag, alpha, beta):
    """MadNLP._kktmul! (called src/KKT/normalkkt.jl:217)."""
    w.xp[:] += alpha * reg * v.xp
    w.y[:] += alpha * du_diag * v.y
    w.xp[w.ind_lb] -= alpha * v.zl
    w.xp[w.ind_ub] += alpha * v.zu
    w.zl[:] = beta * w.zl + alpha * (v.xp[v.ind_lb] * l_lower - v.zl * l_diag)
    w.zu[:] = beta * w.zu + alpha * (v.xp[v.ind_ub] * u_lower + v.zu * u_diag)


def adjust_boundary(x, xl, xu, ind_lb, ind_ub, mu):
    """MadNLP.adjust_boundary! (called src/solver.jl:342)."""
    c1 = EPS * mu
    c2 = EPS ** 0.75
    x_lr, xl_r = x[ind_lb], xl[ind_lb]
    xl[ind_lb] = np.where(x_lr - xl_r < c1, xl_r - c2 * np.maximum(1.0, np.abs(x_lr)), xl_r)
    x_ur, xu_r = x[ind_ub], xu[ind_ub]
    xu[ind_ub] = np.where(xu_r - x_ur < c1, xu_r + c2 * np.maximum(1.0, np.abs(x_ur)), xu_r)


# --------------------------------------------------------------------------
# dense KKT systems
# --------------------------------------------------------------------------
class DenseKKT:
    """Fields read generically by MadIPM: src/kernels.jl:135-144, src/solver.jl:16-18."""

    def __init__(self, solver):
        s = self.s = solver
        n, m = s.n, s.m
        self.reg = np.zeros(n)
        self.pr_diag = np.zeros(n)
        self.du_diag = np.zeros(m)
        self.l_diag = np.zeros(s.nlb)
        self.u_diag = np.zeros(s.nub)
        self.l_lower = np.zeros(s.nlb)
        self.u_lower = np.zeros(s.nub)
        self.factorized = False
        self.n_factorizations = 0

    def initialize(self):  # src/KKT/normalkkt.jl:136-147
        self.reg[:] = 1.0
        self.pr_diag[:] = 1.0
        self.du_diag[:] = 0.0
        self.l_lower[:] = 0.0
        self.u_lower[:] = 0.0
        self.l_diag[:] = 1.0
        self.u_diag[:] = 1.0

    def jtprod(self, y):  # src/KKT/normalkkt.jl:162-164
        return self.s.A_full.T @ y

    def is_factorized(self):  # src/utils.jl:54-62
        return self.factorized

    def mul(self, w: KKTVec, v: KKTVec, alpha=1.0, beta=0.0):
        """src/KKT/normalkkt.jl:207-219 (+ H for a QP, as MadNLP's SparseKKTSystem mul!)."""
        s = self.s
        w.xp[:] = alpha * (s.A_full.T @ v.y) + beta * w.xp
        w.xp[: s.nx] += alpha * (s.H @ v.xp[: s.nx])
        w.y[:] = alpha * (s.A_full @ v.xp) + beta * w.y
        kktmul(w, v, self.reg, self.du_diag, self.l_lower, self.u_lower, self.l_diag, self.u_diag,
               alpha, beta)
        return w


class K2KKT(DenseKKT):
    """Augmented system; MadNLP SparseKKTSystem semantics (default, src/utils.jl:108)."""

    def build_and_factorize(self):
        s = self.s
        n, m = s.n, s.m
        K = np.zeros((n + m, n + m))
        K[: s.nx, : s.nx] = s.H
        K[np.arange(n), np.arange(n)] += self.pr_diag
        K[n:, :n] = s.A_full
        K[:n, n:] = s.A_full.T
        K[np.arange(n, n + m), np.arange(n, n + m)] = self.du_diag
        self.K = K
        self.n_factorizations += 1
        try:
            self.lu = sla.lu_factor(K)
            self.factorized = bool(np.all(np.isfinite(self.lu[0])))
        except Exception:
            self.factorized = False

    def solve(self, w: KKTVec):
        s = self.s
        reduce_rhs(w, self.l_diag, self.u_diag)
        sol = sla.lu_solve(self.lu, w.values[: s.n + s.m])
        w.values[: s.n + s.m] = sol
        finish_aug_solve(w, self.l_lower, self.u_lower, self.l_diag, self.u_diag)
        return w


class K25KKT(DenseKKT):
    """K2.5: MadNLP's ``ScaledSparseKKTSystem`` as MadIPM drives it (src/kernels.jl:149-165, the scaling kernel of
    scripts/cuda_wrapper.jl:90-116; equality of K2.5 and K2 results: test/runtests.jl:95-115).

    *MadNLP-recall* for ``_set_aug_diagonal!`` / ``solve!`` / ``mul!`` of that type, restated from the algebra they
    implement.  Sign convention of src/kernels.jl:157-158: ``l_diag = x - xl > 0``, ``u_diag = xu - x > 0``.  With
    ``a_j = l_diag`` (1 without a lower bound), ``b_j = u_diag`` (1 without an upper bound):

        scaling_j = sqrt(a_j b_j),      pr_diag_j = zl_j b_j + zu_j a_j + del_w a_j b_j   ( = scaling^2 (del_w + Sigma) )

    and the matrix that is factorised is the symmetric scaling of the K2 matrix, entry by entry as the COO kernel does
    it (:96-110): pr_diag as it stands, Hessian entries times scaling_i scaling_j, Jacobian entries times scaling_j,
    du_diag as it stands.  Its entries stay bounded as the iterates converge (Sigma itself blows up on active bounds).
    Unreduced system (the rows of mul!):  zl dx_lr + l_diag dzl = p_zl,   zu dx_ur - u_diag dzu = p_zu.
    """

    def __init__(self, solver):
        super().__init__(solver)
        self.scaling_factor = np.ones(solver.n)

    def initialize(self):
        super().initialize()
        self.scaling_factor[:] = 1.0

    def set_aug_diagonal(self, zl, zu, del_w):  # MadNLP._set_aug_diagonal!(kkt) (src/kernels.jl:163)
        s = self.s
        a, b = np.ones(s.n), np.ones(s.n)
        a[s.ind_lb] = self.l_diag
        b[s.ind_ub] = self.u_diag
        self.scaling_factor[:] = np.sqrt(a * b)
        self.pr_diag[:] = zl * b + zu * a + del_w * (a * b)

    def build_and_factorize(self):
        s = self.s
        n, m, sf = s.n, s.m, self.scaling_factor
        K = np.zeros((n + m, n + m))
        K[: s.nx, : s.nx] = s.H * sf[: s.nx, None] * sf[None, : s.nx]  # Hessian block: V * scaling[i] * scaling[j]
        K[np.arange(n), np.arange(n)] += self.pr_diag  # primal diagonal: copied
        K[n:, :n] = s.A_full * sf[None, :]  # Jacobian block: V * scaling[j]
        K[:n, n:] = K[n:, :n].T
        K[np.arange(n, n + m), np.arange(n, n + m)] = self.du_diag  # dual regularization: copied
        self.K = K
        self.n_factorizations += 1
        try:
            self.lu = sla.lu_factor(K)
            self.factorized = bool(np.all(np.isfinite(self.lu[0])))
        except Exception:
            self.factorized = False

    def solve(self, w: KKTVec):
        s, sf = self.s, self.scaling_factor
        # reduce: r1 = p_x + p_zl / l_diag + p_zu / u_diag, then the symmetric scaling of the primal block
        w.xp[w.ind_lb] += w.zl / self.l_diag
        w.xp[w.ind_ub] += w.zu / self.u_diag
        rhs = np.concatenate([sf * w.xp, w.y])
        sol = sla.lu_solve(self.lu, rhs)
        w.xp[:] = sf * sol[: s.n]
        w.y[:] = sol[s.n :]
        w.zl[:] = (w.zl - self.l_lower * w.xp[w.ind_lb]) / self.l_diag
        w.zu[:] = (self.u_lower * w.xp[w.ind_ub] - w.zu) / self.u_diag
        return w

    def mul(self, w: KKTVec, v: KKTVec, alpha=1.0, beta=0.0):
        s = self.s
        w.xp[:] = alpha * (s.A_full.T @ v.y) + beta * w.xp
        w.xp[: s.nx] += alpha * (s.H @ v.xp[: s.nx])
        w.y[:] = alpha * (s.A_full @ v.xp) + beta * w.y
        w.xp[:] += alpha * self.reg * v.xp
        w.y[:] += alpha * self.du_diag * v.y
        w.xp[w.ind_lb] -= alpha * v.zl
        w.xp[w.ind_ub] += alpha * v.zu
        w.zl[:] = beta * w.zl + alpha * (v.xp[v.ind_lb] * self.l_lower + v.zl * self.l_diag)
        w.zu[:] = beta * w.zu + alpha * (v.xp[v.ind_ub] * self.u_lower - v.zu * self.u_diag)
        return w


class NormalKKT(DenseKKT):
    """src/KKT/normalkkt.jl verbatim, dense storage; LP only (:45-48)."""

    def __init__(self, solver):
        super().__init__(solver)
        if np.any(solver.H):
            raise ValueError("The KKT system NormalKKTSystem supports only linear programs.")

    def build_and_factorize(self):
        s = self.s
        D = 1.0 / self.pr_diag  # normalkkt.jl:177
        # assemble_normal_system!, src/utils.jl:266-298 (du_diag is NOT added)
        self.S = (s.A_full * D) @ s.A_full.T
        self.n_factorizations += 1
        try:
            self.chol = sla.cho_factor(self.S, lower=True)
            self.factorized = True
        except sla.LinAlgError:
            self.factorized = False

    def solve(self, w: KKTVec):  # normalkkt.jl:182-205
        s = self.s
        reduce_rhs(w, self.l_diag, self.u_diag)
        Sig = self.pr_diag
        wx, wy = w.xp, w.y
        r1 = wx / Sig
        r2 = s.A_full @ r1 - wy
        dy = sla.cho_solve(self.chol, r2)
        wy[:] = dy
        r1 = wx - s.A_full.T @ dy
        wx[:] = r1 / Sig
        finish_aug_solve(w, self.l_lower, self.u_lower, self.l_diag, self.u_diag)
        return w


class CondensedKKT(DenseKKT):
    """``K = H + S_x + A_x' Th A_x``; what the HIP path implements.

    Inequality row i with slack k: ``Th_i = S_s,k / (1 - dc_i S_s,k)``; equality
    row: ``Th_i = -1/dc_i`` (needs dc < 0).
    """

    def theta(self):
        s = self.s
        th = np.empty(s.m)
        Ss = self.pr_diag[s.nx :]
        th[s.ind_ineq] = Ss / (1.0 - self.du_diag[s.ind_ineq] * Ss)
        if len(s.ind_eq):
            if np.any(self.du_diag[s.ind_eq] >= 0.0):
                raise ValueError("condensed KKT needs dual regularization < 0 on equality rows")
            th[s.ind_eq] = -1.0 / self.du_diag[s.ind_eq]
        return th

    def build_and_factorize(self):
        s = self.s
        th = self.th = self.theta()
        Ax = s.A_full[:, : s.nx]
        K = s.H + (Ax.T * th) @ Ax
        K[np.arange(s.nx), np.arange(s.nx)] += self.pr_diag[: s.nx]
        self.K = K
        self.n_factorizations += 1
        try:
            self.chol = sla.cho_factor(K, lower=True)
            self.factorized = True
        except sla.LinAlgError:
            self.factorized = False

    def solve(self, w: KKTVec):
        s = self.s
        nx = s.nx
        reduce_rhs(w, self.l_diag, self.u_diag)
        Sig = self.pr_diag
        Ax = s.A_full[:, :nx]
        r1x, r1s, r2 = w.xp[:nx].copy(), w.xp[nx:].copy(), w.y.copy()
        t = r2.copy()
        t[s.ind_ineq] += r1s / Sig[nx:]
        rhs = r1x + Ax.T @ (self.th * t)
        dx = sla.cho_solve(self.chol, rhs)
        dy = self.th * (Ax @ dx - t)
        ds = (r1s + dy[s.ind_ineq]) / Sig[nx:]
        w.xp[:nx], w.xp[nx:], w.y[:] = dx, ds, dy
        finish_aug_solve(w, self.l_lower, self.u_lower, self.l_diag, self.u_diag)
        return w


_KKT = {"K2": K2KKT, "K2.5": K25KKT, "normal": NormalKKT, "condensed": CondensedKKT}


# --------------------------------------------------------------------------
# solver state -- src/structure.jl:1-176
# --------------------------------------------------------------------------
class MPCSolver:
    def __init__(self, qp: DenseQP, **opts):
        self.qp = qp
        self.opt = IPMOptions(**opts)
        ic = get_index_constraints(qp.lvar, qp.uvar, qp.lcon, qp.ucon, self.opt.fixed_variable_treatment)
        self.ind_ineq, self.ind_eq = ic["ind_ineq"], ic["ind_eq"]
        self.ind_lb, self.ind_ub = ic["ind_lb"], ic["ind_ub"]
        self.nx, self.ns = qp.nvar, len(self.ind_ineq)
        self.n, self.m = self.nx + self.ns, qp.ncon
        self.nlb, self.nub = len(self.ind_lb), len(self.ind_ub)
        n, m = self.n, self.m
        self.x, self.xl, self.xu = np.zeros(n), np.zeros(n), np.zeros(n)
        self.zl, self.zu, self.f = np.zeros(n), np.zeros(n), np.zeros(n)
        self.y, self.c, self.rhs, self.jacl = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(n)
        mk = lambda: KKTVec(n, m, self.nlb, self.nub, self.ind_lb, self.ind_ub)
        self.d, self.p, self._w1, self._w2 = mk(), mk(), mk(), mk()
        self.correction_lb, self.correction_ub = np.zeros(self.nlb), np.zeros(self.nub)
        self.obj_scale, self.con_scale = 1.0, np.ones(m)
        # scaled problem data (set in initialize)
        self.H, self.q = qp.H.copy(), qp.q.copy()
        self.A_full = np.zeros((m, n))
        self.A_full[:, : self.nx] = qp.A
        self.A_full[self.ind_ineq, self.nx + np.arange(self.ns)] = -1.0  # normalkkt.jl:76-77,151
        self.kkt = _KKT[self.opt.kkt_system](self)
        self.obj_val = 0.0
        self.inf_pr = self.inf_du = self.inf_compl = 0.0
        self.norm_b = self.norm_c = 0.0
        self.mu = 0.0
        self.alpha_p = self.alpha_d = 0.0
        self.del_w = self.del_c = 0.0
        self.k = 0
        self.status = None
        self.trace = []

    # gather views of src/structure.jl:144-151
    x_lr = property(lambda s: s.x[s.ind_lb])
    x_ur = property(lambda s: s.x[s.ind_ub])
    xl_r = property(lambda s: s.xl[s.ind_lb])
    xu_r = property(lambda s: s.xu[s.ind_ub])
    zl_r = property(lambda s: s.zl[s.ind_lb])
    zu_r = property(lambda s: s.zu[s.ind_ub])
    dx_lr = property(lambda s: s.d.xp[s.ind_lb])
    dx_ur = property(lambda s: s.d.xp[s.ind_ub])

    # ---- callbacks (MadNLP eval_*_wrapper; src/solver.jl:166-170, 338-340) ----
    def eval_f(self):
        xv = self.x[: self.nx]
        return self.obj_scale * self.qp.c0 + self.q @ xv + 0.5 * (xv @ (self.H @ xv))

    def eval_grad_f(self):
        self.f[: self.nx] = self.H @ self.x[: self.nx] + self.q
        self.f[self.nx :] = 0.0

    def eval_cons(self):
        self.c[:] = self.A_full @ self.x - self.rhs

    # ---- src/kernels.jl ----
    def set_initial_primal_rhs(self):  # kernels.jl:1-9
        self.p.values[:] = 0.0
        self.p.y[:] = -self.c

    def set_initial_dual_rhs(self):  # kernels.jl:11-19
        self.p.values[:] = 0.0
        self.p.xp[:] = -self.f

    def set_predictive_rhs(self):  # kernels.jl:21-41
        p = self.p
        p.values[:] = 0.0
        p.xp[:] = -self.f + self.zl - self.zu - self.jacl
        p.y[:] = -self.c
        p.zl[:] = (self.xl_r - self.x_lr) * self.zl_r
        p.zu[:] = (self.xu_r - self.x_ur) * self.zu_r

    def set_correction_rhs(self, mu):  # kernels.jl:43-61
        p = self.p
        p.xp[:] = -self.f + self.zl - self.zu - self.jacl
        p.y[:] = -self.c
        p.zl[:] = (self.xl_r - self.x_lr) * self.zl_r + mu - self.correction_lb
        p.zu[:] = (self.xu_r - self.x_ur) * self.zu_r - mu - self.correction_ub

    def get_correction(self):  # kernels.jl:63-75
        self.correction_lb[:] = self.dx_lr * self.d.zl
        self.correction_ub[:] = self.dx_ur * self.d.zu

    def set_extra_correction(self, alpha_p, alpha_d, bmin, bmax, mu):  # kernels.jl:78-126
        tmin, tmax = bmin * mu, bmax * mu
        v = (self.x_lr + alpha_p * self.dx_lr - self.xl_r) * (self.zl_r + alpha_d * self.d.zl)
        dl = np.where(v < tmin, tmin - v, np.where(v > tmax, tmax - v, 0.0))
        self.correction_lb[:] = self.correction_lb - dl
        v = (self.xu_r - alpha_p * self.dx_ur - self.x_ur) * (self.zu_r + alpha_d * self.d.zu)
        du = np.where(v < tmin, tmin - v, np.where(v > tmax, tmax - v, 0.0))
        self.correction_ub[:] = self.correction_ub + du

    def set_aug_diagonal_reg(self):  # kernels.jl:128-146; ScaledSparseKKTSystem: :149-165
        k = self.kkt
        k.reg[:] = self.del_w
        k.du_diag[:] = self.del_c
        if isinstance(k, K25KKT):
            k.l_diag[:] = self.x_lr - self.xl_r  # :157 (X - Xl)
            k.u_diag[:] = self.xu_r - self.x_ur  # :158 (Xu - X)
            k.l_lower[:] = self.zl_r
            k.u_lower[:] = self.zu_r
            k.set_aug_diagonal(self.zl, self.zu, self.del_w)  # :163
            return
        k.l_diag[:] = self.xl_r - self.x_lr
        k.u_diag[:] = self.x_ur - self.xu_r
        k.l_lower[:] = self.zl_r
        k.u_lower[:] = self.zu_r
        k.pr_diag[:] = k.reg
        k.pr_diag[self.ind_lb] -= k.l_lower / k.l_diag
        k.pr_diag[self.ind_ub] -= k.u_lower / k.u_diag

    def get_complementarity_measure(self):  # kernels.jl:171-190
        if self.nlb + self.nub == 0:
            return 0.0
        l = np.sum((self.x_lr - self.xl_r) * self.zl_r)
        u = np.sum((self.xu_r - self.x_ur) * self.zu_r)
        return (l + u) / (self.nlb + self.nub)

    def get_affine_complementarity_measure(self, alpha_p, alpha_d):  # kernels.jl:192-224
        if self.nlb + self.nub == 0:
            return 0.0
        l = np.sum(((self.x_lr + alpha_p * self.dx_lr) - self.xl_r) * (self.zl_r + alpha_d * self.d.zl))
        u = np.sum((self.xu_r - (self.x_ur + alpha_p * self.dx_ur)) * (self.zu_r + alpha_d * self.d.zu))
        return (l + u) / (self.nlb + self.nub)

    def update_barrier(self, mu_affine):  # kernels.jl:226-236 (+ field-order quirk)
        has_inequalities = (self.nlb + self.nub) > 0
        mu_curr = self.get_complementarity_measure()
        if has_inequalities:
            t = mu_affine / mu_curr
            sigma = min(max(t * t * t, 1e-6), 10.0)  # (mu_affine / mu_curr)^3: Julia lowers a literal power of 3 to x*x*x
        else:
            sigma = 1.0
        self.mu = max(self.opt.mu_min, sigma * mu_curr)
        return mu_curr

    @staticmethod
    def _argmin_last(val):
        """mapreduce with the reducer ``elem1[1] < elem2[1] ? elem1 : elem2`` and init (1.0, 0), kernels.jl:243-251:
        folded from the left the RIGHT element survives every exact tie, so the blocking index is the LAST index that
        attains the minimum; (1.0, "nothing blocks") unless the minimum is below 1 (the index is only ever read when
        it is, kernels.jl:351-368)."""
        if val.size == 0:
            return 1.0, -1
        i = val.size - 1 - int(np.argmin(val[::-1]))  # last minimum
        return (float(val[i]), i) if val[i] < 1.0 else (1.0, -1)

    def get_alpha_max_primal(self, tau):  # kernels.jl:242-264
        with np.errstate(divide="ignore", invalid="ignore"):
            dxl, dxu = self.dx_lr, self.dx_ur
            vl = np.where(dxl < 0, (-self.x_lr + self.xl_r) * tau / dxl, np.inf)
            vu = np.where(dxu > 0, (-self.x_ur + self.xu_r) * tau / dxu, np.inf)
        (al, il), (au, iu) = self._argmin_last(vl), self._argmin_last(vu)
        return al, au, il, iu

    def get_alpha_max_dual(self, tau):  # kernels.jl:266-288
        with np.errstate(divide="ignore", invalid="ignore"):
            dzl, dzu = self.d.zl, self.d.zu
            vl = np.where(dzl < 0, (-self.zl_r) * tau / dzl, np.inf)
            vu = np.where((dzu < 0) & (self.zu_r + dzu < 0), (-self.zu_r) * tau / dzu, np.inf)
        (al, il), (au, iu) = self._argmin_last(vl), self._argmin_last(vu)
        return al, au, il, iu

    def get_fraction_to_boundary_step(self, tau):  # kernels.jl:290-305
        axl, axu, _, _ = self.get_alpha_max_primal(tau)
        azl, azu, _, _ = self.get_alpha_max_dual(tau)
        return min(axl, axu), min(azl, azu)

    def update_step(self):  # kernels.jl:307-374
        rule = self.opt.step_rule
        if isinstance(rule, ConservativeStep):
            self.alpha_p, self.alpha_d = self.get_fraction_to_boundary_step(rule.tau)
        elif isinstance(rule, AdaptiveStep):
            tau = max(1 - self.mu, rule.tau_min)
            self.alpha_p, self.alpha_d = self.get_fraction_to_boundary_step(tau)
        else:  # MehrotraAdaptiveStep, kernels.jl:325-374
            gamma_a = 1.0 / (1.0 - rule.gamma_f)
            d_zl, d_zu = self.d.zl, self.d.zu
            axl, axu, i_xl, i_xu = self.get_alpha_max_primal(1.0)
            azl, azu, i_zl, i_zu = self.get_alpha_max_dual(1.0)
            max_ap, max_ad = min(axl, axu), min(azl, azu)
            mu_full = self.get_affine_complementarity_measure(max_ap, max_ad) / gamma_a
            alpha_p = alpha_d = 1.0
            if max_ap < 1.0:
                if axl <= axu:
                    tmp = mu_full / (self.zl_r[i_xl] + max_ad * d_zl[i_xl])
                    alpha_p = (self.x_lr[i_xl] - self.xl_r[i_xl] - tmp) / (-self.dx_lr[i_xl])
                else:
                    tmp = mu_full / (self.zu_r[i_xu] + max_ad * d_zu[i_xu])
                    alpha_p = (self.xu_r[i_xu] - self.x_ur[i_xu] - tmp) / (self.dx_ur[i_xu])
            if max_ad < 1.0:
                if azl <= azu:
                    tmp = mu_full / (self.x_lr[i_zl] + max_ap * self.dx_lr[i_zl] - self.xl_r[i_zl])
                    alpha_d = -(self.zl_r[i_zl] - tmp) / d_zl[i_zl]
                else:
                    tmp = mu_full / (self.xu_r[i_zu] - self.x_ur[i_zu] - max_ap * self.dx_ur[i_zu])
                    alpha_d = -(self.zu_r[i_zu] - tmp) / d_zu[i_zu]
            self.alpha_p = max(alpha_p, rule.gamma_f * max_ap)
            self.alpha_d = max(alpha_d, rule.gamma_f * max_ad)

    def init_regularization(self):  # kernels.jl:380-408
        reg = self.opt.regularization
        self.del_w = 1.0
        self.del_c = 0.0 if isinstance(reg, NoRegularization) else reg.delta_d

    def update_regularization(self):  # kernels.jl:386-417
        reg = self.opt.regularization
        if isinstance(reg, NoRegularization):
            self.del_w, self.del_c = 0.0, 0.0
        elif isinstance(reg, FixedRegularization):
            self.del_w, self.del_c = reg.delta_p, reg.delta_d
        else:
            reg.delta_p = max(reg.delta_p / 10.0, reg.delta_min)
            reg.delta_d = min(reg.delta_d / 10.0, -reg.delta_min)
            self.del_w, self.del_c = reg.delta_p, reg.delta_d

    def get_optimality_gap(self):  # kernels.jl:435-446 -> MadNLP.get_inf_compl(mu=0, sc=1)
        l = np.max(np.abs((self.x_lr - self.xl_r) * self.zl_r), initial=0.0)
        u = np.max(np.abs((self.xu_r - self.x_ur) * self.zu_r), initial=0.0)
        return max(l, u)

    # ---- src/linear_solver.jl ----
    def factorize_regularized_system(self):  # linear_solver.jl:6-17
        for _ in range(3):
            self.set_aug_diagonal_reg()
            self.kkt.build_and_factorize()  # MadNLP.factorize_wrapper!
            if self.kkt.is_factorized():
                break
            self.del_w *= 100.0
            self.del_c *= 100.0

    def solve_system(self):  # linear_solver.jl:19-45
        d, p, w = self.d, self.p, self._w1
        d.values[:] = p.values
        self.kkt.solve(d)
        w.values[:] = p.values
        self.kkt.mul(w, d, -1.0, 1.0)
        for _ in range(self.opt.refine_steps):  # extension, off by default (see IPMOptions)
            self.kkt.solve(w)
            d.values += w.values
            w.values[:] = p.values
            self.kkt.mul(w, d, -1.0, 1.0)
        norm_w = np.max(np.abs(w.values), initial=0.0)
        norm_p = np.max(np.abs(p.values), initial=0.0)
        ratio = norm_w / max(1.0, norm_p)
        self.last_residual_ratio = ratio
        if math.isnan(ratio) or (self.opt.check_residual and ratio > self.opt.tol_linear_solve):
            raise SolveException()
        return d

    # ---- src/solver.jl ----
    def init_starting_point(self):  # solver.jl:6-125
        x, l, u = self.x, self.xl, self.xu
        ilb, iub = self.ind_lb, self.ind_ub
        k = self.kkt
        k.reg[:] = self.del_w
        k.pr_diag[:] = self.del_w
        k.du_diag[:] = self.del_c
        k.build_and_factorize()  # :21
        self.set_initial_primal_rhs()
        self.solve_system()
        x += 1.0 * self.d.xp  # :28
        self.set_initial_dual_rhs()
        self.solve_system()
        self.y[:] = self.d.y  # :33
        res = k.jtprod(self.y)  # :37
        res += 1.0 * self.f  # :39
        fl, fu = np.isfinite(l), np.isfinite(u)
        self.zl[:] = np.where(fl & fu, 0.5 * res, np.where(fl, res, self.zl))  # :41-53
        self.zu[:] = np.where(fl & fu, -0.5 * res, np.where(fu, -res, self.zu))  # :54-66
        delta_x = max(0.0, -1.5 * np.min(x[ilb] - l[ilb], initial=0.0),
                      -1.5 * np.min(u[iub] - x[iub], initial=0.0))  # :68-72
        delta_s = max(0.0, -1.5 * np.min(self.zl[ilb], initial=0.0),
                      -1.5 * np.min(self.zu[iub], initial=0.0))  # :74-78
        x[ilb] = x[ilb] + delta_x  # :80  (x_lr and x_ur alias the same x: shifts cancel on
        x[iub] = x[iub] - delta_x  # :81   two-sided variables)
        self.zl[ilb] += 1.0 + delta_s  # :82
        self.zu[iub] += 1.0 + delta_s  # :83
        mu = 0.0
        if self.nlb > 0:
            mu += x[ilb] @ self.zl[ilb] - l[ilb] @ self.zl[ilb]  # :87
        if self.nub > 0:
            mu += u[iub] @ self.zu[iub] - x[iub] @ self.zu[iub]  # :90
        with np.errstate(invalid="ignore", divide="ignore"):  # no bound at all: 0 / 0 = NaN, as in Julia, added to empty views below
            delta_x2 = np.float64(mu) / (2 * (np.sum(self.zl[ilb]) + np.sum(self.zu[iub])))  # :93
            delta_s2 = np.float64(mu) / (2 * (np.sum(x[ilb] - l[ilb]) + np.sum(u[iub] - x[iub])))  # :94
        x[ilb] += delta_x2  # :96
        x[iub] -= delta_x2  # :97
        self.zl[ilb] += delta_s2
        self.zu[iub] += delta_s2
        kappa = self.opt.bound_fac  # :102-118
        with np.errstate(invalid="ignore"):
            pl = np.minimum(kappa * np.maximum(1.0, l), kappa * (u - l))
            pu = np.minimum(kappa * np.maximum(1.0, u), kappa * (u - l))
            x[:] = np.where(x < l, l + pl, np.where(u < x, u - pu, x))
        assert np.all(self.zl_r > 0.0) and np.all(self.zu_r > 0.0)  # :120-123
        assert np.all(self.x_lr > self.xl_r) and np.all(self.x_ur < self.xu_r)

    def initialize(self, start=None):  # solver.jl:127-182
        """``start``: None = the reference's start point (init_starting_point, one factorisation + two solves); a dict
        with x, y, zl, zu (scaled iterates of length n, m, n, n) = take these instead -- bench.py's cpu_baseline leg
        times ONE iteration at the metric size from the start point the device computed (equal to this file's to 1e-9,
        tests/test_gpu_solver.py) without paying a second 90 s factorisation for it."""
        qp, opt = self.qp, self.opt
        nx = self.nx
        # MadNLP.initialize!(cb, ...) (MadNLP-recall; :131-142)
        self.x[:nx] = qp.x0
        self.x[nx:] = 0.0
        self.y[:] = qp.y0
        self.xl[:nx], self.xu[:nx] = qp.lvar, qp.uvar
        self.xl[nx:], self.xu[nx:] = qp.lcon[self.ind_ineq], qp.ucon[self.ind_ineq]
        self.rhs[:] = np.where(qp.lcon == qp.ucon, qp.lcon, 0.0)
        tol = opt.bound_relax_factor
        fl, fu = np.isfinite(self.xl), np.isfinite(self.xu)
        self.xl[fl] -= np.maximum(1.0, np.abs(self.xl[fl])) * tol
        self.xu[fu] += np.maximum(1.0, np.abs(self.xu[fu])) * tol
        self.x[:] = _initialize_variable(self.x, self.xl, self.xu, opt.bound_push, opt.bound_fac)
        self.jacl[:] = 0.0  # :144
        if opt.scaling:  # MadNLP.set_scaling!(…, 100) (MadNLP-recall; :148-159)
            rowmax = np.max(np.abs(qp.A), axis=1, initial=0.0)
            with np.errstate(divide="ignore"):
                self.con_scale = np.minimum(1.0, 100.0 / rowmax)
            g = np.max(np.abs(qp.grad(self.x[:nx])), initial=0.0)
            self.obj_scale = min(1.0, 100.0 / g) if g > 0 else 1.0
            cs_slk = self.con_scale[self.ind_ineq]
            self.y /= self.con_scale
            self.rhs *= self.con_scale
            self.x[nx:] *= cs_slk
            self.xl[nx:] *= cs_slk
            self.xu[nx:] *= cs_slk
            self.H = self.obj_scale * qp.H
            self.q = self.obj_scale * qp.q
            self.A_full[:, :nx] = self.con_scale[:, None] * qp.A
        self.kkt.initialize()  # :162
        self.init_regularization()  # :163
        self.obj_val = self.eval_f()  # :166
        self.eval_grad_f()  # :168
        self.eval_cons()  # :169
        self.norm_b = np.max(np.abs(self.rhs), initial=0.0)  # :173
        self.norm_c = np.max(np.abs(self.f), initial=0.0)  # :174
        if start is None:
            self.init_starting_point()  # :177
        else:
            self.x[:], self.y[:], self.zl[:], self.zu[:] = start["x"], start["y"], start["zl"], start["zu"]
        self.mu = opt.mu_init  # :179

    def affine_direction(self):  # solver.jl:188-192
        self.set_predictive_rhs()
        self.solve_system()

    def mehrotra_correction_direction(self):  # solver.jl:194-198
        self.set_correction_rhs(self.mu)
        self.solve_system()

    def gondzio_correction_direction(self, mu_curr, max_ncorr):  # solver.jl:200-251
        delta, bmin, bmax, tau = 0.1, 0.1, 10.0, 0.995
        dp = self._w2.values
        alpha_p, alpha_d = self.get_fraction_to_boundary_step(tau)
        for _ in range(max_ncorr):
            ta_p, ta_d = min(alpha_p + delta, 1.0), min(alpha_d + delta, 1.0)
            ga = self.get_affine_complementarity_measure(ta_p, ta_d)
            mu = (ga / mu_curr) ** 2 * ga
            self.set_extra_correction(ta_p, ta_d, bmin, bmax, mu)
            self.set_correction_rhs(mu)
            dp[:] = self.d.values
            self.solve_system()
            ha_p, ha_d = self.get_fraction_to_boundary_step(tau)
            if ha_p < 1.005 * alpha_p or ha_d < 1.005 * alpha_d:
                self.d.values[:] = dp
                break
            alpha_p, alpha_d = ha_p, ha_d
        return alpha_p, alpha_d

    def record(self):
        self.trace.append(dict(
            k=self.k, obj=self.obj_val / self.obj_scale, inf_pr=self.inf_pr, inf_du=self.inf_du,
            inf_compl=self.inf_compl, mu=self.mu,
            dnorm=0.0 if self.k == 0 else float(np.max(np.abs(self.d.xp), initial=0.0)),
            del_w=self.del_w, alpha_d=self.alpha_d, alpha_p=self.alpha_p))

    def iteration_head(self):
        """solver.jl:259-283: residuals + termination test.  Returns a status or None."""
        self.jacl[:] = self.kkt.jtprod(self.y)  # :259
        self.inf_pr = np.max(np.abs(self.c), initial=0.0) / max(1.0, self.norm_b)  # :264
        self.inf_du = np.max(np.abs(self.f - self.zl + self.zu + self.jacl), initial=0.0) / max(
            1.0, self.norm_c)  # :265-271
        self.inf_compl = self.get_optimality_gap() / max(1.0, self.norm_c)  # :272
        self.record()
        if max(self.inf_pr, self.inf_du, self.inf_compl) <= self.opt.tol:  # :279
            return SOLVE_SUCCEEDED
        if self.k >= self.opt.max_iter:
            return MAXIMUM_ITERATIONS_EXCEEDED
        return None

    def iteration_body(self):
        """solver.jl:288-343: one predictor-corrector step."""
        self.update_regularization()  # :288
        self.factorize_regularized_system()  # :289
        self.affine_direction()  # :294
        a_aff_p, a_aff_d = self.get_fraction_to_boundary_step(1.0)  # :295
        mu_affine = self.get_affine_complementarity_measure(a_aff_p, a_aff_d)  # :296
        self.get_correction()  # :297
        mu_curr = self.update_barrier(mu_affine)  # :302
        self.mehrotra_correction_direction()  # :307
        if self.opt.max_ncorr > 0:  # :316-324
            self.gondzio_correction_direction(mu_curr, self.opt.max_ncorr)
        self.update_step()  # :329
        self.x += self.alpha_p * self.d.xp  # :332
        self.y += self.alpha_d * self.d.y  # :333
        self.zl[self.ind_lb] += self.alpha_d * self.d.zl  # :334
        self.zu[self.ind_ub] += self.alpha_d * self.d.zu  # :335
        self.obj_val = self.eval_f()  # :338
        self.eval_cons()  # :339
        self.eval_grad_f()  # :340
        adjust_boundary(self.x, self.xl, self.xu, self.ind_lb, self.ind_ub, self.mu)  # :342
        self.k += 1

    def mpc(self):  # solver.jl:254-345
        while True:
            st = self.iteration_head()
            if st is not None:
                return st
            self.iteration_body()

    def solve(self):  # solver.jl:347-403
        try:
            self.initialize()
            self.status = self.mpc()
        except SolveException:
            self.status = ERROR_IN_STEP_COMPUTATION
        except AssertionError:
            self.status = INTERNAL_ERROR
        return self.result()

    def result(self):
        """MadNLP.update!(stats, solver): unscaled solution, objective, multipliers."""
        return dict(
            status=self.status, iter=self.k, objective=self.obj_val / self.obj_scale,
            solution=self.x[: self.nx].copy(),
            constraints=self.qp.A @ self.x[: self.nx],
            multipliers=self.y * self.con_scale / self.obj_scale,
            multipliers_L=self.zl[: self.nx] / self.obj_scale,
            multipliers_U=self.zu[: self.nx] / self.obj_scale,
            n_factorizations=self.kkt.n_factorizations,
            trace=self.trace,
        )


def solve(qp: DenseQP, **opts):
    return MPCSolver(qp, **opts).solve()
