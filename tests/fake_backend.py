"""Test double for ``madqp_jl_amd.backend.HipBackend`` (CPU, numpy).

TEST INFRASTRUCTURE ONLY.  It lets the ``-m "not gpu"`` suite drive the product's host
logic (``madqp_jl_amd/solver.py`` and ``kkt.py``: control flow, retries, step rules,
state plumbing) in a container without a GPU.  Every method restates one C-ABI entry
point of ``include/madqp.h`` in numpy on torch CPU tensors.  Nothing under
``madqp_jl_amd/`` imports this file and the product never selects it.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import torch

from madqp_jl_amd.backend import State

EPS = np.finfo(np.float64).eps


def _np(t):
    return t.numpy()


class FakeKKT:
    def __init__(self, nx, m, ind_ineq, H, A, normal=False, augmented=False):
        self.normal, self.augmented = normal, augmented
        self.hdiag = None
        self.nx, self.m = nx, m
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ns = len(self.ind_ineq)
        self.slot = -np.ones(m, dtype=np.int64)
        self.slot[self.ind_ineq] = np.arange(self.ns)
        self.H = None if H is None else _np(H)
        self.A = _np(A)
        self.K = None
        self.chol = None


class FakeBackend:
    name = "fake-cpu"

    def __init__(self):
        self.device = torch.device("cpu")
        self.fail_factorizations = 0  # inject factorization failures (retry-path tests)

    def new_state(self, n, m, ind_lb, ind_ub):
        return State(n, m, ind_lb, ind_ub, self.device)

    def close(self):
        pass

    def sync(self):
        pass

    # helpers
    @staticmethod
    def _parts(st, v):
        a = _np(v)
        n, m, nlb = st.n, st.m, st.nlb
        return a[:n], a[n:n + m], a[n + m:n + m + nlb], a[n + m + nlb:]

    @staticmethod
    def _idx(st):
        return _np(st.ind_lb), _np(st.ind_ub)

    # ---- vector kernels ----
    def set_aug_diagonal_reg(self, st, del_w, del_c):
        ilb, iub = self._idx(st)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        _np(st.reg)[:] = del_w
        _np(st.du_diag)[:] = del_c
        _np(st.l_diag)[:] = xl[ilb] - x[ilb]
        _np(st.u_diag)[:] = x[iub] - xu[iub]
        _np(st.l_lower)[:] = zl[ilb]
        _np(st.u_lower)[:] = zu[iub]
        pr = _np(st.pr_diag)
        pr[:] = del_w
        pr[ilb] -= _np(st.l_lower) / _np(st.l_diag)
        pr[iub] -= _np(st.u_lower) / _np(st.u_diag)

    def _rhs(self, st, mode, mu=0.0):
        ilb, iub = self._idx(st)
        px, py, pzl, pzu = self._parts(st, st.p)
        x, xl, xu, zl, zu, f, jacl, c = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu, st.f, st.jacl, st.c))
        _np(st.p)[:] = 0.0
        if mode <= 1:
            px[:] = -f + zl - zu - jacl
            py[:] = -c
            pzl[:] = (xl[ilb] - x[ilb]) * zl[ilb]
            pzu[:] = (xu[iub] - x[iub]) * zu[iub]
            if mode == 1:
                pzl[:] = pzl + mu - _np(st.correction_lb)
                pzu[:] = pzu - mu - _np(st.correction_ub)
        elif mode == 2:
            py[:] = -c
        else:
            px[:] = -f

    def set_initial_primal_rhs(self, st):
        self._rhs(st, 2)

    def set_initial_dual_rhs(self, st):
        self._rhs(st, 3)

    def set_predictive_rhs(self, st):
        self._rhs(st, 0)

    def set_correction_rhs(self, st, mu):
        self._rhs(st, 1, mu)

    def get_correction(self, st):
        ilb, iub = self._idx(st)
        dx, _, dzl, dzu = self._parts(st, st.d)
        _np(st.correction_lb)[:] = dx[ilb] * dzl
        _np(st.correction_ub)[:] = dx[iub] * dzu

    def set_extra_correction(self, st, ap, ad, bmin, bmax, mu):
        ilb, iub = self._idx(st)
        dx, _, dzl, dzu = self._parts(st, st.d)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        tmin, tmax = bmin * mu, bmax * mu
        v = (x[ilb] + ap * dx[ilb] - xl[ilb]) * (zl[ilb] + ad * dzl)
        _np(st.correction_lb)[:] -= np.where(v < tmin, tmin - v, np.where(v > tmax, tmax - v, 0.0))
        v = (xu[iub] - ap * dx[iub] - x[iub]) * (zu[iub] + ad * dzu)
        _np(st.correction_ub)[:] += np.where(v < tmin, tmin - v, np.where(v > tmax, tmax - v, 0.0))

    def get_complementarity_measure(self, st):
        if st.nlb + st.nub == 0:
            return 0.0
        ilb, iub = self._idx(st)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        return (np.sum((x[ilb] - xl[ilb]) * zl[ilb]) + np.sum((xu[iub] - x[iub]) * zu[iub])) / (st.nlb + st.nub)

    def get_affine_complementarity_measure(self, st, ap, ad):
        if st.nlb + st.nub == 0:
            return 0.0
        ilb, iub = self._idx(st)
        dx, _, dzl, dzu = self._parts(st, st.d)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        l = np.sum(((x[ilb] + ap * dx[ilb]) - xl[ilb]) * (zl[ilb] + ad * dzl))
        u = np.sum((xu[iub] - (x[iub] + ap * dx[iub])) * (zu[iub] + ad * dzu))
        return (l + u) / (st.nlb + st.nub)

    def get_alpha_max(self, st, tau):
        ilb, iub = self._idx(st)
        dx, _, dzl, dzu = self._parts(st, st.d)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        with np.errstate(divide="ignore", invalid="ignore"):
            v = [
                np.where(dx[ilb] < 0, (-x[ilb] + xl[ilb]) * tau / dx[ilb], np.inf),
                np.where(dx[iub] > 0, (-x[iub] + xu[iub]) * tau / dx[iub], np.inf),
                np.where(dzl < 0, (-zl[ilb]) * tau / dzl, np.inf),
                np.where((dzu < 0) & (zu[iub] + dzu < 0), (-zu[iub]) * tau / dzu, np.inf),
            ]
        al, ib = [], []
        for a in v:
            if a.size and a.min() < 1.0:
                i = a.size - 1 - int(np.argmin(a[::-1]))  # the LAST minimum (src/kernels.jl:248 keeps the right element on a tie)
                al.append(float(a[i]))
                ib.append(i)
            else:
                al.append(1.0)
                ib.append(-1)
        return al, ib

    def update_iterates(self, st, ap, ad):
        ilb, iub = self._idx(st)
        dx, dy, dzl, dzu = self._parts(st, st.d)
        _np(st.x)[:] += ap * dx
        _np(st.y)[:] += ad * dy
        _np(st.zl)[ilb] += ad * dzl
        _np(st.zu)[iub] += ad * dzu

    def get_inf(self, st):
        ilb, iub = self._idx(st)
        x, xl, xu, zl, zu, f, jacl, c = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu, st.f, st.jacl, st.c))
        a = np.max(np.abs(c), initial=0.0)
        b = np.max(np.abs(f - zl + zu + jacl), initial=0.0)
        cl = np.max(np.abs((x[ilb] - xl[ilb]) * zl[ilb]), initial=0.0)
        cu = np.max(np.abs((xu[iub] - x[iub]) * zu[iub]), initial=0.0)
        return float(a), float(b), float(max(cl, cu))

    def adjust_boundary(self, st, mu):
        ilb, iub = self._idx(st)
        x, xl, xu = map(_np, (st.x, st.xl, st.xu))
        c1, c2 = EPS * mu, EPS ** 0.75
        xl[ilb] = np.where(x[ilb] - xl[ilb] < c1, xl[ilb] - c2 * np.maximum(1.0, np.abs(x[ilb])), xl[ilb])
        xu[iub] = np.where(xu[iub] - x[iub] < c1, xu[iub] + c2 * np.maximum(1.0, np.abs(x[iub])), xu[iub])

    def reduce_rhs(self, st, w):
        ilb, iub = self._idx(st)
        wx, _, wzl, wzu = self._parts(st, w)
        wx[ilb] -= wzl / _np(st.l_diag)
        wx[iub] -= wzu / _np(st.u_diag)

    def finish_aug_solve(self, st, w):
        ilb, iub = self._idx(st)
        wx, _, wzl, wzu = self._parts(st, w)
        wzl[:] = (-wzl + _np(st.l_lower) * wx[ilb]) / _np(st.l_diag)
        wzu[:] = (wzu - _np(st.u_lower) * wx[iub]) / _np(st.u_diag)

    def kktmul(self, st, w, v, alpha, beta):
        ilb, iub = self._idx(st)
        wx, wy, wzl, wzu = self._parts(st, w)
        vx, vy, vzl, vzu = self._parts(st, v)
        wx[:] += alpha * _np(st.reg) * vx
        wy[:] += alpha * _np(st.du_diag) * vy
        wx[ilb] -= alpha * vzl
        wx[iub] += alpha * vzu
        wzl[:] = beta * wzl + alpha * (vx[ilb] * _np(st.l_lower) - vzl * _np(st.l_diag))
        wzu[:] = beta * wzu + alpha * (vx[iub] * _np(st.u_lower) + vzu * _np(st.u_diag))

    def norm_inf3(self, a, b, c):
        f = lambda t: float(np.max(np.abs(_np(t)), initial=0.0)) if t is not None else 0.0
        return f(a), f(b), f(c)

    def norm_inf(self, a):
        return self.norm_inf3(a, None, None)[0]

    def axpy(self, alpha, x, y):
        _np(y)[:] += alpha * _np(x)

    def copy(self, src, dst):
        _np(dst)[:] = _np(src)

    def fill(self, value, dst):
        _np(dst)[:] = value

    def gemv(self, trans, rows, cols, alpha, A, lda, x, beta, y):
        Am = _np(A).reshape(-1)[: rows * lda].reshape(rows, lda)[:, :cols]
        xin = _np(x)[: (rows if trans else cols)]
        yout = _np(y)[: (cols if trans else rows)]
        prod = (Am.T @ xin) if trans else (Am @ xin)
        yout[:] = alpha * prod + (beta * yout if beta != 0 else 0.0)

    # ---- starting point ----
    def sp_init_duals(self, st):
        res, l, u = map(_np, (st.jacl, st.xl, st.xu))
        fl, fu = np.isfinite(l), np.isfinite(u)
        zl, zu = _np(st.zl), _np(st.zu)
        zl[:] = np.where(fl & fu, 0.5 * res, np.where(fl, res, zl))
        zu[:] = np.where(fl & fu, -0.5 * res, np.where(fu, -res, zu))

    def sp_mins(self, st):
        ilb, iub = self._idx(st)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        return [float(np.min(x[ilb] - xl[ilb], initial=0.0)), float(np.min(xu[iub] - x[iub], initial=0.0)),
                float(np.min(zl[ilb], initial=0.0)), float(np.min(zu[iub], initial=0.0))]

    def sp_shift(self, st, dx, dz):
        ilb, iub = self._idx(st)
        x, zl, zu = map(_np, (st.x, st.zl, st.zu))
        x[ilb] = x[ilb] + dx
        x[iub] = x[iub] - dx
        zl[ilb] += dz
        zu[iub] += dz

    def sp_sums(self, st):
        ilb, iub = self._idx(st)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        return [float(v) for v in (
            x[ilb] @ zl[ilb], xl[ilb] @ zl[ilb], xu[iub] @ zu[iub], x[iub] @ zu[iub],
            np.sum(zl[ilb]), np.sum(zu[iub]), np.sum(x[ilb] - xl[ilb]), np.sum(xu[iub] - x[iub]))]

    def sp_project(self, st, kappa):
        x, l, u = map(_np, (st.x, st.xl, st.xu))
        with np.errstate(invalid="ignore"):
            pl = np.minimum(kappa * np.maximum(1.0, l), kappa * (u - l))
            pu = np.minimum(kappa * np.maximum(1.0, u), kappa * (u - l))
            x[:] = np.where(x < l, l + pl, np.where(u < x, u - pu, x))

    def sp_check(self, st):
        ilb, iub = self._idx(st)
        x, xl, xu, zl, zu = map(_np, (st.x, st.xl, st.xu, st.zl, st.zu))
        return bool(np.all(zl[ilb] > 0) and np.all(zu[iub] > 0) and np.all(x[ilb] > xl[ilb])
                    and np.all(x[iub] < xu[iub]))

    # ---- condensed KKT ----
    def kkt_create(self, nx, m, ind_ineq, H, ldh, A, lda):
        return FakeKKT(nx, m, ind_ineq, H, A)

    def kkt_create_normal(self, nx, m, ind_ineq, At, ldat):
        return FakeKKT(nx, m, ind_ineq, None, At.t(), normal=True)

    def kkt_create_augmented(self, nx, m, ind_ineq, H, ldh, A, lda):
        return FakeKKT(nx, m, ind_ineq, H, A, augmented=True)

    def kkt_set_hdiag(self, h, hdiag):
        h.hdiag = _np(hdiag)
        h.H = np.diag(h.hdiag)

    def kkt_destroy(self, h):
        pass

    def _theta(self, h, st):
        S = _np(st.pr_diag)[h.nx:]
        du = _np(st.du_diag)
        th = np.empty(h.m)
        isq = h.slot >= 0
        th[isq] = S[h.slot[isq]] / (1.0 - du[isq] * S[h.slot[isq]])
        with np.errstate(divide="ignore"):
            th[~isq] = -1.0 / du[~isq]
        return th

    def kkt_build(self, h, st):
        if h.augmented:  # [H + Sigma_x, A'; A, -D] (csrc/kkt.hip, mode AUGMENTED)
            S, du = _np(st.pr_diag), _np(st.du_diag)
            isq = h.slot >= 0
            D = -du.copy()
            D[isq] += 1.0 / S[h.nx + h.slot[isq]]
            K = np.zeros((h.nx + h.m, h.nx + h.m))
            if h.H is not None:
                K[: h.nx, : h.nx] = h.H
            K[np.arange(h.nx), np.arange(h.nx)] += S[: h.nx]
            K[h.nx:, : h.nx] = h.A
            K[: h.nx, h.nx:] = h.A.T
            K[np.arange(h.nx, h.nx + h.m), np.arange(h.nx, h.nx + h.m)] = -D
            h.K = K
            return
        if h.normal:  # src/KKT/normalkkt.jl:166-180
            D = 1.0 / _np(st.pr_diag)
            K = (h.A * D[: h.nx]) @ h.A.T
            isq = h.slot >= 0
            K[np.flatnonzero(isq), np.flatnonzero(isq)] += D[h.nx + h.slot[isq]]
            h.K = K
            return
        h.theta = self._theta(h, st)
        K = (h.A.T * h.theta) @ h.A
        if h.H is not None:
            K = K + h.H
        K[np.arange(h.nx), np.arange(h.nx)] += _np(st.pr_diag)[: h.nx]
        h.K = K

    def kkt_factorize(self, h):
        if self.fail_factorizations > 0:
            self.fail_factorizations -= 1
            return 1
        if h.augmented:  # quasi-definite: L diag(I, -I) L' exists iff both Cholesky factorisations below do
            try:
                nx = h.nx
                L11 = sla.cholesky(h.K[:nx, :nx], lower=True, check_finite=False)
                W = sla.solve_triangular(L11, h.K[:nx, nx:], lower=True, check_finite=False).T
                L22 = sla.cholesky(W @ W.T - h.K[nx:, nx:], lower=True, check_finite=False) if h.m else np.zeros((0, 0))
                h.chol = (L11, W, L22)
                return 0
            except sla.LinAlgError:
                return 1
        try:
            h.chol = sla.cho_factor(h.K, lower=True, check_finite=False)
            return 0
        except sla.LinAlgError as e:
            return 1

    def kkt_solve(self, h, st, w):
        self.reduce_rhs(st, w)
        wx, wy, _, _ = self._parts(st, w)
        nx, S = h.nx, _np(st.pr_diag)
        isq = h.slot >= 0
        if h.augmented:
            L11, W, L22 = h.chol
            b2 = wy.copy()
            b2[isq] += wx[nx + h.slot[isq]] / S[nx + h.slot[isq]]
            z1 = sla.solve_triangular(L11, wx[:nx], lower=True, check_finite=False)
            z2 = -sla.solve_triangular(L22, b2 - W @ z1, lower=True, check_finite=False) if h.m else b2
            dy = sla.solve_triangular(L22, z2, lower=True, trans="T", check_finite=False) if h.m else b2
            wx[:nx] = sla.solve_triangular(L11, z1 - W.T @ dy, lower=True, trans="T", check_finite=False)
            wy[:] = dy
            wx[nx + h.slot[isq]] = (wx[nx + h.slot[isq]] + dy[isq]) / S[nx + h.slot[isq]]
            self.finish_aug_solve(st, w)
            return
        if h.normal:  # src/KKT/normalkkt.jl:185-201
            r1 = wx / S
            u = h.A @ r1[:nx]
            u[isq] -= r1[nx + h.slot[isq]]
            wy[:] = sla.cho_solve(h.chol, u - wy, check_finite=False)
            t = np.empty(st.n)
            t[:nx] = h.A.T @ wy
            t[nx:] = -wy[h.ind_ineq]
            wx[:] = (wx - t) / S
            self.finish_aug_solve(st, w)
            return
        t = wy.copy()
        t[isq] += wx[nx + h.slot[isq]] / S[nx + h.slot[isq]]
        wx[:nx] = wx[:nx] + h.A.T @ (h.theta * t)
        wx[:nx] = sla.cho_solve(h.chol, wx[:nx], check_finite=False)
        dy = h.theta * (h.A @ wx[:nx] - t)
        wy[:] = dy
        wx[nx + h.slot[isq]] = (wx[nx + h.slot[isq]] + dy[isq]) / S[nx + h.slot[isq]]
        self.finish_aug_solve(st, w)

    def kkt_jtprod(self, h, out, y):
        o, yy = _np(out), _np(y)
        o[: h.nx] = h.A.T @ yy
        o[h.nx:] = -yy[h.ind_ineq]

    def kkt_mul(self, h, st, w, v, alpha, beta, solved=False):
        wx, wy, _, _ = self._parts(st, w)
        vx, vy, _, _ = self._parts(st, v)
        nx = h.nx
        isq = h.slot >= 0
        wx[:nx] = alpha * (h.A.T @ vy) + beta * wx[:nx]
        if h.H is not None:
            wx[:nx] += alpha * (h.H @ vx[:nx])
        wx[nx:] = alpha * (-vy[h.ind_ineq]) + beta * wx[nx:]
        u = h.A @ vx[:nx]
        u[isq] -= vx[nx + h.slot[isq]]
        wy[:] = alpha * u + beta * wy
        self.kktmul(st, w, v, alpha, beta)

    def kkt_eval(self, h, st, q, rhs, c0):
        x, f, c = _np(st.x), _np(st.f), _np(st.c)
        nx = h.nx
        hx = h.H @ x[:nx] if h.H is not None else np.zeros(nx)
        qq = _np(q)
        f[:nx] = hx + qq
        f[nx:] = 0.0
        u = h.A @ x[:nx]
        isq = h.slot >= 0
        u[isq] -= x[nx + h.slot[isq]]
        c[:] = u - _np(rhs)
        return float(c0 + qq @ x[:nx] + 0.5 * (x[:nx] @ hx))
