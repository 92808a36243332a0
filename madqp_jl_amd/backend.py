"""HIP backend: the one object through which the host code reaches the device.

Thin, typed wrappers over the C ABI (``include/madqp.h``).  Vectors and matrices
are ``torch`` CUDA tensors (float64 / int64) whose ``data_ptr()`` is handed to
the library; the library launches on torch's current stream, so torch ops and
library calls are stream ordered.  No compute happens in Python and nothing
here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import CState, MadQPError, ptr


class State:
    """Device-resident solver state: the tensors behind ``madqp_state``.

    Layout follows ``MPCSolver`` (src/structure.jl:1-75): iterates of length
    ``n = nx + ns``, multipliers of length ``m``, UnreducedKKTVectors
    ``[x(n) | y(m) | zl(nlb) | zu(nub)]`` contiguous.
    """

    VEC_N = ("x", "xl", "xu", "zl", "zu", "f", "jacl", "reg", "pr_diag")
    VEC_M = ("y", "c", "du_diag", "rhs")
    KKT = ("d", "p", "w1", "w2")

    def __init__(self, n, m, ind_lb, ind_ub, device):
        self.n, self.m = int(n), int(m)
        self.device = torch.device(device)
        self.ind_lb = torch.as_tensor(ind_lb, dtype=torch.int64, device=self.device).contiguous()
        self.ind_ub = torch.as_tensor(ind_ub, dtype=torch.int64, device=self.device).contiguous()
        self.nlb, self.nub = self.ind_lb.numel(), self.ind_ub.numel()
        z = lambda k: torch.zeros(max(int(k), 1), dtype=torch.float64, device=self.device)[: int(k)]
        for name in self.VEC_N:
            setattr(self, name, z(self.n))
        for name in self.VEC_M:
            setattr(self, name, z(self.m))
        self.ntot = self.n + self.m + self.nlb + self.nub
        for name in self.KKT:
            setattr(self, name, z(self.ntot))
        self.correction_lb, self.l_diag, self.l_lower = z(self.nlb), z(self.nlb), z(self.nlb)
        self.correction_ub, self.u_diag, self.u_lower = z(self.nub), z(self.nub), z(self.nub)
        self._c = None

    # views of an UnreducedKKTVector (MadNLP.primal / dual / dual_lb / dual_ub)
    def primal(self, v):
        return v[: self.n]

    def dual(self, v):
        return v[self.n : self.n + self.m]

    def dual_lb(self, v):
        return v[self.n + self.m : self.n + self.m + self.nlb]

    def dual_ub(self, v):
        return v[self.n + self.m + self.nlb :]

    @property
    def cstruct(self) -> CState:
        if self._c is None:
            c = CState()
            c.n, c.m, c.nlb, c.nub = self.n, self.m, self.nlb, self.nub
            c.ind_lb, c.ind_ub = ptr(self.ind_lb), ptr(self.ind_ub)
            for k in ("x", "xl", "xu", "zl", "zu", "f", "y", "c", "jacl", "d", "p",
                      "correction_lb", "correction_ub", "reg", "pr_diag", "du_diag",
                      "l_diag", "l_lower", "u_diag", "u_lower"):
                setattr(c, k, ptr(getattr(self, k)))
            self._c = c
        return self._c


class HipBackend:
    """One context = one device + one stream (Threading)."""

    name = "hip"

    def __init__(self, device_index: int = 0, stream=None):
        """``stream``: a ``torch.cuda.Stream`` to launch on (default: torch's current stream).  Torch ops
        issued under ``torch.cuda.stream(stream)`` and library calls are then ordered on it."""
        if not torch.cuda.is_available():
            raise MadQPError("no HIP device visible: the MI355X path has no CPU fallback")
        self.lib = _lib.load_cdll()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.stream = stream
        stream = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        h = C.c_void_p()
        rc = self.lib.madqp_ctx_create(device_index, C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise MadQPError(f"madqp_ctx_create failed ({rc})")
        self.ctx = h

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.madqp_ctx_destroy(self.ctx)
            self.ctx = None

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.madqp_last_error(self.ctx)
            raise MadQPError(f"libmadqp_hip error {rc}: {msg.decode() if msg else ''}")

    def new_state(self, n, m, ind_lb, ind_ub) -> State:
        return State(n, m, ind_lb, ind_ub, self.device)

    def sync(self):
        self._ck(self.lib.madqp_ctx_sync(self.ctx))

    # ---- profiling ----
    def prof_enable(self, classes=_lib.PROF_CLASSES):
        """Enable the device timers of the named kernel classes (empty = off)."""
        mask = 0
        for c in classes or ():
            mask |= 1 << _lib.PROF_CLASSES.index(c)
        self._ck(self.lib.madqp_prof_enable(self.ctx, mask))

    def probe_mfma_f64(self, iters=20000) -> float:
        out = C.c_double()
        self._ck(self.lib.madqp_probe_mfma_f64(self.ctx, iters, C.byref(out)))
        return out.value

    def prof_reset(self):
        self._ck(self.lib.madqp_prof_reset(self.ctx))

    def prof_get(self):
        out = {}
        for i, name in enumerate(_lib.PROF_CLASSES):
            ms, cnt = C.c_double(), C.c_int64()
            self._ck(self.lib.madqp_prof_get(self.ctx, i, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    # ---- synthetic data ----
    def gen_normal(self, key, idx0, out):
        self._ck(self.lib.madqp_gen_normal(self.ctx, key, idx0, out.numel(), ptr(out)))

    def gen_wigner(self, key, n, inv_sqrt_n, H):
        self._ck(self.lib.madqp_gen_wigner(self.ctx, key, n, inv_sqrt_n, ptr(H), H.stride(0) if n else 1))

    # ---- dense linear algebra ----
    def syrk_assemble(self, n, kdim, B, ldb, w, base, ldbase, dvec, Cmat, ldc):
        self._ck(self.lib.madqp_syrk_assemble(self.ctx, n, kdim, ptr(B), ldb, ptr(w), ptr(base),
                                              ldbase, ptr(dvec), ptr(Cmat), ldc))

    def chol_create(self, n):
        h = C.c_void_p()
        self._ck(self.lib.madqp_chol_create(self.ctx, n, C.byref(h)))
        return h

    def chol_destroy(self, h):
        self.lib.madqp_chol_destroy(h)

    def chol_set_signature(self, h, npos):
        """Quasi-definite mode: the stored [P, .; B, Q] stands for [P, B'; B, -Q], P of order ``npos``."""
        self._ck(self.lib.madqp_chol_set_signature(h, npos))

    def chol_factor(self, h, A, lda) -> int:
        info = C.c_int32()
        self._ck(self.lib.madqp_chol_factor(h, ptr(A), lda, C.byref(info)))
        return info.value

    def chol_solve(self, h, rhs):
        self._ck(self.lib.madqp_chol_solve(h, ptr(rhs)))

    def gemv(self, trans, rows, cols, alpha, A, lda, x, beta, y):
        self._ck(self.lib.madqp_gemv(self.ctx, trans, rows, cols, alpha, ptr(A), lda, ptr(x), beta, ptr(y)))

    # ---- src/kernels.jl ----
    def set_aug_diagonal_reg(self, st, del_w, del_c):
        self._ck(self.lib.madqp_set_aug_diagonal_reg(self.ctx, C.byref(st.cstruct), del_w, del_c))

    def set_initial_primal_rhs(self, st):
        self._ck(self.lib.madqp_set_initial_primal_rhs(self.ctx, C.byref(st.cstruct)))

    def set_initial_dual_rhs(self, st):
        self._ck(self.lib.madqp_set_initial_dual_rhs(self.ctx, C.byref(st.cstruct)))

    def set_predictive_rhs(self, st):
        self._ck(self.lib.madqp_set_predictive_rhs(self.ctx, C.byref(st.cstruct)))

    def set_correction_rhs(self, st, mu):
        self._ck(self.lib.madqp_set_correction_rhs(self.ctx, C.byref(st.cstruct), mu))

    def get_correction(self, st):
        self._ck(self.lib.madqp_get_correction(self.ctx, C.byref(st.cstruct)))

    def set_extra_correction(self, st, alpha_p, alpha_d, bmin, bmax, mu):
        self._ck(self.lib.madqp_set_extra_correction(self.ctx, C.byref(st.cstruct), alpha_p, alpha_d,
                                                     bmin, bmax, mu))

    def get_complementarity_measure(self, st) -> float:
        out = C.c_double()
        self._ck(self.lib.madqp_get_complementarity_measure(self.ctx, C.byref(st.cstruct), C.byref(out)))
        return out.value

    def get_affine_complementarity_measure(self, st, alpha_p, alpha_d) -> float:
        out = C.c_double()
        self._ck(self.lib.madqp_get_affine_complementarity_measure(
            self.ctx, C.byref(st.cstruct), alpha_p, alpha_d, C.byref(out)))
        return out.value

    def get_alpha_max(self, st, tau):
        a, ib = (C.c_double * 4)(), (C.c_int64 * 4)()
        self._ck(self.lib.madqp_get_alpha_max(self.ctx, C.byref(st.cstruct), tau, a, ib))
        return list(a), list(ib)

    def update_iterates(self, st, alpha_p, alpha_d):
        self._ck(self.lib.madqp_update_iterates(self.ctx, C.byref(st.cstruct), alpha_p, alpha_d))

    def get_inf(self, st):
        out = (C.c_double * 3)()
        self._ck(self.lib.madqp_get_inf(self.ctx, C.byref(st.cstruct), out))
        return out[0], out[1], out[2]

    def adjust_boundary(self, st, mu):
        self._ck(self.lib.madqp_adjust_boundary(self.ctx, C.byref(st.cstruct), mu))

    def reduce_rhs(self, st, w):
        self._ck(self.lib.madqp_reduce_rhs(self.ctx, C.byref(st.cstruct), ptr(w)))

    def finish_aug_solve(self, st, w):
        self._ck(self.lib.madqp_finish_aug_solve(self.ctx, C.byref(st.cstruct), ptr(w)))

    def kktmul(self, st, w, v, alpha, beta):
        self._ck(self.lib.madqp_kktmul(self.ctx, C.byref(st.cstruct), ptr(w), ptr(v), alpha, beta))

    def norm_inf3(self, a, b, c):
        out = (C.c_double * 3)()
        self._ck(self.lib.madqp_norm_inf3(self.ctx, a.numel(), ptr(a), ptr(b), ptr(c), out))
        return out[0], out[1], out[2]

    def norm_inf(self, a) -> float:
        out = C.c_double()
        self._ck(self.lib.madqp_norm_inf(self.ctx, a.numel(), ptr(a), C.byref(out)))
        return out.value

    def axpy(self, alpha, x, y):
        self._ck(self.lib.madqp_axpy(self.ctx, x.numel(), alpha, ptr(x), ptr(y)))

    def copy(self, src, dst):
        self._ck(self.lib.madqp_copy(self.ctx, src.numel(), ptr(src), ptr(dst)))

    def fill(self, value, dst):
        self._ck(self.lib.madqp_fill(self.ctx, dst.numel(), value, ptr(dst)))

    # ---- init_starting_point! helpers ----
    def sp_init_duals(self, st):
        self._ck(self.lib.madqp_sp_init_duals(self.ctx, C.byref(st.cstruct)))

    def sp_mins(self, st):
        out = (C.c_double * 4)()
        self._ck(self.lib.madqp_sp_mins(self.ctx, C.byref(st.cstruct), out))
        return list(out)

    def sp_shift(self, st, dx, dz):
        self._ck(self.lib.madqp_sp_shift(self.ctx, C.byref(st.cstruct), dx, dz))

    def sp_sums(self, st):
        out = (C.c_double * 8)()
        self._ck(self.lib.madqp_sp_sums(self.ctx, C.byref(st.cstruct), out))
        return list(out)

    def sp_project(self, st, kappa):
        self._ck(self.lib.madqp_sp_project(self.ctx, C.byref(st.cstruct), kappa))

    def sp_check(self, st) -> bool:
        ok = C.c_int32()
        self._ck(self.lib.madqp_sp_check(self.ctx, C.byref(st.cstruct), C.byref(ok)))
        return bool(ok.value)

    # ---- condensed KKT object ----
    def kkt_create(self, nx, m, ind_ineq, H, ldh, A, lda):
        ns = len(ind_ineq)
        arr = (C.c_int64 * max(ns, 1))(*[int(i) for i in ind_ineq])
        h = C.c_void_p()
        self._ck(self.lib.madqp_kkt_create(self.ctx, nx, m, ns, arr, ptr(H), ldh, ptr(A), lda, C.byref(h)))
        return h

    def kkt_create_augmented(self, nx, m, ind_ineq, H, ldh, A, lda):
        ns = len(ind_ineq)
        arr = (C.c_int64 * max(ns, 1))(*[int(i) for i in ind_ineq])
        h = C.c_void_p()
        self._ck(self.lib.madqp_kkt_create_augmented(self.ctx, nx, m, ns, arr, ptr(H), ldh, ptr(A), lda, C.byref(h)))
        return h

    def kkt_create_scaled_augmented(self, nx, m, ind_ineq, H, ldh, A, lda):
        ns = len(ind_ineq)
        arr = (C.c_int64 * max(ns, 1))(*[int(i) for i in ind_ineq])
        h = C.c_void_p()
        self._ck(self.lib.madqp_kkt_create_scaled_augmented(self.ctx, nx, m, ns, arr, ptr(H), ldh, ptr(A), lda,
                                                            C.byref(h)))
        return h

    def kkt_set_aug_diagonal_reg(self, h, st, del_w, del_c):
        self._ck(self.lib.madqp_kkt_set_aug_diagonal_reg(h, C.byref(st.cstruct), del_w, del_c))

    def kkt_initialize(self, h, st):
        self._ck(self.lib.madqp_kkt_initialize(h, C.byref(st.cstruct)))

    def kkt_create_normal(self, nx, m, ind_ineq, At, ldat):
        ns = len(ind_ineq)
        arr = (C.c_int64 * max(ns, 1))(*[int(i) for i in ind_ineq])
        h = C.c_void_p()
        self._ck(self.lib.madqp_kkt_create_normal(self.ctx, nx, m, ns, arr, ptr(At), ldat, C.byref(h)))
        return h

    def kkt_create_sparse(self, mode, nx, m, ind_ineq, H, ldh, csr, t_val):
        """``csr``: :class:`DeviceCSR`; ``t_val``: values of A' in its CSR order (kept alive by the caller)."""
        h = C.c_void_p()
        ineq = (C.c_int64 * max(1, len(ind_ineq)))(*[int(i) for i in ind_ineq])
        self._ck(self.lib.madqp_kkt_create_sparse(self.ctx, mode, nx, m, len(ind_ineq), ineq, ptr(H), ldh,
                                                  ptr(csr.ptr), ptr(csr.col), ptr(csr.val), ptr(csr.t_ptr),
                                                  ptr(csr.t_col), ptr(t_val), C.byref(h)))
        return h

    def kkt_set_hdiag(self, h, hdiag):
        self._ck(self.lib.madqp_kkt_set_hdiag(h, ptr(hdiag)))

    def kkt_destroy(self, h):
        self.lib.madqp_kkt_destroy(h)

    def kkt_build(self, h, st):
        self._ck(self.lib.madqp_kkt_build(h, C.byref(st.cstruct)))

    def kkt_factorize(self, h) -> int:
        info = C.c_int32()
        self._ck(self.lib.madqp_kkt_factorize(h, C.byref(info)))
        return info.value

    def kkt_solve(self, h, st, w):
        self._ck(self.lib.madqp_kkt_solve(h, C.byref(st.cstruct), ptr(w)))

    def kkt_set_refine(self, h, steps):
        self._ck(self.lib.madqp_kkt_set_refine(h, int(steps)))

    def kkt_mul(self, h, st, w, v, alpha, beta, solved=False):
        f = self.lib.madqp_kkt_mul_solved if solved else self.lib.madqp_kkt_mul
        self._ck(f(h, C.byref(st.cstruct), ptr(w), ptr(v), alpha, beta))

    def kkt_jtprod(self, h, out, y):
        self._ck(self.lib.madqp_kkt_jtprod(h, ptr(out), ptr(y)))

    def kkt_eval(self, h, st, q, rhs, c0) -> float:
        obj = C.c_double()
        self._ck(self.lib.madqp_kkt_eval(h, C.byref(st.cstruct), ptr(q), ptr(rhs), c0, C.byref(obj)))
        return obj.value

    def kkt_chol(self, h):
        """(linear solver handle, order of its matrix) of a KKT object."""
        ch, n = C.c_void_p(), C.c_int64()
        self._ck(self.lib.madqp_kkt_chol(h, C.byref(ch), C.byref(n)))
        return ch, n.value

    # ---- native driver of one MPC iteration (csrc/mpc.hip) ----
    def mpc_create(self, kkt_h, st, w1, w2, q, rhs, c0, norm_b, norm_c, copt):
        h = C.c_void_p()
        self._ck(self.lib.madqp_mpc_create(kkt_h, C.byref(st.cstruct), ptr(w1), ptr(w2), ptr(q), ptr(rhs),
                                           c0, norm_b, norm_c, C.byref(copt), C.byref(h)))
        return h

    def mpc_destroy(self, h):
        self.lib.madqp_mpc_destroy(h)

    def mpc_set_scalars(self, h, mu, del_w, del_c, obj, k):
        self._ck(self.lib.madqp_mpc_set_scalars(h, mu, del_w, del_c, obj, k))

    def mpc_head(self, h, info):
        status = C.c_int32()
        self._ck(self.lib.madqp_mpc_head(h, C.byref(info), C.byref(status)))
        return status.value

    def mpc_body(self, h, info) -> int:
        """Returns the raw status: 0 ok, MADQP_NUM_NAN (>0) = SolveException; <0 raises."""
        rc = self.lib.madqp_mpc_body(h, C.byref(info))
        if rc < 0:
            self._ck(rc)
        return rc

    def mpc_readbacks(self, h) -> int:
        """Blocking scalar read-backs madqp_mpc_head / madqp_mpc_body have issued (the fused form counts them)."""
        n = C.c_int64()
        self._ck(self.lib.madqp_mpc_readbacks(h, C.byref(n)))
        return n.value

    def mpc_ahead_stats(self, h):
        """(assemblies madqp_mpc_body queued ahead for the next pass, how many of them that pass took over)."""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.madqp_mpc_ahead_stats(h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kkt_matrix(self, h, nx):
        """Torch view (nx x ld, row = column of K) of the library-owned K for inspection."""
        p, ld = C.c_void_p(), C.c_int64()
        self._ck(self.lib.madqp_kkt_matrix(h, C.byref(p), C.byref(ld)))
        return p.value, ld.value

    def read_doubles(self, dev_ptr: int, count: int):
        """Blocking device-to-host copy of ``count`` doubles at a raw device address."""
        import numpy as np

        buf = np.empty(count, dtype=np.float64)
        self._ck(self.lib.madqp_memcpy_d2h(self.ctx, buf.ctypes.data, dev_ptr, count * 8))
        return buf
