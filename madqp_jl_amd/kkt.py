"""Host-side mirror of the MadNLP plugin surface for the MI355X path.

``HIPCondensedKKTSystem`` plays the role of ``NormalKKTSystem <:
MadNLP.AbstractKKTSystem`` (src/KKT/normalkkt.jl) and ``HIPCholeskySolver`` that
of the ``MadNLP.AbstractLinearSolver`` it owns (``MadNLP.LapackCPUSolver`` in
test/runtests.jl:151).  Method names follow the reference's generic functions
(``build_kkt!`` -> ``build_kkt`` ...); all arithmetic happens in
``libmadqp_hip.so``.  The Julia glue that binds the same C ABI is
``julia/MadQPHIP.jl``.
"""
from __future__ import annotations

from .backend import State


class HIPCholeskySolver:
    """AbstractLinearSolver contract: ctor keeps the matrix it re-reads at every
    ``factorize`` (src/KKT/normalkkt.jl:99-101); ``solve`` is in place (:196)."""

    def __init__(self, backend, kkt_handle):
        self.be, self._kkt = backend, kkt_handle
        self.info = 0

    def introduce(self) -> str:  # MadNLP.introduce, src/solver.jl:360
        return "madqp-hip blocked left-looking fp64 Cholesky (MFMA, gfx950)"

    def factorize(self):  # MadNLP.factorize!
        self.info = self.be.kkt_factorize(self._kkt)
        return self

    def is_factorized(self) -> bool:  # MadIPM.is_factorized, src/utils.jl:54-62
        return self.info == 0

    def is_inertia(self) -> bool:
        return True

    def inertia(self, n):  # (pos, zero, neg) of the SPD condensed matrix
        return (n, 0, 0) if self.info == 0 else (self.info - 1, 0, 1)


class HIPQuasiDefiniteSolver(HIPCholeskySolver):
    """The same blocked factorisation in quasi-definite mode (``madqp_chol_set_signature``): ``L diag(I, -I) L'`` of
    the augmented matrix, inertia (nx, 0, m) when every pivot has the expected sign."""

    def __init__(self, backend, kkt_handle, nx, m):
        super().__init__(backend, kkt_handle)
        self.nx, self.m = nx, m

    def introduce(self) -> str:
        return "madqp-hip blocked left-looking fp64 L diag(I,-I) L' (quasi-definite, MFMA, gfx950)"

    def inertia(self, n=None):
        return (self.nx, 0, self.m) if self.info == 0 else (0, 1, 0)


class HIPCondensedKKTSystem:
    """Dense condensed KKT system ``K = H + Sigma_x + A' Theta A`` on the device.

    Fields that MadIPM reads generically (``reg, pr_diag, du_diag, l_diag, u_diag,
    l_lower, u_lower, ind_lb, ind_ub, linear_solver``; src/kernels.jl:135-144,
    src/solver.jl:16-18) live in the shared :class:`State` and are exposed as
    attributes.
    """

    def __init__(self, backend, st: State, nx, ind_ineq, H, A):
        """``create_kkt_system`` (src/KKT/normalkkt.jl:29-126).  ``H``: (nx, nx) symmetric
        tensor or None (LP); ``A``: (m, nx) row-major tensor; both are borrowed."""
        self.be, self.st = backend, st
        self.nx, self.m = int(nx), st.m
        self.ind_ineq = [int(i) for i in ind_ineq]
        self.ns = len(self.ind_ineq)
        assert st.n == self.nx + self.ns
        self.H, self.A = H, A
        if H is not None:
            assert H.is_contiguous() and H.shape == (nx, nx)
        assert A.is_contiguous() and tuple(A.shape) == (self.m, self.nx)
        self._h = backend.kkt_create(self.nx, self.m, self.ind_ineq, H, max(self.nx, 1), A,
                                     max(self.nx, 1))
        self.linear_solver = HIPCholeskySolver(backend, self._h)
        self.n_factorizations = 0

    def close(self):
        if self._h is not None:
            self.be.kkt_destroy(self._h)
            self._h = None

    # generic field access, as MadIPM does on any AbstractKKTSystem
    reg = property(lambda s: s.st.reg)
    pr_diag = property(lambda s: s.st.pr_diag)
    du_diag = property(lambda s: s.st.du_diag)
    l_diag = property(lambda s: s.st.l_diag)
    u_diag = property(lambda s: s.st.u_diag)
    l_lower = property(lambda s: s.st.l_lower)
    u_lower = property(lambda s: s.st.u_lower)
    ind_lb = property(lambda s: s.st.ind_lb)
    ind_ub = property(lambda s: s.st.ind_ub)

    def num_variables(self):  # src/KKT/normalkkt.jl:128
        return self.st.n

    def is_inertia_correct(self, num_pos, num_zero, num_neg):  # :132-134
        return num_zero == 0 and num_pos == self.nx

    def initialize(self):  # MadNLP.initialize!(kkt), :136-147
        st, be = self.st, self.be
        for t, v in ((st.reg, 1.0), (st.pr_diag, 1.0), (st.du_diag, 0.0), (st.l_lower, 0.0),
                     (st.u_lower, 0.0), (st.l_diag, 1.0), (st.u_diag, 1.0)):
            be.fill(v, t)

    def set_aug_diagonal_reg(self, del_w, del_c):
        """``set_aug_diagonal_reg!(kkt, solver)`` -- dispatched on the KKT type (src/kernels.jl:128-146; the
        ``ScaledSparseKKTSystem`` method of :149-165 is :class:`HIPScaledAugmentedKKTSystem`)."""
        self.be.set_aug_diagonal_reg(self.st, del_w, del_c)

    def jtprod(self, out, y):  # MadNLP.jtprod!, :162-164
        self.be.kkt_jtprod(self._h, out, y)

    def build_kkt(self):  # MadNLP.build_kkt!, :166-180
        self.be.kkt_build(self._h, self.st)

    def factorize_wrapper(self):
        """MadNLP.factorize_wrapper! = build_kkt! then factorize! (src/linear_solver.jl:10)."""
        self.build_kkt()
        self.linear_solver.factorize()
        self.n_factorizations += 1

    def solve(self, w):  # MadNLP.solve!(kkt, w), :182-205
        self.be.kkt_solve(self._h, self.st, w)
        return w

    def set_refine(self, steps):
        """Refinement steps ``solve`` runs itself (``madqp_kkt_set_refine``; -1 = the AUTO rule by order).  For hosts whose
        loop calls ``solve!`` once (MadIPM's ``solve_system!``: what the Julia glue asks for); the drivers of this package
        refine in their own ``solve_system`` (option ``refine_steps``) and leave this at 0."""
        self.be.kkt_set_refine(self._h, steps)

    def mul(self, w, v, alpha=1.0, beta=0.0):  # MadNLP.mul!, :207-219
        self.be.kkt_mul(self._h, self.st, w, v, alpha, beta)
        return w

    def mul_solved(self, w, v, alpha=1.0, beta=0.0):
        """``mul`` for the residual check of solve_system! (src/linear_solver.jl:26-31): ``v`` is what the last
        ``solve`` returned, unmodified -- the condensed solve's own ``A dx`` is taken instead of a second pass over A
        (``madqp_kkt_mul_solved``: bitwise the result of ``mul``)."""
        self.be.kkt_mul(self._h, self.st, w, v, alpha, beta, solved=True)
        return w

    def eval_model(self, q, rhs, c0) -> float:
        """obj / grad! / cons! callbacks of the loop (src/solver.jl:166-169, 338-340)."""
        return self.be.kkt_eval(self._h, self.st, q, rhs, c0)


class HIPNormalKKTSystem(HIPCondensedKKTSystem):
    """The reference's own ``NormalKKTSystem`` (src/KKT/normalkkt.jl) on the device: normal equations
    ``A Sigma^-1 A'`` (m x m), LP only (:45-48); equality rows need no dual regularization.

    ``At``: (nx, m) tensor, row k = variable k contiguous (a Julia ``m x nx`` matrix as is); borrowed.
    """

    def __init__(self, backend, st: State, nx, ind_ineq, H, At):
        if H is not None:
            raise ValueError("The KKT system NormalKKTSystem supports only linear programs.")  # :45-48
        self.be, self.st = backend, st
        self.nx, self.m = int(nx), st.m
        self.ind_ineq = [int(i) for i in ind_ineq]
        self.ns = len(self.ind_ineq)
        assert st.n == self.nx + self.ns
        self.H, self.A, self.At = None, None, At
        assert At.is_contiguous() and tuple(At.shape) == (self.nx, self.m)
        self._h = backend.kkt_create_normal(self.nx, self.m, self.ind_ineq, At, max(self.m, 1))
        self.linear_solver = HIPCholeskySolver(backend, self._h)
        self.n_factorizations = 0

    def is_inertia_correct(self, num_pos, num_zero, num_neg):  # src/KKT/normalkkt.jl:132-134
        return num_zero == 0 and num_pos == self.m


class HIPAugmentedKKTSystem(HIPCondensedKKTSystem):
    """The K2 form of MadNLP's default ``SparseKKTSystem`` (src/utils.jl:108; the system the reference's tests
    compare everything against, test/runtests.jl:102-115,165-180) with the slack block eliminated, dense on
    the device: ``[H + Sigma_x, A'; A, -D]`` of order ``ceil128(nx) + m``, factorised as ``L diag(I, -I) L'``
    without pivoting (quasi-definite).  Equality rows enter exactly (``D_i = -dc_i``, which may be 0 when the
    equality rows are linearly independent) -- no ``-1/dc`` weights as in the condensed form.

    ``H``: (nx, nx) tensor, a 1-D tensor (diagonal of H) or None; ``A``: (m, nx) row-major tensor; borrowed.
    """

    def __init__(self, backend, st: State, nx, ind_ineq, H, A):
        self.be, self.st = backend, st
        self.nx, self.m = int(nx), st.m
        self.ind_ineq = [int(i) for i in ind_ineq]
        self.ns = len(self.ind_ineq)
        assert st.n == self.nx + self.ns
        self.H, self.A = H, A
        diag = H is not None and H.dim() == 1
        if H is not None:
            assert H.is_contiguous() and tuple(H.shape) in ((nx, nx), (nx,))
        assert A.is_contiguous() and tuple(A.shape) == (self.m, self.nx)
        self._h = backend.kkt_create_augmented(self.nx, self.m, self.ind_ineq, None if diag else H,
                                               max(self.nx, 1), A, max(self.nx, 1))
        if diag:
            backend.kkt_set_hdiag(self._h, H)
        self.linear_solver = HIPQuasiDefiniteSolver(backend, self._h, self.nx, self.m)
        self.n_factorizations = 0

    def is_inertia_correct(self, num_pos, num_zero, num_neg):  # src/KKT/normalkkt.jl:132-134, K2 inertia
        return num_zero == 0 and num_neg == self.m


class HIPScaledAugmentedKKTSystem(HIPAugmentedKKTSystem):
    """K2.5: MadNLP's ``ScaledSparseKKTSystem`` (``kkt_system=MadNLP.ScaledSparseKKTSystem`` in test/runtests.jl:95-115)
    on the dense quasi-definite path: the augmented matrix scaled symmetrically by ``sqrt((x - xl)(xu - x))`` per variable
    (scripts/cuda_wrapper.jl:90-116), ``l_diag = x - xl``, ``u_diag = xu - x`` positive (src/kernels.jl:149-165).  Same
    iterates as the K2 form; every entry of the matrix stays bounded as the iterates converge."""

    def __init__(self, backend, st: State, nx, ind_ineq, H, A):
        self.be, self.st = backend, st
        self.nx, self.m = int(nx), st.m
        self.ind_ineq = [int(i) for i in ind_ineq]
        self.ns = len(self.ind_ineq)
        assert st.n == self.nx + self.ns
        self.H, self.A = H, A
        if H is not None:
            assert H.is_contiguous() and tuple(H.shape) == (nx, nx)
        assert A.is_contiguous() and tuple(A.shape) == (self.m, self.nx)
        self._h = backend.kkt_create_scaled_augmented(self.nx, self.m, self.ind_ineq, H, max(self.nx, 1), A,
                                                      max(self.nx, 1))
        self.linear_solver = HIPQuasiDefiniteSolver(backend, self._h, self.nx, self.m)
        self.n_factorizations = 0

    def initialize(self):  # MadNLP.initialize!(kkt): also the scaling factor (library owned) = 1
        self.be.kkt_initialize(self._h, self.st)

    def set_aug_diagonal_reg(self, del_w, del_c):  # src/kernels.jl:149-165
        self.be.kkt_set_aug_diagonal_reg(self._h, self.st, del_w, del_c)


class _SparseMixin:
    """The Jacobian stays sparse (``DeviceCSR``), the factorised matrix stays dense (SURVEY.md 8f rank 1;
    reference: src/utils.jl:148-298, src/KKT/normalkkt.jl:51-101)."""

    def _init_sparse(self, backend, st, nx, ind_ineq, H, csr, mode):
        """``H``: None, a dense (nx, nx) tensor (condensed mode) or a 1-D tensor = diagonal of H (either mode)."""
        self.be, self.st = backend, st
        self.nx, self.m = int(nx), st.m
        self.ind_ineq = [int(i) for i in ind_ineq]
        self.ns = len(self.ind_ineq)
        assert st.n == self.nx + self.ns and (csr.m, csr.n) == (self.m, self.nx)
        self.H, self.A, self.csr = H, None, csr
        self._t_val = csr.t_val  # values of A' in CSR order, borrowed by the library
        diag = H is not None and H.dim() == 1
        self._h = backend.kkt_create_sparse(mode, self.nx, self.m, self.ind_ineq, None if diag else H,
                                            max(self.nx, 1), csr, self._t_val)
        if diag:
            assert H.is_contiguous() and H.numel() == self.nx
            backend.kkt_set_hdiag(self._h, H)
        self.linear_solver = HIPCholeskySolver(backend, self._h)
        self.n_factorizations = 0


class HIPSparseCondensedKKTSystem(_SparseMixin, HIPCondensedKKTSystem):
    def __init__(self, backend, st, nx, ind_ineq, H, csr):
        if H is not None and H.dim() == 2:
            assert H.is_contiguous() and H.shape == (nx, nx)
        self._init_sparse(backend, st, nx, ind_ineq, H, csr, 0)


class HIPSparseNormalKKTSystem(_SparseMixin, HIPNormalKKTSystem):
    """Normal equations A (H + Sigma)^-1 A' with H = 0 (the reference's LP-only NormalKKTSystem) or H diagonal
    (1-D tensor; SURVEY.md 8a-note -- what CONT-type QPs need)."""

    def __init__(self, backend, st, nx, ind_ineq, H, csr):
        if H is not None and H.dim() != 1:
            raise ValueError("The KKT system NormalKKTSystem supports only linear programs "
                             "(or a diagonal Hessian given as a vector).")  # normalkkt.jl:45-48
        self._init_sparse(backend, st, nx, ind_ineq, H, csr, 1)


class HIPSparseAugmentedKKTSystem(_SparseMixin, HIPAugmentedKKTSystem):
    """Augmented system with a sparse Jacobian (``DeviceCSR``) scattered into the dense quasi-definite matrix:
    the exact treatment of equality rows for a QP whose Hessian is dense (or diagonal, 1-D tensor)."""

    def __init__(self, backend, st, nx, ind_ineq, H, csr):
        if H is not None and H.dim() == 2:
            assert H.is_contiguous() and H.shape == (nx, nx)
        self._init_sparse(backend, st, nx, ind_ineq, H, csr, 2)
        self.linear_solver = HIPQuasiDefiniteSolver(backend, self._h, self.nx, self.m)
