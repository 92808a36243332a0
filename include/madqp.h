/*
 * madqp.h -- C ABI of libmadqp_hip.so: the MI355X (gfx950) implementation of the
 * Mehrotra predictor-corrector KKT hot path of MadIPM / MadQP.jl.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes
 * plain pointers and sizes, returns an int32 status and never throws or aborts:
 *     0  MADQP_OK
 *    <0  usage / runtime error (text via madqp_last_error)
 *    >0  numerical condition (not positive definite at column j, NaN detected)
 *
 * Conventions
 *   - all `double*` / `int64_t*` arguments are DEVICE pointers unless the name ends in
 *     `_host`; scalar results are returned through host pointers (the call synchronises
 *     the context's stream when it returns a scalar; calls that return nothing are
 *     asynchronous and stream ordered).
 *   - symmetric matrices / Cholesky factors: column-major, lower triangle, LAPACK 'L'
 *     (element (i,j), i>=j, at  ptr[i + j*ld]) -- byte-identical to a Julia `Matrix` handed
 *     to `LAPACK.potrf!('L', ...)`.
 *   - the constraint matrix A (m x nx) is given with ROW k contiguous:  A[k*lda + i]
 *     (a numpy C-order array; from Julia pass `permutedims(A)`), because the MFMA kernels
 *     stage k-major tiles whose fast index is the variable index.
 *   - index vectors are 0-based int64 (MadNLP's are 1-based Int: the Julia glue passes
 *     `ind .- 1`).
 *   - one context = one device + one HIP stream; calls on a context are stream ordered and
 *     must come from one host thread at a time (the reference loop is single threaded,
 *     src/solver.jl:254-345).
 *
 * Each function names the reference interface (file:line under /root/reference) it replaces.
 */
#ifndef MADQP_H
#define MADQP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADQP_OK 0
#define MADQP_ERR_ARG (-1)
#define MADQP_ERR_HIP (-2)
#define MADQP_ERR_ALLOC (-3)
#define MADQP_ERR_STATE (-4)
#define MADQP_NUM_NAN 1000000001 /* NaN detected (src/linear_solver.jl:41-43) */

typedef struct madqp_ctx madqp_ctx;
typedef struct madqp_chol madqp_chol;
typedef struct madqp_kkt madqp_kkt;

/* ------------------------------------------------------------------ context */
int32_t madqp_version(void);
/* stream: a hipStream_t to launch on (NULL = the device's default stream). */
int32_t madqp_ctx_create(int32_t device, void* stream, madqp_ctx** out);
int32_t madqp_ctx_destroy(madqp_ctx* ctx);
const char* madqp_last_error(madqp_ctx* ctx);
int32_t madqp_ctx_sync(madqp_ctx* ctx);
/* Device faults.  The triangular sweeps of madqp_chol_solve hand solved blocks from workgroup to workgroup; a
 * consumer that waits longer than its spin limit sets a fault word on the device and finishes with NaNs.  The word
 * travels with every scalar read-back (madqp_norm_inf3, madqp_get_alpha_max, ...): that call then returns
 * MADQP_ERR_HIP ("triangular sweep hand-off timed out") and clears the word -- a hardware / scheduling fault is not
 * reported as the numerical failure MadNLP.SolveException stands for (src/linear_solver.jl:41-43).
 * madqp_debug_inject_fault sets the word (tests). */
int32_t madqp_debug_inject_fault(madqp_ctx* ctx);
/* device memory helpers for hosts without their own allocator (Julia glue, C++) */
int32_t madqp_malloc(madqp_ctx* ctx, size_t bytes, void** out);
int32_t madqp_free(madqp_ctx* ctx, void* ptr);
int32_t madqp_memcpy_h2d(madqp_ctx* ctx, void* dst, const void* src_host, size_t bytes);
int32_t madqp_memcpy_d2h(madqp_ctx* ctx, void* dst_host, const void* src, size_t bytes);

/* Per-kernel-class device timers (hipEvent pairs on the context's stream). */
enum {
    MADQP_PROF_SYRK = 0,      /* gemm core, assembly instance            */
    MADQP_PROF_POTRF_GEMM,    /* gemm core, left-looking panel updates   */
    MADQP_PROF_POTRF_DIAG,    /* 128x128 diagonal block factor + inverse */
    MADQP_PROF_POTRF_TRSM,    /* gemm core, panel times inverse block    */
    MADQP_PROF_TRSV,          /* triangular sweeps of the solves         */
    MADQP_PROF_GEMV,
    MADQP_PROF_VEC,           /* fused vector / reduction kernels        */
    MADQP_PROF_COUNT
};
/* mask: bit (1 << class) enables the timers of that class; 0 = off, 0x7F = all.  Timing the
 * launch-bound classes (TRSV, VEC) perturbs them; the MFMA classes are safe to time always. */
int32_t madqp_prof_enable(madqp_ctx* ctx, int32_t mask);
int32_t madqp_prof_reset(madqp_ctx* ctx);
/* total device milliseconds and launch count of a class since the last reset (syncs) */
int32_t madqp_prof_get(madqp_ctx* ctx, int32_t cls, double* ms_host, int64_t* launches_host);

/* Hardware probe: back-to-back v_mfma_f64_16x16x4_f64 on every SIMD (no memory traffic);
 * tflops_host = measured dense fp64 MFMA rate of this device, the ceiling the GEMM core is
 * judged against. */
int32_t madqp_probe_mfma_f64(madqp_ctx* ctx, int32_t iters, double* tflops_host);

/* ------------------------------------------------------------ synthetic data */
/* out[i] = g(key, idx0 + i): the position-addressable generator of oracle/qp.py */
int32_t madqp_gen_normal(madqp_ctx* ctx, uint64_t key, uint64_t idx0, int64_t count, double* out);
/* dense symmetric "wigner" H (full storage): oracle/qp.py:gen_H_wigner */
int32_t madqp_gen_wigner(madqp_ctx* ctx, uint64_t key, int64_t n, double inv_sqrt_n, double* H,
                         int64_t ld);

/* the same data as the block-cyclic pieces one rank of a P x Q grid holds (madqp_dist_*, madqp_dkkt_*): local column c of
 * a direction with modulus R and residue r is global column ((c / nb) * R + r) * nb + c % nb.
 * gen_normal_cyclic: out[k*ldo + c] = A[k, global(c)] (m rows; entries beyond nx are 0);
 * gen_wigner_cyclic: H[i + j*ld] = H[global_P(i), global_Q(j)], all local tiles complete (both triangles). */
int32_t madqp_gen_normal_cyclic(madqp_ctx* ctx, uint64_t key, int64_t m, int64_t nx, int64_t nb, int32_t R, int32_t r,
                                int64_t ncols_local, double* out, int64_t ldo);
int32_t madqp_gen_wigner_cyclic(madqp_ctx* ctx, uint64_t key, int64_t n, double inv_sqrt_n, int64_t nb, int32_t P,
                                int32_t p, int32_t Q, int32_t q, int64_t mloc, int64_t nloc, double* H, int64_t ld);

/* ------------------------------------------------------- dense linear algebra */
/* C(lower) = base(lower, may be NULL) + diag(dvec, may be NULL) + B' diag(w) B
 * B: kdim rows of length n (row k at B + k*ldb); w may be NULL (= ones).
 * Replaces assemble_normal_system! (src/utils.jl:266-298) / build_kkt! (src/KKT/normalkkt.jl:166-180)
 * for dense data: the SYRK of SURVEY.md 8a-2.  MFMA v_mfma_f64_16x16x4_f64. */
int32_t madqp_syrk_assemble(madqp_ctx* ctx, int64_t n, int64_t kdim, const double* B, int64_t ldb,
                            const double* w, const double* base, int64_t ldbase,
                            const double* dvec, double* C, int64_t ldc);

/* Cholesky linear solver: replaces MadNLP.LapackCPUSolver / the AbstractLinearSolver
 * contract used at src/KKT/normalkkt.jl:99-101,196 and src/linear_solver.jl:10-11. */
int32_t madqp_chol_create(madqp_ctx* ctx, int64_t n, madqp_chol** out);
int32_t madqp_chol_destroy(madqp_chol* s);
/* Quasi-definite mode (the inertia MadNLP's K2 systems have, is_inertia_correct at src/KKT/normalkkt.jl:132-134
 * generalised to (npos, 0, n - npos)): the matrix given to madqp_chol_factor holds [P, .; B, Q] (lower triangle,
 * P of order npos, P and Q positive definite) and stands for M = [P, B'; B, -Q]; factor computes M = L diag(I, -I) L',
 * solve applies M^-1.  npos: a multiple of 128, or n (plain Cholesky, the default).  info as for Cholesky
 * (the first column whose pivot is not positive). */
int32_t madqp_chol_set_signature(madqp_chol* s, int64_t npos);
/* MadNLP.factorize!: in-place lower Cholesky of the n x n matrix at A (blocked left-looking,
 * MFMA panel updates).  info_host: 0 = success, j>0 = leading minor of order j not positive
 * definite (LAPACK dpotrf convention; maps to is_factorized, src/utils.jl:54-62). */
int32_t madqp_chol_factor(madqp_chol* s, double* A, int64_t lda, int32_t* info_host);
/* MadNLP.solve!(linear_solver, rhs): rhs <- (L L')^-1 rhs in place (two triangular sweeps) */
int32_t madqp_chol_solve(madqp_chol* s, double* rhs);

/* y = alpha*op(A) x + beta*y ; A has `rows` rows of length `cols`, row r at A + r*lda.
 * trans=0: y(rows) = A x(cols);  trans=1: y(cols) = A' x(rows).
 * Replaces mul!(y, kkt.AT, x) / mul!(y, kkt.AT', x) (src/KKT/normalkkt.jl:162-164,194,200,214-215). */
int32_t madqp_gemv(madqp_ctx* ctx, int32_t trans, int64_t rows, int64_t cols, double alpha,
                   const double* A, int64_t lda, const double* x, double beta, double* y);

/* ----------------------------------------------------------- solver state view */
/* Device-pointer view of MPCSolver (src/structure.jl:1-75) + the diagonal fields of the
 * KKT system that MadIPM reads generically (src/kernels.jl:135-144).  d and p are
 * UnreducedKKTVector.values: [x(n) | y(m) | zl(nlb) | zu(nub)] contiguous. */
typedef struct madqp_state {
    int64_t n, m, nlb, nub;
    const int64_t* ind_lb; /* nlb, 0-based, strictly increasing */
    const int64_t* ind_ub; /* nub */
    /* (The two lists are the problem's bound pattern: their CONTENTS do not change while a KKT object that has seen
     * them lives -- the condensed KKT system keeps their inverse, csrc/kkt.hip: ensure_pos, keyed on the pointers and
     * lengths; hand a different pattern over in different arrays.) */
    double *x, *xl, *xu, *zl, *zu, *f; /* n */
    double *y, *c;                     /* m */
    double* jacl;                      /* n */
    double *d, *p;                     /* n+m+nlb+nub */
    double *correction_lb, *correction_ub;
    double *reg, *pr_diag; /* n */
    double* du_diag;       /* m */
    double *l_diag, *l_lower; /* nlb */
    double *u_diag, *u_lower; /* nub */
} madqp_state;

/* ---------------------------------------------- src/kernels.jl, fused on device */
/* set_aug_diagonal_reg! (src/kernels.jl:128-146) */
int32_t madqp_set_aug_diagonal_reg(madqp_ctx* ctx, const madqp_state* st, double del_w,
                                   double del_c);
/* set_initial_primal_rhs! / set_initial_dual_rhs! (src/kernels.jl:1-19) */
int32_t madqp_set_initial_primal_rhs(madqp_ctx* ctx, const madqp_state* st);
int32_t madqp_set_initial_dual_rhs(madqp_ctx* ctx, const madqp_state* st);
/* set_predictive_rhs! (src/kernels.jl:21-41) */
int32_t madqp_set_predictive_rhs(madqp_ctx* ctx, const madqp_state* st);
/* set_correction_rhs! (src/kernels.jl:43-61) */
int32_t madqp_set_correction_rhs(madqp_ctx* ctx, const madqp_state* st, double mu);
/* get_correction! (src/kernels.jl:63-75) */
int32_t madqp_get_correction(madqp_ctx* ctx, const madqp_state* st);
/* set_extra_correction! (src/kernels.jl:78-126) */
int32_t madqp_set_extra_correction(madqp_ctx* ctx, const madqp_state* st, double alpha_p,
                                   double alpha_d, double beta_min, double beta_max, double mu);
/* get_complementarity_measure (src/kernels.jl:171-190) */
int32_t madqp_get_complementarity_measure(madqp_ctx* ctx, const madqp_state* st, double* mu_host);
/* get_affine_complementarity_measure (src/kernels.jl:192-224) */
int32_t madqp_get_affine_complementarity_measure(madqp_ctx* ctx, const madqp_state* st,
                                                 double alpha_p, double alpha_d,
                                                 double* mu_host);
/* get_alpha_max_primal + get_alpha_max_dual (src/kernels.jl:242-288) in one launch.
 * alpha_host[4] = (alpha_xl, alpha_xu, alpha_zl, alpha_zu); iblock_host[4] = 0-based blocking
 * index into the lb/ub lists, -1 when nothing blocks (the reference's init (1.0, 0)).
 * Exact ties resolve to the LAST index among the minima: src/kernels.jl:248 compares `a[1] < b[1] ? a : b` and mapreduce
 * folds from the left, so a later element with the same ratio replaces the earlier one (tests/test_gpu_kernels.py folds
 * the reference's reducer literally). */
int32_t madqp_get_alpha_max(madqp_ctx* ctx, const madqp_state* st, double tau,
                            double* alpha_host, int64_t* iblock_host);
/* axpy! x4 of src/solver.jl:332-335 */
int32_t madqp_update_iterates(madqp_ctx* ctx, const madqp_state* st, double alpha_p,
                              double alpha_d);
/* MadNLP.get_inf_pr / get_inf_du / get_inf_compl as called at src/solver.jl:264-272,
 * src/kernels.jl:435-446: out_host[3] = (||c||inf, ||f - zl + zu + jacl||inf, max compl product) */
int32_t madqp_get_inf(madqp_ctx* ctx, const madqp_state* st, double* out_host);
/* MadNLP.adjust_boundary! (src/solver.jl:342) */
int32_t madqp_adjust_boundary(madqp_ctx* ctx, const madqp_state* st, double mu);
/* MadNLP.reduce_rhs! / finish_aug_solve! / _kktmul! on an UnreducedKKTVector
 * (src/KKT/normalkkt.jl:183,203,217) */
int32_t madqp_reduce_rhs(madqp_ctx* ctx, const madqp_state* st, double* w);
int32_t madqp_finish_aug_solve(madqp_ctx* ctx, const madqp_state* st, double* w);
int32_t madqp_kktmul(madqp_ctx* ctx, const madqp_state* st, double* w, const double* v,
                     double alpha, double beta);
/* the three inf-norms of solve_system! (src/linear_solver.jl:29-35):
 * out_host[3] = (||a||inf, ||b||inf, ||c||inf) over `len` entries each; NaN propagates. */
int32_t madqp_norm_inf3(madqp_ctx* ctx, int64_t len, const double* a, const double* b,
                        const double* c, double* out_host);
int32_t madqp_norm_inf(madqp_ctx* ctx, int64_t len, const double* a, double* out_host);
int32_t madqp_axpy(madqp_ctx* ctx, int64_t len, double alpha, const double* x, double* y);
int32_t madqp_copy(madqp_ctx* ctx, int64_t len, const double* src, double* dst);
int32_t madqp_fill(madqp_ctx* ctx, int64_t len, double value, double* dst);

/* ---------------------------------- init_starting_point! (src/solver.jl:6-125) */
/* multipliers from res = A'y + f by bound type (src/solver.jl:41-66); res is st->jacl */
int32_t madqp_sp_init_duals(madqp_ctx* ctx, const madqp_state* st);
/* out_host[4] = min(0, min(x_lr-xl_r)), min(0, min(xu_r-x_ur)), min(0, min zl_r), min(0, min zu_r)
 * (src/solver.jl:68-78) */
int32_t madqp_sp_mins(madqp_ctx* ctx, const madqp_state* st, double* out_host);
/* x_lr += dx; then x_ur -= dx; zl_r += dz; zu_r += dz (src/solver.jl:80-83, 96-99);
 * the two primal passes are ordered so the aliasing of x_lr / x_ur is reproduced */
int32_t madqp_sp_shift(madqp_ctx* ctx, const madqp_state* st, double dx, double dz);
/* out_host[8] = dot(x_lr,zl_r), dot(xl_r,zl_r), dot(xu_r,zu_r), dot(x_ur,zu_r),
 *               sum(zl_r), sum(zu_r), sum(x_lr-xl_r), sum(xu_r-x_ur)   (src/solver.jl:85-94) */
int32_t madqp_sp_sums(madqp_ctx* ctx, const madqp_state* st, double* out_host);
/* Ipopt-style projection (src/solver.jl:101-118) */
int32_t madqp_sp_project(madqp_ctx* ctx, const madqp_state* st, double kappa);
/* the four asserts (src/solver.jl:120-123): ok_host = 1 when all hold */
int32_t madqp_sp_check(madqp_ctx* ctx, const madqp_state* st, int32_t* ok_host);

/* ------------------------------------------------- condensed dense KKT system */
/* HIPCondensedKKTSystem: K = H + Sigma_x + A' Theta A  (SURVEY.md 8a-note), modelled on
 * NormalKKTSystem (src/KKT/normalkkt.jl:1-27).  H (nx x nx, full symmetric storage, may be
 * NULL for an LP) and A (m x nx, row k contiguous) are BORROWED device pointers that must
 * outlive the object; K, its factor and all work vectors are owned by the library.
 * ind_ineq_host: ns 0-based rows that carry a slack (host pointer, copied). */
int32_t madqp_kkt_create(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                         const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                         const double* A, int64_t lda, madqp_kkt** out);
/* The reference's own formulation, NormalKKTSystem (src/KKT/normalkkt.jl:29-126): normal equations
 * S = A Sigma_x^-1 A' + diag(Sigma_s^-1) (m x m), LP only (:45-48), equality rows need no
 * regularization.  At: nx x m with row k (variable k) contiguous = a Julia m x nx `Matrix` A as is;
 * borrowed.  The object answers the same madqp_kkt_* calls below. */
int32_t madqp_kkt_create_normal(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                const int64_t* ind_ineq_host, const double* At, int64_t ldat,
                                madqp_kkt** out);
/* The K2 form of MadNLP's default SparseKKTSystem (src/utils.jl:108; augmented system of
 * test/runtests.jl:102-115,165-180) with the slack block eliminated: [H + Sigma_x, A'; A, -D] of order
 * ceil128(nx) + m, D_i = 1/Sigma_s,k - dc_i (inequality row) or -dc_i (equality row).  Quasi-definite:
 * factorised as L diag(I, -I) L' without pivoting (madqp_chol_set_signature); equality rows need no dual
 * regularization when they are linearly independent.  Operands as madqp_kkt_create (H may be NULL; a
 * diagonal Hessian goes through madqp_kkt_set_hdiag).  Answers the same madqp_kkt_* calls. */
int32_t madqp_kkt_create_augmented(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                   const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                   const double* A, int64_t lda, madqp_kkt** out);
/* K2.5: MadNLP's ScaledSparseKKTSystem as MadIPM drives it (set_aug_diagonal_reg! src/kernels.jl:149-165, the scaling
 * of the augmented matrix scripts/cuda_wrapper.jl:90-116, K2.5 == K2 in test/runtests.jl:95-115) on the dense
 * quasi-definite path: the matrix of madqp_kkt_create_augmented scaled symmetrically by sqrt((x - xl)(xu - x)) per
 * variable, with l_diag = x - xl > 0, u_diag = xu - x > 0 and pr_diag = scaling^2 (del_w + Sigma).  The per-variable
 * state must be set through madqp_kkt_set_aug_diagonal_reg / madqp_kkt_initialize. */
int32_t madqp_kkt_create_scaled_augmented(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                          const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                          const double* A, int64_t lda, madqp_kkt** out);
int32_t madqp_kkt_destroy(madqp_kkt* kkt);
/* set_aug_diagonal_reg!(kkt, solver) dispatched on the KKT type, as the reference does (src/kernels.jl:128 for any
 * AbstractKKTSystem = madqp_set_aug_diagonal_reg; :149 for ScaledSparseKKTSystem, with MadNLP._set_aug_diagonal!) */
int32_t madqp_kkt_set_aug_diagonal_reg(madqp_kkt* kkt, const madqp_state* st, double del_w, double del_c);
/* MadNLP.initialize!(kkt) (src/KKT/normalkkt.jl:136-147; K2.5: also scaling factor = 1) */
int32_t madqp_kkt_initialize(madqp_kkt* kkt, const madqp_state* st);
/* MadNLP.build_kkt! (src/KKT/normalkkt.jl:166-180): Theta from pr_diag/du_diag, then the SYRK */
int32_t madqp_kkt_build(madqp_kkt* kkt, const madqp_state* st);
/* MadNLP.factorize!(kkt.linear_solver); info as madqp_chol_factor */
int32_t madqp_kkt_factorize(madqp_kkt* kkt, int32_t* info_host);
/* MadNLP.solve!(kkt, w) (src/KKT/normalkkt.jl:182-205): reduce, condense, two triangular
 * sweeps, decondense, finish; w is an UnreducedKKTVector.values, solved in place */
int32_t madqp_kkt_solve(madqp_kkt* kkt, const madqp_state* st, double* w);
/* Steps of iterative refinement that madqp_kkt_solve runs ITSELF: w = K^-1 p, then w += K^-1 (p - K w) per step, with
 * madqp_kkt_mul for the residual.  0 (default): none -- MadNLP.solve!(kkt, w) as src/KKT/normalkkt.jl:182-205 states it.
 * -1: the AUTO rule -- one step while the factorised matrix has order <= 1024 (MADQP_REFINE_AUTO_MAX), none above.  For
 * hosts whose loop is not ours: MadIPM's solve_system! (src/linear_solver.jl:19-45) calls solve! once and only looks at the
 * residual, so the Julia glue asks for -1; the drivers of this repository refine in their own solve_system (option
 * refine_steps, same rule) and leave this at 0.  Why: DESIGN.md section 4.2 (16 x 16 inverse products against LAPACK's scalar
 * substitution on small ill-conditioned problems). */
int32_t madqp_kkt_set_refine(madqp_kkt* kkt, int32_t steps);
/* MadNLP.mul!(w, kkt, v, alpha, beta) (src/KKT/normalkkt.jl:207-219), with H for a QP */
int32_t madqp_kkt_mul(madqp_kkt* kkt, const madqp_state* st, double* w, const double* v,
                      double alpha, double beta);
/* The same product for the residual check of solve_system! (src/linear_solver.jl:26-31: solve!(kkt, d), then
 * mul!(w, kkt, d, -1, 1)): v must be the vector the LAST madqp_kkt_solve on this object returned, unmodified since.
 * The condensed solve ends with A dx (its decondensation needs it); this call takes that product instead of streaming
 * A a second time -- same kernel, same operands, bitwise the result of madqp_kkt_mul.  Other KKT forms, or a solve
 * that left nothing to reuse: identical to madqp_kkt_mul. */
int32_t madqp_kkt_mul_solved(madqp_kkt* kkt, const madqp_state* st, double* w, const double* v,
                             double alpha, double beta);
/* MadNLP.jtprod!(out, kkt, y) (src/KKT/normalkkt.jl:162-164): out(n) = [A' y ; -y[ind_ineq]] */
int32_t madqp_kkt_jtprod(madqp_kkt* kkt, double* out, const double* y);
/* model callbacks of the loop (src/solver.jl:166-169,338-340; formulas scripts/qp_gpu.jl:29-40):
 * f = [H x + q ; 0], c = A x - s - rhs, obj_host = c0 + q'x + x'Hx/2 (scaled data) */
int32_t madqp_kkt_eval(madqp_kkt* kkt, const madqp_state* st, const double* q, const double* rhs,
                       double c0, double* obj_host);
/* device pointer to the assembled / factored K (nx x nx, ld = madqp_kkt_ld) for inspection */
int32_t madqp_kkt_matrix(madqp_kkt* kkt, double** K, int64_t* ld);

/* ----------------------------------------- sparse-A front end */
/* create_kkt_system with the Jacobian kept sparse, as the reference does (coo_to_csr src/utils.jl:148-197,
 * src/KKT/normalkkt.jl:51-101): A in CSR (a_*: m rows) and A' in CSR (at_*: nx rows = the CSC of A), device
 * int64 / double arrays, column indices ascending within a row, borrowed.  mode 0: condensed
 * K = H + Sigma_x + A' Theta A (H dense or NULL); mode 1: normal equations A Sigma^-1 A' (LP only,
 * assemble_normal_system! src/utils.jl:266-298); mode 2: the augmented system of madqp_kkt_create_augmented
 * (the CSR entries are scattered into its constraint rows).  The factorised matrix stays dense; every madqp_kkt_*
 * call works on the returned object (products with A / A' become CSR mat-vecs). */
int32_t madqp_kkt_create_sparse(madqp_ctx* ctx, int32_t mode, int64_t nx, int64_t m, int64_t ns,
                                const int64_t* ind_ineq_host, const double* H, int64_t ldh, const int64_t* a_ptr,
                                const int64_t* a_col, const double* a_val, const int64_t* at_ptr,
                                const int64_t* at_col, const double* at_val, madqp_kkt** out);

/* H = diag(hdiag) (nx entries, device, borrowed) for a KKT object created without a dense H, either mode:
 * condensed K = diag(hdiag) + Sigma_x + A' Theta A; normal equations A (H + Sigma)^-1 A' -- the diagonal-H
 * extension of the LP-only NormalKKTSystem (src/KKT/normalkkt.jl:45-48) that CONT-type QPs need. */
int32_t madqp_kkt_set_hdiag(madqp_kkt* kkt, const double* hdiag);

/* ----------------------------------------- callback buffers -> dense operands */
/* MadNLP.SparseCallback hands the Jacobian / Hessian values as nnz-long buffers in the order of the model's COO
 * pattern (get_jacobian / get_hessian, src/KKT/normalkkt.jl:129-130, filled at src/solver.jl:167,170).
 * compress_jacobian! (src/KKT/normalkkt.jl:149-158, through A_csr_map) and compress_hessian! (the scatter-add
 * transfer! of scripts/cuda_wrapper.jl:9-34) move them into the matrix storage; for the DENSE operands of this
 * library that is a madqp_coo_map: built once from the pattern (I_host, J_host: 1-based, as MadNLP keeps them),
 * applied after every callback evaluation.  apply: dst(i, j) at dst[i*ld + j] (row i contiguous -- A as
 * madqp_kkt_create wants it) = sum of the values whose pattern entry is (i+1, j+1), added in COO order by ONE lane
 * (no atomics, reproducible); every other entry of the nrows x ncols target is set to 0.  symmetric != 0: the pattern
 * is one triangle of a symmetric matrix (MadNLP's Hessians are lower triangular) and both halves are written. */
typedef struct madqp_coo_map madqp_coo_map;
int32_t madqp_coo_map_create(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host, const int32_t* J_host,
                             int64_t nrows, int64_t ncols, int32_t symmetric, madqp_coo_map** out);
/* The same for the pieces one rank (p, q) of a P x Q grid holds of a QP shared by all ranks (madqp_dkkt_create):
 *   cols_cyclic   Jacobian entries whose column lies in a tile (width nb) of residue r mod R -> dst[i*ld + local
 *                 column]; (R, r) = (P, p) gives A_I, (Q, q) gives A_J; all other entries are dropped;
 *   tiles_cyclic  the lower tiles (I, J), I = p mod P, J = q mod Q, of a symmetric Hessian whose pattern is one
 *                 triangle -> dst[local column * ld + local row] (the layout of the local K); diagonal tiles get both
 *                 triangles.
 * apply as above (target: nrows x local columns, resp. local columns x local rows). */
int32_t madqp_coo_map_create_cols_cyclic(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host, const int32_t* J_host,
                                         int64_t nrows, int64_t ncols, int64_t nb, int32_t R, int32_t r,
                                         madqp_coo_map** out);
int32_t madqp_coo_map_create_tiles_cyclic(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host, const int32_t* J_host,
                                          int64_t n, int64_t nb, int32_t P, int32_t p, int32_t Q, int32_t q,
                                          madqp_coo_map** out);
int32_t madqp_coo_map_apply(madqp_coo_map* map, const double* vals, double* dst, int64_t ld);
int32_t madqp_coo_map_destroy(madqp_coo_map* map);

/* ----------------------------------------- factorisation pieces */
/* One block column ("panel": start and width multiples of 128, the last one may be short) of a matrix the caller owns:
 * the tile factorisation inside the P x Q distributed Cholesky below is built from these (csrc/dist.hip), and a host
 * that schedules panels itself can use them the same way.  All calls are asynchronous on the context's stream. */
/* the linear solver object of a KKT system and the order of its matrix (borrowed handle) */
int32_t madqp_kkt_chol(madqp_kkt* kkt, madqp_chol** chol, int64_t* order);
/* start a factorisation of the column-major lower matrix A (clears info) */
int32_t madqp_chol_factor_begin(madqp_chol* s, double* A, int64_t lda);
/* factor columns [j0, j0+w), which already carry the updates of all columns < j0 (all rows below too) */
int32_t madqp_chol_factor_panel(madqp_chol* s, int64_t j0, int64_t w);
/* packed image of a factored panel: [info, 0 | inverse diagonal blocks | L[j0:n, j0:j0+w]],
 * 2 + ceil(w/128) * 2 * 128 * 128 + w * (n - j0) doubles */
int32_t madqp_chol_panel_pack(madqp_chol* s, int64_t j0, int64_t w, double* buf);

/* ----------------------------------------- one dense KKT matrix on a P x Q grid of GPUs */
/* 2-D block-cyclic distributed Cholesky + triangular solves, one process per GPU: the AbstractLinearSolver contract
 * (factorize! / solve!, src/KKT/normalkkt.jl:99-101,196; src/linear_solver.jl:10) for a matrix that no longer fits --
 * or is no longer worth factorising on -- one GPU.  The reference has no counterpart (single process).
 *   grid     rank = p*Q + q, 0 <= p < P, 0 <= q < Q, P*Q = world (8 GPUs: 2 x 4);
 *   layout   tile (I, J) of the lower triangle (nb x nb, nb a multiple of 128) lives on rank (I mod P, J mod Q) at
 *            position (I div P, J div Q) of that rank's column-major local matrix (madqp_dist_matrix): the caller (or
 *            madqp_dkkt_*) fills the local tiles with K, factor overwrites them with L;
 *   factor   over tile columns with look-ahead 1 in the panel phase: diagonal tile -> broadcast down its process column ->
 *            panel solves -> the panel is broadcast along process rows and, transposed, down process columns; the
 *            UPDATES are left-looking and lazy: a rank keeps the operands of all steps and brings a tile column up to
 *            date in one wide-K MFMA GEMM when it is needed (csrc/dist_core.inc); info as LAPACK dpotrf, identical on
 *            all ranks;
 *   solve    rhs: n doubles, replicated on every rank, overwritten with the solution.  Forward and backward sweeps
 *            over GROUPS of G tiles (G nb = 4096 rows by default): during the factorisation the G x G tile triangle on
 *            the diagonal of each group is collected on one rank (group g -> rank g mod world; point-to-point, <= G
 *            tiles per step, behind the operand broadcasts), so a sweep costs one reduction of the group's partial sums
 *            to that rank, one local triangular solve of order G nb and one broadcast per GROUP, not per tile.
 * Collectives: RCCL over xGMI -- rank 0 draws an id with madqp_dist_unique_id and the caller ships those 128 bytes to
 * every rank (any channel), create builds the world / row / column communicators; the calls run on internal streams
 * of the context, ordered after what the context's stream held at the call and finished (joined) before it goes on.
 * For rehearsals with several ranks on one GPU (RCCL refuses that) the caller passes host-staged collectives instead
 * (madqp_comm_ops; tests use torch.distributed gloo): group 0 = world, 1 = my process row, 2 = my process column;
 * root = rank inside the group (q in a row group, p in a column group); send / recv: blocking point-to-point between
 * world ranks (never to oneself). */
typedef struct madqp_dist madqp_dist;
typedef struct madqp_comm_ops {
    void* user;
    int32_t (*bcast)(void* user, void* host_buf, int64_t bytes, int32_t root, int32_t group);
    int32_t (*reduce_sum)(void* user, double* host_buf, int64_t count, int32_t root, int32_t group);
    int32_t (*allreduce_sum)(void* user, double* host_buf, int64_t count, int32_t group);
    int32_t (*send)(void* user, const void* host_buf, int64_t bytes, int32_t dst_world_rank);
    int32_t (*recv)(void* user, void* host_buf, int64_t bytes, int32_t src_world_rank);
} madqp_comm_ops;
int32_t madqp_dist_unique_id(madqp_ctx* ctx, void* id128);
/* nccl_id128: the 128 bytes of madqp_dist_unique_id (ignored when world == 1 or ops != NULL); ops: NULL = RCCL */
int32_t madqp_dist_create(madqp_ctx* ctx, int32_t rank, int32_t world, int32_t P, int32_t Q, int64_t n, int64_t nb,
                          const void* nccl_id128, const madqp_comm_ops* ops, madqp_dist** out);
int32_t madqp_dist_destroy(madqp_dist* d);
/* out8 = (p, q, local tile rows, local tile columns, local rows, local columns, leading dimension, padded columns) */
int32_t madqp_dist_layout(madqp_dist* d, int64_t* out8);
int32_t madqp_dist_matrix(madqp_dist* d, double** Kloc, int64_t* ld);
int32_t madqp_dist_factor(madqp_dist* d, int32_t* info_host);
int32_t madqp_dist_solve(madqp_dist* d, double* rhs);
int32_t madqp_dist_bytes_sent(madqp_dist* d, int64_t* bytes_host);
/* who carries the collectives, as the library sees it: out8 = (backend: 0 = one rank, 1 = RCCL, 2 = madqp_comm_ops |
 * ranks in the world communicator (RCCL: ncclCommCount) | in my process-row communicator | in my process-column
 * communicator | my rank in the world communicator | workgroup slots left to the collectives' kernels | internal
 * streams (0 / 2) | 0).  bench.py puts it into every N > 1 line: a run labelled N GPUs ran its collectives on N ranks. */
int32_t madqp_dist_comm_info(madqp_dist* d, int64_t* out8);
/* device bytes the handle holds: out8 = (total | local matrix | stored row operands | stored column operands | solve
 * bands | broadcast images, staging, vectors | levels of the operand staircases | 0).  The total is known before
 * anything is allocated: madqp_dist_create returns MADQP_ERR_ALLOC with these figures in madqp_last_error when the
 * device has less free memory than that. */
int32_t madqp_dist_memory(madqp_dist* d, int64_t* out8);

/* The condensed KKT system K = H + Sigma_x + A' Theta A (madqp_kkt_create) with K on the grid of `d` (order nx): the
 * same plugin methods -- build_kkt! / factorize! / solve! / mul! / jtprod! (src/KKT/normalkkt.jl:162-219) and the
 * model callbacks (src/solver.jl:166-169,338-340) -- for one QP shared by all ranks.  A rank holds (borrowed, device):
 *   Hloc  its tiles of H in the layout of the local K (ldh >= leading dimension of madqp_dist_layout; lower tiles,
 *         diagonal tiles complete), or NULL for an LP;
 *   A_I   the columns of A of its tile ROWS:    ceil16(m) rows of length ld_ai >= ld, row k contiguous, zero padded;
 *   A_J   the columns of A of its tile COLUMNS: ceil16(m) rows of length ld_aj >= padded column count, zero padded
 * i.e. 2/(PQ) of H and K and (1/P + 1/Q) of A; with the operands the factorisation stores for its lazy updates --
 * (1/P + 1/Q) n^2/2 doubles, the factor replicated Q-fold along process rows and P-fold along columns -- C5 on 2 x 4 is
 * 84 GB per rank (madqp_dist_memory; 224 GB on one GPU).  Assembly needs no
 * communication; iterates and scalars are replicated (madqp_state as for madqp_kkt_*), products with A, A', H are
 * summed with one all-reduce each, so every rank sees bitwise the same vectors and scalars. */
typedef struct madqp_dkkt madqp_dkkt;
int32_t madqp_dkkt_create(madqp_dist* d, int64_t nx, int64_t m, int64_t ns, const int64_t* ind_ineq_host,
                          const double* Hloc, int64_t ldh, const double* A_I, int64_t ld_ai, const double* A_J,
                          int64_t ld_aj, madqp_dkkt** out);
int32_t madqp_dkkt_destroy(madqp_dkkt* kkt);
int32_t madqp_dkkt_build(madqp_dkkt* kkt, const madqp_state* st);
int32_t madqp_dkkt_factorize(madqp_dkkt* kkt, int32_t* info_host);
int32_t madqp_dkkt_solve(madqp_dkkt* kkt, const madqp_state* st, double* w);
int32_t madqp_dkkt_mul(madqp_dkkt* kkt, const madqp_state* st, double* w, const double* v, double alpha, double beta);
int32_t madqp_dkkt_jtprod(madqp_dkkt* kkt, double* out, const double* y);
int32_t madqp_dkkt_eval(madqp_dkkt* kkt, const madqp_state* st, const double* q, const double* rhs, double c0,
                        double* obj_host);

/* ----------------------------------------- native driver of one MPC iteration */
/* The loop body of mpc! (src/solver.jl:254-345) above the entry points of this header, for hosts
 * that want one foreign call per iteration (small problems, batches).  The Julia glue does not
 * need it: MadIPM's own mpc! drives the plugin methods.  Start-up (src/solver.jl:127-182) stays
 * with the host. */
typedef struct madqp_mpc madqp_mpc;
typedef struct madqp_mpc_options { /* src/utils.jl:69-103 */
    double tol;
    int64_t max_iter;
    int32_t max_ncorr;
    int32_t step_rule;      /* 0 ConservativeStep(tau), 1 AdaptiveStep(tau_min), 2 MehrotraAdaptiveStep(gamma_f) */
    double step_param;
    int32_t regularization; /* 0 NoRegularization, 1 FixedRegularization, 2 AdaptiveRegularization */
    int32_t check_residual;
    double delta_p, delta_d, delta_min;
    double mu_min;
    double tol_linear_solve;
    int32_t refine_steps;   /* extension, default 0 = the reference's solve_system!: steps of iterative refinement
                               d += K^-1 (p - K d) with the residual src/linear_solver.jl:29-31 already forms */
    int32_t kkt_form;       /* madqp_batch_* only: 0 condensed K = H + Sigma_x + A' Theta A, 1 the reference's normal
                               equations A Sigma^-1 A' (NormalKKTSystem, LP only) */
} madqp_mpc_options;
typedef struct madqp_mpc_info { /* print_iter tuple (src/structure.jl:178-195) + counters */
    int64_t k;
    double obj, inf_pr, inf_du, inf_compl, mu, dnorm, del_w, del_c, alpha_p, alpha_d, residual_ratio;
    int64_t n_factorizations;
    int32_t factor_info;
} madqp_mpc_info;
/* w1, w2: the _w1/_w2 UnreducedKKTVector buffers (src/structure.jl:36-37); q, rhs, c0: scaled model
 * data as for madqp_kkt_eval; norm_b, norm_c: src/solver.jl:173-174.  All pointers are borrowed. */
int32_t madqp_mpc_create(madqp_kkt* kkt, const madqp_state* st, double* w1, double* w2, const double* q,
                         const double* rhs, double c0, double norm_b, double norm_c,
                         const madqp_mpc_options* opt, madqp_mpc** out);
int32_t madqp_mpc_destroy(madqp_mpc* mpc);
/* scalars produced by the host start-up: mu = mu_init, del_w/del_c of init_regularization!, objective, k */
int32_t madqp_mpc_set_scalars(madqp_mpc* mpc, double mu, double del_w, double del_c, double obj, int64_t k);
/* src/solver.jl:259-283; status_host: 0 continue, 1 SOLVE_SUCCEEDED, 6 MAXIMUM_ITERATIONS_EXCEEDED */
int32_t madqp_mpc_head(madqp_mpc* mpc, madqp_mpc_info* info_host, int32_t* status_host);
/* src/solver.jl:288-343; returns MADQP_NUM_NAN for MadNLP.SolveException (src/linear_solver.jl:41-43).
 * Unless Mehrotra's adaptive step rule is selected, the reductions of an iteration are queued in the context's result
 * block and fetched twice per iteration, plus once per tried Gondzio correction beyond the first (csrc/mpc.hip,
 * body_fused; the iterates are bitwise those of the sequential form, which MADQP_MPC_FUSED=0 selects when the object is
 * created).  Neither fetch leaves the device idle: the decisions the host takes from the first one (residual verdicts,
 * step lengths, keep or drop the first Gondzio trial) are taken by a one-thread kernel as well, and the update of the
 * iterates, the model evaluation and the next termination test's norms are queued behind it as kernels that do
 * nothing when that verdict is "the host takes over" (failed factorisation, failed verdict, a further trial); before
 * the second fetch the NEXT iteration's diagonal and KKT assembly are queued (K is rebuilt from scratch every
 * iteration, so a loop that ends there has lost nothing but that assembly).  Where the host takes over -- and where the
 * call returns MADQP_NUM_NAN -- x, y, zl, zu and the bounds are as they were when the call began; f, c and A'y have
 * been re-evaluated at that same x (the values they had, for any state the loop itself produced).  The host checks every decision of the
 * device against its own and fails with MADQP_ERR_HIP should they differ.  MADQP_MPC_AHEAD=0 at creation: the round-4
 * form, nothing queued behind a fetch.  The residual norms of the next termination test come with the second fetch,
 * so madqp_mpc_head behind a body does not synchronise at all. */
int32_t madqp_mpc_body(madqp_mpc* mpc, madqp_mpc_info* info_host);
/* blocking scalar read-backs issued so far by madqp_mpc_head / madqp_mpc_body of the fused form (the reference's loop
 * has about 20 implicit synchronisations per iteration, src/solver.jl:264-343) */
int32_t madqp_mpc_readbacks(const madqp_mpc* mpc, int64_t* count);
/* assemblies queued ahead by madqp_mpc_body for the pass after it, and how many of them the next pass took over */
int32_t madqp_mpc_ahead_stats(const madqp_mpc* mpc, int64_t* queued, int64_t* used);

/* ----------------------------------------- batches of small, equally shaped QPs */
/* B problems with the same (nx, m) and the same bound / inequality pattern advance in lock step:
 * assembly and Cholesky are batched launches of the MFMA kernels, the rest of an iteration is one
 * workgroup per problem; scalars stay on the device, a finished problem is masked out by its status
 * word (csrc/batch.hip).  Options: every step rule and regularization of madqp_mpc_options, Gondzio corrections;
 * a failed factorisation is retried per problem with del_w, del_c x 100, three trials in all
 * (src/linear_solver.jl:6-17); a problem still not factorised then ends with status -3. */
#define MADQP_BATCH_SCALARS 16 /* per problem: mu, alpha_p, alpha_d, obj, inf_pr, inf_du, inf_compl, dnorm,
                                  norm_b, norm_c, del_w, del_c, residual_ratio, reg state (2), number of factorisations */
typedef struct madqp_batch madqp_batch;
typedef struct madqp_batch_data { /* caller-owned device arrays, problem b at offset b * length; borrowed */
    const double* H;   /* [B][nx][nx] symmetric, or NULL for LPs */
    const double* A;   /* [B][m][nx], row k contiguous */
    const double* q;   /* [B][nx] */
    const double* rhs; /* [B][m] */
    const double* c0;  /* [B] */
    double *x, *xl, *xu, *zl, *zu; /* [B][n], n = nx + ns: iterates after src/solver.jl:131-159 */
    double* y;                     /* [B][m] */
} madqp_batch_data;
/* ind_lb / ind_ub: device, 0-based, shared by all problems; ind_ineq_host: host, strictly increasing */
int32_t madqp_batch_create(madqp_ctx* ctx, int64_t B, int64_t nx, int64_t m, int64_t ns,
                           const int64_t* ind_ineq_host, int64_t nlb, const int64_t* ind_lb, int64_t nub,
                           const int64_t* ind_ub, const madqp_batch_data* data, const madqp_mpc_options* opt,
                           madqp_batch** out);
int32_t madqp_batch_destroy(madqp_batch* b);
/* src/solver.jl:162-179 for every problem: initialize!(kkt), model evaluation, norms, starting point */
int32_t madqp_batch_init(madqp_batch* b, double mu_init, double bound_fac);
/* up to max_steps lock-step iterations of mpc! (src/solver.jl:254-345); one 4-byte read-back every
 * check_every steps; n_active_host: problems still active on return */
int32_t madqp_batch_iterate(madqp_batch* b, int32_t max_steps, int32_t check_every, int32_t* n_active_host);
/* per problem: status (0 active, 1 SOLVE_SUCCEEDED, 6 MAXIMUM_ITERATIONS_EXCEEDED,
 * -3 ERROR_IN_STEP_COMPUTATION, -1 INTERNAL_ERROR), iterations, MADQP_BATCH_SCALARS scalars */
int32_t madqp_batch_results(madqp_batch* b, int32_t* status_host, int32_t* iters_host, double* scal_host);

#ifdef __cplusplus
}
#endif
#endif /* MADQP_H */
