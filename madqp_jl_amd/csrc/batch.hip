// Batches of small, equally shaped QPs (BASELINE configs[3], SURVEY.md 8e "persistent kernel or per-QP
// convergence mask"): B problems advance in lock step, a handful of launches per iteration for the
// whole batch instead of ~60 launches (and ~12 host round trips) per problem.
//
//   * dense MFMA work is batched: ONE launch of the GEMM core assembles K_b = H_b + Sigma_b + A_b' Theta_b A_b
//     for all b (grid.y = problem), the recursive Cholesky runs as ~10 launches for all b
//     (madqp_chol_factor_batched: diagonal kernel with grid = B, TRSM / update GEMMs with grid.y = B);
//   * everything else of an iteration -- residuals, termination test, right-hand sides, the
//     condensed solves with their triangular sweeps, the residual check of solve_system!, step
//     lengths, centering, iterate update, model callbacks -- is ONE workgroup per problem running the
//     kernel bodies of vec_kernels.inc / kkt_kernels.inc back to back (compiled here as device
//     functions with a workgroup-stride loop): the per-problem scalars (mu, alpha, norms) never leave
//     the chip, and a problem that has converged (or failed) is masked out by its status word;
//   * the host only reads the status words (one small copy per call of madqp_batch_iterate).
//
// Same arithmetic as the single-problem path (src/solver.jl:127-182, 254-345 in the same order);
// sums are accumulated in a different order, so results agree to rounding, not bitwise.  All step
// rules and Gondzio's corrections are in; not covered here (use the per-problem driver): the x100
// regularization retry (a failed factorisation ends that problem with status -3), normal equations.
#include <algorithm>
#include <cmath>

#include "common.h"

#define TPB 256

typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {
constexpr int NB = 128;
constexpr int64_t WBLK = 2 * NB * NB;

namespace wg {
#define MQ_KERNEL __device__ void
#define MQ_BLOCK 0
#define GRID_STRIDE(i, len) for (int64_t i = threadIdx.x; i < (len); i += TPB)
#include "vec_kernels.inc"
#include "kkt_kernels.inc"
#undef MQ_KERNEL
#undef MQ_BLOCK
#undef GRID_STRIDE
}  // namespace wg

enum {
    S_MU = 0, S_ALPHA_P, S_ALPHA_D, S_OBJ, S_INF_PR, S_INF_DU, S_INF_COMPL, S_DNORM, S_NORM_B, S_NORM_C,
    S_DEL_W, S_DEL_C, S_RATIO, S_REG_P, S_REG_D, S_SPARE, S_COUNT
};
static_assert(S_COUNT == MADQP_BATCH_SCALARS, "scalar block layout is part of the ABI");

enum { ST_ACTIVE = 0, ST_SOLVED = 1, ST_MAXITER = 6, ST_STEP_ERROR = -3, ST_INTERNAL = -1 };

struct BQ {  // device view of the batch (by value in the kernel arguments); problem b at offset b * length
    int64_t B, nx, m, ns, n, nlb, nub, ntot, ldk, npad, kpad, nblk;
    const int64_t *ind_lb, *ind_ub, *ind_ineq, *slot;
    const double *H, *A, *q, *rhs, *c0;
    double *x, *xl, *xu, *zl, *zu, *y;
    double *f, *c, *jacl, *reg, *pr_diag, *du_diag, *d, *p, *w1, *w2;
    double *l_diag, *l_lower, *u_diag, *u_lower, *corr_lb, *corr_ub;
    double *theta, *t, *u, *K, *S, *winv, *tmp;
    double* scal;
    int32_t *status, *iters, *info;
    madqp_mpc_options opt;
    double mu_init, bound_fac;
};

__device__ __forceinline__ madqp_state state_of(const BQ& q, int64_t b) {
    madqp_state s;
    s.n = q.n;
    s.m = q.m;
    s.nlb = q.nlb;
    s.nub = q.nub;
    s.ind_lb = q.ind_lb;
    s.ind_ub = q.ind_ub;
    s.x = q.x + b * q.n;
    s.xl = q.xl + b * q.n;
    s.xu = q.xu + b * q.n;
    s.zl = q.zl + b * q.n;
    s.zu = q.zu + b * q.n;
    s.f = q.f + b * q.n;
    s.y = q.y + b * q.m;
    s.c = q.c + b * q.m;
    s.jacl = q.jacl + b * q.n;
    s.d = q.d + b * q.ntot;
    s.p = q.p + b * q.ntot;
    s.correction_lb = q.corr_lb + b * q.nlb;
    s.correction_ub = q.corr_ub + b * q.nub;
    s.reg = q.reg + b * q.n;
    s.pr_diag = q.pr_diag + b * q.n;
    s.du_diag = q.du_diag + b * q.m;
    s.l_diag = q.l_diag + b * q.nlb;
    s.l_lower = q.l_lower + b * q.nlb;
    s.u_diag = q.u_diag + b * q.nub;
    s.u_lower = q.u_lower + b * q.nub;
    return s;
}

// per-problem pointers that are not part of madqp_state
struct Prob {
    const double *H, *A, *qv, *rhs;
    double *theta, *t, *u, *K, *S, *winv, *tmp, *w1, *scal;
    double c0;
};
__device__ __forceinline__ Prob prob_of(const BQ& q, int64_t b) {
    Prob p;
    p.H = q.H ? q.H + b * q.nx * q.nx : nullptr;
    p.A = q.A + b * q.m * q.nx;
    p.qv = q.q + b * q.nx;
    p.rhs = q.rhs + b * q.m;
    p.theta = q.theta + b * q.m;
    p.t = q.t + b * q.m;
    p.u = q.u + b * q.m;
    p.K = q.K + b * q.ldk * q.ldk;
    p.S = q.S + b * q.kpad * q.npad;
    p.winv = q.winv + b * q.nblk * WBLK;
    p.tmp = q.tmp + b * q.npad;
    p.w1 = q.w1 + b * q.ntot;
    p.scal = q.scal + b * S_COUNT;
    p.c0 = q.c0[b];
    return p;
}

#define WG_SYNC() __syncthreads()

// ---- workgroup-level dense helpers (one problem: matrices of a few MB, vectors of a few KB) ----
// y(rows) = alpha * M x + beta * y,  M row-major rows x cols (row k contiguous): a wave per row
__device__ void wg_gemv_n(int64_t rows, int64_t cols, double alpha, const double* __restrict__ M,
                          const double* __restrict__ x, double beta, double* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = TPB / 64, R = 4;  // 4 rows per wave pass: 4x the loads in flight
    for (int64_t k0 = (int64_t)wave * R; k0 < rows; k0 += NW * R) {
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        for (int64_t j = lane; j < cols; j += 64) {
            const double xj = x[j];
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (k0 + r < rows) acc[r] += M[(k0 + r) * cols + j] * xj;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double a = acc[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0 && k0 + r < rows) y[k0 + r] = (beta == 0.0) ? alpha * a : alpha * a + beta * y[k0 + r];
        }
    }
}
// out(cols) = alpha * M' v + beta * out,  M row-major rows x cols: a thread per column
__device__ void wg_gemv_t(int64_t rows, int64_t cols, double alpha, const double* __restrict__ M,
                          const double* __restrict__ v, double beta, double* __restrict__ out) {
    for (int64_t j = threadIdx.x; j < cols; j += TPB) {
        double a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = 0.0;
        int64_t k = 0;
        for (; k + 8 <= rows; k += 8) {  // 8 independent loads in flight per thread
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] += M[(k + r) * cols + j] * v[k + r];
        }
        for (; k < rows; ++k) a[0] += M[k * cols + j] * v[k];
        const double acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        out[j] = (beta == 0.0) ? alpha * acc : alpha * acc + beta * out[j];
    }
}
// xs(128, LDS) = img * vs,  img: 128 x 128, fast index = output (the Wcm / Wrm images of chol.hip)
__device__ void wg_block_matvec(const double* __restrict__ img, const double* vs, double* part, double* xs) {
    const int i = threadIdx.x & (NB - 1), h = threadIdx.x >> 7;
    double acc = 0.0;
#pragma unroll 8
    for (int c = h * 64; c < h * 64 + 64; ++c) acc += img[i + c * NB] * vs[c];
    part[h * NB + i] = acc;
    WG_SYNC();
    if (threadIdx.x < NB) xs[i] = part[i] + part[NB + i];
    WG_SYNC();
}
// rhs(n) <- (L L')^-1 rhs with the inverse diagonal blocks of potf2_inv_kernel; tmp: n doubles
__device__ void wg_chol_solve(const double* __restrict__ L, int64_t lda, const double* __restrict__ winv,
                              int64_t n, double* rhs, double* tmp, double* lds /* 4 * NB doubles */) {
    double *vs = lds, *xs = lds + NB, *part = lds + 2 * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // forward: L y = b (b updated in place below the block, y collected in tmp)
    for (int64_t jb = 0; jb < n; jb += NB) {
        const int w = (int)((n - jb < NB) ? (n - jb) : NB);
        if (threadIdx.x < NB) vs[threadIdx.x] = (threadIdx.x < w) ? rhs[jb + threadIdx.x] : 0.0;
        WG_SYNC();
        wg_block_matvec(winv + (jb / NB) * WBLK, vs, part, xs);
        if (threadIdx.x < w) tmp[jb + threadIdx.x] = xs[threadIdx.x];
        for (int64_t row = jb + w + threadIdx.x; row < n; row += TPB) {
            const double* Lp = L + row + jb * lda;
            double acc = 0.0;
            for (int c = 0; c < w; ++c) acc += Lp[(int64_t)c * lda] * xs[c];
            rhs[row] -= acc;
        }
        WG_SYNC();
    }
    // backward: L' x = y (y = tmp updated in place left of the block, x written to rhs)
    const int64_t last = ((n - 1) / NB) * NB;
    for (int64_t jb = last; jb >= 0; jb -= NB) {
        const int w = (int)((n - jb < NB) ? (n - jb) : NB);
        if (threadIdx.x < NB) vs[threadIdx.x] = (threadIdx.x < w) ? tmp[jb + threadIdx.x] : 0.0;
        WG_SYNC();
        wg_block_matvec(winv + (jb / NB) * WBLK + NB * NB, vs, part, xs);
        if (threadIdx.x < w) rhs[jb + threadIdx.x] = xs[threadIdx.x];
        for (int64_t c = wave; c < jb; c += TPB / 64) {  // a wave per column left of the block
            const double* Lp = L + jb + c * lda;
            double acc = 0.0;
            for (int r = lane; r < w; r += 64) acc += Lp[r] * xs[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (lane == 0) tmp[c] -= acc;
        }
        WG_SYNC();
    }
}

__device__ void wg_copy(int64_t len, const double* __restrict__ src, double* __restrict__ dst) {
    for (int64_t i = threadIdx.x; i < len; i += TPB) dst[i] = src[i];
}

// MadNLP.jtprod!(out, kkt, y): out = [A' y ; -y[ind_ineq]]   (src/KKT/normalkkt.jl:162-164)
__device__ void wg_jtprod(const BQ& q, const Prob& pb, const double* y, double* out) {
    wg_gemv_t(q.m, q.nx, 1.0, pb.A, y, 0.0, out);
    wg::jt_slack_kernel(q.ns, q.ind_ineq, y, out + q.nx, 1.0, 0.0);
    WG_SYNC();
}

// MadNLP.solve!(kkt, w), condensed form (kkt.hip: madqp_kkt_solve)
__device__ void wg_kkt_solve(const BQ& q, const madqp_state& s, const Prob& pb, double* w, double* lds) {
    double* wx = w;
    double* wy = w + s.n;
    if (s.nlb) wg::reduce_rhs_kernel(s.nlb, s.ind_lb, w, w + s.n + s.m, s.l_diag);
    WG_SYNC();
    if (s.nub) wg::reduce_rhs_kernel(s.nub, s.ind_ub, w, w + s.n + s.m + s.nlb, s.u_diag);
    WG_SYNC();
    if (q.m) {
        wg::condense_kernel(q.m, q.nx, q.slot, s.pr_diag, pb.theta, wx, wy, pb.t, pb.u);
        WG_SYNC();
        wg_gemv_t(q.m, q.nx, 1.0, pb.A, pb.u, 1.0, wx);
        WG_SYNC();
    }
    wg_chol_solve(pb.K, q.ldk, pb.winv, q.nx, wx, pb.tmp, lds);
    if (q.m) {
        wg_gemv_n(q.m, q.nx, 1.0, pb.A, wx, 0.0, pb.u);
        WG_SYNC();
        wg::decondense_kernel(q.m, q.nx, q.slot, s.pr_diag, pb.theta, pb.t, pb.u, wx, wy);
        WG_SYNC();
    }
    if (s.nlb || s.nub) wg::finish_aug_solve_kernel(s, w);
    WG_SYNC();
}

// MadNLP.mul!(w, kkt, v, alpha, beta) (kkt.hip: madqp_kkt_mul + madqp_kktmul)
__device__ void wg_kkt_mul(const BQ& q, const madqp_state& s, const Prob& pb, double* w, const double* v,
                           double alpha, double beta) {
    const int64_t nx = q.nx, n = s.n;
    wg_gemv_t(q.m, nx, alpha, pb.A, v + n, beta, w);
    WG_SYNC();
    if (pb.H && nx) {
        wg_gemv_t(nx, nx, alpha, pb.H, v, 1.0, w);
        WG_SYNC();
    }
    if (q.ns) wg::jt_slack_kernel(q.ns, q.ind_ineq, v + n, w + nx, alpha, beta);
    if (q.m) {
        wg_gemv_n(q.m, nx, 1.0, pb.A, v, 0.0, pb.u);
        WG_SYNC();
        wg::mul_rows_kernel(q.m, q.slot, pb.u, v + nx, w + n, alpha, beta);
    }
    WG_SYNC();
    wg::kktmul_diag_kernel(s, w, v, alpha);
    WG_SYNC();
    if (s.nlb) wg::kktmul_lb_kernel(s, w, v, alpha, beta);
    WG_SYNC();
    if (s.nub) wg::kktmul_ub_kernel(s, w, v, alpha, beta);
    WG_SYNC();
}

// solve_system! (src/linear_solver.jl:19-45); returns false for MadNLP.SolveException
__device__ bool wg_solve_system(const BQ& q, const madqp_state& s, const Prob& pb, double* lds, double* red) {
    wg_copy(q.ntot, s.p, s.d);
    WG_SYNC();
    wg_kkt_solve(q, s, pb, s.d, lds);
    wg_copy(q.ntot, s.p, pb.w1);
    WG_SYNC();
    wg_kkt_mul(q, s, pb, pb.w1, s.d, -1.0, 1.0);
    wg::norm_inf3_kernel(q.ntot, pb.w1, s.p, s.d, red);
    WG_SYNC();
    const double ratio = red[0] / fmax(1.0, red[1]);
    WG_SYNC();
    if (threadIdx.x == 0) pb.scal[S_RATIO] = ratio;
    return !((ratio != ratio) || (q.opt.check_residual && ratio > q.opt.tol_linear_solve));
}

// callbacks obj / grad! / cons! (kkt.hip: madqp_kkt_eval)
__device__ double wg_eval_model(const BQ& q, const madqp_state& s, const Prob& pb, double* red) {
    if (pb.H && q.nx) wg_gemv_t(q.nx, q.nx, 1.0, pb.H, s.x, 0.0, s.f);
    WG_SYNC();
    wg::eval_grad_kernel(s.n, q.nx, (pb.H && q.nx) ? 1 : 0, pb.qv, s.x, s.f, red);
    WG_SYNC();
    const double obj = pb.c0 + red[0] + 0.5 * red[1];
    WG_SYNC();
    if (q.m) {
        wg_gemv_n(q.m, q.nx, 1.0, pb.A, s.x, 0.0, s.c);
        WG_SYNC();
        wg::eval_cons_kernel(q.m, q.slot, s.x + q.nx, pb.rhs, s.c);
        WG_SYNC();
    }
    return obj;
}

__device__ double wg_compl(const madqp_state& s, int affine, double ap, double ad, double* red) {
    if (s.nlb + s.nub == 0) return 0.0;
    wg::compl_kernel(s, affine, ap, ad, red);
    WG_SYNC();
    const double v = (red[0] + red[1]) / (double)(s.nlb + s.nub);
    WG_SYNC();
    return v;
}

// the four (value, blocking index) pairs of get_alpha_max_primal / _dual (src/kernels.jl:242-288)
__device__ void wg_alpha_max(const madqp_state& s, double tau, double* red, double (&a)[4], int64_t (&ib)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = 1.0;
        ib[k] = -1;
    }
    if (s.nlb + s.nub == 0) return;
    wg::alpha_max_kernel(s, tau, red);
    WG_SYNC();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (red[2 * k] < 1.0) {  // init = (1.0, 0): alpha <= 1, "nothing blocks" otherwise
            a[k] = red[2 * k];
            ib[k] = (int64_t)red[2 * k + 1];
        }
    WG_SYNC();
}
// (alpha_p, alpha_d) of get_fraction_to_boundary_step (src/kernels.jl:290-305)
__device__ void wg_fraction_to_boundary(const madqp_state& s, double tau, double* red, double& ap, double& ad) {
    double a[4];
    int64_t ib[4];
    wg_alpha_max(s, tau, red, a, ib);
    ap = fmin(a[0], a[1]);
    ad = fmin(a[2], a[3]);
}

// update_step!(::MehrotraAdaptiveStep) (src/kernels.jl:325-374): element reads at the blocking indices
__device__ void wg_mehrotra_adaptive_step(const madqp_state& s, double gamma_f, double* red, double& alpha_p,
                                          double& alpha_d) {
    const double gamma_a = 1.0 / (1.0 - gamma_f);
    double a[4];
    int64_t ib[4];
    wg_alpha_max(s, 1.0, red, a, ib);
    const double max_ap = fmin(a[0], a[1]), max_ad = fmin(a[2], a[3]);
    const double mu_full = wg_compl(s, 1, max_ap, max_ad, red) / gamma_a;
    const double* dx = s.d;
    const double* dzl = s.d + s.n + s.m;
    const double* dzu = dzl + s.nlb;
    alpha_p = 1.0;
    alpha_d = 1.0;
    if (max_ap < 1.0) {
        if (a[0] <= a[1]) {
            const int64_t i = ib[0], j = s.ind_lb[i];
            const double tmp = mu_full / (s.zl[j] + max_ad * dzl[i]);
            alpha_p = (s.x[j] - s.xl[j] - tmp) / (-dx[j]);
        } else {
            const int64_t i = ib[1], j = s.ind_ub[i];
            const double tmp = mu_full / (s.zu[j] + max_ad * dzu[i]);
            alpha_p = (s.xu[j] - s.x[j] - tmp) / dx[j];
        }
    }
    if (max_ad < 1.0) {
        if (a[2] <= a[3]) {
            const int64_t i = ib[2], j = s.ind_lb[i];
            const double tmp = mu_full / (s.x[j] + max_ap * dx[j] - s.xl[j]);
            alpha_d = -(s.zl[j] - tmp) / dzl[i];
        } else {
            const int64_t i = ib[3], j = s.ind_ub[i];
            const double tmp = mu_full / (s.xu[j] - s.x[j] - max_ap * dx[j]);
            alpha_d = -(s.zu[j] - tmp) / dzu[i];
        }
    }
    alpha_p = fmax(alpha_p, gamma_f * max_ap);
    alpha_d = fmax(alpha_d, gamma_f * max_ad);
}

// Theta and the scaled operand S = sqrt(Theta) A (zero padded to kpad x npad) of build_kkt!
__device__ void wg_build_operands(const BQ& q, const madqp_state& s, const Prob& pb) {
    if (q.m) wg::theta_kernel(q.m, q.nx, q.slot, s.pr_diag, s.du_diag, pb.theta);
    WG_SYNC();
    for (int64_t k = 0; k < q.kpad; ++k) {
        double* dst = pb.S + k * q.npad;
        if (k < q.m) {
            const double wk = sqrt(pb.theta[k]);
            const double* src = pb.A + k * q.nx;
            for (int64_t i = threadIdx.x; i < q.npad; i += TPB) dst[i] = (i < q.nx) ? src[i] * wk : 0.0;
        } else {
            for (int64_t i = threadIdx.x; i < q.npad; i += TPB) dst[i] = 0.0;
        }
    }
}

__device__ void wg_fill(int64_t len, double v, double* dst) {
    for (int64_t i = threadIdx.x; i < len; i += TPB) dst[i] = v;
}

// ---- start: src/solver.jl:162-174 and the first half of init_starting_point! (:6-21) ----------
__global__ __launch_bounds__(TPB) void bq_init_pre_kernel(BQ q) {
    __shared__ double red[32];
    const int64_t b = blockIdx.x;
    const madqp_state s = state_of(q, b);
    const Prob pb = prob_of(q, b);
    if (threadIdx.x == 0) {
        q.status[b] = ST_ACTIVE;
        q.iters[b] = 0;
    }
    // MadNLP.initialize!(kkt) (src/KKT/normalkkt.jl:136-147)
    wg_fill(s.n, 1.0, s.reg);
    wg_fill(s.nlb, 0.0, s.l_lower);
    wg_fill(s.nub, 0.0, s.u_lower);
    wg_fill(s.nlb, 1.0, s.l_diag);
    wg_fill(s.nub, 1.0, s.u_diag);
    wg_fill(s.n, 0.0, s.jacl);
    WG_SYNC();
    // init_regularization! (src/kernels.jl:380-384)
    const double del_w = 1.0, del_c = (q.opt.regularization == 0) ? 0.0 : q.opt.delta_d;
    const double obj = wg_eval_model(q, s, pb, red);  // :166-169
    wg::norm_inf3_kernel(s.m, pb.rhs, nullptr, nullptr, red);
    WG_SYNC();
    const double norm_b = red[0];
    WG_SYNC();
    wg::norm_inf3_kernel(s.n, s.f, nullptr, nullptr, red);
    WG_SYNC();
    const double norm_c = red[0];
    WG_SYNC();
    // init_starting_point! :16-18
    wg_fill(s.n, del_w, s.reg);
    wg_fill(s.n, del_w, s.pr_diag);
    wg_fill(s.m, del_c, s.du_diag);
    WG_SYNC();
    wg_build_operands(q, s, pb);
    if (threadIdx.x == 0) {
        double* sc = pb.scal;
        sc[S_MU] = q.mu_init;
        sc[S_ALPHA_P] = sc[S_ALPHA_D] = 0.0;
        sc[S_OBJ] = obj;
        sc[S_INF_PR] = sc[S_INF_DU] = sc[S_INF_COMPL] = sc[S_DNORM] = 0.0;
        sc[S_NORM_B] = norm_b;
        sc[S_NORM_C] = norm_c;
        sc[S_DEL_W] = del_w;
        sc[S_DEL_C] = del_c;
        sc[S_RATIO] = 0.0;
        sc[S_REG_P] = q.opt.delta_p;
        sc[S_REG_D] = q.opt.delta_d;
        sc[S_SPARE] = 0.0;
    }
}

// ---- second half of init_starting_point! (src/solver.jl:25-123) ---------------------------------
__global__ __launch_bounds__(TPB) void bq_init_post_kernel(BQ q) {
    __shared__ double red[32];
    __shared__ double lds[4 * NB];
    const int64_t b = blockIdx.x;
    const madqp_state s = state_of(q, b);
    const Prob pb = prob_of(q, b);
    int status = ST_ACTIVE;
    if (q.info[b] != 0) status = ST_INTERNAL;  // the start matrix (Sigma = 1) must be positive definite
    if (status == ST_ACTIVE) {
        wg::rhs_kernel(s, 2, 0.0);  // set_initial_primal_rhs! :25
        WG_SYNC();
        if (!wg_solve_system(q, s, pb, lds, red)) status = ST_STEP_ERROR;
    }
    if (status == ST_ACTIVE) {
        wg::axpy_kernel(s.n, 1.0, s.d, s.x);  // :28
        WG_SYNC();
        wg::rhs_kernel(s, 3, 0.0);  // set_initial_dual_rhs! :31
        WG_SYNC();
        if (!wg_solve_system(q, s, pb, lds, red)) status = ST_STEP_ERROR;
    }
    if (status == ST_ACTIVE) {
        wg_copy(s.m, s.d + s.n, s.y);  // :33
        WG_SYNC();
        wg_jtprod(q, pb, s.y, s.jacl);  // :37
        wg::axpy_kernel(s.n, 1.0, s.f, s.jacl);  // :39
        WG_SYNC();
        wg::sp_init_duals_kernel(s);  // :41-66
        WG_SYNC();
        double mn[4] = {0.0, 0.0, 0.0, 0.0};
        if (s.nlb || s.nub) {
            wg::sp_mins_kernel(s, red);  // :68-78
            WG_SYNC();
            for (int k = 0; k < 4; ++k) mn[k] = red[k];
            WG_SYNC();
        }
        const double delta_x = fmax(0.0, fmax(-1.5 * mn[0], -1.5 * mn[1]));
        const double delta_s = fmax(0.0, fmax(-1.5 * mn[2], -1.5 * mn[3]));
        auto shift = [&](double dx, double dz) {  // :80-83 (madqp_sp_shift: four ordered passes)
            if (s.nlb) wg::sp_shift_kernel(s.nlb, s.ind_lb, s.x, dx);
            WG_SYNC();
            if (s.nub) wg::sp_shift_kernel(s.nub, s.ind_ub, s.x, -dx);
            WG_SYNC();
            if (s.nlb) wg::sp_shift_kernel(s.nlb, s.ind_lb, s.zl, dz);
            if (s.nub) wg::sp_shift_kernel(s.nub, s.ind_ub, s.zu, dz);
            WG_SYNC();
        };
        shift(delta_x, 1.0 + delta_s);
        double sm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (s.nlb || s.nub) {
            wg::sp_sums_kernel(s, red);  // :85-94
            WG_SYNC();
            for (int k = 0; k < 8; ++k) sm[k] = red[k];
            WG_SYNC();
        }
        double mu = 0.0;
        if (s.nlb > 0) mu += sm[0] - sm[1];
        if (s.nub > 0) mu += sm[2] - sm[3];
        shift(mu / (2 * (sm[4] + sm[5])), mu / (2 * (sm[6] + sm[7])));  // :96-99
        wg::sp_project_kernel(s, q.bound_fac);  // :101-118
        WG_SYNC();
        if (s.nlb || s.nub) {
            wg::sp_check_kernel(s, red);  // :120-123
            WG_SYNC();
            if (red[0] != 0.0) status = ST_INTERNAL;
            WG_SYNC();
        }
    }
    if (threadIdx.x == 0) q.status[b] = status;
}

// ---- loop head + build_kkt! operands: src/solver.jl:259-289 -----------------------------------
__global__ __launch_bounds__(TPB) void bq_iter_pre_kernel(BQ q) {
    __shared__ double red[32];
    const int64_t b = blockIdx.x;
    if (q.status[b] != ST_ACTIVE) return;
    const madqp_state s = state_of(q, b);
    const Prob pb = prob_of(q, b);
    double* sc = pb.scal;
    wg_jtprod(q, pb, s.y, s.jacl);  // :259
    wg::inf_kernel(s, red);
    WG_SYNC();
    const double nc = red[0], nd = red[1];
    const double ncompl = (red[2] != red[2]) ? red[2] : ((red[3] != red[3]) ? red[3] : fmax(red[2], red[3]));
    WG_SYNC();
    const double inf_pr = nc / fmax(1.0, sc[S_NORM_B]);        // :264
    const double inf_du = nd / fmax(1.0, sc[S_NORM_C]);        // :265-271
    const double inf_compl = ncompl / fmax(1.0, sc[S_NORM_C]);  // :272
    int status = ST_ACTIVE;
    if (fmax(inf_pr, fmax(inf_du, inf_compl)) <= q.opt.tol)  // :279
        status = ST_SOLVED;
    else if (q.iters[b] >= q.opt.max_iter)
        status = ST_MAXITER;
    // update_regularization! (src/kernels.jl:386-417)
    double del_w, del_c, rp = sc[S_REG_P], rd = sc[S_REG_D];
    if (q.opt.regularization == 0) {
        del_w = 0.0;
        del_c = 0.0;
    } else if (q.opt.regularization == 1) {
        del_w = q.opt.delta_p;
        del_c = q.opt.delta_d;
    } else {
        rp = fmax(rp / 10.0, q.opt.delta_min);
        rd = fmin(rd / 10.0, -q.opt.delta_min);
        del_w = rp;
        del_c = rd;
    }
    WG_SYNC();
    if (threadIdx.x == 0) {
        sc[S_INF_PR] = inf_pr;
        sc[S_INF_DU] = inf_du;
        sc[S_INF_COMPL] = inf_compl;
        q.status[b] = status;
        if (status == ST_ACTIVE) {
            sc[S_DEL_W] = del_w;
            sc[S_DEL_C] = del_c;
            sc[S_REG_P] = rp;
            sc[S_REG_D] = rd;
        }
    }
    if (status != ST_ACTIVE) return;
    // set_aug_diagonal_reg! (src/kernels.jl:128-146)
    wg::aug_diag_fill_kernel(s, del_w, del_c);
    WG_SYNC();
    if (s.nlb) wg::aug_diag_lb_kernel(s);
    WG_SYNC();
    if (s.nub) wg::aug_diag_ub_kernel(s);
    WG_SYNC();
    wg_build_operands(q, s, pb);
}

// ---- the rest of the iteration after factorize!: src/solver.jl:294-343 ------------------------
__global__ __launch_bounds__(TPB) void bq_iter_post_kernel(BQ q) {
    __shared__ double red[32];
    __shared__ double lds[4 * NB];
    const int64_t b = blockIdx.x;
    if (q.status[b] != ST_ACTIVE) return;
    const madqp_state s = state_of(q, b);
    const Prob pb = prob_of(q, b);
    double* sc = pb.scal;
    int status = ST_ACTIVE;
    if (q.info[b] != 0) status = ST_STEP_ERROR;  // not factorized: no x100 retry in the batched driver
    double mu = sc[S_MU], alpha_p = 0.0, alpha_d = 0.0, dnorm = 0.0, obj = sc[S_OBJ];
    if (status == ST_ACTIVE) {
        wg::rhs_kernel(s, 0, 0.0);  // set_predictive_rhs! :294
        WG_SYNC();
        if (!wg_solve_system(q, s, pb, lds, red)) status = ST_STEP_ERROR;
    }
    if (status == ST_ACTIVE) {
        double a_aff_p, a_aff_d;
        wg_fraction_to_boundary(s, 1.0, red, a_aff_p, a_aff_d);           // :295
        const double mu_affine = wg_compl(s, 1, a_aff_p, a_aff_d, red);   // :296
        wg::correction_kernel(s);                                         // :297
        WG_SYNC();
        const double mu_curr = wg_compl(s, 0, 0.0, 0.0, red);  // update_barrier! src/kernels.jl:226-236
        double sigma = 1.0;
        if (s.nlb + s.nub > 0) sigma = fmin(fmax(pow(mu_affine / mu_curr, 3.0), 1e-6), 10.0);
        mu = fmax(q.opt.mu_min, sigma * mu_curr);
        wg::rhs_kernel(s, 1, mu);  // set_correction_rhs! :307
        WG_SYNC();
        if (!wg_solve_system(q, s, pb, lds, red)) status = ST_STEP_ERROR;
        // gondzio_correction_direction! (src/solver.jl:200-251)
        if (status == ST_ACTIVE && q.opt.max_ncorr > 0) {
            const double delta = 0.1, bmin = 0.1, bmax = 10.0, tau_g = 0.995;
            double* w2 = q.w2 + b * q.ntot;
            double ap, ad;
            wg_fraction_to_boundary(s, tau_g, red, ap, ad);
            for (int c = 0; c < q.opt.max_ncorr; ++c) {
                const double ta_p = fmin(ap + delta, 1.0), ta_d = fmin(ad + delta, 1.0);
                const double ga = wg_compl(s, 1, ta_p, ta_d, red);
                const double mu_g = (ga / mu_curr) * (ga / mu_curr) * ga;
                wg::extra_correction_kernel(s, ta_p, ta_d, bmin * mu_g, bmax * mu_g);
                WG_SYNC();
                wg::rhs_kernel(s, 1, mu_g);
                wg_copy(q.ntot, s.d, w2);
                WG_SYNC();
                if (!wg_solve_system(q, s, pb, lds, red)) {
                    status = ST_STEP_ERROR;
                    break;
                }
                double ha_p, ha_d;
                wg_fraction_to_boundary(s, tau_g, red, ha_p, ha_d);
                if (ha_p < 1.005 * ap || ha_d < 1.005 * ad) {
                    wg_copy(q.ntot, w2, s.d);
                    WG_SYNC();
                    break;
                }
                ap = ha_p;
                ad = ha_d;
            }
        }
    }
    if (status == ST_ACTIVE) {
        // update_step! (src/kernels.jl:307-374)
        if (q.opt.step_rule == 2) {
            wg_mehrotra_adaptive_step(s, q.opt.step_param, red, alpha_p, alpha_d);
        } else {
            const double tau = (q.opt.step_rule == 0) ? q.opt.step_param : fmax(1.0 - mu, q.opt.step_param);
            wg_fraction_to_boundary(s, tau, red, alpha_p, alpha_d);
        }
        wg::norm_inf3_kernel(s.n, s.d, nullptr, nullptr, red);  // print_iter, src/structure.jl:190
        WG_SYNC();
        dnorm = red[0];
        WG_SYNC();
        wg::update_iterates_kernel(s, alpha_p, alpha_d);  // :332-335
        WG_SYNC();
        obj = wg_eval_model(q, s, pb, red);  // :338-340
        const double eps = 2.220446049250313e-16;
        if (s.nlb || s.nub) wg::adjust_boundary_kernel(s, eps * mu, 1.8189894035458565e-12);  // :342
        WG_SYNC();
    }
    if (threadIdx.x == 0) {
        q.status[b] = status;
        if (status == ST_ACTIVE) {
            sc[S_MU] = mu;
            sc[S_ALPHA_P] = alpha_p;
            sc[S_ALPHA_D] = alpha_d;
            sc[S_DNORM] = dnorm;
            sc[S_OBJ] = obj;
            q.iters[b] += 1;
        }
    }
}

__global__ void bq_count_active_kernel(const int32_t* __restrict__ status, int64_t B, int32_t* out) {
    int cnt = 0;
    for (int64_t i = threadIdx.x; i < B; i += blockDim.x) cnt += (status[i] == ST_ACTIVE);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (threadIdx.x == 0) *out = cnt;
}
}  // namespace

struct madqp_batch {
    madqp_ctx* ctx;
    BQ q;
    std::vector<void*> owned;
    int32_t* d_active;
};

namespace {
template <class T>
int32_t dalloc(madqp_batch* b, T** p, int64_t count, bool zero = false) {
    *p = nullptr;
    const size_t bytes = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess)
        return madqp_fail(b->ctx, MADQP_ERR_ALLOC, "madqp_batch_create: hipMalloc(%zu): %s", bytes,
                          hipGetErrorString(e));
    b->owned.push_back(*p);
    if (zero) {
        e = hipMemsetAsync(*p, 0, bytes, b->ctx->stream);
        if (e != hipSuccess) return madqp_fail(b->ctx, MADQP_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
    }
    return MADQP_OK;
}

// build_kkt! + factorize! for every active problem: one assembly launch, ~2 nx/128 launches of Cholesky
int32_t factor_all(madqp_batch* b) {
    madqp_ctx* ctx = b->ctx;
    const BQ& q = b->q;
    if (q.nx == 0) return hipMemsetAsync(q.info, 0, q.B * sizeof(int32_t), ctx->stream) == hipSuccess
                              ? MADQP_OK
                              : MADQP_ERR_HIP;
    GemmArgs g{};
    g.X = q.S;
    g.ldx = q.npad;
    g.Y = q.S;
    g.ldy = q.npad;
    g.K = q.kpad;
    g.Mread = g.Nread = q.npad;
    g.C = q.K;
    g.ldc = q.ldk;
    g.Cin = q.H;
    g.ldcin = q.nx;
    g.dvec = q.pr_diag;
    g.alpha = 1.0;
    g.beta = 1.0;
    g.M = q.nx;
    g.N = q.nx;
    g.lower_only = 1;
    if (q.m == 0) {  // no constraints: K = H + Sigma through a K = 0 product is not worth a special case
        g.K = 0;
        g.X = g.Y = q.K;
        g.Mread = g.Nread = 0;
    }
    GemmBatch bt{q.B, q.kpad * q.npad, q.kpad * q.npad, q.ldk * q.ldk, q.nx * q.nx, q.n, q.status};
    int32_t r = madqp_gemm_tn(ctx, g, MADQP_PROF_SYRK, nullptr, 0, &bt);
    if (r) return r;
    return madqp_chol_factor_batched(ctx, q.K, q.ldk, q.nx, q.ldk * q.ldk, q.winv, q.nblk * WBLK, q.info, q.B,
                                     q.status);
}
}  // namespace

extern "C" int32_t madqp_batch_destroy(madqp_batch* b) {
    if (!b) return MADQP_OK;
    (void)hipStreamSynchronize(b->ctx->stream);
    for (void* p : b->owned) (void)hipFree(p);
    delete b;
    return MADQP_OK;
}

extern "C" int32_t madqp_batch_create(madqp_ctx* ctx, int64_t B, int64_t nx, int64_t m, int64_t ns,
                                      const int64_t* ind_ineq_host, int64_t nlb, const int64_t* ind_lb,
                                      int64_t nub, const int64_t* ind_ub, const madqp_batch_data* data,
                                      const madqp_mpc_options* opt, madqp_batch** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, out && data && opt && B >= 1 && B <= 65535 && nx >= 0 && m >= 0 && ns >= 0 && ns <= m);
    ARG_TRY(ctx, nlb >= 0 && nub >= 0 && (nlb == 0 || ind_lb) && (nub == 0 || ind_ub) && (ns == 0 || ind_ineq_host));
    ARG_TRY(ctx, (nx == 0 || (data->q && data->x && data->xl && data->xu && data->zl && data->zu)) &&
                     (m == 0 || (data->A && data->rhs && data->y)) && data->c0);
    ARG_TRY(ctx, opt->step_rule >= 0 && opt->step_rule <= 2 && opt->max_ncorr >= 0 && opt->regularization >= 0 &&
                     opt->regularization <= 2);
    // the condensed form needs delta_d < 0 on equality rows (conventions)
    ARG_TRY(ctx, ns == m || (opt->regularization != 0 && opt->delta_d < 0.0));
    *out = nullptr;
    madqp_batch* b = new (std::nothrow) madqp_batch();
    if (!b) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    b->ctx = ctx;
    BQ& q = b->q;
    memset(&q, 0, sizeof(q));
    q.B = B;
    q.nx = nx;
    q.m = m;
    q.ns = ns;
    q.n = nx + ns;
    q.nlb = nlb;
    q.nub = nub;
    q.ntot = q.n + m + nlb + nub;
    q.npad = std::max<int64_t>(128, (nx + 127) / 128 * 128);
    q.ldk = q.npad;
    q.kpad = std::max<int64_t>(16, (m + 15) / 16 * 16);
    q.nblk = q.npad / 128;
    q.ind_lb = ind_lb;
    q.ind_ub = ind_ub;
    q.H = data->H;
    q.A = data->A;
    q.q = data->q;
    q.rhs = data->rhs;
    q.c0 = data->c0;
    q.x = data->x;
    q.xl = data->xl;
    q.xu = data->xu;
    q.zl = data->zl;
    q.zu = data->zu;
    q.y = data->y;
    q.opt = *opt;
    std::vector<int64_t> slot((size_t)std::max<int64_t>(m, 1), -1);
    for (int64_t k = 0; k < ns; ++k) {
        const int64_t r = ind_ineq_host[k];
        if (!(r >= 0 && r < m && slot[r] < 0 && (k == 0 || ind_ineq_host[k - 1] < r))) {
            delete b;
            return madqp_fail(ctx, MADQP_ERR_ARG, "ind_ineq must be strictly increasing row indices");
        }
        slot[r] = k;
    }
    int32_t r = MADQP_OK;
    int64_t *d_ineq = nullptr, *d_slot = nullptr;
#define BALLOC(ptr, count, ...)                                   \
    if (r == MADQP_OK) r = dalloc(b, &(ptr), (count), ##__VA_ARGS__)
    BALLOC(d_ineq, ns);
    BALLOC(d_slot, m);
    BALLOC(q.f, B * q.n);
    BALLOC(q.c, B * m);
    BALLOC(q.jacl, B * q.n);
    BALLOC(q.reg, B * q.n);
    BALLOC(q.pr_diag, B * q.n);
    BALLOC(q.du_diag, B * m);
    BALLOC(q.d, B * q.ntot);
    BALLOC(q.p, B * q.ntot);
    BALLOC(q.w1, B * q.ntot);
    BALLOC(q.w2, opt->max_ncorr > 0 ? B * q.ntot : 1);
    BALLOC(q.l_diag, B * nlb);
    BALLOC(q.l_lower, B * nlb);
    BALLOC(q.u_diag, B * nub);
    BALLOC(q.u_lower, B * nub);
    BALLOC(q.corr_lb, B * nlb);
    BALLOC(q.corr_ub, B * nub);
    BALLOC(q.theta, B * m);
    BALLOC(q.t, B * m);
    BALLOC(q.u, B * m);
    BALLOC(q.K, B * q.ldk * q.ldk + 128, true);
    BALLOC(q.S, B * q.kpad * q.npad);
    BALLOC(q.winv, B * q.nblk * WBLK, true);  // potf2_inv_kernel writes the lower parts only
    BALLOC(q.tmp, B * q.npad);
    BALLOC(q.scal, B * S_COUNT, true);
    BALLOC(q.status, B, true);
    BALLOC(q.iters, B, true);
    BALLOC(q.info, B, true);
    BALLOC(b->d_active, 1, true);
#undef BALLOC
    if (r == MADQP_OK && ns &&
        hipMemcpy(d_ineq, ind_ineq_host, ns * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
        r = madqp_fail(ctx, MADQP_ERR_HIP, "copy of ind_ineq failed");
    if (r == MADQP_OK && m && hipMemcpy(d_slot, slot.data(), m * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
        r = madqp_fail(ctx, MADQP_ERR_HIP, "copy of the slack map failed");
    if (r != MADQP_OK) {
        madqp_batch_destroy(b);
        return r;
    }
    q.ind_ineq = d_ineq;
    q.slot = d_slot;
    *out = b;
    return MADQP_OK;
}

// src/solver.jl:162-179 for every problem (the caller has done :127-159: bounds, interior push, scaling)
extern "C" int32_t madqp_batch_init(madqp_batch* b, double mu_init, double bound_fac) {
    if (!b) return MADQP_ERR_ARG;
    madqp_ctx* ctx = b->ctx;
    b->q.mu_init = mu_init;
    b->q.bound_fac = bound_fac;
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(bq_init_pre_kernel, dim3((unsigned)b->q.B), dim3(TPB), 0, ctx->stream, b->q);
        LAUNCH_CHECK(ctx);
    }
    int32_t r = factor_all(b);
    if (r) return r;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(bq_init_post_kernel, dim3((unsigned)b->q.B), dim3(TPB), 0, ctx->stream, b->q);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

// Up to max_steps lock-step iterations of mpc! (src/solver.jl:254-345); stops as soon as no problem
// is active (checked every `check_every` steps with one 4-byte read-back).  n_active_host: problems
// still active on return.
extern "C" int32_t madqp_batch_iterate(madqp_batch* b, int32_t max_steps, int32_t check_every,
                                       int32_t* n_active_host) {
    if (!b) return MADQP_ERR_ARG;
    madqp_ctx* ctx = b->ctx;
    ARG_TRY(ctx, max_steps >= 0 && check_every >= 1 && n_active_host);
    const BQ& q = b->q;
    int32_t active = -1;
    for (int32_t it = 0; it < max_steps; ++it) {
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            hipLaunchKernelGGL(bq_iter_pre_kernel, dim3((unsigned)q.B), dim3(TPB), 0, ctx->stream, q);
            LAUNCH_CHECK(ctx);
        }
        int32_t r = factor_all(b);
        if (r) return r;
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            hipLaunchKernelGGL(bq_iter_post_kernel, dim3((unsigned)q.B), dim3(TPB), 0, ctx->stream, q);
            LAUNCH_CHECK(ctx);
        }
        if ((it + 1) % check_every == 0 || it + 1 == max_steps) {
            hipLaunchKernelGGL(bq_count_active_kernel, dim3(1), dim3(64), 0, ctx->stream, q.status, q.B,
                               b->d_active);
            LAUNCH_CHECK(ctx);
            HIP_TRY(ctx, hipMemcpyAsync(&active, b->d_active, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (active == 0) break;
        }
    }
    if (active < 0) {
        hipLaunchKernelGGL(bq_count_active_kernel, dim3(1), dim3(64), 0, ctx->stream, q.status, q.B, b->d_active);
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, hipMemcpyAsync(&active, b->d_active, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_active_host = active;
    return MADQP_OK;
}

// status (0 active, 1 SOLVE_SUCCEEDED, 6 MAXIMUM_ITERATIONS_EXCEEDED, -3 ERROR_IN_STEP_COMPUTATION,
// -1 INTERNAL_ERROR), iteration count and the MADQP_BATCH_SCALARS scalars of every problem
extern "C" int32_t madqp_batch_results(madqp_batch* b, int32_t* status_host, int32_t* iters_host,
                                       double* scal_host) {
    if (!b) return MADQP_ERR_ARG;
    madqp_ctx* ctx = b->ctx;
    const BQ& q = b->q;
    if (status_host)
        HIP_TRY(ctx, hipMemcpyAsync(status_host, q.status, q.B * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (iters_host)
        HIP_TRY(ctx, hipMemcpyAsync(iters_host, q.iters, q.B * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (scal_host)
        HIP_TRY(ctx, hipMemcpyAsync(scal_host, q.scal, q.B * S_COUNT * sizeof(double), hipMemcpyDeviceToHost,
                                    ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MADQP_OK;
}
