// HIPCondensedKKTSystem: dense condensed KKT  K = H + Sigma_x + A' Theta A  (SURVEY.md 8a-note).
//
// The MI355X counterpart of NormalKKTSystem (src/KKT/normalkkt.jl): same plugin methods
// (build_kkt!, solve!, mul!, jtprod!), same sign conventions, but the slack block is eliminated
// so that the system stays n_x x n_x and positive definite for a QP with dense Hessian:
//   inequality row i with slack k:  Theta_i = S_k / (1 - dc_i S_k),  S_k = pr_diag[nx + k]
//   equality   row i             :  Theta_i = -1 / dc_i              (requires dc_i < 0)
//   rhs_x = r1_x + A' Theta (r2 + r1_s / S),   K dx = rhs_x,
//   dy = Theta (A dx - r2 - r1_s / S),          ds = (r1_s + dy) / S
//
// The same object also provides the reference's own formulation (mode NORMAL): the normal
// equations  S = A_full Sigma^-1 A_full' = A Sigma_x^-1 A' + diag(Sigma_s^-1)  (m x m) of
// src/KKT/normalkkt.jl:166-205, LP only as there (:45-48), dual regularization not added to the
// matrix.  It needs A' with row k (a variable) contiguous, which is again the
// k-major layout the GEMM core consumes; equality rows need no regularization in this form.
//
// Mode AUGMENTED is the K2 form of MadNLP's default SparseKKTSystem (src/utils.jl:108) with only the slack block
// eliminated:  [H + Sigma_x, A'; A, -D],  D_i = 1/S_k - dc_i (inequality row) or -dc_i (equality row), order
// nx + m.  It is quasi-definite, so L diag(I, -I) L' without pivoting is stable (madqp_chol_set_signature), and
// equality rows need no dual regularization when A_eq has full row rank:
//   [H + Sigma_x, A'; A, -D] [dx; dy] = [r1_x; r2 + r1_s / S],   ds = (r1_s + dy) / S
//
// Mode AUGMENTED with `scaled` set is K2.5, MadNLP's ScaledSparseKKTSystem as MadIPM drives it (src/kernels.jl:149-165;
// test/runtests.jl:95-115): l_diag = x - xl > 0, u_diag = xu - x > 0 (signs flipped against K2), a_j = l_diag or 1,
// b_j = u_diag or 1,  scaling_j = sqrt(a_j b_j),  pr_diag_j = zl_j b_j + zu_j a_j + dw a_j b_j = scaling^2 (dw + Sigma).
// The matrix is the symmetric scaling of K2 entry by entry as scripts/cuda_wrapper.jl:90-116 does it for the COO
// values -- pr_diag as it stands, Hessian entries times scaling_i scaling_j, Jacobian entries times scaling_j, du_diag
// as it stands -- with the slack block eliminated as in the unscaled case (1/S_k becomes scaling_k^2 / pr_diag_k):
// every entry stays bounded as the iterates converge.
#include <algorithm>

#include "common.h"

#define TPB 256
#define MADQP_MAX_BLOCKS 1024

enum { KKT_CONDENSED = 0, KKT_NORMAL = 1, KKT_AUGMENTED = 2 };

struct madqp_kkt {
    madqp_ctx* ctx;
    int mode;
    int64_t nx, m, ns;
    const double* H;
    int64_t ldh;
    const double* A;  // m x nx, row k = constraint k contiguous (condensed mode)
    int64_t lda;
    const double* At;  // nx x m, row k = variable k contiguous (normal mode)
    int64_t ldat;
    // sparse front end (either mode): CSR of A (m rows) and CSR of A' (nx rows); A == At == nullptr then
    const int64_t *a_ptr, *a_col, *at_ptr, *at_col;
    const double *a_val, *at_val;
    const double* hdiag;  // diagonal Hessian (nx), instead of the dense H; borrowed
    double* sg;           // Sigma + [hdiag; 0] (n), owned; the solves divide by it
    double *dn, *tn;  // normal mode: 1/Sigma (n) and an n-vector of scratch
    int64_t* d_ind_ineq;  // ns
    int64_t* d_slot;      // m: slack slot of a row, -1 for an equality row
    double* K;
    int64_t ldk;
    double *theta, *t, *u;  // m
    const double* u_is_A_of;  // condensed mode: u = A x for the primal part of this vector (the last solve's), or nullptr
    int64_t np;             // augmented mode: nx rounded up to a multiple of 128 = row of the first constraint
    double* b;              // augmented mode: right-hand side of order np + m
    int scaled;             // augmented mode: K2.5 (see the header comment)
    double *sfac, *sb;      // K2.5: scaling factor (n) and a scratch n-vector
    madqp_chol* chol;
    // refinement steps madqp_kkt_solve runs itself (madqp_kkt_set_refine; 0 = none, the default) and their two work vectors
    int32_t refine;
    double *rf_p, *rf_r;
    int64_t rf_len;
    // fused per-variable passes of the condensed mode (solve_pre_kernel / solve_post_kernel / resid_tail_kernel below):
    // where each variable sits in the two bound lists (-1: not there), built when the lists are first seen, and the
    // reduced right-hand side of the slacks kept between the two passes around the sweeps
    int32_t *pos_lb, *pos_ub;
    const int64_t *pos_key_lb, *pos_key_ub;
    int64_t pos_nlb, pos_nub, pos_n;
    double* rs;
    bool fuse;  // MADQP_KKT_FUSE != 0
};

namespace {
#define MQ_KERNEL __global__ __launch_bounds__(TPB) void
#define MQ_BLOCK blockIdx.x
#define GRID_STRIDE(i, len) \
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < (len); i += (int64_t)gridDim.x * TPB)
inline int grid_for(int64_t len) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((len + TPB - 1) / TPB, MADQP_MAX_BLOCKS));
}

#include "kkt_kernels.inc"

// ---- fused per-variable passes of the condensed mode ---------------------------------------------------------------------
// An iteration at n_x = 5 000 spent ~0.5 ms in ~100 launches of 2 us kernels that follow each other at the 4.7 us interval of
// dependent dispatches.  Three runs of them -- in front of the sweeps, behind them, and the tail of the residual
// p - K d -- touch each entry independently EXCEPT for the scatters of the bound lists into the variables
// (reduce_rhs!, _kktmul!), which is why they were separate launches in list order.  With the inverse of the lists
// (pos_lb / pos_ub: the place of a variable in each list) every variable gathers its two terms itself, in the order
// the scatters applied them -- the same operations on the same operands, so the results are bitwise those of the
// separate kernels (tests/test_gpu_solver.py compares the two forms; MADQP_KKT_FUSE=0 selects the separate ones).
__global__ __launch_bounds__(TPB) void pos_fill_kernel(int64_t n, int32_t* __restrict__ a, int32_t* __restrict__ b) {
    GRID_STRIDE(i, n) a[i] = b[i] = -1;
}
__global__ __launch_bounds__(TPB) void pos_scatter_kernel(int64_t cnt, const int64_t* __restrict__ ind, int32_t* __restrict__ pos) {
    GRID_STRIDE(i, cnt) pos[ind[i]] = (int32_t)i;
}
// reduce_rhs! for one variable (src/kernels.jl:150-161 order: lower list, then upper list)
__device__ __forceinline__ double reduced_rhs(const madqp_state& s, const double* __restrict__ p, int64_t j,
                                              const int32_t* __restrict__ pos_lb, const int32_t* __restrict__ pos_ub) {
    double v = p[j];
    const int32_t pl = pos_lb[j], pu = pos_ub[j];
    if (pl >= 0) v -= p[s.n + s.m + pl] / s.l_diag[pl];
    if (pu >= 0) v -= p[s.n + s.m + s.nlb + pu] / s.u_diag[pu];
    return v;
}
// w <- p with reduce_rhs! applied, t / u of condense_kernel, rs <- the reduced right-hand side of the slacks
__global__ __launch_bounds__(TPB) void solve_pre_kernel(madqp_state s, const double* __restrict__ p, double* __restrict__ w,
                                                        const int32_t* __restrict__ pos_lb, const int32_t* __restrict__ pos_ub,
                                                        int64_t nx, const int64_t* __restrict__ slot,
                                                        const double* __restrict__ theta, double* __restrict__ t,
                                                        double* __restrict__ u, double* __restrict__ rs) {
    int64_t L = s.n > s.m ? s.n : s.m;
    if (s.nlb > L) L = s.nlb;
    if (s.nub > L) L = s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.n) {
            const double v = reduced_rhs(s, p, i, pos_lb, pos_ub);
            w[i] = v;
            if (i >= nx) rs[i - nx] = v;
        }
        if (i < s.m) {
            const int64_t k = slot[i];
            double ti = p[s.n + i];
            if (k >= 0) ti += reduced_rhs(s, p, nx + k, pos_lb, pos_ub) / s.pr_diag[nx + k];
            t[i] = ti;
            u[i] = theta[i] * ti;
            w[s.n + i] = p[s.n + i];
        }
        if (i < s.nlb) w[s.n + s.m + i] = p[s.n + s.m + i];
        if (i < s.nub) w[s.n + s.m + s.nlb + i] = p[s.n + s.m + s.nlb + i];
    }
}
// decondense_kernel + finish_aug_solve_kernel (u holds A dx, w[0 .. nx) the solved dx), optionally pcopy <- p
__global__ __launch_bounds__(TPB) void solve_post_kernel(madqp_state s, double* __restrict__ w, int64_t nx,
                                                         const int64_t* __restrict__ slot, const int64_t* __restrict__ ind_ineq,
                                                         const double* __restrict__ theta, const double* __restrict__ t,
                                                         const double* __restrict__ u, const double* __restrict__ rs,
                                                         const double* __restrict__ p, double* __restrict__ pcopy) {
    int64_t L = s.n > s.m ? s.n : s.m;
    if (s.nlb > L) L = s.nlb;
    if (s.nub > L) L = s.nub;
    double* wzl = w + s.n + s.m;
    double* wzu = wzl + s.nlb;
    auto dx_of = [&](int64_t j) {  // the entry of dx the separate kernels would read at j
        if (j < nx) return w[j];
        const int64_t k = j - nx, r = ind_ineq[k];
        const double dy = theta[r] * (u[r] - t[r]);
        return (rs[k] + dy) / s.pr_diag[nx + k];
    };
    GRID_STRIDE(i, L) {
        if (i < s.m) {
            const int64_t k = slot[i];
            const double dy = theta[i] * (u[i] - t[i]);
            w[s.n + i] = dy;
            if (k >= 0) w[nx + k] = (rs[k] + dy) / s.pr_diag[nx + k];
        }
        if (i < s.nlb) wzl[i] = (-wzl[i] + s.l_lower[i] * dx_of(s.ind_lb[i])) / s.l_diag[i];
        if (i < s.nub) wzu[i] = (wzu[i] - s.u_lower[i] * dx_of(s.ind_ub[i])) / s.u_diag[i];
        if (pcopy) {
            if (i < s.n) pcopy[i] = p[i];
            if (i < s.m) pcopy[s.n + i] = p[s.n + i];
            if (i < s.nlb) pcopy[s.n + s.m + i] = p[s.n + s.m + i];
            if (i < s.nub) pcopy[s.n + s.m + s.nlb + i] = p[s.n + s.m + s.nlb + i];
        }
    }
}
// the tail of w = alpha K v + beta w behind the matrix products: jt_slack_kernel, mul_rows_kernel and the three passes of
// _kktmul! (vec_kernels.inc) in one
__global__ __launch_bounds__(TPB) void resid_tail_kernel(madqp_state s, double* __restrict__ w, const double* __restrict__ v,
                                                         double alpha, double beta, int64_t nx,
                                                         const int64_t* __restrict__ slot, const int64_t* __restrict__ ind_ineq,
                                                         const double* __restrict__ u, const int32_t* __restrict__ pos_lb,
                                                         const int32_t* __restrict__ pos_ub) {
    int64_t L = s.n > s.m ? s.n : s.m;
    if (s.nlb > L) L = s.nlb;
    if (s.nub > L) L = s.nub;
    double* wzl = w + s.n + s.m;
    double* wzu = wzl + s.nlb;
    const double* vzl = v + s.n + s.m;
    const double* vzu = vzl + s.nlb;
    GRID_STRIDE(i, L) {
        if (i < s.n) {
            double wi = w[i];
            if (i >= nx) {  // jt_slack_kernel
                const double val = alpha * (-v[s.n + ind_ineq[i - nx]]);
                wi = (beta == 0.0) ? val : val + beta * wi;
            }
            wi += alpha * s.reg[i] * v[i];  // kktmul_diag_kernel
            const int32_t pl = pos_lb[i], pu = pos_ub[i];
            if (pl >= 0) wi -= alpha * vzl[pl];  // kktmul_lb_kernel
            if (pu >= 0) wi += alpha * vzu[pu];  // kktmul_ub_kernel
            w[i] = wi;
        }
        if (i < s.m) {
            const int64_t k = slot[i];
            double a = u[i];  // mul_rows_kernel
            if (k >= 0) a -= v[nx + k];
            double wy = (beta == 0.0) ? alpha * a : alpha * a + beta * w[s.n + i];
            wy += alpha * s.du_diag[i] * v[s.n + i];
            w[s.n + i] = wy;
        }
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            wzl[i] = beta * wzl[i] + alpha * (v[j] * s.l_lower[i] - vzl[i] * s.l_diag[i]);
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            wzu[i] = beta * wzu[i] + alpha * (v[j] * s.u_lower[i] + vzu[i] * s.u_diag[i]);
        }
    }
}

// Lower triangle of the augmented matrix, one 64 x 64 tile per workgroup (blockIdx.x = tile row, .y = tile column):
//   rows/cols [0, nx): H + diag(dx);  [nx, np): identity (padding up to a block boundary);
//   rows np + r, cols [0, nx): A[r, :] (A is row-major: transposed through LDS);  rows/cols np + r: diag(dd).
__global__ __launch_bounds__(256) void aug_fill_kernel(int64_t nx, int64_t np, int64_t m, const double* __restrict__ H,
                                                       int64_t ldh, const double* __restrict__ hdiag,
                                                       const double* __restrict__ dx, const double* __restrict__ A,
                                                       int64_t lda, const double* __restrict__ dd,
                                                       const double* __restrict__ sf,  // K2.5 scaling or nullptr
                                                       double* __restrict__ K, int64_t ldk) {
    const int64_t ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;
    const int64_t i0 = ti * 64, j0 = tj * 64, N = np + m;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    __shared__ double tile[64][65];
    if (i0 >= np && j0 < np && !A) {  // sparse Jacobian: zero here, aug_scatter_kernel adds the entries
        const int64_t i = i0 + tx;
        if (i < N)
            for (int c = ty; c < 64; c += 4) K[i + (j0 + c) * ldk] = 0.0;
        return;
    }
    if (i0 >= np && j0 < np) {  // constraint rows x variable columns
        const int64_t r0 = i0 - np;
        for (int rr = ty; rr < 64; rr += 4) {
            const int64_t r = r0 + rr, j = j0 + tx;
            tile[rr][tx] = (r < m && j < nx) ? (sf ? A[r * lda + j] * sf[j] : A[r * lda + j]) : 0.0;
        }
        __syncthreads();
        const int64_t i = i0 + tx;
        if (i < N)
            for (int c = ty; c < 64; c += 4) K[i + (j0 + c) * ldk] = tile[tx][c];
        return;
    }
    const int64_t i = i0 + tx;
    if (i >= N) return;
    for (int c = ty; c < 64; c += 4) {
        const int64_t j = j0 + c;
        if (j > i) continue;
        double v = 0.0;
        if (i < nx) {
            if (H) v = sf ? H[i + j * ldh] * sf[i] * sf[j] : H[i + j * ldh];
            if (i == j) v += dx[i] + (hdiag ? (sf ? hdiag[i] * sf[i] * sf[i] : hdiag[i]) : 0.0);
        } else if (i == j) {
            v = (i < np) ? 1.0 : dd[i - np];
        }
        K[i + j * ldk] = v;
    }
}

// sparse Jacobian (CSR of A): K[np + r, col] = val, 16 lanes per row
__global__ __launch_bounds__(256) void aug_scatter_kernel(int64_t m, int64_t np, const int64_t* __restrict__ ptr,
                                                          const int64_t* __restrict__ col,
                                                          const double* __restrict__ val,
                                                          const double* __restrict__ sf, double* __restrict__ K,
                                                          int64_t ldk) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (r >= m) return;
    for (int64_t e = ptr[r] + (threadIdx.x & 15); e < ptr[r + 1]; e += 16)
        K[(np + r) + col[e] * ldk] = sf ? val[e] * sf[col[e]] : val[e];
}

// ---- K2.5 (scaled augmented system): set_aug_diagonal_reg!(::ScaledSparseKKTSystem) (src/kernels.jl:149-165) with
// MadNLP._set_aug_diagonal! folded in, and the sign-flipped reduce_rhs! / finish_aug_solve! / _kktmul! rows
__global__ __launch_bounds__(TPB) void k25_fill_kernel(madqp_state s, double del_w, double del_c, double* a, double* b) {
    const int64_t L = s.n > s.m ? s.n : s.m;
    GRID_STRIDE(i, L) {
        if (i < s.n) {
            s.reg[i] = del_w;
            a[i] = 1.0;
            b[i] = 1.0;
        }
        if (i < s.m) s.du_diag[i] = del_c;
    }
}
__global__ __launch_bounds__(TPB) void k25_bounds_kernel(madqp_state s, double* a, double* b) {
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            const double ld = s.x[j] - s.xl[j];  // (X - Xl), src/kernels.jl:157
            s.l_diag[i] = ld;
            s.l_lower[i] = s.zl[j];
            a[j] = ld;
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            const double ud = s.xu[j] - s.x[j];  // (Xu - X), :158
            s.u_diag[i] = ud;
            s.u_lower[i] = s.zu[j];
            b[j] = ud;
        }
    }
}
__global__ __launch_bounds__(TPB) void k25_final_kernel(madqp_state s, double del_w, double* a, const double* b) {
    GRID_STRIDE(j, s.n) {
        const double aj = a[j], bj = b[j];
        s.pr_diag[j] = s.zl[j] * bj + s.zu[j] * aj + del_w * (aj * bj);
        a[j] = sqrt(aj * bj);  // a becomes the scaling factor
    }
}
__global__ __launch_bounds__(TPB) void k25_reduce_kernel(madqp_state s, double* w) {
    const double* wzl = w + s.n + s.m;
    const double* wzu = wzl + s.nlb;
    // the two lists may name the same variable: two passes in one launch would race, so lb here, ub below
    GRID_STRIDE(i, s.nlb) w[s.ind_lb[i]] += wzl[i] / s.l_diag[i];
    (void)wzu;
}
__global__ __launch_bounds__(TPB) void k25_reduce_ub_kernel(madqp_state s, double* w) {
    const double* wzu = w + s.n + s.m + s.nlb;
    GRID_STRIDE(i, s.nub) w[s.ind_ub[i]] += wzu[i] / s.u_diag[i];
}
__global__ __launch_bounds__(TPB) void k25_finish_kernel(madqp_state s, double* w) {
    const double* wx = w;
    double* wzl = w + s.n + s.m;
    double* wzu = wzl + s.nlb;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) wzl[i] = (wzl[i] - s.l_lower[i] * wx[s.ind_lb[i]]) / s.l_diag[i];
        if (i < s.nub) wzu[i] = (s.u_lower[i] * wx[s.ind_ub[i]] - wzu[i]) / s.u_diag[i];
    }
}
__global__ __launch_bounds__(TPB) void k25_kktmul_diag_kernel(madqp_state s, double* w, const double* v, double alpha) {
    const int64_t L = s.n > s.m ? s.n : s.m;
    GRID_STRIDE(i, L) {
        if (i < s.n) w[i] += alpha * s.reg[i] * v[i];
        if (i < s.m) w[s.n + i] += alpha * s.du_diag[i] * v[s.n + i];
    }
}
__global__ __launch_bounds__(TPB) void k25_kktmul_lb_kernel(madqp_state s, double* w, const double* v, double alpha,
                                                            double beta) {
    double* wzl = w + s.n + s.m;
    const double* vzl = v + s.n + s.m;
    GRID_STRIDE(i, s.nlb) {
        const int64_t j = s.ind_lb[i];
        w[j] -= alpha * vzl[i];
        wzl[i] = beta * wzl[i] + alpha * (v[j] * s.l_lower[i] + vzl[i] * s.l_diag[i]);
    }
}
__global__ __launch_bounds__(TPB) void k25_kktmul_ub_kernel(madqp_state s, double* w, const double* v, double alpha,
                                                            double beta) {
    double* wzu = w + s.n + s.m + s.nlb;
    const double* vzu = v + s.n + s.m + s.nlb;
    GRID_STRIDE(i, s.nub) {
        const int64_t j = s.ind_ub[i];
        w[j] += alpha * vzu[i];
        wzu[i] = beta * wzu[i] + alpha * (v[j] * s.u_lower[i] - vzu[i] * s.u_diag[i]);
    }
}
__global__ __launch_bounds__(TPB) void k25_ones_kernel(int64_t n, double* a) {
    GRID_STRIDE(i, n) a[i] = 1.0;
}
}  // namespace

#define KLAUNCH(kern, len, ...)                                                                 \
    do {                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid_for(len)), dim3(TPB), 0, ctx->stream, __VA_ARGS__);  \
        LAUNCH_CHECK(ctx);                                                                      \
    } while (0)

// y(m) = alpha A x(nx) + beta y   /   y(nx) = alpha A' x(m) + beta y, from whichever layout is held
static int32_t apply_A(madqp_kkt* k, double alpha, const double* x, double beta, double* y) {
    if (k->a_ptr)
        return madqp_spmv_csr(k->ctx, k->m, k->a_ptr, k->a_col, k->a_val, alpha, x, beta, y, MADQP_PROF_GEMV);
    if (k->A) return madqp_gemv_impl(k->ctx, 0, k->m, k->nx, alpha, k->A, k->lda, x, beta, y, MADQP_PROF_GEMV);
    return madqp_gemv_impl(k->ctx, 1, k->nx, k->m, alpha, k->At, k->ldat, x, beta, y, MADQP_PROF_GEMV);
}
static int32_t apply_At(madqp_kkt* k, double alpha, const double* x, double beta, double* y) {
    if (k->a_ptr)
        return madqp_spmv_csr(k->ctx, k->nx, k->at_ptr, k->at_col, k->at_val, alpha, x, beta, y, MADQP_PROF_GEMV);
    if (k->A) return madqp_gemv_impl(k->ctx, 1, k->m, k->nx, alpha, k->A, k->lda, x, beta, y, MADQP_PROF_GEMV);
    return madqp_gemv_impl(k->ctx, 0, k->nx, k->m, alpha, k->At, k->ldat, x, beta, y, MADQP_PROF_GEMV);
}

extern "C" int32_t madqp_kkt_destroy(madqp_kkt* k) {
    if (!k) return MADQP_OK;
    (void)hipStreamSynchronize(k->ctx->stream);
    if (k->chol) madqp_chol_destroy(k->chol);
    if (k->d_ind_ineq) (void)hipFree(k->d_ind_ineq);
    if (k->d_slot) (void)hipFree(k->d_slot);
    if (k->K) (void)hipFree(k->K);
    if (k->theta) (void)hipFree(k->theta);
    if (k->t) (void)hipFree(k->t);
    if (k->u) (void)hipFree(k->u);
    if (k->sg) (void)hipFree(k->sg);
    if (k->dn) (void)hipFree(k->dn);
    if (k->tn) (void)hipFree(k->tn);
    if (k->b) (void)hipFree(k->b);
    if (k->sfac) (void)hipFree(k->sfac);
    if (k->sb) (void)hipFree(k->sb);
    if (k->rf_p) (void)hipFree(k->rf_p);
    if (k->rf_r) (void)hipFree(k->rf_r);
    if (k->pos_lb) (void)hipFree(k->pos_lb);
    if (k->pos_ub) (void)hipFree(k->pos_ub);
    if (k->rs) (void)hipFree(k->rs);
    delete k;
    return MADQP_OK;
}

static int32_t kkt_create_common(madqp_ctx* ctx, int mode, int64_t nx, int64_t m, int64_t ns,
                                 const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                 const double* A, int64_t lda, const double* At, int64_t ldat,
                                 madqp_kkt** out) {
    ARG_TRY(ctx, out && nx >= 0 && m >= 0 && ns >= 0 && ns <= m);
    ARG_TRY(ctx, ns == 0 || ind_ineq_host);
    *out = nullptr;
    std::vector<int64_t> slot((size_t)std::max<int64_t>(m, 1), -1);
    for (int64_t k = 0; k < ns; ++k) {
        const int64_t r = ind_ineq_host[k];
        ARG_TRY(ctx, r >= 0 && r < m && slot[r] < 0);
        ARG_TRY(ctx, k == 0 || ind_ineq_host[k - 1] < r);
        slot[r] = k;
    }
    madqp_kkt* k = new (std::nothrow) madqp_kkt();
    if (!k) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    memset(k, 0, sizeof(*k));
    k->ctx = ctx;
    k->mode = mode;
    k->nx = nx;
    k->m = m;
    k->ns = ns;
    {
        const char* e_fuse = getenv("MADQP_KKT_FUSE");  // 0: the separate per-variable kernels (A/B tests)
        k->fuse = !(e_fuse && e_fuse[0] == '0');
    }
    k->H = H;
    k->ldh = ldh;
    k->A = A;
    k->lda = lda;
    k->At = At;
    k->ldat = ldat;
    k->np = (nx + 127) / 128 * 128;
    // order of the matrix that is factorised
    const int64_t dim = (mode == KKT_NORMAL) ? m : (mode == KKT_AUGMENTED) ? k->np + m : nx;
    const int64_t n = nx + ns;
    // leading dimension and column count padded to a multiple of 128 (zero filled): every GEMM tile
    // of the factorisation is a full tile (see GemmArgs::Mread)
    const int64_t dpad = std::max<int64_t>(128, (dim + 127) / 128 * 128);
    k->ldk = dpad;
    const size_t mb = (size_t)std::max<int64_t>(m, 1) * sizeof(double);
    const size_t nb = (size_t)std::max<int64_t>(n, 1) * sizeof(double);
    // +128 doubles of slack after the last column (room for 16-byte tile loads at the edge)
    hipError_t e = hipMalloc(&k->K, ((size_t)k->ldk * dpad + 128) * sizeof(double));
    if (e == hipSuccess) e = hipMemset(k->K, 0, ((size_t)k->ldk * dpad + 128) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&k->d_ind_ineq, (size_t)std::max<int64_t>(ns, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->d_slot, (size_t)std::max<int64_t>(m, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->theta, mb);
    if (e == hipSuccess) e = hipMalloc(&k->t, mb);
    if (e == hipSuccess) e = hipMalloc(&k->u, mb);
    if (e == hipSuccess && mode == KKT_NORMAL) e = hipMalloc(&k->dn, nb);
    if (e == hipSuccess && mode == KKT_NORMAL) e = hipMalloc(&k->tn, nb);
    if (e == hipSuccess && mode == KKT_AUGMENTED) e = hipMalloc(&k->b, (size_t)std::max<int64_t>(dim, 1) * sizeof(double));
    if (e == hipSuccess && ns)
        e = hipMemcpy(k->d_ind_ineq, ind_ineq_host, ns * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && m)
        e = hipMemcpy(k->d_slot, slot.data(), m * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        madqp_kkt_destroy(k);
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_kkt_create(nx=%lld, m=%lld): %s",
                          (long long)nx, (long long)m, hipGetErrorString(e));
    }
    int32_t r = madqp_chol_create(ctx, dim, &k->chol);
    if (!r && mode == KKT_AUGMENTED) r = madqp_chol_set_signature(k->chol, k->np);
    if (r) {
        madqp_kkt_destroy(k);
        return r;
    }
    *out = k;
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_create(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                    const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                    const double* A, int64_t lda, madqp_kkt** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, !H || ldh >= nx);
    ARG_TRY(ctx, m == 0 || nx == 0 || (A && lda >= nx));
    return kkt_create_common(ctx, KKT_CONDENSED, nx, m, ns, ind_ineq_host, H, ldh, A, lda, nullptr, 0, out);
}

// Sparse front end: A as CSR (a_*: m rows, column indices < nx, ascending within a row) and A' as CSR
// (at_*: nx rows, indices < m), both device, int64, borrowed.  mode 0: condensed K = H + Sigma_x + A' Theta A
// (H dense or NULL), mode 1: the reference's normal equations A Sigma^-1 A' (LP only), mode 2: the augmented
// system [H + Sigma_x, A'; A, -D] (H dense, NULL or diagonal through madqp_kkt_set_hdiag).
extern "C" int32_t madqp_kkt_create_sparse(madqp_ctx* ctx, int32_t mode, int64_t nx, int64_t m, int64_t ns,
                                           const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                           const int64_t* a_ptr, const int64_t* a_col, const double* a_val,
                                           const int64_t* at_ptr, const int64_t* at_col, const double* at_val,
                                           madqp_kkt** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, (mode >= 0 && mode <= 2) && a_ptr && at_ptr && (!H || ldh >= nx) && !(mode == 1 && H));
    int32_t r = kkt_create_common(ctx, mode == 1 ? KKT_NORMAL : mode == 2 ? KKT_AUGMENTED : KKT_CONDENSED, nx, m, ns,
                                  ind_ineq_host, H, ldh,
                                  nullptr, 0, nullptr, 0, out);
    if (r) return r;
    madqp_kkt* k = *out;
    k->a_ptr = a_ptr;
    k->a_col = a_col;
    k->a_val = a_val;
    k->at_ptr = at_ptr;
    k->at_col = at_col;
    k->at_val = at_val;
    return MADQP_OK;
}

// Diagonal Hessian H = diag(hdiag) (nx entries, device, borrowed) for a KKT object created without a dense
// H: the condensed matrix becomes diag(hdiag) + Sigma_x + A' Theta A, the normal equations A (H + Sigma)^-1 A'
// ("the alternative of row 6 applies verbatim with Sigma -> H + Sigma") -- the reference's
// NormalKKTSystem is LP only (src/KKT/normalkkt.jl:45-48); this is the diagonal-H extension CONT-type QPs need.
extern "C" int32_t madqp_kkt_set_hdiag(madqp_kkt* k, const double* hdiag) {
    if (!k) return MADQP_ERR_ARG;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, hdiag && !k->H);
    if (!k->sg) {
        const size_t nb = (size_t)std::max<int64_t>(k->nx + k->ns, 1) * sizeof(double);
        hipError_t e = hipMalloc(&k->sg, nb);
        if (e != hipSuccess) return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_kkt_set_hdiag: %s", hipGetErrorString(e));
    }
    k->hdiag = hdiag;
    return MADQP_OK;
}

// Augmented (K2) system [H + Sigma_x, A'; A, -D] of order ceil128(nx) + m; same operands as madqp_kkt_create.
// H may be NULL (LP, or a diagonal Hessian through madqp_kkt_set_hdiag).
extern "C" int32_t madqp_kkt_create_augmented(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                              const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                              const double* A, int64_t lda, madqp_kkt** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, !H || ldh >= nx);
    ARG_TRY(ctx, m == 0 || nx == 0 || (A && lda >= nx));
    return kkt_create_common(ctx, KKT_AUGMENTED, nx, m, ns, ind_ineq_host, H, ldh, A, lda, nullptr, 0, out);
}

// K2.5: the augmented system symmetrically scaled (MadNLP's ScaledSparseKKTSystem; see the header comment).
extern "C" int32_t madqp_kkt_create_scaled_augmented(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                                     const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                                     const double* A, int64_t lda, madqp_kkt** out) {
    int32_t r = madqp_kkt_create_augmented(ctx, nx, m, ns, ind_ineq_host, H, ldh, A, lda, out);
    if (r) return r;
    madqp_kkt* k = *out;
    const int64_t n = nx + ns;
    const size_t nb = (size_t)std::max<int64_t>(n, 1) * sizeof(double);
    hipError_t e = hipMalloc(&k->sfac, nb);
    if (e == hipSuccess) e = hipMalloc(&k->sb, nb);
    if (e != hipSuccess) {
        madqp_kkt_destroy(k);
        *out = nullptr;
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_kkt_create_scaled_augmented: %s", hipGetErrorString(e));
    }
    k->scaled = 1;
    if (n) KLAUNCH(k25_ones_kernel, n, n, k->sfac);
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_create_normal(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                           const int64_t* ind_ineq_host, const double* At,
                                           int64_t ldat, madqp_kkt** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, m == 0 || nx == 0 || (At && ldat >= m));
    return kkt_create_common(ctx, KKT_NORMAL, nx, m, ns, ind_ineq_host, nullptr, 0, nullptr, 0, At, ldat, out);
}

static int32_t check_kkt_state(madqp_kkt* k, const madqp_state* st) {
    if (!k) return MADQP_ERR_ARG;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, st != nullptr);
    ARG_TRY(ctx, st->n == k->nx + k->ns && st->m == k->m);
    return MADQP_OK;
}

int32_t madqp_syrk_assemble_ranges(madqp_ctx* ctx, int64_t n, int64_t kdim, const double* B, int64_t ldb,
                                   const double* w, const double* base, int64_t ldbase,
                                   const double* dvec, double* C, int64_t ldc, int64_t nranges,
                                   const int64_t* ranges);  // gemm_f64.hip; ranges == nullptr: everything

static int32_t kkt_build_impl(madqp_kkt* k, const madqp_state* st, int64_t nranges, const int64_t* ranges) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    if (k->mode == KKT_NORMAL) {
        // S = A diag(D_x) A' + diag(D_s on inequality rows),  D = 1/Sigma; du_diag is NOT added
        // (src/KKT/normalkkt.jl:166-180, src/utils.jl:266-298)
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            if (k->hdiag && st->n) KLAUNCH(sigma_h_kernel, st->n, st->n, k->nx, st->pr_diag, k->hdiag, k->sg);
            if (st->n) KLAUNCH(recip_kernel, st->n, st->n, k->hdiag ? k->sg : st->pr_diag, k->dn);
            if (k->m) KLAUNCH(slack_diag_kernel, k->m, k->m, k->nx, k->d_slot, k->dn, k->theta);
        }
        if (k->a_ptr) {  // V = A (rows = constraints), weights 1/Sigma over the variables
            ARG_TRY(ctx, ranges == nullptr);
            return madqp_sparse_gram(ctx, k->m, k->a_ptr, k->a_col, k->a_val, k->at_ptr, k->at_col, k->at_val, k->dn,
                                     nullptr, 0, k->theta, k->K, k->ldk);
        }
        return madqp_syrk_assemble_ranges(ctx, k->m, k->nx, k->At, k->ldat, k->dn, nullptr, 0, k->theta,
                                          k->K, k->ldk, nranges, ranges);
    }
    if (k->mode == KKT_AUGMENTED) {
        ARG_TRY(ctx, ranges == nullptr);
        const int64_t N = k->np + k->m;
        if (N == 0) return MADQP_OK;
        ProfScope ps(ctx, MADQP_PROF_SYRK);
        const double* sf = k->scaled ? k->sfac : nullptr;
        if (k->m) KLAUNCH(aug_diag_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, st->du_diag, sf, k->theta);
        const unsigned tiles = (unsigned)((N + 63) / 64);
        hipLaunchKernelGGL(aug_fill_kernel, dim3(tiles, tiles), dim3(256), 0, ctx->stream, k->nx, k->np, k->m, k->H,
                           k->ldh, k->hdiag, st->pr_diag, k->A, k->lda, k->theta, sf, k->K, k->ldk);
        LAUNCH_CHECK(ctx);
        if (k->a_ptr && k->m) {
            hipLaunchKernelGGL(aug_scatter_kernel, dim3((unsigned)((k->m * 16 + 255) / 256)), dim3(256), 0, ctx->stream,
                               k->m, k->np, k->a_ptr, k->a_col, k->a_val, sf, k->K, k->ldk);
            LAUNCH_CHECK(ctx);
        }
        return MADQP_OK;
    }
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (k->m) KLAUNCH(theta_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, st->du_diag, k->theta);
        if (k->hdiag && st->n) KLAUNCH(sigma_h_kernel, st->n, st->n, k->nx, st->pr_diag, k->hdiag, k->sg);
    }
    const double* dvec = k->hdiag ? k->sg : st->pr_diag;  // diagonal of H + Sigma_x
    if (k->a_ptr) {  // V = A' (rows = variables), weights Theta over the constraints
        ARG_TRY(ctx, ranges == nullptr);
        return madqp_sparse_gram(ctx, k->nx, k->at_ptr, k->at_col, k->at_val, k->a_ptr, k->a_col, k->a_val, k->theta,
                                 k->H, k->ldh, dvec, k->K, k->ldk);
    }
    return madqp_syrk_assemble_ranges(ctx, k->nx, k->m, k->A, k->lda, k->theta, k->H, k->ldh, dvec,
                                      k->K, k->ldk, nranges, ranges);
}

// set_aug_diagonal_reg!(kkt, solver), dispatched on the KKT type as the reference does (src/kernels.jl:128 for every
// AbstractKKTSystem, :149 for ScaledSparseKKTSystem)
extern "C" int32_t madqp_kkt_set_aug_diagonal_reg(madqp_kkt* k, const madqp_state* st, double del_w, double del_c) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    if (!k->scaled) return madqp_set_aug_diagonal_reg(ctx, st, del_w, del_c);
    ARG_TRY(ctx, st->x && st->xl && st->xu && st->zl && st->zu);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (std::max(st->n, st->m) > 0) KLAUNCH(k25_fill_kernel, std::max(st->n, st->m), *st, del_w, del_c, k->sfac, k->sb);
    if (std::max(st->nlb, st->nub) > 0) KLAUNCH(k25_bounds_kernel, std::max(st->nlb, st->nub), *st, k->sfac, k->sb);
    if (st->n) KLAUNCH(k25_final_kernel, st->n, *st, del_w, k->sfac, k->sb);
    return MADQP_OK;
}

// MadNLP.initialize!(kkt) (src/KKT/normalkkt.jl:136-147): reg = pr_diag = 1, du_diag = 0, l_lower = u_lower = 0,
// l_diag = u_diag = 1 -- and, for K2.5, scaling factor 1
extern "C" int32_t madqp_kkt_initialize(madqp_kkt* k, const madqp_state* st) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    if ((r = madqp_fill(ctx, st->n, 1.0, st->reg)) || (r = madqp_fill(ctx, st->n, 1.0, st->pr_diag)) ||
        (r = madqp_fill(ctx, st->m, 0.0, st->du_diag)) || (r = madqp_fill(ctx, st->nlb, 0.0, st->l_lower)) ||
        (r = madqp_fill(ctx, st->nub, 0.0, st->u_lower)) || (r = madqp_fill(ctx, st->nlb, 1.0, st->l_diag)) ||
        (r = madqp_fill(ctx, st->nub, 1.0, st->u_diag)))
        return r;
    if (k->scaled && st->n) KLAUNCH(k25_ones_kernel, st->n, st->n, k->sfac);
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_build(madqp_kkt* k, const madqp_state* st) {
    return kkt_build_impl(k, st, 0, nullptr);
}

// build_kkt! restricted to the block columns [ranges[2r], ranges[2r+1]) of the lower triangle
// (multi-GPU path: a rank assembles the panels it owns).  Theta and the vectors the solves read
// are computed in full, so solve!/mul! work on every rank.
extern "C" int32_t madqp_kkt_chol(madqp_kkt* k, madqp_chol** chol, int64_t* order) {
    if (!k || !chol) return MADQP_ERR_ARG;
    *chol = k->chol;
    if (order) *order = (k->mode == KKT_NORMAL) ? k->m : (k->mode == KKT_AUGMENTED) ? k->np + k->m : k->nx;
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_factorize(madqp_kkt* k, int32_t* info_host) {
    if (!k) return MADQP_ERR_ARG;
    return madqp_chol_factor(k->chol, k->K, k->ldk, info_host);
}
// queued form: the factorisation is enqueued, its info lands in the result block; madqp_kkt_factor_result hands the
// value the caller read back to the solver object
int32_t madqp_q_kkt_factorize(madqp_kkt* k, int slot0) {
    if (!k) return MADQP_ERR_ARG;
    return madqp_chol_factor_q(k->chol, k->K, k->ldk, k->ctx->d_res + slot0);
}
int32_t madqp_kkt_factor_result(madqp_kkt* k, int32_t info) {
    if (!k) return MADQP_ERR_ARG;
    madqp_chol_factor_result(k->chol, info);
    return MADQP_OK;
}

static int32_t kkt_solve_once(madqp_kkt* k, const madqp_state* st, double* w);

// The fused passes serve the condensed mode with dense or sparse A, unscaled; the lists of st must be the ones the
// inverse maps were made from (same pointers and lengths: the state layout is fixed while a KKT object lives,
// include/madqp.h) -- they are made again when that changes.
static bool kkt_fusable(const madqp_kkt* k) { return k->fuse && k->mode == KKT_CONDENSED && !k->scaled && k->m > 0; }
static int32_t ensure_pos(madqp_kkt* k, const madqp_state* st) {
    madqp_ctx* ctx = k->ctx;
    if (k->pos_lb && k->pos_key_lb == st->ind_lb && k->pos_key_ub == st->ind_ub && k->pos_nlb == st->nlb &&
        k->pos_nub == st->nub && k->pos_n == st->n)
        return MADQP_OK;
    if (k->pos_n != st->n || !k->pos_lb) {
        (void)hipStreamSynchronize(ctx->stream);
        if (k->pos_lb) (void)hipFree(k->pos_lb);
        if (k->pos_ub) (void)hipFree(k->pos_ub);
        if (k->rs) (void)hipFree(k->rs);
        k->pos_lb = k->pos_ub = nullptr;
        k->rs = nullptr;
        const size_t nn = (size_t)std::max<int64_t>(st->n, 1);
        if (hipMalloc(&k->pos_lb, nn * sizeof(int32_t)) != hipSuccess || hipMalloc(&k->pos_ub, nn * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(&k->rs, (size_t)std::max<int64_t>(k->ns, 1) * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            return madqp_fail(ctx, MADQP_ERR_ALLOC, "inverse bound lists (%lld variables)", (long long)st->n);
        }
    }
    ARG_TRY(ctx, st->n < (int64_t)1 << 31 && st->nlb < (int64_t)1 << 31 && st->nub < (int64_t)1 << 31);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (st->n) KLAUNCH(pos_fill_kernel, st->n, st->n, k->pos_lb, k->pos_ub);
    if (st->nlb) KLAUNCH(pos_scatter_kernel, st->nlb, st->nlb, st->ind_lb, k->pos_lb);
    if (st->nub) KLAUNCH(pos_scatter_kernel, st->nub, st->nub, st->ind_ub, k->pos_ub);
    k->pos_key_lb = st->ind_lb;
    k->pos_key_ub = st->ind_ub;
    k->pos_nlb = st->nlb;
    k->pos_nub = st->nub;
    k->pos_n = st->n;
    return MADQP_OK;
}

// w = K^-1 p (p untouched; w a different vector), optionally pcopy = p on the way: solve_system!'s copy, solve! and the
// copy in front of its residual (src/linear_solver.jl:19-35) -- in the condensed mode with the per-variable passes fused
int32_t madqp_kkt_solve_from(madqp_kkt* k, const madqp_state* st, const double* p, double* w, double* pcopy) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, p && w && p != w);
    const int64_t len = st->n + st->m + st->nlb + st->nub;
    if (!kkt_fusable(k) || k->refine > 0) {
        if ((r = madqp_copy(ctx, len, p, w))) return r;
        if ((r = madqp_kkt_solve(k, st, w))) return r;
        return pcopy ? madqp_copy(ctx, len, p, pcopy) : MADQP_OK;
    }
    if ((r = ensure_pos(k, st))) return r;
    k->u_is_A_of = nullptr;
    const int64_t L = std::max(std::max(st->n, st->m), std::max(st->nlb, st->nub));
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(solve_pre_kernel, L, *st, p, w, k->pos_lb, k->pos_ub, k->nx, k->d_slot, k->theta, k->t, k->u, k->rs);
    }
    if ((r = apply_At(k, 1.0, k->u, 1.0, w))) return r;  // rhs_x = r1_x + A' (theta t)
    if ((r = madqp_chol_solve(k->chol, w))) return r;
    if ((r = apply_A(k, 1.0, w, 0.0, k->u))) return r;
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(solve_post_kernel, L, *st, w, k->nx, k->d_slot, k->d_ind_ineq, k->theta, k->t, k->u, k->rs, p, pcopy);
    }
    k->u_is_A_of = w;  // dx is final since the sweeps: u = A dx (madqp_kkt_mul_solved)
    return MADQP_OK;
}

// Steps of iterative refinement INSIDE solve!: w = K^-1 p, then w += K^-1 (p - K w) -- for hosts whose loop is not ours.
// MadIPM's solve_system! (src/linear_solver.jl:19-45) calls solve!(kkt, d) once and only LOOKS at the residual; the device
// factorisation multiplies with explicit inverses of 16 x 16 sub-blocks where LAPACK substitutes scalar by scalar, which on
// small ill-conditioned problems shows in the per-iteration traces and which one step removes.  The
// Julia glue asks for the AUTO rule (steps = -1: one step while the factorised matrix has order <= 1024, none above -- the
// rule madqp_jl_amd/options.py applies in the drivers of this repository, which refine in their own solve_system with the
// residual they form anyway and leave this at 0).  Arithmetic: operation for operation the drivers' (bitwise equal results).
extern "C" int32_t madqp_kkt_set_refine(madqp_kkt* k, int32_t steps) {
    if (!k) return MADQP_ERR_ARG;
    ARG_TRY(k->ctx, steps >= -1 && steps <= 8);
    if (steps < 0) {
        const int64_t order = k->mode == KKT_NORMAL ? k->m : k->mode == KKT_AUGMENTED ? k->nx + k->m : k->nx;
        static const int64_t auto_max = getenv("MADQP_REFINE_AUTO_MAX") ? atoll(getenv("MADQP_REFINE_AUTO_MAX")) : 1024;
        steps = order <= auto_max ? 1 : 0;
    }
    k->refine = steps;
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_solve(madqp_kkt* k, const madqp_state* st, double* w) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    ARG_TRY(k->ctx, w != nullptr);
    if (k->refine <= 0) return kkt_solve_once(k, st, w);
    madqp_ctx* ctx = k->ctx;
    const int64_t len = st->n + st->m + st->nlb + st->nub;
    if (len > k->rf_len) {
        (void)hipStreamSynchronize(ctx->stream);
        if (k->rf_p) (void)hipFree(k->rf_p);
        if (k->rf_r) (void)hipFree(k->rf_r);
        k->rf_p = k->rf_r = nullptr;
        k->rf_len = 0;
        if (hipMalloc(&k->rf_p, len * sizeof(double)) != hipSuccess || hipMalloc(&k->rf_r, len * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_kkt_solve: work vectors of the refinement steps (%lld doubles)", (long long)len);
        }
        k->rf_len = len;
    }
    if ((r = madqp_copy(ctx, len, w, k->rf_p))) return r;  // p
    if ((r = kkt_solve_once(k, st, w))) return r;           // w = K^-1 p
    for (int32_t it = 0; it < k->refine; ++it) {
        if ((r = madqp_copy(ctx, len, k->rf_p, k->rf_r))) return r;
        if ((r = madqp_kkt_mul(k, st, k->rf_r, w, -1.0, 1.0))) return r;  // r = p - K w
        if ((r = kkt_solve_once(k, st, k->rf_r))) return r;
        if ((r = madqp_axpy(ctx, len, 1.0, k->rf_r, w))) return r;         // w += K^-1 r
    }
    k->u_is_A_of = nullptr;  // (u belongs to the last correction, not to w)
    return MADQP_OK;
}

static int32_t kkt_solve_once(madqp_kkt* k, const madqp_state* st, double* w) {
    int32_t r = 0;
    madqp_ctx* ctx = k->ctx;
    k->u_is_A_of = nullptr;
    double* wx = w;
    double* wy = w + st->n;
    if (k->scaled) {  // r1 = p_x + p_zl / l_diag + p_zu / u_diag (signs of src/kernels.jl:157-158)
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (st->nlb) KLAUNCH(k25_reduce_kernel, st->nlb, *st, w);
        if (st->nub) KLAUNCH(k25_reduce_ub_kernel, st->nub, *st, w);
    } else if ((r = madqp_reduce_rhs(ctx, st, w))) {
        return r;
    }
    if (k->mode == KKT_NORMAL) {  // src/KKT/normalkkt.jl:185-201
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            if (st->n) KLAUNCH(div_kernel, st->n, st->n, wx, k->hdiag ? k->sg : st->pr_diag, k->tn);  // r1 = (H + Sigma)^-1 wx
        }
        if ((r = apply_A(k, 1.0, k->tn, 0.0, k->u))) return r;
        if (k->m) {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            KLAUNCH(normal_rhs_kernel, k->m, k->m, k->d_slot, k->u, k->tn + k->nx, wy, wy);  // A r1 - r2
        }
        if ((r = madqp_chol_solve(k->chol, wy))) return r;  // wy = dy
        if ((r = apply_At(k, 1.0, wy, 0.0, k->tn))) return r;
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            if (k->ns) KLAUNCH(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, wy, k->tn + k->nx, 1.0, 0.0);
            if (st->n) KLAUNCH(normal_back_kernel, st->n, st->n, k->tn, k->hdiag ? k->sg : st->pr_diag, wx);
        }
        return madqp_finish_aug_solve(ctx, st, w);
    }
    if (k->mode == KKT_AUGMENTED) {
        const int64_t N = k->np + k->m;
        const double* sf = k->scaled ? k->sfac : nullptr;
        if (N) {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            KLAUNCH(aug_rhs_kernel, N, k->nx, k->np, k->m, k->d_slot, st->pr_diag, sf, wx, wy, k->b);
        }
        if ((r = madqp_chol_solve(k->chol, k->b))) return r;
        if (k->nx + k->m) {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            KLAUNCH(aug_back_kernel, k->nx + k->m, k->nx, k->np, k->m, k->d_slot, st->pr_diag, sf, k->b, wx, wy);
        }
        if (k->scaled) {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            if (std::max(st->nlb, st->nub) > 0) KLAUNCH(k25_finish_kernel, std::max(st->nlb, st->nub), *st, w);
            return MADQP_OK;
        }
        return madqp_finish_aug_solve(ctx, st, w);
    }
    if (k->m) {
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            KLAUNCH(condense_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, k->theta, wx, wy,
                    k->t, k->u);
        }
        // rhs_x = r1_x + A' (theta t)
        if ((r = apply_At(k, 1.0, k->u, 1.0, wx))) return r;
    }
    if ((r = madqp_chol_solve(k->chol, wx))) return r;
    if (k->m) {
        if ((r = apply_A(k, 1.0, wx, 0.0, k->u))) return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(decondense_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, k->theta, k->t, k->u,
                wx, wy);
    }
    if ((r = madqp_finish_aug_solve(ctx, st, w))) return r;
    if (k->m) k->u_is_A_of = w;  // dx is final since the sweeps: u = A dx (madqp_kkt_mul_solved)
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_jtprod(madqp_kkt* k, double* out, const double* y) {
    if (!k) return MADQP_ERR_ARG;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, (out || k->nx + k->ns == 0) && (y || k->m == 0));  // no constraints: y is empty
    int32_t r;
    if ((r = apply_At(k, 1.0, y, 0.0, out))) return r;
    if (k->ns) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, y, out + k->nx, 1.0, 0.0);
    }
    return MADQP_OK;
}

static int32_t kkt_mul_impl(madqp_kkt* k, const madqp_state* st, double* w, const double* v, double alpha, double beta,
                            bool v_is_last_solution);
extern "C" int32_t madqp_kkt_mul(madqp_kkt* k, const madqp_state* st, double* w, const double* v,
                                 double alpha, double beta) {
    return kkt_mul_impl(k, st, w, v, alpha, beta, false);
}
extern "C" int32_t madqp_kkt_mul_solved(madqp_kkt* k, const madqp_state* st, double* w, const double* v,
                                        double alpha, double beta) {
    return kkt_mul_impl(k, st, w, v, alpha, beta, true);
}
static int32_t kkt_mul_impl(madqp_kkt* k, const madqp_state* st, double* w, const double* v, double alpha, double beta,
                            bool v_is_last_solution) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, w && v);
    const int64_t nx = k->nx, n = st->n;
    const bool have_Av = v_is_last_solution && k->m && k->u_is_A_of == v;  // the solve's own A dx, same kernel and operands
    k->u_is_A_of = nullptr;
    // wx = alpha A_full' vy + beta wx  (+ alpha H vx)
    if ((r = apply_At(k, alpha, v + n, beta, w))) return r;
    if (k->H && nx)  // H is symmetric: its lower triangle is enough (gemv.hip: madqp_symv_lower)
        if ((r = madqp_symv_lower(ctx, nx, alpha, k->H, k->ldh, v, 1.0, w, MADQP_PROF_GEMV)))
            return r;
    if (k->hdiag && nx) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(hdiag_axpy_kernel, nx, nx, alpha, k->hdiag, v, w);
    }
    if (kkt_fusable(k)) {  // the five per-variable passes that follow, in one (resid_tail_kernel)
        if (!have_Av && (r = apply_A(k, 1.0, v, 0.0, k->u))) return r;
        if ((r = ensure_pos(k, st))) return r;
        const int64_t L = std::max(std::max(st->n, st->m), std::max(st->nlb, st->nub));
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(resid_tail_kernel, L, *st, w, v, alpha, beta, nx, k->d_slot, k->d_ind_ineq, k->u, k->pos_lb, k->pos_ub);
        return MADQP_OK;
    }
    if (k->ns) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, v + n, w + nx, alpha, beta);
    }
    // wy = alpha A_full vx + beta wy
    if (k->m) {
        if (!have_Av && (r = apply_A(k, 1.0, v, 0.0, k->u))) return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(mul_rows_kernel, k->m, k->m, k->d_slot, k->u, v + nx, w + n, alpha, beta);
    }
    if (k->scaled) {  // mul!(w, ::ScaledSparseKKTSystem, v): zl dx + l_diag dzl, zu dx - u_diag dzu
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (std::max(st->n, st->m) > 0) KLAUNCH(k25_kktmul_diag_kernel, std::max(st->n, st->m), *st, w, v, alpha);
        if (st->nlb) KLAUNCH(k25_kktmul_lb_kernel, st->nlb, *st, w, v, alpha, beta);
        if (st->nub) KLAUNCH(k25_kktmul_ub_kernel, st->nub, *st, w, v, alpha, beta);
        return MADQP_OK;
    }
    return madqp_kktmul(ctx, st, w, v, alpha, beta);
}

// objective pieces into 2 slots (q'x and x'Hx; obj = c0 + s0 + s1/2), gradient and constraint values into st
int32_t madqp_q_kkt_eval(madqp_kkt* k, const madqp_state* st, const double* q, const double* rhs, int slot0) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, (k->nx == 0 || q) && (k->m == 0 || rhs));
    const int64_t nx = k->nx, n = st->n;
    if (k->H && nx)
        if ((r = madqp_symv_lower(ctx, nx, 1.0, k->H, k->ldh, st->x, 0.0, st->f, MADQP_PROF_GEMV)))
            return r;
    if (k->hdiag && nx) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(hdiag_mul_kernel, nx, nx, k->hdiag, st->x, st->f);
    }
    if (n) {
        const int nb = grid_for(n);
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            hipLaunchKernelGGL(eval_grad_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, n, nx,
                               ((k->H || k->hdiag) && nx) ? 1 : 0, q, st->x, st->f, ctx->d_part);
            LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(sum2_final_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part, nb,
                               ctx->d_res + slot0);
            LAUNCH_CHECK(ctx);
        }
    } else {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_res + slot0, 0, 2 * sizeof(double), ctx->stream));
    }
    if (k->m) {
        if ((r = apply_A(k, 1.0, st->x, 0.0, st->c))) return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(eval_cons_kernel, k->m, k->m, k->d_slot, st->x + nx, rhs, st->c);
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_eval(madqp_kkt* k, const madqp_state* st, const double* q,
                                  const double* rhs, double c0, double* obj_host) {
    if (!k) return MADQP_ERR_ARG;
    ARG_TRY(k->ctx, obj_host != nullptr);
    int32_t r = madqp_q_kkt_eval(k, st, q, rhs, 0);
    if (r) return r;
    double sums[2] = {0.0, 0.0};
    if ((r = madqp_read_results(k->ctx, 2, sums))) return r;
    *obj_host = c0 + sums[0] + 0.5 * sums[1];
    return MADQP_OK;
}

madqp_ctx* madqp_kkt_ctx(madqp_kkt* kkt) { return kkt->ctx; }

extern "C" int32_t madqp_kkt_matrix(madqp_kkt* k, double** K, int64_t* ld) {
    if (!k || !K || !ld) return MADQP_ERR_ARG;
    *K = k->K;
    *ld = k->ldk;
    return MADQP_OK;
}
