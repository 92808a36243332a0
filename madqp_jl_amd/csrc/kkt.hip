// HIPCondensedKKTSystem: dense condensed KKT  K = H + Sigma_x + A' Theta A  (SURVEY.md 8a-note).
//
// The MI355X counterpart of NormalKKTSystem (src/KKT/normalkkt.jl): same plugin methods
// (build_kkt!, solve!, mul!, jtprod!), same sign conventions, but the slack block is eliminated
// so that the system stays n_x x n_x and positive definite for a QP with dense Hessian:
//   inequality row i with slack k:  Theta_i = S_k / (1 - dc_i S_k),  S_k = pr_diag[nx + k]
//   equality   row i             :  Theta_i = -1 / dc_i              (requires dc_i < 0)
//   rhs_x = r1_x + A' Theta (r2 + r1_s / S),   K dx = rhs_x,
//   dy = Theta (A dx - r2 - r1_s / S),          ds = (r1_s + dy) / S
#include <algorithm>

#include "common.h"

#define TPB 256
#define MADQP_MAX_BLOCKS 1024

struct madqp_kkt {
    madqp_ctx* ctx;
    int64_t nx, m, ns;
    const double* H;
    int64_t ldh;
    const double* A;
    int64_t lda;
    int64_t* d_ind_ineq;  // ns
    int64_t* d_slot;      // m: slack slot of a row, -1 for an equality row
    double* K;
    int64_t ldk;
    double *theta, *t, *u;  // m
    madqp_chol* chol;
};

namespace {
#define GRID_STRIDE(i, len) \
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < (len); i += (int64_t)gridDim.x * TPB)
inline int grid_for(int64_t len) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((len + TPB - 1) / TPB, MADQP_MAX_BLOCKS));
}

__global__ __launch_bounds__(TPB) void theta_kernel(int64_t m, int64_t nx,
                                                    const int64_t* __restrict__ slot,
                                                    const double* __restrict__ pr_diag,
                                                    const double* __restrict__ du_diag,
                                                    double* __restrict__ theta) {
    GRID_STRIDE(i, m) {
        const int64_t k = slot[i];
        if (k >= 0) {
            const double S = pr_diag[nx + k];
            theta[i] = S / (1.0 - du_diag[i] * S);
        } else {
            theta[i] = -1.0 / du_diag[i];
        }
    }
}

// t = r2 + r1_s / S ;  u = theta * t
__global__ __launch_bounds__(TPB) void condense_kernel(int64_t m, int64_t nx,
                                                       const int64_t* __restrict__ slot,
                                                       const double* __restrict__ pr_diag,
                                                       const double* __restrict__ theta,
                                                       const double* __restrict__ wx,
                                                       const double* __restrict__ wy,
                                                       double* __restrict__ t, double* __restrict__ u) {
    GRID_STRIDE(i, m) {
        const int64_t k = slot[i];
        double ti = wy[i];
        if (k >= 0) ti += wx[nx + k] / pr_diag[nx + k];
        t[i] = ti;
        u[i] = theta[i] * ti;
    }
}

// dy = theta (u - t), ds = (r1_s + dy) / S       (u holds A dx on entry)
__global__ __launch_bounds__(TPB) void decondense_kernel(int64_t m, int64_t nx,
                                                         const int64_t* __restrict__ slot,
                                                         const double* __restrict__ pr_diag,
                                                         const double* __restrict__ theta,
                                                         const double* __restrict__ t,
                                                         const double* __restrict__ u,
                                                         double* __restrict__ wx, double* __restrict__ wy) {
    GRID_STRIDE(i, m) {
        const int64_t k = slot[i];
        const double dy = theta[i] * (u[i] - t[i]);
        wy[i] = dy;
        if (k >= 0) wx[nx + k] = (wx[nx + k] + dy) / pr_diag[nx + k];
    }
}

__global__ __launch_bounds__(TPB) void jt_slack_kernel(int64_t ns, const int64_t* __restrict__ ind,
                                                       const double* __restrict__ y,
                                                       double* __restrict__ out_s, double alpha,
                                                       double beta) {
    GRID_STRIDE(k, ns) {
        const double v = alpha * (-y[ind[k]]);
        out_s[k] = (beta == 0.0) ? v : v + beta * out_s[k];
    }
}

// wy_i = alpha (u_i - vx_s[slot]) + beta wy_i
__global__ __launch_bounds__(TPB) void mul_rows_kernel(int64_t m, const int64_t* __restrict__ slot,
                                                       const double* __restrict__ u,
                                                       const double* __restrict__ vx_s,
                                                       double* __restrict__ wy, double alpha,
                                                       double beta) {
    GRID_STRIDE(i, m) {
        const int64_t k = slot[i];
        double a = u[i];
        if (k >= 0) a -= vx_s[k];
        wy[i] = (beta == 0.0) ? alpha * a : alpha * a + beta * wy[i];
    }
}

// f <- Hx + q (first nx), 0 (slacks);  partial sums of q'x and x'Hx
__global__ __launch_bounds__(TPB) void eval_grad_kernel(int64_t n, int64_t nx, int has_h,
                                                        const double* __restrict__ q,
                                                        const double* __restrict__ x,
                                                        double* __restrict__ f,
                                                        double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    double s1 = 0.0, s2 = 0.0;
    GRID_STRIDE(i, n) {
        if (i < nx) {
            const double hx = has_h ? f[i] : 0.0;
            const double xi = x[i];
            s1 += q[i] * xi;
            s2 += xi * hx;
            f[i] = hx + q[i];
        } else {
            f[i] = 0.0;
        }
    }
    for (int which = 0; which < 2; ++which) {
        double v = which ? s2 : s1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) part[blockIdx.x * 2 + which] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}
__global__ __launch_bounds__(TPB) void sum2_final_kernel(const double* __restrict__ part, int nblocks,
                                                         double* __restrict__ res) {
    __shared__ double sm[TPB / 64];
    for (int which = 0; which < 2; ++which) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += TPB) v += part[b * 2 + which];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) res[which] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

// c_i = (u_i - s_slot) - rhs_i
__global__ __launch_bounds__(TPB) void eval_cons_kernel(int64_t m, const int64_t* __restrict__ slot,
                                                        const double* __restrict__ xs,
                                                        const double* __restrict__ rhs,
                                                        double* __restrict__ c) {
    GRID_STRIDE(i, m) {
        const int64_t k = slot[i];
        double a = c[i];
        if (k >= 0) a -= xs[k];
        c[i] = a - rhs[i];
    }
}
}  // namespace

#define KLAUNCH(kern, len, ...)                                                                 \
    do {                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid_for(len)), dim3(TPB), 0, ctx->stream, __VA_ARGS__);  \
        LAUNCH_CHECK(ctx);                                                                      \
    } while (0)

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
    delete k;
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_create(madqp_ctx* ctx, int64_t nx, int64_t m, int64_t ns,
                                    const int64_t* ind_ineq_host, const double* H, int64_t ldh,
                                    const double* A, int64_t lda, madqp_kkt** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, out && nx >= 0 && m >= 0 && ns >= 0 && ns <= m);
    ARG_TRY(ctx, ns == 0 || ind_ineq_host);
    ARG_TRY(ctx, !H || ldh >= nx);
    ARG_TRY(ctx, m == 0 || nx == 0 || (A && lda >= nx));
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
    k->nx = nx;
    k->m = m;
    k->ns = ns;
    k->H = H;
    k->ldh = ldh;
    k->A = A;
    k->lda = lda;
    k->ldk = std::max<int64_t>(16, (nx + 15) / 16 * 16);
    const size_t mb = (size_t)std::max<int64_t>(m, 1) * sizeof(double);
    hipError_t e = hipMalloc(&k->K, (size_t)k->ldk * std::max<int64_t>(nx, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&k->d_ind_ineq, (size_t)std::max<int64_t>(ns, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->d_slot, (size_t)std::max<int64_t>(m, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->theta, mb);
    if (e == hipSuccess) e = hipMalloc(&k->t, mb);
    if (e == hipSuccess) e = hipMalloc(&k->u, mb);
    if (e == hipSuccess && ns)
        e = hipMemcpy(k->d_ind_ineq, ind_ineq_host, ns * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && m)
        e = hipMemcpy(k->d_slot, slot.data(), m * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        madqp_kkt_destroy(k);
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_kkt_create(nx=%lld, m=%lld): %s",
                          (long long)nx, (long long)m, hipGetErrorString(e));
    }
    int32_t r = madqp_chol_create(ctx, nx, &k->chol);
    if (r) {
        madqp_kkt_destroy(k);
        return r;
    }
    *out = k;
    return MADQP_OK;
}

static int32_t check_kkt_state(madqp_kkt* k, const madqp_state* st) {
    if (!k) return MADQP_ERR_ARG;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, st != nullptr);
    ARG_TRY(ctx, st->n == k->nx + k->ns && st->m == k->m);
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_build(madqp_kkt* k, const madqp_state* st) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    if (k->m) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(theta_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, st->du_diag, k->theta);
    }
    return madqp_syrk_assemble(ctx, k->nx, k->m, k->A, k->lda, k->theta, k->H, k->ldh, st->pr_diag,
                               k->K, k->ldk);
}

extern "C" int32_t madqp_kkt_factorize(madqp_kkt* k, int32_t* info_host) {
    if (!k) return MADQP_ERR_ARG;
    return madqp_chol_factor(k->chol, k->K, k->ldk, info_host);
}

extern "C" int32_t madqp_kkt_solve(madqp_kkt* k, const madqp_state* st, double* w) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, w != nullptr);
    double* wx = w;
    double* wy = w + st->n;
    if ((r = madqp_reduce_rhs(ctx, st, w))) return r;
    if (k->m) {
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            KLAUNCH(condense_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, k->theta, wx, wy,
                    k->t, k->u);
        }
        // rhs_x = r1_x + A' (theta t)
        if ((r = madqp_gemv_impl(ctx, 1, k->m, k->nx, 1.0, k->A, k->lda, k->u, 1.0, wx, MADQP_PROF_GEMV)))
            return r;
    }
    if ((r = madqp_chol_solve(k->chol, wx))) return r;
    if (k->m) {
        if ((r = madqp_gemv_impl(ctx, 0, k->m, k->nx, 1.0, k->A, k->lda, wx, 0.0, k->u, MADQP_PROF_GEMV)))
            return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(decondense_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, k->theta, k->t, k->u,
                wx, wy);
    }
    return madqp_finish_aug_solve(ctx, st, w);
}

extern "C" int32_t madqp_kkt_jtprod(madqp_kkt* k, double* out, const double* y) {
    if (!k) return MADQP_ERR_ARG;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, (out && y) || (k->nx + k->ns == 0));
    int32_t r;
    if ((r = madqp_gemv_impl(ctx, 1, k->m, k->nx, 1.0, k->A, k->lda, y, 0.0, out, MADQP_PROF_GEMV)))
        return r;
    if (k->ns) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, y, out + k->nx, 1.0, 0.0);
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_mul(madqp_kkt* k, const madqp_state* st, double* w, const double* v,
                                 double alpha, double beta) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, w && v);
    const int64_t nx = k->nx, n = st->n;
    // wx = alpha A_full' vy + beta wx  (+ alpha H vx)
    if ((r = madqp_gemv_impl(ctx, 1, k->m, nx, alpha, k->A, k->lda, v + n, beta, w, MADQP_PROF_GEMV)))
        return r;
    if (k->H && nx)
        if ((r = madqp_gemv_impl(ctx, 0, nx, nx, alpha, k->H, k->ldh, v, 1.0, w, MADQP_PROF_GEMV)))
            return r;
    if (k->ns) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, v + n, w + nx, alpha, beta);
    }
    // wy = alpha A_full vx + beta wy
    if (k->m) {
        if ((r = madqp_gemv_impl(ctx, 0, k->m, nx, 1.0, k->A, k->lda, v, 0.0, k->u, MADQP_PROF_GEMV)))
            return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(mul_rows_kernel, k->m, k->m, k->d_slot, k->u, v + nx, w + n, alpha, beta);
    }
    return madqp_kktmul(ctx, st, w, v, alpha, beta);
}

extern "C" int32_t madqp_kkt_eval(madqp_kkt* k, const madqp_state* st, const double* q,
                                  const double* rhs, double c0, double* obj_host) {
    int32_t r = check_kkt_state(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, obj_host && (k->nx == 0 || q) && (k->m == 0 || rhs));
    const int64_t nx = k->nx, n = st->n;
    if (k->H && nx)
        if ((r = madqp_gemv_impl(ctx, 0, nx, nx, 1.0, k->H, k->ldh, st->x, 0.0, st->f, MADQP_PROF_GEMV)))
            return r;
    double sums[2] = {0.0, 0.0};
    if (n) {
        const int nb = grid_for(n);
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            hipLaunchKernelGGL(eval_grad_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, n, nx,
                               (k->H && nx) ? 1 : 0, q, st->x, st->f, ctx->d_part);
            LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(sum2_final_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part, nb,
                               ctx->d_res);
            LAUNCH_CHECK(ctx);
        }
    }
    if (k->m) {
        if ((r = madqp_gemv_impl(ctx, 0, k->m, nx, 1.0, k->A, k->lda, st->x, 0.0, st->c, MADQP_PROF_GEMV)))
            return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        KLAUNCH(eval_cons_kernel, k->m, k->m, k->d_slot, st->x + nx, rhs, st->c);
    }
    if (n) {
        if ((r = madqp_read_results(ctx, 2, sums))) return r;
    } else {
        if ((r = madqp_ctx_sync(ctx))) return r;
    }
    *obj_host = c0 + sums[0] + 0.5 * sums[1];
    return MADQP_OK;
}

extern "C" int32_t madqp_kkt_matrix(madqp_kkt* k, double** K, int64_t* ld) {
    if (!k || !K || !ld) return MADQP_ERR_ARG;
    *K = k->K;
    *ld = k->ldk;
    return MADQP_OK;
}
