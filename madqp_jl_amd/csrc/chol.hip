// Blocked left-looking fp64 Cholesky (lower, column-major, in place) and the triangular solves.
//
// Replaces the AbstractLinearSolver used behind NormalKKTSystem (MadNLP.LapackCPUSolver -> LAPACK
// dpotrf/dpotrs; reference call sites src/KKT/normalkkt.jl:99-101,196, src/linear_solver.jl:10-11).
//
// Factorisation, two levels of left-looking blocking so that >96 % of the flops run in wide GEMMs:
//   for each outer panel J (NBO = 1024 columns)
//     C[J0:n, J]  -= L[J0:n, 0:J0] * L[J, 0:J0]'            gemm core, N = 1024, K = J0   (MFMA)
//     for each 128-column block jb inside J
//       C[jb:n, jb] -= L[jb:n, J0:jb] * L[jb, J0:jb]'       gemm core, N = 128,  K <= 896 (MFMA)
//       L_jj = chol(C_jj),  W_jj = L_jj^-1                   one workgroup, block resident in LDS
//       L[jb+128:n, jb] = C[jb+128:n, jb] * W_jj'            gemm core, K = N = 128        (MFMA)
// The inverse diagonal blocks W are kept (two images, 2 x n x 128 doubles) and turn the diagonal
// solves of the two triangular sweeps into 128 x 128 mat-vecs; the sweeps are HBM bound
// (4 n^2 bytes each).
#include <algorithm>

#include "common.h"

namespace {
constexpr int NB = 128;    // diagonal block
constexpr int NBO = 1024;  // outer panel
constexpr int LDS_LD = NB + 1;
constexpr int64_t WBLK = 2 * NB * NB;  // doubles per block in chol->winv: [Wcm | Wrm]

// Unblocked right-looking Cholesky of one nb x nb (nb <= 128) diagonal block held in LDS,
// followed by the inversion of the triangular factor.  info: the first failing column
// (1-based, LAPACK dpotrf convention) is recorded once; the block is then completed with a
// unit pivot so that the launch always terminates.
// Outputs: L_jj in place; Wcm[r + c*NB] = W(r,c) (column-major) and Wrm[c + r*NB] = W(r,c)
// (row-major), both zero padded to 128 x 128.
__global__ __launch_bounds__(256) void potf2_inv_kernel(double* __restrict__ A, int64_t lda, int nb,
                                                        double* __restrict__ Wcm,
                                                        double* __restrict__ Wrm,
                                                        int32_t* __restrict__ info, int32_t col0) {
    __shared__ double S[NB * LDS_LD];  // S[c*LDS_LD + r] = element (r, c)
    __shared__ double dinv[NB];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int c = idx / NB, r = idx % NB;
        double v = 0.0;
        if (r < nb && c < nb && r >= c) v = A[r + (int64_t)c * lda];
        S[c * LDS_LD + r] = v;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    for (int k = 0; k < nb; ++k) {
        double akk = S[k * LDS_LD + k];
        if (!(akk > 0.0)) {  // also catches NaN
            if (tid == 0) atomicCAS(info, 0, col0 + k + 1);
            akk = 1.0;
        }
        const double d = sqrt(akk);
        const double rd = 1.0 / d;
        __syncthreads();  // every thread has read the pivot
        for (int r = k + 1 + tid; r < nb; r += 256) S[k * LDS_LD + r] *= rd;
        if (tid == 0) S[k * LDS_LD + k] = d;
        __syncthreads();
        for (int c = k + 1 + ty; c < nb; c += 8) {
            const double lck = S[k * LDS_LD + c];
            for (int r = c + tx; r < nb; r += 32) S[c * LDS_LD + r] -= S[k * LDS_LD + r] * lck;
        }
        __syncthreads();
    }
    // factor -> global (lower triangle only)
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int c = idx / nb, r = idx % nb;
        if (r >= c) A[r + (int64_t)c * lda] = S[c * LDS_LD + r];
    }
    __syncthreads();
    // W = L^-1, column c by thread c; entry (r, c), r > c, is parked in the unused upper part
    // at S[r*LDS_LD + c].  Loops are wave uniform so that L[r,k] is an LDS broadcast.
    if (tid < NB) {
        const int c = tid;
        const double xc = (c < nb) ? 1.0 / S[c * LDS_LD + c] : 0.0;
        dinv[c] = xc;
        for (int r = 1; r < nb; ++r) {
            const double lrr = S[r * LDS_LD + r];
            double acc = 0.0;
            for (int k = 0; k < r; ++k) {
                const double l = S[k * LDS_LD + r];
                const double up = S[k * LDS_LD + c];
                const double xk = (k > c) ? up : (k == c ? xc : 0.0);
                acc += l * xk;
            }
            if (r > c) S[r * LDS_LD + c] = -acc / lrr;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        {  // column-major image: idx = c*NB + r
            const int c = idx / NB, r = idx % NB;
            double v = 0.0;
            if (r < nb && c < nb) v = (r == c) ? dinv[c] : (r > c ? S[r * LDS_LD + c] : 0.0);
            Wcm[idx] = v;
        }
        {  // row-major image: idx = r*NB + c
            const int r = idx / NB, c = idx % NB;
            double v = 0.0;
            if (r < nb && c < nb) v = (r == c) ? dinv[c] : (r > c ? S[r * LDS_LD + c] : 0.0);
            Wrm[idx] = v;
        }
    }
}

// forward diagonal step:  b_j <- W_jj b_j       (y[r] = sum_{c<=r} W(r,c) b[c]), lanes over r
__global__ __launch_bounds__(128) void trsv_diag_fwd_kernel(const double* __restrict__ Wcm,
                                                            double* __restrict__ b, int nb) {
    __shared__ double bs[NB];
    const int r = threadIdx.x;
    bs[r] = (r < nb) ? b[r] : 0.0;
    __syncthreads();
    double acc = 0.0;
    for (int c = 0; c < nb; ++c) acc += Wcm[r + c * NB] * bs[c];  // W is zero above the diagonal
    if (r < nb) b[r] = acc;
}

// backward diagonal step:  b_j <- W_jj' (b_j - t)   (y[c] = sum_{r>=c} W(r,c) (b[r]-t[r])), lanes over c
__global__ __launch_bounds__(128) void trsv_diag_bwd_kernel(const double* __restrict__ Wrm,
                                                            double* __restrict__ b,
                                                            const double* __restrict__ t, int nb) {
    __shared__ double bs[NB];
    const int c = threadIdx.x;
    bs[c] = (c < nb) ? (b[c] - (t ? t[c] : 0.0)) : 0.0;
    __syncthreads();
    double acc = 0.0;
    for (int r = 0; r < nb; ++r) acc += Wrm[c + r * NB] * bs[r];
    if (c < nb) b[c] = acc;
}
}  // namespace

extern "C" int32_t madqp_chol_create(madqp_ctx* ctx, int64_t n, madqp_chol** out) {
    ARG_TRY(ctx, ctx && out && n >= 0);
    *out = nullptr;
    madqp_chol* s = new (std::nothrow) madqp_chol();
    if (!s) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    s->ctx = ctx;
    s->n = n;
    s->factored = false;
    s->A = nullptr;
    s->lda = 0;
    s->winv = nullptr;
    s->tmp = nullptr;
    s->d_info = nullptr;
    const int64_t nblk = std::max<int64_t>(1, (n + NB - 1) / NB);
    hipError_t e = hipMalloc(&s->winv, nblk * WBLK * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->tmp, std::max<int64_t>(NB, n) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->d_info, sizeof(int32_t));
    if (e != hipSuccess) {
        madqp_chol_destroy(s);
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_chol_create(%lld): %s", (long long)n,
                          hipGetErrorString(e));
    }
    *out = s;
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_destroy(madqp_chol* s) {
    if (!s) return MADQP_OK;
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->winv) (void)hipFree(s->winv);
    if (s->tmp) (void)hipFree(s->tmp);
    if (s->d_info) (void)hipFree(s->d_info);
    delete s;
    return MADQP_OK;
}

static int32_t panel_update(madqp_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t row0,
                            int64_t k0, int64_t width) {
    // C[row0:n, row0:row0+width] -= L[row0:n, k0:row0] * L[row0:row0+width, k0:row0]'
    GemmArgs g{};
    g.X = A + row0 + k0 * lda;
    g.ldx = lda;
    g.Y = g.X;
    g.ldy = lda;
    g.C = A + row0 + row0 * lda;
    g.ldc = lda;
    g.Cin = g.C;
    g.ldcin = lda;
    g.alpha = -1.0;
    g.beta = 1.0;
    g.M = n - row0;
    g.N = width;
    g.K = row0 - k0;
    g.diag_off = 0;
    g.lower_only = 1;
    return madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_GEMM);
}

extern "C" int32_t madqp_chol_factor(madqp_chol* s, double* A, int64_t lda, int32_t* info_host) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, A && info_host && lda >= s->n);
    const int64_t n = s->n;
    s->factored = false;
    s->A = A;
    s->lda = lda;
    *info_host = 0;
    if (n == 0) {
        s->factored = true;
        return MADQP_OK;
    }
    HIP_TRY(ctx, hipMemsetAsync(s->d_info, 0, sizeof(int32_t), ctx->stream));
    for (int64_t J0 = 0; J0 < n; J0 += NBO) {
        const int64_t W = std::min<int64_t>(NBO, n - J0);
        if (J0 > 0) {
            int32_t r = panel_update(ctx, A, lda, n, J0, 0, W);
            if (r) return r;
        }
        for (int64_t jb = J0; jb < J0 + W; jb += NB) {
            const int64_t w = std::min<int64_t>(NB, n - jb);
            if (jb > J0) {
                int32_t r = panel_update(ctx, A, lda, n, jb, J0, w);
                if (r) return r;
            }
            double* Wcm = s->winv + (jb / NB) * WBLK;
            double* Wrm = Wcm + NB * NB;
            {
                ProfScope ps(ctx, MADQP_PROF_POTRF_DIAG);
                hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, ctx->stream,
                                   A + jb + jb * lda, lda, (int)w, Wcm, Wrm, s->d_info,
                                   (int32_t)jb);
                LAUNCH_CHECK(ctx);
            }
            if (jb + w < n) {
                // L[jb+w:n, jb:jb+w] = C[jb+w:n, jb:jb+w] * W',  out[i,j] = sum_k C[i,k] W(j,k);
                // in place: a single tile column, every workgroup reads exactly the rows it writes
                // and finishes reading (K = w, all stages) before its epilogue stores.
                GemmArgs g{};
                g.X = A + (jb + w) + jb * lda;
                g.ldx = lda;
                g.Y = Wcm;  // Y[j + k*NB] = W(j,k)
                g.ldy = NB;
                g.C = A + (jb + w) + jb * lda;
                g.ldc = lda;
                g.alpha = 1.0;
                g.beta = 0.0;
                g.M = n - jb - w;
                g.N = w;
                g.K = w;
                int32_t r = madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_TRSM);
                if (r) return r;
            }
        }
    }
    int32_t info = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&info, s->d_info, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *info_host = info;
    s->factored = (info == 0);
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_solve(madqp_chol* s, double* rhs) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, rhs != nullptr || s->n == 0);
    if (!s->A) return madqp_fail(ctx, MADQP_ERR_STATE, "madqp_chol_solve before madqp_chol_factor");
    const int64_t n = s->n, lda = s->lda;
    const double* A = s->A;
    // forward: L y = b
    for (int64_t jb = 0; jb < n; jb += NB) {
        const int64_t w = std::min<int64_t>(NB, n - jb);
        const double* Wcm = s->winv + (jb / NB) * WBLK;
        {
            ProfScope ps(ctx, MADQP_PROF_TRSV);
            hipLaunchKernelGGL(trsv_diag_fwd_kernel, dim3(1), dim3(NB), 0, ctx->stream, Wcm,
                               rhs + jb, (int)w);
            LAUNCH_CHECK(ctx);
        }
        const int64_t below = n - jb - w;
        if (below > 0) {
            // b[below] -= L[below, jb:jb+w] * y_j : memory rows = the w columns, each `below` long
            int32_t r = madqp_gemv_impl(ctx, 1, w, below, -1.0, A + (jb + w) + jb * lda, lda,
                                        rhs + jb, 1.0, rhs + jb + w, MADQP_PROF_TRSV);
            if (r) return r;
        }
    }
    // backward: L' x = y
    const int64_t last = ((n - 1) / NB) * NB;
    for (int64_t jb = last; jb >= 0; jb -= NB) {
        const int64_t w = std::min<int64_t>(NB, n - jb);
        const double* Wrm = s->winv + (jb / NB) * WBLK + NB * NB;
        const int64_t below = n - jb - w;
        const double* t = nullptr;
        if (below > 0) {
            // t = L[below, jb:jb+w]' x[below]
            int32_t r = madqp_gemv_impl(ctx, 0, w, below, 1.0, A + (jb + w) + jb * lda, lda,
                                        rhs + jb + w, 0.0, s->tmp, MADQP_PROF_TRSV);
            if (r) return r;
            t = s->tmp;
        }
        ProfScope ps(ctx, MADQP_PROF_TRSV);
        hipLaunchKernelGGL(trsv_diag_bwd_kernel, dim3(1), dim3(NB), 0, ctx->stream, Wrm, rhs + jb,
                           t, (int)w);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}
