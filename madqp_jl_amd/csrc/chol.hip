// Blocked left-looking fp64 Cholesky (lower, column-major, in place) and the triangular solves.
//
// Replaces the AbstractLinearSolver used behind NormalKKTSystem (MadNLP.LapackCPUSolver -> LAPACK
// dpotrf/dpotrs; reference call sites src/KKT/normalkkt.jl:99-101,196, src/linear_solver.jl:10-11).
//
// Factorisation, two levels of left-looking blocking so that >96 % of the flops run in wide GEMMs:
//   for each outer panel J (768..2048 columns, chosen to fill the last round of workgroups)
//     C[J0:n, J]  -= L[J0:n, 0:J0] * L[J, 0:J0]'            gemm core, N = |J|,  K = J0   (MFMA)
//     inside J, recursively: factor the first half, update the second half with it
//       C[h:n, h:w] -= L[h:n, 0:h] * L[h:w, 0:h]'           gemm core, N = K = w/2        (MFMA)
//     down to 128-column blocks jb:
//       L_jj = chol(C_jj),  W_jj = L_jj^-1                   one workgroup, block resident in LDS
//       L[jb+128:n, jb] = C[jb+128:n, jb] * W_jj'            gemm core, K = N = 128        (MFMA)
// The inverse diagonal blocks W are kept (two images, 2 x n x 128 doubles) and turn the diagonal
// solves of the two triangular sweeps into 128 x 128 mat-vecs; the sweeps are HBM bound
// (4 n^2 bytes each).
#include <algorithm>

#include <vector>

#include "common.h"

namespace {
constexpr int NB = 128;    // diagonal block
constexpr int NBO = 1024;  // outer panel
constexpr int LDS_LD = NB + 1;
constexpr int64_t WBLK = 2 * NB * NB;  // doubles per block in chol->winv: [Wcm | Wrm]

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int SB = 16;             // sub-block of the diagonal kernel (one MFMA tile)
constexpr int NSB = NB / SB;       // 8 sub-block columns
constexpr int WD_LD = SB + 1;      // row stride of the inverse diagonal sub-blocks

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One wave: Cholesky of the 16 x 16 sub-block at (b, b) of S and the inverse of its factor.
// Lane r (= lane & 15) keeps row r of the sub-block in registers; column k is broadcast with
// v_readlane, so the 16 elimination steps need no LDS round trip and no barrier.
// Writes L (lower) back to S and W = L^-1 to Wd[r*WD_LD + c] (zero above the diagonal).
__device__ __forceinline__ void diag16_factor_invert(double* __restrict__ S, int b,
                                                     double* __restrict__ Wd,
                                                     int32_t* __restrict__ info, int32_t col0,
                                                     int lane) {
    const int r = lane & 15;
    double a[SB], rdk[SB];  // rdk[k] = 1 / L_kk (wave uniform), reused by the inversion
#pragma unroll
    for (int c = 0; c < SB; ++c) a[c] = S[(b + c) * LDS_LD + b + r];
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        double akk = readlane_f64(a[k], k);
        if (!(akk > 0.0)) {  // not positive definite (or NaN): record the first column, go on
            if (lane == 0) atomicCAS(info, 0, col0 + b + k + 1);
            akk = 1.0;
        }
        // this 16-step chain is the critical path of the whole factorisation: one rsqrt instead of
        // sqrt + divide, fused multiply-adds for the rank-1 update
        const double rd = rsqrt(akk);
        const double d = akk * rd;
        rdk[k] = rd;
        const double lk = (r == k) ? d : a[k] * rd;
        a[k] = lk;
#pragma unroll
        for (int c = k + 1; c < SB; ++c) {
            const double lck = readlane_f64(lk, c);
            a[c] = __builtin_fma(-lk, lck, a[c]);
        }
    }
    if (lane < SB) {
#pragma unroll
        for (int c = 0; c < SB; ++c)
            if (r >= c) S[(b + c) * LDS_LD + b + r] = a[c];
    }
    // inverse: lane cc computes column cc of W by forward substitution over the rows
    const int cc = r;
    double w[SB];
#pragma unroll
    for (int rr = 0; rr < SB; ++rr) {
        const double inv = rdk[rr];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) acc = __builtin_fma(readlane_f64(a[k], rr), w[k], acc);
        w[rr] = (rr == cc) ? inv : ((rr > cc) ? -(acc * inv) : 0.0);
    }
    if (lane < SB) {
#pragma unroll
        for (int rr = 0; rr < SB; ++rr) Wd[rr * WD_LD + cc] = w[rr];
    }
}

// Cholesky of one nb x nb (nb <= 128) diagonal block, resident in LDS, and the inverse of its
// factor -- blocked by 16 so that everything off the 16 x 16 diagonal sub-blocks is
// v_mfma_f64_16x16x4_f64 work with operands read straight from the LDS image:
//   for J = 0..7:  wave 0: L_JJ, W_JJ = L_JJ^-1 (registers + v_readlane)
//                  all waves: L_IJ = A_IJ W_JJ'  (I > J),   A_IK -= L_IJ L_KJ'  (I >= K > J)
//   then W = L^-1 block column by block column (columns are independent -> one per wave, no
//   barrier):  W_IJ = -W_II * sum_{K=J}^{I-1} L_IK W_KJ.  The f64 accumulator layout
//   (row = (lane>>4) + 4v) is exactly the B-operand layout (k = 4s + (lane>>4)), so the inner
//   sum feeds the second product without leaving the registers.
// W_IJ (I > J) is parked in the unused upper triangle at S[R*LDS_LD + C] (R > C global indices).
// info: the first failing column (1-based, LAPACK dpotrf convention) is recorded once; the block
// is then completed with a unit pivot so that the launch always terminates.
// Outputs: L_jj in place; Wcm[r + c*NB] = W(r,c) (column-major) and Wrm[c + r*NB] = W(r,c)
// (row-major), both zero padded to 128 x 128.
__global__ __launch_bounds__(256) void potf2_inv_kernel(double* __restrict__ A, int64_t lda, int nb,
                                                        double* __restrict__ Wcm,
                                                        double* __restrict__ Wrm,
                                                        int32_t* __restrict__ info, int32_t col0) {
    __shared__ double S[NB * LDS_LD];           // S[c*LDS_LD + r] = element (r, c)
    __shared__ double Wd[NSB * SB * WD_LD];     // inverse diagonal sub-blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int c = idx / NB, r = idx % NB;
        double v = 0.0;
        if (r < nb && c < nb) {
            if (r >= c) v = A[r + (int64_t)c * lda];
        } else if (r == c) {
            v = 1.0;  // identity padding keeps the padded block positive definite
        }
        S[c * LDS_LD + r] = v;
    }
    __syncthreads();
    for (int J = 0; J < NSB; ++J) {
        const int b = J * SB;
        double* WdJ = Wd + J * SB * WD_LD;
        if (wave == 0) diag16_factor_invert(S, b, WdJ, info, col0, lane);
        __syncthreads();
        // panel: L_IJ = A_IJ * W_JJ'
        for (int I = J + 1 + wave; I < NSB; I += 4) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double av = S[(b + 4 * s + hi) * LDS_LD + SB * I + lo];  // A_IJ[lo][4s+hi]
                const double bv = WdJ[lo * WD_LD + 4 * s + hi];                // W_JJ[lo][4s+hi]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) S[(b + lo) * LDS_LD + SB * I + hi + 4 * v] = acc[v];
        }
        __syncthreads();
        // trailing update: A_IK -= L_IJ * L_KJ'   (J < K <= I)
        int t = 0;
        for (int K = J + 1; K < NSB; ++K) {
            for (int I = K; I < NSB; ++I, ++t) {
                if ((t & 3) != wave) continue;
                double4_t acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = S[(SB * K + lo) * LDS_LD + SB * I + hi + 4 * v];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double av = -S[(b + 4 * s + hi) * LDS_LD + SB * I + lo];  // -L_IJ[lo][k]
                    const double bv = S[(b + 4 * s + hi) * LDS_LD + SB * K + lo];   //  L_KJ[lo][k]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) S[(SB * K + lo) * LDS_LD + SB * I + hi + 4 * v] = acc[v];
            }
        }
        __syncthreads();
    }
    // factor -> global (lower triangle only)
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int c = idx / nb, r = idx % nb;
        if (r >= c) A[r + (int64_t)c * lda] = S[c * LDS_LD + r];
    }
    // W = L^-1: block columns {0}, {1,6}, {2,5,7}, {3,4} on waves 0..3 (balanced MFMA counts)
    for (int q = 0; q < 3; ++q) {
        int J;
        if (q == 0)
            J = wave;
        else if (q == 1)
            J = (wave == 0) ? -1 : 7 - wave;
        else
            J = (wave == 2) ? 7 : -1;
        if (J < 0) continue;
        for (int I = J + 1; I < NSB; ++I) {
            double4_t T = {0.0, 0.0, 0.0, 0.0};
            for (int K = J; K < I; ++K) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = 4 * s + hi;
                    const double av = S[(SB * K + k) * LDS_LD + SB * I + lo];  // L_IK[lo][k]
                    const double bv = (K == J) ? Wd[(J * SB + k) * WD_LD + lo]  // W_JJ[k][lo]
                                               : S[(SB * K + k) * LDS_LD + SB * J + lo];  // W_KJ[k][lo]
                    T = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, T, 0, 0, 0);
                }
            }
            double4_t R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double av = Wd[(I * SB + lo) * WD_LD + 4 * s + hi];  // W_II[lo][4s+hi]
                R = __builtin_amdgcn_mfma_f64_16x16x4f64(av, T[s], R, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) S[(SB * I + hi + 4 * v) * LDS_LD + SB * J + lo] = -R[v];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        {  // column-major image: idx = c*NB + r
            const int c = idx / NB, r = idx % NB;
            double v = 0.0;
            if (r < nb && c < nb && r >= c)
                v = (r / SB == c / SB) ? Wd[((r / SB) * SB + r % SB) * WD_LD + c % SB] : S[r * LDS_LD + c];
            Wcm[idx] = v;
        }
        {  // row-major image: idx = r*NB + c
            const int r = idx / NB, c = idx % NB;
            double v = 0.0;
            if (r < nb && c < nb && r >= c)
                v = (r / SB == c / SB) ? Wd[((r / SB) * SB + r % SB) * WD_LD + c % SB] : S[r * LDS_LD + c];
            Wrm[idx] = v;
        }
    }
}

// ---- triangular sweeps: one launch per 128-column block ------------------------------------
// Both sweeps are "right-looking": as soon as the solution of block j is known it is applied to
// the part of the vector not yet solved, so every workgroup of the launch finishes its outputs
// alone (no cross-workgroup reduction).  Each workgroup recomputes the small diagonal product
//   x_j = W_jj v_j (forward, W column-major image)   /   x_j = W_jj' v_j (backward, row-major image)
// itself (128 x 128, the inverse block is shared through L2) instead of waiting for a separate
// launch; workgroup 0 stores x_j, workgroups g >= 1 update one 128-entry slice of the vector:
//   forward :  b[R] -= L[R, J] x_j      R = 128 rows below the block
//   backward:  y[C] -= L[J, C]' x_j     C = 128 columns left of the block
// The vector being updated and the vector receiving the solution are different arrays, so no
// workgroup reads what another one writes.  4 n^2 bytes of L per sweep: HBM bound.
__device__ __forceinline__ void block_matvec(const double* __restrict__ img, const double* __restrict__ v,
                                             int w, double* vs, double (*part)[NB], double* xs) {
    const int i = threadIdx.x & (NB - 1), p = threadIdx.x >> 7;
    if (threadIdx.x < NB) vs[i] = (i < w) ? v[i] : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = p * 16 + q;
        acc += img[i + j * NB] * vs[j];
    }
    part[p][i] = acc;
    __syncthreads();
    if (threadIdx.x < NB) {
        double sum = part[0][i];
#pragma unroll
        for (int q = 1; q < 8; ++q) sum += part[q][i];
        xs[i] = sum;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void trsv_fwd_step_kernel(const double* __restrict__ L, int64_t lda,
                                                             const double* __restrict__ Wcm,
                                                             double* __restrict__ b,
                                                             double* __restrict__ yout, int64_t jb,
                                                             int w, int64_t n) {
    __shared__ double vs[NB], xs[NB];
    __shared__ double part[8][NB];
    block_matvec(Wcm, b + jb, w, vs, part, xs);
    const int i = threadIdx.x & (NB - 1), p = threadIdx.x >> 7;
    if (blockIdx.x == 0) {
        if (threadIdx.x < w) yout[jb + threadIdx.x] = xs[threadIdx.x];
        return;
    }
    const int64_t row = jb + w + (int64_t)(blockIdx.x - 1) * NB + i;
    double acc = 0.0;
    if (row < n) {
        const double* Lp = L + row + jb * lda;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = p * 16 + q;
            if (c < w) acc += Lp[(int64_t)c * lda] * xs[c];
        }
    }
    part[p][i] = acc;
    __syncthreads();
    if (threadIdx.x < NB && row < n) {
        double sum = part[0][i];
#pragma unroll
        for (int q = 1; q < 8; ++q) sum += part[q][i];
        b[row] -= sum;
    }
}

__global__ __launch_bounds__(1024) void trsv_bwd_step_kernel(const double* __restrict__ L, int64_t lda,
                                                             const double* __restrict__ Wrm,
                                                             double* __restrict__ y,
                                                             double* __restrict__ xout, int64_t jb,
                                                             int w) {
    __shared__ double vs[NB], xs[NB];
    __shared__ double part[8][NB];
    block_matvec(Wrm, y + jb, w, vs, part, xs);
    if (blockIdx.x == 0) {
        if (threadIdx.x < w) xout[jb + threadIdx.x] = xs[threadIdx.x];
        return;
    }
    // 16 waves: wave v sums rows [64h, 64h+64) of the block for columns [16g, 16g+16), h = v&1, g = v>>1
    const int lane = threadIdx.x & 63, v = threadIdx.x >> 6;
    const int h = v & 1, g = v >> 1;
    const int r = 64 * h + lane;
    const int64_t c0 = (int64_t)(blockIdx.x - 1) * NB + 16 * g;  // columns left of the block (< jb)
    const double xr = (r < w) ? xs[r] : 0.0;
    const double* Lp = L + (jb + r) + c0 * lda;
    double sums[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) sums[q] = (r < w) ? Lp[(int64_t)q * lda] * xr : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        double t = sums[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0) part[h][16 * g + q] = t;
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        const int64_t c = (int64_t)(blockIdx.x - 1) * NB + threadIdx.x;
        y[c] -= part[0][threadIdx.x] + part[1][threadIdx.x];
    }
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

// Width (multiple of 128, 768..2048) of the next outer panel: the wide update GEMM runs
// (rows/128) x (W/128) tiles of equal cost on `slots` resident workgroups, so W is chosen to
// make the last round of tiles as full as possible (tail quantisation is the main loss of a
// left-looking factorisation; 8 fixed tile columns leave the last round 5-50 % full).
static int64_t outer_panel_width(int64_t rows, bool has_update, int64_t slots) {
    const int64_t mt = (rows + NB - 1) / NB;
    if (!has_update || mt <= 6) return NBO;
    int64_t best = NBO / NB;
    double best_eff = -1.0;
    for (int64_t wt = 6; wt <= 16 && wt <= mt; ++wt) {
        const int64_t tiles = mt * wt - wt * (wt - 1) / 2;
        const int64_t rounds = (tiles + slots - 1) / slots;
        const double eff = (double)tiles / (double)(rounds * slots);
        if (eff > best_eff + 1e-9 || (eff > best_eff - 0.01 && wt > best && eff > 0.97)) {
            best_eff = std::max(eff, best_eff);
            best = wt;
        }
    }
    return best * NB;
}

static int32_t panel_update(madqp_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t row0,
                            int64_t k0, int64_t width, int64_t kend = -1) {
    // C[row0:n, row0:row0+width] -= L[row0:n, k0:kend] * L[row0:row0+width, k0:kend]'   (kend = row0
    // unless given: the distributed factorisation applies one received panel at a time)
    if (kend < 0) kend = row0;
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
    g.K = kend - k0;
    // rows up to the padded order may be read when the leading dimension covers them (the KKT
    // object allocates K that way): no partial tiles, stores stay masked to M x N
    const int64_t npad = (n + NB - 1) / NB * NB;
    if (lda >= npad) {
        g.Mread = npad - row0;
        g.Nread = std::min<int64_t>(npad - row0, (width + NB - 1) / NB * NB);
    }
    g.diag_off = 0;
    g.lower_only = 1;
    return madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_GEMM);
}

// One 128-column block whose entries already carry every update from the columns to its left:
// factor the diagonal block (and invert it), then L[below, jb] = C[below, jb] * W_jj'.
static int32_t factor_block(madqp_chol* s, double* A, int64_t lda, int64_t jb, int64_t w) {
    madqp_ctx* ctx = s->ctx;
    const int64_t n = s->n;
    double* Wcm = s->winv + (jb / NB) * WBLK;
    double* Wrm = Wcm + NB * NB;
    {
        ProfScope ps(ctx, MADQP_PROF_POTRF_DIAG);
        hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, ctx->stream, A + jb + jb * lda, lda,
                           (int)w, Wcm, Wrm, s->d_info, (int32_t)jb);
        LAUNCH_CHECK(ctx);
    }
    if (jb + w < n) {
        // out[i,j] = sum_k C[i,k] W(j,k); in place: a single tile column, every workgroup reads
        // exactly the rows it writes and finishes reading (K = w, all stages) before its stores.
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
        const int64_t npad = (n + NB - 1) / NB * NB;
        if (lda >= npad) g.Mread = npad - jb - w;
        g.Nread = NB;  // the inverse block image is always 128 x 128, zero padded
        return madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_TRSM);
    }
    return MADQP_OK;
}

// Recursive left-looking factorisation of the columns [j0, j0+w), which already carry the updates
// of all columns < j0: factor the first half, apply it to the second half with ONE wide GEMM
// (N = K = w/2), recurse.  Compared with a flat loop over 128-column blocks (N = 128, K up to
// w - 128) this moves the in-panel flops into well-filled launches; only the leaves are narrow.
static int32_t factor_range(madqp_chol* s, double* A, int64_t lda, int64_t j0, int64_t w) {
    if (w <= NB) return factor_block(s, A, lda, j0, w);
    const int64_t h = ((w + NB - 1) / NB + 1) / 2 * NB;  // first half, in whole blocks
    int32_t r = factor_range(s, A, lda, j0, h);
    if (r) return r;
    if ((r = panel_update(s->ctx, A, lda, s->n, j0 + h, j0, w - h))) return r;
    return factor_range(s, A, lda, j0 + h, w - h);
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
    int64_t W = 0;
    for (int64_t J0 = 0; J0 < n; J0 += W) {
        W = std::min<int64_t>(outer_panel_width(n - J0, J0 > 0, ctx->gemm_slots), n - J0);
        if (J0 > 0) {
            int32_t r = panel_update(ctx, A, lda, n, J0, 0, W);
            if (r) return r;
        }
        int32_t r = factor_range(s, A, lda, J0, W);
        if (r) return r;
    }
    int32_t info = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&info, s->d_info, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *info_host = info;
    s->factored = (info == 0);
    return MADQP_OK;
}

// ---------------------------------------------------------------------------------------------
// Pieces of the factorisation for the multi-GPU path (madqp_jl_amd/dist.py): block
// columns ("panels", width a multiple of 128) are dealt round-robin to the ranks; a rank factors a
// panel it owns once every panel to its left has been applied to it, packs it, the host broadcasts
// the packed image (RCCL), the other ranks unpack it into their copy of L and apply it to the
// panels they own.  All calls are asynchronous on the context's stream; only factor_end reads back.
namespace {
__global__ void info_store_kernel(const int32_t* __restrict__ info, double* __restrict__ hdr) {
    hdr[0] = (double)*info;
    hdr[1] = 0.0;
}
__global__ void info_merge_kernel(int32_t* __restrict__ info, const double* __restrict__ hdr) {
    const int32_t in = (int32_t)hdr[0];
    if (in != 0) atomicCAS(info, 0, in);  // keep the first failing column, as on one GPU
}
constexpr int64_t PACK_HDR = 2;  // doubles: [info, 0] (keeps the payload 16-byte aligned)

bool panel_ok(const madqp_chol* s, int64_t j0, int64_t w) {
    return s->A && j0 >= 0 && w > 0 && j0 % NB == 0 && j0 + w <= s->n && (w % NB == 0 || j0 + w == s->n);
}
}  // namespace

extern "C" int32_t madqp_chol_factor_begin(madqp_chol* s, double* A, int64_t lda) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, A && lda >= s->n);
    s->factored = false;
    s->A = A;
    s->lda = lda;
    HIP_TRY(ctx, hipMemsetAsync(s->d_info, 0, sizeof(int32_t), ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_factor_panel(madqp_chol* s, int64_t j0, int64_t w) {
    if (!s) return MADQP_ERR_ARG;
    ARG_TRY(s->ctx, panel_ok(s, j0, w));
    return factor_range(s, s->A, s->lda, j0, w);
}

extern "C" int32_t madqp_chol_update_cols(madqp_chol* s, int64_t c0, int64_t cw, int64_t p0, int64_t pw) {
    if (!s) return MADQP_ERR_ARG;
    ARG_TRY(s->ctx, panel_ok(s, c0, cw) && panel_ok(s, p0, pw) && p0 + pw <= c0);
    return panel_update(s->ctx, s->A, s->lda, s->n, c0, p0, cw, p0 + pw);
}

// The same update for several panels of the caller in ONE launch: cols_host = ncols pairs
// (start, width), ascending; all of them right of the source panel [p0, p0+pw).
extern "C" int32_t madqp_chol_update_multi(madqp_chol* s, int64_t ncols, const int64_t* cols_host, int64_t p0,
                                           int64_t pw) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, ncols >= 0 && (ncols == 0 || cols_host) && panel_ok(s, p0, pw));
    if (ncols == 0) return MADQP_OK;
    const int64_t n = s->n, lda = s->lda, cmin = cols_host[0];
    std::vector<int64_t> rel((size_t)(2 * ncols));
    for (int64_t r = 0; r < ncols; ++r) {
        const int64_t c0 = cols_host[2 * r], cw = cols_host[2 * r + 1];
        ARG_TRY(ctx, panel_ok(s, c0, cw) && p0 + pw <= c0 && (r == 0 || c0 >= cols_host[2 * r - 2] + cols_host[2 * r - 1]));
        rel[(size_t)(2 * r)] = c0 - cmin;
        rel[(size_t)(2 * r + 1)] = c0 + cw - cmin;
    }
    GemmArgs g{};
    g.X = s->A + cmin + p0 * lda;
    g.ldx = lda;
    g.Y = g.X;
    g.ldy = lda;
    g.C = s->A + cmin + cmin * lda;
    g.ldc = lda;
    g.Cin = g.C;
    g.ldcin = lda;
    g.alpha = -1.0;
    g.beta = 1.0;
    g.M = n - cmin;
    g.N = n - cmin;
    g.K = pw;
    const int64_t npad = (n + NB - 1) / NB * NB;
    if (lda >= npad) g.Mread = g.Nread = npad - cmin;
    g.diag_off = 0;
    g.lower_only = 1;
    return madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_GEMM, rel.data(), ncols);
}

extern "C" int32_t madqp_chol_panel_doubles(madqp_chol* s, int64_t j0, int64_t w, int64_t* count_host) {
    if (!s || !count_host) return MADQP_ERR_ARG;
    ARG_TRY(s->ctx, j0 >= 0 && w > 0 && j0 % NB == 0 && j0 + w <= s->n);
    const int64_t nblk = (w + NB - 1) / NB;
    *count_host = PACK_HDR + nblk * WBLK + w * (s->n - j0);
    return MADQP_OK;
}

// buf = [info, 0 | inverse diagonal blocks of the panel | L[j0:n, j0+c] for c = 0..w-1]
extern "C" int32_t madqp_chol_panel_pack(madqp_chol* s, int64_t j0, int64_t w, double* buf) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, panel_ok(s, j0, w) && buf);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    const int64_t nblk = (w + NB - 1) / NB, rows = s->n - j0;
    hipLaunchKernelGGL(info_store_kernel, dim3(1), dim3(1), 0, ctx->stream, s->d_info, buf);
    LAUNCH_CHECK(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(buf + PACK_HDR, s->winv + (j0 / NB) * WBLK, nblk * WBLK * sizeof(double),
                                hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpy2DAsync(buf + PACK_HDR + nblk * WBLK, rows * sizeof(double),
                                  s->A + j0 + j0 * s->lda, s->lda * sizeof(double), rows * sizeof(double),
                                  (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_panel_unpack(madqp_chol* s, int64_t j0, int64_t w, const double* buf) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, panel_ok(s, j0, w) && buf);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    const int64_t nblk = (w + NB - 1) / NB, rows = s->n - j0;
    hipLaunchKernelGGL(info_merge_kernel, dim3(1), dim3(1), 0, ctx->stream, s->d_info, buf);
    LAUNCH_CHECK(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(s->winv + (j0 / NB) * WBLK, buf + PACK_HDR, nblk * WBLK * sizeof(double),
                                hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpy2DAsync(s->A + j0 + j0 * s->lda, s->lda * sizeof(double),
                                  buf + PACK_HDR + nblk * WBLK, rows * sizeof(double), rows * sizeof(double),
                                  (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_factor_end(madqp_chol* s, int32_t* info_host) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, s->A && info_host);
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
    // forward: L y = b   (b updated in place below the block, y collected in s->tmp)
    ProfScope ps(ctx, MADQP_PROF_TRSV);
    for (int64_t jb = 0; jb < n; jb += NB) {
        const int64_t w = std::min<int64_t>(NB, n - jb);
        const double* Wcm = s->winv + (jb / NB) * WBLK;
        const int64_t below = n - jb - w;
        hipLaunchKernelGGL(trsv_fwd_step_kernel, dim3((unsigned)(1 + (below + NB - 1) / NB)), dim3(1024), 0,
                           ctx->stream, A, lda, Wcm, rhs, s->tmp, jb, (int)w, n);
        LAUNCH_CHECK(ctx);
    }
    // backward: L' x = y  (y = s->tmp updated in place left of the block, x written to rhs)
    const int64_t last = ((n - 1) / NB) * NB;
    for (int64_t jb = last; jb >= 0; jb -= NB) {
        const int64_t w = std::min<int64_t>(NB, n - jb);
        const double* Wrm = s->winv + (jb / NB) * WBLK + NB * NB;
        hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3((unsigned)(1 + jb / NB)), dim3(1024), 0, ctx->stream,
                           A, lda, Wrm, s->tmp, rhs, jb, (int)w);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}
