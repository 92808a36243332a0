// HBM-bound dense mat-vec kernels (row r of the operand is contiguous: A[r*lda + c]).
//
// Replace mul!(y, kkt.AT, x) / mul!(y, kkt.AT', x) of the reference (src/KKT/normalkkt.jl:162-164,
// 194,200,214-215), the H x / A x products of the model callbacks (scripts/qp_gpu.jl:29-40) and the
// off-diagonal updates of the triangular sweeps (chol.hip).  Algorithmic traffic: 8*rows*cols bytes.
//
//   trans = 0 : y_r = alpha * sum_c A[r,c] x_c + beta * y_r   one wave (many rows) or one workgroup
//                                                              (few long rows) per row, 16-byte loads
//   trans = 1 : y_c = alpha * sum_r A[r,c] x_r + beta * y_c   workgroup = 128 columns x a row chunk,
//                                                              the 4 waves stride the rows; chunk
//                                                              partials are combined by a second
//                                                              kernel in a fixed order (deterministic)
#include <algorithm>
#include <cstdlib>

#include "common.h"

typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <bool VEC>
__device__ __forceinline__ double row_dot(const double* __restrict__ a, const double* __restrict__ x,
                                          int64_t cols, int t, int nt) {
    double acc0 = 0.0, acc1 = 0.0;
    if (VEC) {
        const int64_t pairs = cols >> 1;
#pragma unroll 4
        for (int64_t p = t; p < pairs; p += nt) {
            const double2_t av = *reinterpret_cast<const double2_t*>(a + 2 * p);
            const double2_t xv = *reinterpret_cast<const double2_t*>(x + 2 * p);
            acc0 = fma(av.x, xv.x, acc0);
            acc1 = fma(av.y, xv.y, acc1);
        }
        if ((cols & 1) && t == 0) acc0 = fma(a[cols - 1], x[cols - 1], acc0);
    } else {
#pragma unroll 4
        for (int64_t c = t; c < cols; c += nt) acc0 = fma(a[c], x[c], acc0);
    }
    return acc0 + acc1;
}

template <bool VEC>
__global__ __launch_bounds__(256) void gemv_n_wave_kernel(int64_t rows, int64_t cols, double alpha,
                                                          const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ x, double beta,
                                                          double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nwaves) {
        double s = wave_sum(row_dot<VEC>(A + r * lda, x, cols, lane, 64));
        if (lane == 0) y[r] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[r];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gemv_n_block_kernel(int64_t rows, int64_t cols, double alpha,
                                                           const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ x, double beta,
                                                           double* __restrict__ y) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        double s = wave_sum(row_dot<VEC>(A + r * lda, x, cols, threadIdx.x, 256));
        if (lane == 0) part[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double tot = (part[0] + part[1]) + (part[2] + part[3]);
            y[r] = (beta == 0.0) ? alpha * tot : alpha * tot + beta * y[r];
        }
        __syncthreads();
    }
}

// trans = 1.  grid = (column tiles of 128, row chunks).  FINAL: single chunk, write y directly.
template <bool VEC, bool FINAL>
__global__ __launch_bounds__(256) void gemv_t_kernel(int64_t rows, int64_t cols, double alpha,
                                                     const double* __restrict__ A, int64_t lda,
                                                     const double* __restrict__ x, double beta,
                                                     double* __restrict__ y,
                                                     double* __restrict__ partial,
                                                     int64_t rows_per_chunk) {
    __shared__ double red[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 128 + 2 * lane;
    const int64_t rb = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t re = min(rows, rb + rows_per_chunk);
    double a0 = 0.0, a1 = 0.0;
    if (VEC) {
        if (c0 + 1 < cols) {
#pragma unroll 8
            for (int64_t r = rb + w; r < re; r += 4) {
                const double2_t av = *reinterpret_cast<const double2_t*>(A + r * lda + c0);
                const double xr = x[r];
                a0 = fma(av.x, xr, a0);
                a1 = fma(av.y, xr, a1);
            }
        } else if (c0 < cols) {
            for (int64_t r = rb + w; r < re; r += 4) a0 = fma(A[r * lda + c0], x[r], a0);
        }
    } else {
        const bool ok0 = c0 < cols, ok1 = c0 + 1 < cols;
#pragma unroll 4
        for (int64_t r = rb + w; r < re; r += 4) {
            const double xr = x[r];
            if (ok0) a0 = fma(A[r * lda + c0], xr, a0);
            if (ok1) a1 = fma(A[r * lda + c0 + 1], xr, a1);
        }
    }
    red[w][2 * lane] = a0;
    red[w][2 * lane + 1] = a1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int64_t c = (int64_t)blockIdx.x * 128 + threadIdx.x;
        if (c < cols) {
            const double tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                               (red[2][threadIdx.x] + red[3][threadIdx.x]);
            if (FINAL)
                y[c] = (beta == 0.0) ? alpha * tot : alpha * tot + beta * y[c];
            else
                partial[(int64_t)blockIdx.y * cols + c] = tot;
        }
    }
}

// trans = 1 in ONE pass for matrices of a few thousand rows (the Jacobians of the mid-size problems: 2 000 x 5 000): a
// workgroup owns a strip of 16 columns -- one 128-byte line per row -- and walks ALL rows: a wave takes 8 rows per step
// (lane = row slot x column pair, one 16-byte load each, eight steps in flight), the eight row slots meet through
// shuffles, the four waves through LDS.  cols / 16 workgroups fill the chip without row chunks, so there are no partial
// sums to write and no second launch (22 + 9 us -> 17 us at 2 000 x 5 000).  Fixed summation order.
__global__ __launch_bounds__(256) void gemv_t_strip_kernel(int64_t rows, int64_t cols, double alpha,
                                                           const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ x, double beta,
                                                           double* __restrict__ y) {
    __shared__ double red[4][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cp = lane & 7, rs = lane >> 3;  // column pair of the strip, row slot of the wave's 8 rows
    const int64_t c0 = (int64_t)blockIdx.x * 16 + 2 * cp;
    const bool ok0 = c0 < cols, ok1 = c0 + 1 < cols;
    double a0 = 0.0, a1 = 0.0;
    if (ok1) {
#pragma unroll 8
        for (int64_t r = 8 * w + rs; r < rows; r += 32) {
            const double2_t av = *reinterpret_cast<const double2_t*>(A + r * lda + c0);
            const double xr = x[r];
            a0 = fma(av.x, xr, a0);
            a1 = fma(av.y, xr, a1);
        }
    } else if (ok0) {
        for (int64_t r = 8 * w + rs; r < rows; r += 32) a0 = fma(A[r * lda + c0], x[r], a0);
    }
#pragma unroll
    for (int d = 32; d >= 8; d >>= 1) {
        a0 += __shfl_down(a0, d);
        a1 += __shfl_down(a1, d);
    }
    if (rs == 0) {
        red[w][2 * cp] = a0;
        red[w][2 * cp + 1] = a1;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int64_t c = (int64_t)blockIdx.x * 16 + threadIdx.x;
        if (c < cols) {
            const double tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            y[c] = (beta == 0.0) ? alpha * tot : alpha * tot + beta * y[c];
        }
    }
}

__global__ __launch_bounds__(256) void gemv_t_reduce_kernel(int64_t cols, int nchunks, double alpha,
                                                            const double* __restrict__ partial,
                                                            double beta, double* __restrict__ y) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    double tot = 0.0;
    for (int k = 0; k < nchunks; ++k) tot += partial[(int64_t)k * cols + c];
    y[c] = (beta == 0.0) ? alpha * tot : alpha * tot + beta * y[c];
}

__global__ __launch_bounds__(256) void scale_kernel(int64_t n, double beta, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (beta == 0.0) ? 0.0 : beta * y[i];
}
// ---- y = alpha H x + beta y for a SYMMETRIC H from its lower triangle only ------------------------------------
// The Hessian products of the path (mul! with H for a QP, src/KKT/normalkkt.jl:207-219 + the H term; the model
// evaluation H x + q, scripts/qp_gpu.jl:29-40) streamed the full matrix: 20 GB at n_x = 50 000, 3.3 ms, four to six
// times per iteration.  Half of that is the same numbers again.  Here a workgroup owns a tile of SY_TR rows x SY_TC
// columns on or below the diagonal, reads it once, and produces BOTH of its contributions:
//   row part   p_r = sum_{c in tile, c <= r} H[r,c] x[c]   -> y[r]     (lane-local over 4 column groups, one wave sum per row)
//   col part   q_c = sum_{r in tile, r >  c} H[r,c] x[r]   -> y[c]     (per lane, the 4 waves combined through LDS)
// into partial[column tile][r] / partial2[row tile][c]; symv_reduce_kernel adds them in tile order: deterministic,
// no atomics, 8 n^2/2 bytes + 0.2 GB of partials.
constexpr int SY_TR = 256, SY_TC = 512;
// UPPER = false: entries H[r*ldh + c] with c <= r are read (the lower triangle of row-major H -- equally the upper
// triangle of a column-major matrix); UPPER = true: the entries with c >= r (the mirror image: row parts run over
// c >= r, column parts over r < c).  A caller that holds only ONE side of the diagonal picks the side it holds: the
// distributed KKT system stores the tiles I >= J of column-major H, element (i, j) at j*ldh + i, i.e. c >= r here.
template <bool UPPER>
__global__ __launch_bounds__(256) void symv_tri_kernel(int64_t n, const double* __restrict__ H, int64_t ldh,
                                                       const double* __restrict__ x, double* __restrict__ rowpart,
                                                       double* __restrict__ colpart) {
    __shared__ double red[4][SY_TC];
    const int64_t ct = blockIdx.x, rt = blockIdx.y;
    const int64_t r0 = rt * SY_TR, c0 = ct * SY_TC;
    if (!UPPER && r0 + SY_TR - 1 < c0) return;  // wholly above the diagonal
    if (UPPER && c0 + SY_TC - 1 < r0) return;   // wholly below it
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // strictly inside the triangle and the matrix: no masks
    const bool interior = (UPPER ? (c0 >= r0 + SY_TR) : (r0 >= c0 + SY_TC)) && (r0 + SY_TR <= n) && (c0 + SY_TC <= n);
    double xc[4][2], a[4][2];
    int64_t cc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        cc[g] = c0 + g * 128 + 2 * lane;
        xc[g][0] = (cc[g] < n) ? x[cc[g]] : 0.0;
        xc[g][1] = (cc[g] + 1 < n) ? x[cc[g] + 1] : 0.0;
        a[g][0] = a[g][1] = 0.0;
    }
    const int64_t rend = (r0 + SY_TR < n) ? r0 + SY_TR : n;
    for (int64_t r = r0 + w; r < rend; r += 4) {
        const double xr = x[r];
        const double* row = H + r * ldh;
        double t = 0.0;
        if (interior) {
            double2_t v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const double2_t*>(row + cc[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[g][0] = fma(v[g].x, xr, a[g][0]);
                a[g][1] = fma(v[g].y, xr, a[g][1]);
                t = fma(v[g].x, xc[g][0], t);
                t = fma(v[g].y, xc[g][1], t);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t c = cc[g];
                // the held side of row r, inside the matrix (lower: c <= r < n; upper: r <= c < n)
                const bool in0 = UPPER ? (c >= r && c < n) : (c <= r), in1 = UPPER ? (c + 1 >= r && c + 1 < n) : (c + 1 <= r);
                const double h0 = in0 ? row[c] : 0.0;
                const double h1 = in1 ? row[c + 1] : 0.0;
                if (in0 && c != r) a[g][0] = fma(h0, xr, a[g][0]);  // off the diagonal: the mirrored entry's product
                if (in1 && c + 1 != r) a[g][1] = fma(h1, xr, a[g][1]);
                t = fma(h0, xc[g][0], t);
                t = fma(h1, xc[g][1], t);
            }
        }
        t = wave_sum(t);
        if (lane == 0) rowpart[ct * n + r] = t;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        red[w][g * 128 + 2 * lane] = a[g][0];
        red[w][g * 128 + 2 * lane + 1] = a[g][1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SY_TC; i += 256) {
        const int64_t c = c0 + i;
        if (c < n) colpart[rt * n + c] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}
// y_i = alpha (row parts of the column tiles on the held side of i's row tile + column parts of the row tiles on the
// held side of i's column tile, each in tile order) + beta y_i
template <bool UPPER>
__global__ __launch_bounds__(256) void symv_reduce_kernel(int64_t n, double alpha, const double* __restrict__ rowpart,
                                                          const double* __restrict__ colpart, double beta,
                                                          double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t rt_i = i / SY_TR, ct_i = i / SY_TC, nrt = (n + SY_TR - 1) / SY_TR, nct = (n + SY_TC - 1) / SY_TC;
    double s = 0.0;
    if (!UPPER) {
        const int64_t ct_last = (rt_i * SY_TR + SY_TR - 1) / SY_TC;  // column tiles that row tile rt_i has a workgroup for
        for (int64_t ct = 0; ct <= ct_last; ++ct) s += rowpart[ct * n + i];
        for (int64_t rt = 2 * ct_i; rt < nrt; ++rt) s += colpart[rt * n + i];  // rt*256 + 255 >= ct_i*512  <=>  rt >= 2 ct_i
    } else {
        for (int64_t ct = (rt_i * SY_TR) / SY_TC; ct < nct; ++ct) s += rowpart[ct * n + i];  // ct*512 + 511 >= rt_i*256
        const int64_t rt_last = (ct_i * SY_TC + SY_TC - 1) / SY_TR;                           // rt*256 <= ct_i*512 + 511
        for (int64_t rt = 0; rt <= rt_last && rt < nrt; ++rt) s += colpart[rt * n + i];
    }
    y[i] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[i];
}
static_assert(SY_TC == 2 * SY_TR, "symv_reduce_kernel's first row tile of a column tile");

}  // namespace

// y(n) = alpha H x + beta y with H symmetric, n x n, row (= column) r at H + r*ldh: the lower triangle is read.
// Small or unaligned operands go through the general product (both triangles).
// true: madqp_symv_lower touches the lower triangle only (a caller that holds nothing else must ask)
bool madqp_symv_lower_reads_triangle(int64_t n, const double* H, int64_t ldh) {
    // below ~12 000 the 256 x 512 tiles are fewer than the chip's workgroup slots and the second pass is not paid back
    // (n = 5 000: 6.31 against 6.18 ms per iteration; n = 50 000: 1316 against 1323.5)
    static const int64_t nmin = getenv("MADQP_SYMV_MIN") ? atoll(getenv("MADQP_SYMV_MIN")) : 12288;
    const bool vec = (((uintptr_t)H) & 15) == 0 && (ldh % 2 == 0);
    return n >= nmin && vec;
}
static int32_t symv_tri(madqp_ctx* ctx, bool upper, int64_t n, double alpha, const double* H, int64_t ldh, const double* x,
                        double beta, double* y, int prof_cls) {
    const int64_t nrt = (n + SY_TR - 1) / SY_TR, nct = (n + SY_TC - 1) / SY_TC;
    int32_t r = madqp_work_reserve(ctx, (size_t)(nrt + nct) * n * sizeof(double));
    if (r) return r;
    double* rowpart = ctx->d_work;
    double* colpart = rowpart + nct * n;
    ProfScope ps(ctx, prof_cls);
    const dim3 grid((unsigned)nct, (unsigned)nrt), rgrid((unsigned)((n + 255) / 256));
    if (upper) {
        hipLaunchKernelGGL(symv_tri_kernel<true>, grid, dim3(256), 0, ctx->stream, n, H, ldh, x, rowpart, colpart);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(symv_reduce_kernel<true>, rgrid, dim3(256), 0, ctx->stream, n, alpha, rowpart, colpart, beta, y);
    } else {
        hipLaunchKernelGGL(symv_tri_kernel<false>, grid, dim3(256), 0, ctx->stream, n, H, ldh, x, rowpart, colpart);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(symv_reduce_kernel<false>, rgrid, dim3(256), 0, ctx->stream, n, alpha, rowpart, colpart, beta, y);
    }
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
int32_t madqp_symv_lower(madqp_ctx* ctx, int64_t n, double alpha, const double* H, int64_t ldh, const double* x,
                         double beta, double* y, int prof_cls) {
    ARG_TRY(ctx, ctx != nullptr && n >= 0);
    if (n == 0) return MADQP_OK;
    ARG_TRY(ctx, H && x && y && ldh >= n);
    if (!madqp_symv_lower_reads_triangle(n, H, ldh)) return madqp_gemv_impl(ctx, 0, n, n, alpha, H, ldh, x, beta, y, prof_cls);
    return symv_tri(ctx, false, n, alpha, H, ldh, x, beta, y, prof_cls);
}
// The same product from the OTHER side of the diagonal: only the entries H[r*ldh + c] with c >= r are read, at EVERY
// size (the caller holds nothing else: no full-matrix path).  This is the side a column-major matrix stored by its
// lower tiles holds -- element (i, j), i >= j, sits at j*ldh + i -- i.e. Hloc of madqp_dkkt_create on a 1 x 1 grid.
int32_t madqp_symv_upper(madqp_ctx* ctx, int64_t n, double alpha, const double* H, int64_t ldh, const double* x,
                         double beta, double* y, int prof_cls) {
    ARG_TRY(ctx, ctx != nullptr && n >= 0);
    if (n == 0) return MADQP_OK;
    ARG_TRY(ctx, H && x && y && ldh >= n && (((uintptr_t)H) & 15) == 0 && ldh % 2 == 0);
    return symv_tri(ctx, true, n, alpha, H, ldh, x, beta, y, prof_cls);
}

int32_t madqp_gemv_impl(madqp_ctx* ctx, int32_t trans, int64_t rows, int64_t cols, double alpha,
                        const double* A, int64_t lda, const double* x, double beta, double* y,
                        int prof_cls) {
    ARG_TRY(ctx, ctx != nullptr);
    ARG_TRY(ctx, rows >= 0 && cols >= 0 && (trans == 0 || trans == 1));
    const int64_t ylen = trans ? cols : rows, klen = trans ? rows : cols;
    if (ylen == 0) return MADQP_OK;
    ARG_TRY(ctx, y != nullptr);
    if (klen == 0) {  // y = beta*y
        ProfScope ps(ctx, prof_cls);
        hipLaunchKernelGGL(scale_kernel, dim3((ylen + 255) / 256), dim3(256), 0, ctx->stream, ylen,
                           beta, y);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    ARG_TRY(ctx, A && x && lda >= cols);
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (trans == 0) {
        const bool vec = al16(A) && al16(x) && (lda % 2 == 0);
        ProfScope ps(ctx, prof_cls);
        if (rows >= 2048 || cols <= 1024) {
            const int64_t blocks = std::min<int64_t>((rows + 3) / 4, 8192);
            if (vec)
                hipLaunchKernelGGL(gemv_n_wave_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream,
                                   rows, cols, alpha, A, lda, x, beta, y);
            else
                hipLaunchKernelGGL(gemv_n_wave_kernel<false>, dim3(blocks), dim3(256), 0,
                                   ctx->stream, rows, cols, alpha, A, lda, x, beta, y);
        } else {
            const int64_t blocks = std::min<int64_t>(rows, 4096);
            if (vec)
                hipLaunchKernelGGL(gemv_n_block_kernel<true>, dim3(blocks), dim3(256), 0,
                                   ctx->stream, rows, cols, alpha, A, lda, x, beta, y);
            else
                hipLaunchKernelGGL(gemv_n_block_kernel<false>, dim3(blocks), dim3(256), 0,
                                   ctx->stream, rows, cols, alpha, A, lda, x, beta, y);
        }
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    // trans == 1
    const bool vec = al16(A) && (lda % 2 == 0);
    static const bool strip_on = !(getenv("MADQP_GEMV_T_STRIP") && atoi(getenv("MADQP_GEMV_T_STRIP")) == 0);
    if (strip_on && vec && cols >= 2048 && rows >= 64 && rows <= 16384 && rows * cols <= ((int64_t)1 << 27)) {
        ProfScope ps(ctx, prof_cls);
        hipLaunchKernelGGL(gemv_t_strip_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, ctx->stream, rows, cols, alpha,
                           A, lda, x, beta, y);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    const int64_t ctiles = (cols + 127) / 128;
    int64_t nchunks = std::max<int64_t>(1, std::min<int64_t>((2048 + ctiles - 1) / ctiles, rows / 64));
    nchunks = std::min<int64_t>(nchunks, 64);
    int64_t rpc = (rows + nchunks - 1) / nchunks;
    rpc = (rpc + 3) / 4 * 4;
    nchunks = (rows + rpc - 1) / rpc;
    if (nchunks > 1) {
        int32_t r = madqp_work_reserve(ctx, (size_t)nchunks * cols * sizeof(double));
        if (r) return r;
    }
    ProfScope ps(ctx, prof_cls);
    dim3 grid((unsigned)ctiles, (unsigned)nchunks);
    if (nchunks == 1) {
        if (vec)
            hipLaunchKernelGGL((gemv_t_kernel<true, true>), grid, dim3(256), 0, ctx->stream, rows,
                               cols, alpha, A, lda, x, beta, y, (double*)nullptr, rpc);
        else
            hipLaunchKernelGGL((gemv_t_kernel<false, true>), grid, dim3(256), 0, ctx->stream, rows,
                               cols, alpha, A, lda, x, beta, y, (double*)nullptr, rpc);
        LAUNCH_CHECK(ctx);
    } else {
        if (vec)
            hipLaunchKernelGGL((gemv_t_kernel<true, false>), grid, dim3(256), 0, ctx->stream, rows,
                               cols, alpha, A, lda, x, beta, y, ctx->d_work, rpc);
        else
            hipLaunchKernelGGL((gemv_t_kernel<false, false>), grid, dim3(256), 0, ctx->stream, rows,
                               cols, alpha, A, lda, x, beta, y, ctx->d_work, rpc);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(gemv_t_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, ctx->stream,
                           cols, (int)nchunks, alpha, ctx->d_work, beta, y);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_gemv(madqp_ctx* ctx, int32_t trans, int64_t rows, int64_t cols,
                              double alpha, const double* A, int64_t lda, const double* x,
                              double beta, double* y) {
    if (!ctx) return MADQP_ERR_ARG;
    return madqp_gemv_impl(ctx, trans, rows, cols, alpha, A, lda, x, beta, y, MADQP_PROF_GEMV);
}
