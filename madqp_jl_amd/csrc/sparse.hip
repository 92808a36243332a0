// Sparse-A front end: the Jacobian arrives in CSR (the reference keeps it sparse:
// coo_to_csr, src/utils.jl:148-197; NormalKKTSystem constructor, src/KKT/normalkkt.jl:51-101) but the
// matrix that is factorised stays the DENSE condensed / normal matrix of this library, so the MFMA
// Cholesky and the triangular sweeps are unchanged.  Two pieces:
//   * y = alpha M x + beta y for a CSR matrix (products with A use the CSR of A, products with A' the
//     CSR of A' = the CSC of A, built once by the host: no atomics, fixed summation order);
//   * the weighted Gram matrix  C = base + diag(dvec) + V diag(w) V'  (lower triangle, dense, column
//     major) of the sparse rows of V -- assemble_normal_system! (src/utils.jl:266-298) with V = A,
//     w = 1/Sigma for the normal equations, and V = A', w = Theta for the condensed form.  One
//     workgroup per column of C (see sparse_gram_kernel): deterministic, no atomics.
#include <algorithm>

#include "common.h"

namespace {
__global__ __launch_bounds__(256) void spmv_csr_kernel(int64_t rows, const int64_t* __restrict__ rowptr,
                                                       const int64_t* __restrict__ col,
                                                       const double* __restrict__ val, double alpha,
                                                       const double* __restrict__ x, double beta,
                                                       double* __restrict__ y) {
    // a quarter wave (16 lanes) per row: short rows dominate in LP / QP Jacobians
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    double acc = 0.0;
    if (r < rows) {
        const int64_t e = rowptr[r + 1];
        for (int64_t p = rowptr[r] + sub; p < e; p += 16) acc += val[p] * x[col[p]];
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 16);
    if (r < rows && sub == 0) y[r] = (beta == 0.0) ? alpha * acc : alpha * acc + beta * y[r];
}

// Column j of the lower triangle of C = base + diag(dvec) + V diag(w) V' is owned by one workgroup:
//   C[j:n, j] <- base[j:n, j] (+ dvec[j] on the diagonal), one contiguous, coalesced run of the column-major
//   matrix; then for every stored entry (j, k) of row j of V, in ascending k, the column k of V (= row k of V',
//   entries (i, V[i,k]), i ascending) is scattered:  C[i, j] += w[k] V[j,k] V[i,k]  for i >= j.
// The entries of one column are distinct rows, so a pass has no write conflicts; the passes are separated
// by a barrier and ordered: every entry of C is a sum in a fixed order -- no atomics.  Work = nnz(V) x
// (average column length) multiply-adds plus one write of the triangle (n^2/2 x 8 B); the merge of two index
// lists per ENTRY of C, the obvious alternative, costs n^2/2 list merges and is 300 x slower at n = 90 000.
__global__ __launch_bounds__(256) void sparse_gram_kernel(int64_t n, const int64_t* __restrict__ rowptr,
                                                          const int64_t* __restrict__ col,
                                                          const double* __restrict__ val,
                                                          const int64_t* __restrict__ t_ptr,
                                                          const int64_t* __restrict__ t_col,
                                                          const double* __restrict__ t_val,
                                                          const double* __restrict__ w,
                                                          const double* __restrict__ base, int64_t ldbase,
                                                          const double* __restrict__ dvec,
                                                          double* __restrict__ C, int64_t ldc) {
    const int64_t j = blockIdx.x;
    double* Cj = C + j * ldc;
    const double* Bj = base ? base + j * ldbase : nullptr;
    for (int64_t i = j + threadIdx.x; i < n; i += 256) {
        double v = Bj ? Bj[i] : 0.0;
        if (dvec && i == j) v += dvec[j];
        Cj[i] = v;
    }
    __syncthreads();
    for (int64_t p = rowptr[j]; p < rowptr[j + 1]; ++p) {
        const int64_t k = col[p];
        const double f = w[k] * val[p];
        const int64_t e = t_ptr[k + 1];
        for (int64_t q = t_ptr[k] + threadIdx.x; q < e; q += 256) {
            const int64_t i = t_col[q];
            if (i >= j) Cj[i] += f * t_val[q];
        }
        __syncthreads();
    }
}
}  // namespace

int32_t madqp_spmv_csr(madqp_ctx* ctx, int64_t rows, const int64_t* rowptr, const int64_t* col, const double* val,
                       double alpha, const double* x, double beta, double* y, int prof_cls) {
    if (rows == 0) return MADQP_OK;
    ARG_TRY(ctx, rowptr && x && y);
    ProfScope ps(ctx, prof_cls);
    const unsigned grid = (unsigned)((rows * 16 + 255) / 256);
    hipLaunchKernelGGL(spmv_csr_kernel, dim3(grid), dim3(256), 0, ctx->stream, rows, rowptr, col, val, alpha, x,
                       beta, y);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

int32_t madqp_sparse_gram(madqp_ctx* ctx, int64_t n, const int64_t* rowptr, const int64_t* col, const double* val,
                          const int64_t* t_ptr, const int64_t* t_col, const double* t_val, const double* w,
                          const double* base, int64_t ldbase, const double* dvec, double* C, int64_t ldc) {
    if (n == 0) return MADQP_OK;
    ARG_TRY(ctx, rowptr && t_ptr && w && C && ldc >= n && (!base || ldbase >= n));
    ProfScope ps(ctx, MADQP_PROF_SYRK);
    hipLaunchKernelGGL(sparse_gram_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, n, rowptr, col, val, t_ptr,
                       t_col, t_val, w, base, ldbase, dvec, C, ldc);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
