// COO -> dense transfer of the callback buffers (Jacobian / Hessian values in the order of the model's sparsity
// pattern) into the dense operands the KKT object borrows.
//
// Replaces, for a dense target, compress_jacobian! of NormalKKTSystem (src/KKT/normalkkt.jl:149-158: the nnzj values
// that MadNLP.SparseCallback writes into `get_jacobian(kkt)` travel through A_csr_map into AT.nzval) and the
// scatter-add `transfer!` behind compress_hessian! (scripts/cuda_wrapper.jl:9-34, whose kernel adds duplicates
// non-atomically -- "do we need Atomix?").  Here the map is built once on the host: entries are grouped by destination
// (stable, so duplicates keep their COO order) and one lane owns one destination -- a fixed summation order, no
// atomics, bitwise reproducible.
#include <algorithm>
#include <numeric>

#include "common.h"

struct madqp_coo_map {
    madqp_ctx* ctx;
    int64_t nnz, ndest, nrows, ncols;
    int32_t symmetric;
    int64_t* d_perm;  // nnz: source positions grouped by destination
    int64_t* d_seg;   // ndest + 1
    int64_t* d_row;   // ndest (0-based)
    int64_t* d_col;   // ndest
};

namespace {
__global__ __launch_bounds__(256) void coo_apply_kernel(int64_t ndest, const int64_t* __restrict__ perm,
                                                        const int64_t* __restrict__ seg,
                                                        const int64_t* __restrict__ row,
                                                        const int64_t* __restrict__ col,
                                                        const double* __restrict__ vals, double* __restrict__ dst,
                                                        int64_t ld, int symmetric) {
    for (int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x; d < ndest; d += (int64_t)gridDim.x * 256) {
        double s = 0.0;
        for (int64_t k = seg[d]; k < seg[d + 1]; ++k) s += vals[perm[k]];
        const int64_t i = row[d], j = col[d];
        dst[i * ld + j] = s;
        if (symmetric && i != j) dst[j * ld + i] = s;
    }
}
}  // namespace

extern "C" int32_t madqp_coo_map_create(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host, const int32_t* J_host,
                                        int64_t nrows, int64_t ncols, int32_t symmetric, madqp_coo_map** out) {
    ARG_TRY(ctx, ctx && out && nnz >= 0 && nrows >= 0 && ncols >= 0 && (nnz == 0 || (I_host && J_host)));
    ARG_TRY(ctx, !symmetric || nrows == ncols);
    *out = nullptr;
    std::vector<int64_t> key((size_t)nnz), perm((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t i = (int64_t)I_host[k] - 1, j = (int64_t)J_host[k] - 1;  // MadNLP's patterns are 1-based
        if (i < 0 || i >= nrows || j < 0 || j >= ncols)
            return madqp_fail(ctx, MADQP_ERR_ARG, "madqp_coo_map_create: entry %lld = (%lld, %lld) outside %lld x %lld",
                              (long long)k, (long long)i + 1, (long long)j + 1, (long long)nrows, (long long)ncols);
        if (symmetric && i < j) std::swap(i, j);  // one owner per symmetric pair: the lower-triangle position
        key[(size_t)k] = i * ncols + j;
    }
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
    std::vector<int64_t> seg, row, col;
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t kk = key[(size_t)perm[(size_t)k]];
        if (k == 0 || kk != key[(size_t)perm[(size_t)k - 1]]) {
            seg.push_back(k);
            row.push_back(kk / ncols);
            col.push_back(kk % ncols);
        }
    }
    seg.push_back(nnz);
    madqp_coo_map* m = new (std::nothrow) madqp_coo_map();
    if (!m) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    m->ctx = ctx;
    m->nnz = nnz;
    m->ndest = (int64_t)row.size();
    m->nrows = nrows;
    m->ncols = ncols;
    m->symmetric = symmetric;
    m->d_perm = m->d_seg = m->d_row = m->d_col = nullptr;
    auto up = [&](int64_t** d, const std::vector<int64_t>& h) -> hipError_t {
        hipError_t e = hipMalloc(d, std::max<size_t>(1, h.size()) * sizeof(int64_t));
        if (e == hipSuccess && !h.empty())
            e = hipMemcpy(*d, h.data(), h.size() * sizeof(int64_t), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&m->d_perm, perm);
    if (e == hipSuccess) e = up(&m->d_seg, seg);
    if (e == hipSuccess) e = up(&m->d_row, row);
    if (e == hipSuccess) e = up(&m->d_col, col);
    if (e != hipSuccess) {
        madqp_coo_map_destroy(m);
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_coo_map_create: %s", hipGetErrorString(e));
    }
    *out = m;
    return MADQP_OK;
}

extern "C" int32_t madqp_coo_map_apply(madqp_coo_map* m, const double* vals, double* dst, int64_t ld) {
    if (!m) return MADQP_ERR_ARG;
    madqp_ctx* ctx = m->ctx;
    ARG_TRY(ctx, ld >= m->ncols && (dst || m->nrows * m->ncols == 0) && (vals || m->nnz == 0));
    if (m->nrows == 0 || m->ncols == 0) return MADQP_OK;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    HIP_TRY(ctx, hipMemset2DAsync(dst, (size_t)ld * sizeof(double), 0, (size_t)m->ncols * sizeof(double),
                                  (size_t)m->nrows, ctx->stream));
    if (m->ndest) {
        const unsigned grid = (unsigned)std::min<int64_t>((m->ndest + 255) / 256, 4096);
        hipLaunchKernelGGL(coo_apply_kernel, dim3(grid), dim3(256), 0, ctx->stream, m->ndest, m->d_perm, m->d_seg,
                           m->d_row, m->d_col, vals, dst, ld, (int)m->symmetric);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_coo_map_destroy(madqp_coo_map* m) {
    if (!m) return MADQP_OK;
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->d_perm) (void)hipFree(m->d_perm);
    if (m->d_seg) (void)hipFree(m->d_seg);
    if (m->d_row) (void)hipFree(m->d_row);
    if (m->d_col) (void)hipFree(m->d_col);
    delete m;
    return MADQP_OK;
}
