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
                                                        int64_t ld, int symmetric, int64_t nnz) {
    for (int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x; d < ndest; d += (int64_t)gridDim.x * 256) {
        double s = 0.0;
        for (int64_t k = seg[d]; k < seg[d + 1]; ++k) {
            const int64_t src = perm[k];  // (>= nnz: the mirrored copy of entry src - nnz, madqp_coo_map_create_tiles_cyclic)
            s += vals[src < nnz ? src : src - nnz];
        }
        const int64_t i = row[d], j = col[d];
        dst[i * ld + j] = s;
        if (symmetric && i != j) dst[j * ld + i] = s;
    }
}
}  // namespace

namespace {
// builds the map from destination keys (row * ncols + col, or -1 = entry dropped) of the nnz pattern entries
int32_t coo_map_from_keys(madqp_ctx* ctx, int64_t nnz, const std::vector<int64_t>& key, int64_t nrows, int64_t ncols,
                          int32_t symmetric, madqp_coo_map** out) {
    std::vector<int64_t> perm;
    perm.reserve((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k)
        if (key[(size_t)k] >= 0) perm.push_back(k);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
    std::vector<int64_t> seg, row, col;
    for (size_t k = 0; k < perm.size(); ++k) {
        const int64_t kk = key[(size_t)perm[k]];
        if (k == 0 || kk != key[(size_t)perm[k - 1]]) {
            seg.push_back((int64_t)k);
            row.push_back(kk / ncols);
            col.push_back(kk % ncols);
        }
    }
    seg.push_back((int64_t)perm.size());
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
// block-cyclic bookkeeping of dist_core.inc: tiles of width nb dealt to residues mod R
inline int64_t cyc_local_extent(int64_t n, int64_t nb, int64_t R, int64_t r) {
    const int64_t T = (n + nb - 1) / nb;
    if (T <= r) return 0;
    const int64_t cnt = (T - r + R - 1) / R, last = r + (cnt - 1) * R;
    return (cnt - 1) * nb + std::min(nb, n - last * nb);
}
inline int64_t cyc_local_index(int64_t g, int64_t nb, int64_t R) { return (g / nb / R) * nb + g % nb; }
}  // namespace

extern "C" int32_t madqp_coo_map_create(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host, const int32_t* J_host,
                                        int64_t nrows, int64_t ncols, int32_t symmetric, madqp_coo_map** out) {
    ARG_TRY(ctx, ctx && out && nnz >= 0 && nrows >= 0 && ncols >= 0 && (nnz == 0 || (I_host && J_host)));
    ARG_TRY(ctx, !symmetric || nrows == ncols);
    *out = nullptr;
    std::vector<int64_t> key((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t i = (int64_t)I_host[k] - 1, j = (int64_t)J_host[k] - 1;  // MadNLP's patterns are 1-based
        if (i < 0 || i >= nrows || j < 0 || j >= ncols)
            return madqp_fail(ctx, MADQP_ERR_ARG, "madqp_coo_map_create: entry %lld = (%lld, %lld) outside %lld x %lld",
                              (long long)k, (long long)i + 1, (long long)j + 1, (long long)nrows, (long long)ncols);
        if (symmetric && i < j) std::swap(i, j);  // one owner per symmetric pair: the lower-triangle position
        key[(size_t)k] = i * ncols + j;
    }
    return coo_map_from_keys(ctx, nnz, key, nrows, ncols, symmetric, out);
}

// The pieces of a Jacobian that one rank of a P x Q grid holds (madqp_dkkt_create: A_I with (R, r) = (P, p), A_J with
// (Q, q)): the entries whose COLUMN lies in a tile of residue r (mod R) go to dst[i*ld + local column], all others
// are dropped.  The map's target is nrows x (local columns).
extern "C" int32_t madqp_coo_map_create_cols_cyclic(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host,
                                                    const int32_t* J_host, int64_t nrows, int64_t ncols, int64_t nb,
                                                    int32_t R, int32_t r, madqp_coo_map** out) {
    ARG_TRY(ctx, ctx && out && nnz >= 0 && nrows >= 0 && ncols >= 0 && (nnz == 0 || (I_host && J_host)));
    ARG_TRY(ctx, nb >= 1 && R >= 1 && r >= 0 && r < R);
    *out = nullptr;
    const int64_t lcols = cyc_local_extent(ncols, nb, R, r);
    std::vector<int64_t> key((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t i = (int64_t)I_host[k] - 1, j = (int64_t)J_host[k] - 1;
        if (i < 0 || i >= nrows || j < 0 || j >= ncols)
            return madqp_fail(ctx, MADQP_ERR_ARG, "madqp_coo_map_create_cols_cyclic: entry %lld outside %lld x %lld",
                              (long long)k, (long long)nrows, (long long)ncols);
        key[(size_t)k] = ((j / nb) % R == r) ? i * std::max<int64_t>(lcols, 1) + cyc_local_index(j, nb, R) : -1;
    }
    return coo_map_from_keys(ctx, nnz, key, nrows, std::max<int64_t>(lcols, 1), 0, out);
}

// The tiles of a symmetric Hessian (pattern = one triangle, as MadNLP's) that rank (p, q) holds, in the layout of its
// local K (madqp_dkkt_create: Hloc): entry (i, j) of tile (I, J), I >= J, I = p (mod P), J = q (mod Q), goes to
// dst[jl*ld + il] (il, jl: local row / column); diagonal tiles receive both triangles.  Target: (local columns) x
// (local rows), i.e. the local matrix column by column.
extern "C" int32_t madqp_coo_map_create_tiles_cyclic(madqp_ctx* ctx, int64_t nnz, const int32_t* I_host,
                                                     const int32_t* J_host, int64_t n, int64_t nb, int32_t P, int32_t p,
                                                     int32_t Q, int32_t q, madqp_coo_map** out) {
    ARG_TRY(ctx, ctx && out && nnz >= 0 && n >= 0 && (nnz == 0 || (I_host && J_host)));
    ARG_TRY(ctx, nb >= 1 && P >= 1 && Q >= 1 && p >= 0 && p < P && q >= 0 && q < Q);
    *out = nullptr;
    const int64_t mloc = std::max<int64_t>(cyc_local_extent(n, nb, P, p), 1);
    const int64_t nloc = std::max<int64_t>(cyc_local_extent(n, nb, Q, q), 1);
    // an entry may have two destinations (both triangles of a diagonal tile): the pattern is doubled, second half =
    // the mirrored entries; `perm` then indexes a doubled value array -- so mirrored entries are appended as keys of
    // their own and mapped back to their source by the modulus in coo_apply (see the kernel)
    std::vector<int64_t> key((size_t)(2 * nnz), -1);
    auto dest = [&](int64_t i, int64_t j) -> int64_t {  // global (row i, column j), tile row >= tile column required
        const int64_t I = i / nb, J = j / nb;
        if (I < J || I % P != p || J % Q != q) return -1;
        return cyc_local_index(j, nb, Q) * mloc + cyc_local_index(i, nb, P);
    };
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t i = (int64_t)I_host[k] - 1, j = (int64_t)J_host[k] - 1;
        if (i < 0 || i >= n || j < 0 || j >= n)
            return madqp_fail(ctx, MADQP_ERR_ARG, "madqp_coo_map_create_tiles_cyclic: entry %lld outside order %lld",
                              (long long)k, (long long)n);
        const int64_t lo_i = std::max(i, j), lo_j = std::min(i, j);  // the lower-triangle position of the pair
        key[(size_t)k] = dest(lo_i, lo_j);
        if (lo_i != lo_j && lo_i / nb == lo_j / nb) key[(size_t)(nnz + k)] = dest(lo_j, lo_i);  // mirror inside a diagonal tile
    }
    int32_t rc = coo_map_from_keys(ctx, 2 * nnz, key, nloc, mloc, 0, out);
    if (rc == MADQP_OK) (*out)->nnz = nnz;  // sources are taken modulo nnz
    return rc;
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
                           m->d_row, m->d_col, vals, dst, ld, (int)m->symmetric, m->nnz);
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
