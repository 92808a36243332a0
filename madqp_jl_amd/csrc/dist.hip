// madqp_dist_*: one dense KKT matrix over a P x Q grid of GPUs, one process per GPU (BASELINE
// configs[4]) -- the HIP / RCCL instantiation of dist_core.inc (schedule and layout are described there).
//
// Rank-local kernels: the MFMA GEMM core (gemm_f64.hip) for the panel solves and the trailing updates, the diagonal
// tile through the blocked Cholesky of chol.hip (factor_begin / factor_panel / panel_pack on an order-nb object), the
// tile solves through the sweep kernels of chol.hip, GEMVs of gemv.hip.  Collectives: RCCL (ncclBroadcast / ncclReduce
// / ncclAllReduce on the world communicator and on row / column communicators made with ncclCommSplit), resolved at
// run time from the RCCL the process already has (torch's) -- or, when the caller hands in `madqp_comm_ops`, its
// host-staged callbacks (multi-rank rehearsals on one GPU over gloo, where RCCL refuses two ranks per device).
// The collectives of step k+1 run on their own stream beside the trailing update of step k (dist_core.inc).
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "common.h"

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // optional: madqp_dist_comm_info
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;   // optional
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char* names[] = {getenv("MADQP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
    for (const char* nm : names)  // first the copy the process already has (torch's): one RCCL per process
        if (nm && !r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* nm : names)
        if (nm && !r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) return r;
#define RSYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name)
    RSYM(GetUniqueId, "ncclGetUniqueId");
    RSYM(CommInitRank, "ncclCommInitRank");
    RSYM(CommSplit, "ncclCommSplit");
    RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(CommCount, "ncclCommCount");
    RSYM(CommUserRank, "ncclCommUserRank");
    RSYM(Broadcast, "ncclBroadcast");
    RSYM(Reduce, "ncclReduce");
    RSYM(AllReduce, "ncclAllReduce");
    RSYM(Send, "ncclSend");
    RSYM(Recv, "ncclRecv");
    RSYM(GroupStart, "ncclGroupStart");
    RSYM(GroupEnd, "ncclGroupEnd");
    RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommSplit && r.CommDestroy && r.Broadcast && r.Reduce && r.AllReduce &&
           r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    return r;
}

constexpr int DIST_NEV = 8;
struct Dev {
    madqp_ctx* ctx;
    hipStream_t main, sP, sU;  // caller's stream, communication stream, kernel stream
    hipEvent_t ev[DIST_NEV], ev_main;
    int ev_next;
    int free_slots;  // workgroup slots the bulk updates leave to the collectives (0 = ordinary launches)
    madqp_chol* chol_nb;    // order nb: diagonal tiles
    madqp_chol* chol_last;  // order of the (partial) last tile, or nullptr
    madqp_chol* chol_full;  // order n (P == 1): whole panels through the one-GPU panel factorisation, or nullptr
    int64_t nb, wlast;
    int32_t* ctl;           // 4 ints: tile sweeps
    bool two_streams;
    int prof_cls;           // timer class of the masked GEMM (panel updates, or the assembly)
};

#define NCCL_TRY(dev, expr)                                                                                    \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return madqp_fail((dev)->ctx, MADQP_ERR_HIP, "%s failed: %s", #expr,                               \
                              rccl().GetErrorString ? rccl().GetErrorString(r_) : "rccl error");               \
    } while (0)

// ---- memory
void* dop_alloc(Dev* dev, size_t bytes) {
    void* ptr = nullptr;
    if (hipMalloc(&ptr, bytes) != hipSuccess) return nullptr;
    if (hipMemset(ptr, 0, bytes) != hipSuccess) {
        (void)hipFree(ptr);
        return nullptr;
    }
    return ptr;
}
void dop_free(Dev*, void* ptr) { (void)hipFree(ptr); }
// free device memory in bytes (-1: unknown); MADQP_DIST_MEM_FREE overrides the figure (tests of the refusal path)
int64_t dop_mem_free(Dev*) {
    if (const char* e = getenv("MADQP_DIST_MEM_FREE")) return atoll(e);
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return -1;
    return (int64_t)fr;
}
int32_t dop_sync(Dev* dev) {
    HIP_TRY(dev->ctx, hipStreamSynchronize(dev->ctx->stream));
    return 0;
}
int32_t dop_h2d(Dev* dev, void* dst, const void* src, size_t bytes) {
    HIP_TRY(dev->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, dev->ctx->stream));
    HIP_TRY(dev->ctx, hipStreamSynchronize(dev->ctx->stream));
    return 0;
}
int32_t dop_d2h(Dev* dev, void* dst, const void* src, size_t bytes) {
    HIP_TRY(dev->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, dev->ctx->stream));
    HIP_TRY(dev->ctx, hipStreamSynchronize(dev->ctx->stream));
    return 0;
}
int32_t dop_zero(Dev* dev, double* ptr, int64_t count) {
    if (count > 0) HIP_TRY(dev->ctx, hipMemsetAsync(ptr, 0, (size_t)count * sizeof(double), dev->ctx->stream));
    return 0;
}
int32_t dop_copy(Dev* dev, double* dst, const double* src, int64_t count) {
    if (count > 0)
        HIP_TRY(dev->ctx, hipMemcpyAsync(dst, src, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, dev->ctx->stream));
    return 0;
}
int32_t dop_copy2d(Dev* dev, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) {
    if (rows > 0 && cols > 0)
        HIP_TRY(dev->ctx, hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(double), src, (size_t)lds * sizeof(double),
                                           (size_t)rows * sizeof(double), (size_t)cols, hipMemcpyDeviceToDevice,
                                           dev->ctx->stream));
    return 0;
}

// ---- small kernels
__global__ void dist_info_store_kernel(const double* info, double* hdr) {
    hdr[0] = *info;
    hdr[1] = 0.0;
}
__global__ void dist_info_merge_kernel(double* info, const double* hdr) {
    if (*info == 0.0 && hdr[0] != 0.0) *info = hdr[0];
}
__global__ void dist_info_global_kernel(double* hdr, double col0, double* info) {
    if (hdr[0] != 0.0) hdr[0] += col0;  // tile-relative (1-based) -> global column
    if (*info == 0.0 && hdr[0] != 0.0) *info = hdr[0];
}
__global__ __launch_bounds__(256) void dist_vsub_kernel(double* a, const double* b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a[i] -= b[i];
}
// out[i] = loc[((I0 + i/nb) div R) nb + i mod nb] when tile I0 + i/nb belongs to residue r (mod R), else 0
__global__ __launch_bounds__(256) void dist_group_pack_kernel(double* __restrict__ out, const double* __restrict__ loc,
                                                              int64_t I0, int64_t wg, int64_t nb, int64_t R, int64_t r) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < wg; i += (int64_t)gridDim.x * 256) {
        const int64_t I = I0 + i / nb;
        out[i] = (I % R == r) ? loc[(I / R) * nb + i % nb] : 0.0;
    }
}
int32_t dop_group_pack(Dev* dev, double* out, const double* loc, int64_t I0, int64_t wg, int64_t nb, int64_t R, int64_t r) {
    if (wg <= 0) return 0;
    hipLaunchKernelGGL(dist_group_pack_kernel, dim3((unsigned)std::min<int64_t>((wg + 255) / 256, 1024)), dim3(256), 0,
                       dev->ctx->stream, out, loc, I0, wg, nb, R, r);
    LAUNCH_CHECK(dev->ctx);
    return 0;
}
int32_t dop_info_store(Dev* dev, const double* info, double* hdr) {
    hipLaunchKernelGGL(dist_info_store_kernel, dim3(1), dim3(1), 0, dev->ctx->stream, info, hdr);
    LAUNCH_CHECK(dev->ctx);
    return 0;
}
int32_t dop_info_merge(Dev* dev, double* info, const double* hdr) {
    hipLaunchKernelGGL(dist_info_merge_kernel, dim3(1), dim3(1), 0, dev->ctx->stream, info, hdr);
    LAUNCH_CHECK(dev->ctx);
    return 0;
}
int32_t dop_vsub(Dev* dev, double* a, const double* b, int64_t n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dist_vsub_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0,
                       dev->ctx->stream, a, b, n);
    LAUNCH_CHECK(dev->ctx);
    return 0;
}

// ---- rank-local linear algebra
// C = beta C + alpha X Y' on the 128-tiles (ti, tj) with ti >= row0[tj]
int32_t dop_gemm(Dev* dev, double* C, int64_t ldc, const double* Cin, int64_t ldcin, const double* X, int64_t ldx,
                 const double* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, double alpha, double beta, int64_t Mread,
                 int64_t Nread, const int64_t* row0) {
    GemmArgs g{};
    g.X = X;
    g.ldx = ldx;
    g.Y = Y;
    g.ldy = ldy;
    g.C = C;
    g.ldc = ldc;
    g.Cin = (beta != 0.0) ? Cin : nullptr;
    g.ldcin = ldcin;
    g.alpha = alpha;
    g.beta = beta;
    g.M = M;
    g.N = N;
    g.K = K;
    g.Mread = std::max(M, std::min(Mread, (M + 127) / 128 * 128));
    g.Nread = std::max(N, std::min(Nread, (N + 127) / 128 * 128));
    g.tile_row0 = row0;
    return madqp_gemm_tn(dev->ctx, g, dev->prof_cls);
}

// Cholesky of the w x w diagonal tile at T (in place); buf <- [info, 0 | inverse 128-blocks | L (w x w, ld w)];
// info: the tile's first failing column as a GLOBAL 1-based column (col0 = columns before the tile), merged into *info
int32_t dop_potrf_tile(Dev* dev, double* T, int64_t ld, int64_t w, double* buf, int64_t col0, double* info) {
    madqp_chol* s = (w == dev->nb) ? dev->chol_nb : dev->chol_last;
    if (!s || s->n != w) return madqp_fail(dev->ctx, MADQP_ERR_STATE, "no Cholesky object for a tile of order %lld", (long long)w);
    int32_t r;
    if ((r = madqp_chol_factor_begin(s, T, ld))) return r;
    if ((r = madqp_chol_factor_panel(s, 0, w))) return r;
    if ((r = madqp_chol_panel_pack(s, 0, w, buf))) return r;
    hipLaunchKernelGGL(dist_info_global_kernel, dim3(1), dim3(1), 0, dev->ctx->stream, buf, (double)col0, info);
    LAUNCH_CHECK(dev->ctx);
    return 0;
}

// P == 1: the panel of step k -- local column block lc, rows from k nb down -- is factored, solved and updated inside by
// the one-GPU panel factorisation; buf <- [info, 0 | inverse 128-blocks | L_kk (w x w, ld w)], info (LAPACK's, global
// column) merged into *info
__global__ void dist_info_from_int_kernel(const int32_t* chol_info, double* hdr, double* info) {
    hdr[0] = (double)*chol_info;
    hdr[1] = 0.0;
    if (*info == 0.0 && hdr[0] != 0.0) *info = hdr[0];
}
int32_t dop_panel_local(Dev* dev, double* K, int64_t ld, int64_t lc, int64_t k, int64_t nb, int64_t w, double* buf,
                        double* info) {
    madqp_chol* s = dev->chol_full;
    madqp_ctx* ctx = dev->ctx;
    if (!s) return madqp_fail(ctx, MADQP_ERR_STATE, "no order-n Cholesky object (P == 1 expected)");
    // the factorisation addresses a column by its GLOBAL index j0 = k nb: shift the base so that this is local block lc
    // (only the columns of the panel are touched)
    double* base = K + (lc - k) * nb * ld;
    int32_t r;
    if ((r = madqp_chol_factor_begin(s, base, ld))) return r;
    if ((r = madqp_chol_factor_panel(s, k * nb, w))) return r;
    const int64_t nblk = (w + 127) / 128;
    hipLaunchKernelGGL(dist_info_from_int_kernel, dim3(1), dim3(1), 0, ctx->stream, s->d_info, buf, info);
    LAUNCH_CHECK(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(buf + 2, s->winv + (k * nb / 128) * (2 * 128 * 128), (size_t)nblk * 2 * 128 * 128 * sizeof(double),
                                hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpy2DAsync(buf + 2 + nblk * 2 * 128 * 128, (size_t)w * sizeof(double), K + k * nb + lc * nb * ld,
                                  (size_t)ld * sizeof(double), (size_t)w * sizeof(double), (size_t)w, hipMemcpyDeviceToDevice,
                                  ctx->stream));
    return 0;
}

// X (rows x w, leading dimension ldx) <- X L^-T with the inverse 128-blocks W of L's diagonal (recursive, MFMA)
int32_t trsm_range(Dev* dev, double* X, int64_t ldx, int64_t rows, int64_t rread, const double* L, int64_t ldl,
                   const double* W, int64_t j0, int64_t wd) {
    if (wd == 128 && madqp_chol_panel_sub16_on())  // block substitution inside the 128-block (chol.hip: panel_sub16_kernel)
        return madqp_chol_panel_solve128(dev->ctx, X + j0 * ldx, ldx, rows, rread, L + j0 + j0 * ldl, ldl,
                                         W + (j0 / 128) * (2 * 128 * 128));
    if (wd <= 128) {  // (a short last block of a tile: the product with its inverse image, zero padded to 128 x 128)
        GemmArgs g{};
        g.X = X + j0 * ldx;
        g.ldx = ldx;
        g.Y = W + (j0 / 128) * (2 * 128 * 128);  // column-major image: Y[j + k*128] = W(j, k)
        g.ldy = 128;
        g.C = X + j0 * ldx;
        g.ldc = ldx;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.M = rows;
        g.N = wd;
        g.K = wd;
        g.Mread = rread;
        g.Nread = 128;
        return madqp_gemm_tn(dev->ctx, g, MADQP_PROF_POTRF_TRSM);
    }
    const int64_t h = ((wd + 127) / 128 + 1) / 2 * 128;
    int32_t r = trsm_range(dev, X, ldx, rows, rread, L, ldl, W, j0, h);
    if (r) return r;
    GemmArgs g{};
    g.X = X + j0 * ldx;
    g.ldx = ldx;
    g.Y = L + (j0 + h) + j0 * ldl;  // Y[j + k ldl] = L(j0 + h + j, j0 + k)
    g.ldy = ldl;
    g.C = X + (j0 + h) * ldx;
    g.ldc = ldx;
    g.Cin = g.C;
    g.ldcin = ldx;
    g.alpha = -1.0;
    g.beta = 1.0;
    g.M = rows;
    g.N = wd - h;
    g.K = h;
    g.Mread = rread;
    if ((r = madqp_gemm_tn(dev->ctx, g, MADQP_PROF_POTRF_GEMM))) return r;
    return trsm_range(dev, X, ldx, rows, rread, L, ldl, W, j0 + h, wd - h);
}
int32_t dop_trsm(Dev* dev, double* X, int64_t ldx, int64_t rows, int64_t rows_read, const double* L, int64_t ldl,
                 const double* W, int64_t w) {
    if (rows <= 0 || w <= 0) return 0;
    return trsm_range(dev, X, ldx, rows, std::max(rows, std::min(rows_read, (rows + 127) / 128 * 128)), L, ldl, W, 0, w);
}
int32_t dop_gemv(Dev* dev, int32_t trans, int64_t rows, int64_t cols, double alpha, const double* A, int64_t lda,
                 const double* x, double beta, double* y) {
    return madqp_gemv_impl(dev->ctx, trans, rows, cols, alpha, A, lda, x, beta, y, MADQP_PROF_TRSV);
}
// one rank (P = Q = 1): the local matrix IS the factor and chol_full holds the inverse blocks of every panel
// (dop_panel_local): both sweeps in the one-GPU form (chol.hip), one launch each
int32_t dop_solve_local(Dev* dev, double* rhs) {
    if (!dev->chol_full) return madqp_fail(dev->ctx, MADQP_ERR_STATE, "no order-n Cholesky object (1 x 1 grid expected)");
    return madqp_chol_solve(dev->chol_full, rhs);
}
int32_t dop_tile_solve(Dev* dev, int32_t trans, const double* L, int64_t ld, const double* W, double* v, int64_t w,
                       double* scratch) {
    return madqp_trsv_tile(dev->ctx, trans, L, ld, W, v, w, scratch, dev->ctl);
}

// ---- streams (dist_core.inc): kernels on sU, collectives on sP
int32_t dop_stream(Dev* dev, int which) {
    if (dev->two_streams) dev->ctx->stream = which ? dev->sP : dev->sU;
    return 0;
}
// everything queued on `from` so far happens before whatever is queued on `to` from now on
int32_t dop_link(Dev* dev, int from, int to) {
    if (!dev->two_streams) return 0;
    madqp_ctx* ctx = dev->ctx;
    hipEvent_t e = dev->ev[dev->ev_next];
    dev->ev_next = (dev->ev_next + 1) % DIST_NEV;  // a wait captures the record before it: slots may be re-recorded
    HIP_TRY(ctx, hipEventRecord(e, from ? dev->sP : dev->sU));
    HIP_TRY(ctx, hipStreamWaitEvent(to ? dev->sP : dev->sU, e, 0));
    return 0;
}
int32_t dop_fork(Dev* dev) {
    if (!dev->two_streams) return 0;
    madqp_ctx* ctx = dev->ctx;
    HIP_TRY(ctx, hipEventRecord(dev->ev_main, dev->main));
    HIP_TRY(ctx, hipStreamWaitEvent(dev->sP, dev->ev_main, 0));
    HIP_TRY(ctx, hipStreamWaitEvent(dev->sU, dev->ev_main, 0));
    ctx->stream = dev->sU;
    return 0;
}
// the caller's stream waits for both internal streams; the context is back on the caller's stream whatever happens
int32_t dop_join(Dev* dev) {
    if (!dev->two_streams) return 0;
    madqp_ctx* ctx = dev->ctx;
    ctx->stream = dev->main;
    hipError_t e1 = hipEventRecord(dev->ev[0], dev->sP);
    hipError_t e2 = hipEventRecord(dev->ev[1], dev->sU);
    if (e1 == hipSuccess) e1 = hipStreamWaitEvent(dev->main, dev->ev[0], 0);
    if (e2 == hipSuccess) e2 = hipStreamWaitEvent(dev->main, dev->ev[1], 0);
    if (e1 != hipSuccess || e2 != hipSuccess) {  // last resort: drain them on the host
        (void)hipStreamSynchronize(dev->sP);
        (void)hipStreamSynchronize(dev->sU);
    }
    HIP_TRY(ctx, e1);
    HIP_TRY(ctx, e2);
    return 0;
}
inline void dop_mark(Dev*, int, int64_t) {}  // schedule recorder of the CPU build (tests/csrc/dist_cpu.cpp)
// the bulk of a trailing update runs as a persistent launch of (resident slots - free_slots) workgroups, so that the
// kernels of the collectives travelling beside it (RCCL's, the packing kernels) find room on the chip at once
int32_t dop_bulk(Dev* dev, int on) {
    dev->ctx->gemm_cap_slots = on ? dev->free_slots : 0;
    return 0;
}

// dst tile t (at dst + t*dstep, leading dimension ldd) <- src tile t (src + t*sstep, lds): rows_t x w with
// rows_t = clamp(limit - t*lstep, 0, nb); rows rows_t .. nb-1 of the destination are zeroed when zero_pad
__global__ __launch_bounds__(256) void dist_gather_tiles_kernel(double* __restrict__ dst, int64_t ldd, int64_t dstep,
                                                                const double* __restrict__ src, int64_t lds,
                                                                int64_t sstep, int64_t nb, int64_t w, int64_t limit,
                                                                int64_t lstep, int zero_pad) {
    const int64_t t = blockIdx.z;
    int64_t rows = limit - t * lstep;
    rows = rows < 0 ? 0 : (rows > nb ? nb : rows);
    const int64_t rend = zero_pad ? nb : rows;
    const double* S = src + t * sstep;
    double* D = dst + t * dstep;
    for (int64_t c = blockIdx.y; c < w; c += gridDim.y)
        for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rend; r += (int64_t)gridDim.x * 256)
            D[r + c * ldd] = (r < rows) ? S[r + c * lds] : 0.0;
}
int32_t dop_gather_tiles(Dev* dev, double* dst, int64_t ldd, int64_t dstep, const double* src, int64_t lds, int64_t sstep,
                         int64_t count, int64_t nb, int64_t w, int64_t limit, int64_t lstep, int zero_pad) {
    if (count <= 0 || w <= 0) return 0;
    madqp_ctx* ctx = dev->ctx;
    ARG_TRY(ctx, count <= 65535);
    const dim3 grid((unsigned)((nb + 255) / 256), (unsigned)std::min<int64_t>(w, 256), (unsigned)count);
    hipLaunchKernelGGL(dist_gather_tiles_kernel, grid, dim3(256), 0, ctx->stream, dst, ldd, dstep, src, lds, sstep, nb, w,
                       limit, lstep, zero_pad);
    LAUNCH_CHECK(ctx);
    return 0;
}

// ---- RCCL collectives on the current stream
int32_t dop_nccl_bcast(Dev* dev, void* comm, double* buf, int64_t count, int root) {
    NCCL_TRY(dev, rccl().Broadcast(buf, buf, (size_t)count, ncclDouble, root, (ncclComm_t)comm, dev->ctx->stream));
    return 0;
}
int32_t dop_nccl_reduce(Dev* dev, void* comm, double* buf, int64_t count, int root) {
    NCCL_TRY(dev, rccl().Reduce(buf, buf, (size_t)count, ncclDouble, ncclSum, root, (ncclComm_t)comm, dev->ctx->stream));
    return 0;
}
int32_t dop_nccl_allreduce(Dev* dev, void* comm, double* buf, int64_t count) {
    NCCL_TRY(dev, rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm, dev->ctx->stream));
    return 0;
}
int32_t dop_nccl_send(Dev* dev, void* comm, const double* buf, int64_t count, int dst) {
    NCCL_TRY(dev, rccl().Send(buf, (size_t)count, ncclDouble, dst, (ncclComm_t)comm, dev->ctx->stream));
    return 0;
}
int32_t dop_nccl_recv(Dev* dev, void* comm, double* buf, int64_t count, int src) {
    NCCL_TRY(dev, rccl().Recv(buf, (size_t)count, ncclDouble, src, (ncclComm_t)comm, dev->ctx->stream));
    return 0;
}
int32_t dop_nccl_group(Dev* dev, int begin) {
    NCCL_TRY(dev, begin ? rccl().GroupStart() : rccl().GroupEnd());
    return 0;
}
}  // namespace

#include "dist_core.inc"

namespace {
void dev_destroy(Dev* dev) {
    if (!dev) return;
    if (dev->chol_nb) madqp_chol_destroy(dev->chol_nb);
    if (dev->chol_last) madqp_chol_destroy(dev->chol_last);
    if (dev->chol_full) madqp_chol_destroy(dev->chol_full);
    if (dev->ctl) (void)hipFree(dev->ctl);
    for (auto& e : dev->ev)
        if (e) (void)hipEventDestroy(e);
    if (dev->ev_main) (void)hipEventDestroy(dev->ev_main);
    if (dev->sP) (void)hipStreamDestroy(dev->sP);
    if (dev->sU) (void)hipStreamDestroy(dev->sU);
    delete dev;
}
}  // namespace

extern "C" int32_t madqp_dist_unique_id(madqp_ctx* ctx, void* id128) {
    ARG_TRY(ctx, ctx && id128);
    if (!rccl().ok) return madqp_fail(ctx, MADQP_ERR_STATE, "RCCL is not available in this process (librccl.so)");
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return madqp_fail(ctx, MADQP_ERR_HIP, "ncclGetUniqueId: %s", rccl().GetErrorString(r));
    memcpy(id128, &id, sizeof(id));
    return MADQP_OK;
}

extern "C" int32_t madqp_dist_destroy(madqp_dist* d) {
    if (!d) return MADQP_OK;
    Dev* dev = d->dev;
    (void)hipStreamSynchronize(dev->ctx->stream);
    if (dev->sP) (void)hipStreamSynchronize(dev->sP);
    if (dev->sU) (void)hipStreamSynchronize(dev->sU);
    for (int g = 2; g >= 0; --g)
        if (d->nccl[g]) (void)rccl().CommDestroy((ncclComm_t)d->nccl[g]);
    distcore::destroy(d);
    dev_destroy(dev);
    return MADQP_OK;
}

extern "C" int32_t madqp_dist_create(madqp_ctx* ctx, int32_t rank, int32_t world, int32_t P, int32_t Q, int64_t n,
                                     int64_t nb, const void* nccl_id128, const madqp_comm_ops* ops, madqp_dist** out) {
    ARG_TRY(ctx, ctx && out);
    *out = nullptr;
    ARG_TRY(ctx, P >= 1 && Q >= 1 && P * Q == world && rank >= 0 && rank < world && n >= 0 && nb >= 128 && nb % 128 == 0);
    ARG_TRY(ctx, world == 1 || ops || nccl_id128);
    ARG_TRY(ctx, !ops || (ops->bcast && ops->reduce_sum && ops->allreduce_sum && ops->send && ops->recv));
    Dev* dev = new (std::nothrow) Dev();
    if (!dev) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    memset(dev, 0, sizeof(*dev));
    dev->ctx = ctx;
    dev->main = ctx->stream;
    dev->nb = nb;
    dev->prof_cls = MADQP_PROF_POTRF_GEMM;
    const int64_t T = (n + nb - 1) / nb;
    dev->wlast = T ? n - (T - 1) * nb : 0;
    static const int two = getenv("MADQP_DIST_STREAMS") ? atoi(getenv("MADQP_DIST_STREAMS")) : 1;
    dev->two_streams = two != 0 && !ops;  // host-staged collectives synchronise anyway
    hipError_t e = hipMalloc(&dev->ctl, 4 * sizeof(int32_t));
    if (e == hipSuccess && dev->two_streams) {
        e = hipStreamCreateWithFlags(&dev->sP, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&dev->sU, hipStreamNonBlocking);
        for (int i = 0; i < DIST_NEV && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&dev->ev[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&dev->ev_main, hipEventDisableTiming);
    }
    // MADQP_DIST_FREE_SLOTS: of the 2 x #CU resident GEMM workgroups, how many the bulk updates leave free while
    // collectives travel beside them (default: 16 with several ranks over RCCL, 0 otherwise; cost on one GPU: DESIGN.md 7)
    dev->free_slots = getenv("MADQP_DIST_FREE_SLOTS") ? atoi(getenv("MADQP_DIST_FREE_SLOTS")) : ((world > 1 && !ops) ? 16 : 0);
    int32_t r = (e == hipSuccess) ? MADQP_OK : madqp_fail(ctx, MADQP_ERR_HIP, "madqp_dist_create: %s", hipGetErrorString(e));
    if (!r && n > 0) r = madqp_chol_create(ctx, std::min(nb, n), &dev->chol_nb);
    if (!r && dev->wlast && dev->wlast != std::min(nb, n)) r = madqp_chol_create(ctx, dev->wlast, &dev->chol_last);
    if (!r && P == 1 && n > 0) r = madqp_chol_create(ctx, n, &dev->chol_full);
    if (r) {
        dev_destroy(dev);
        return r;
    }
    if (n > 0 && n < nb) dev->nb = n;  // a single partial tile: chol_nb has order n
    // MADQP_DIST_FORCE_RCCL=1: build the communicators and run every collective through RCCL even with one rank (a
    // one-GPU box can then exercise the RCCL calls themselves; the id is drawn here when the caller gave none)
    static const int force = getenv("MADQP_DIST_FORCE_RCCL") ? atoi(getenv("MADQP_DIST_FORCE_RCCL")) : 0;
    const int32_t force_comm = (world == 1 && !ops && force) ? 1 : 0;
    madqp_dist* d = nullptr;
    r = distcore::create(dev, rank, world, P, Q, n, nb, ops, force_comm, &d);  // sizes and this rank's verdict; no allocation yet
    if (r) {
        if (d) madqp_fail(ctx, r, "madqp_dist_create: %s", d->err);
        distcore::destroy(d);
        dev_destroy(dev);
        return r == MADQP_ERR_ARG ? madqp_fail(ctx, r, "madqp_dist_create: bad grid / tile size") : r;
    }
    ncclUniqueId own_id;
    if (force_comm) {
        if (!rccl().ok || rccl().GetUniqueId(&own_id) != ncclSuccess) {
            madqp_dist_destroy(d);
            return madqp_fail(ctx, MADQP_ERR_STATE, "MADQP_DIST_FORCE_RCCL: RCCL is not available in this process");
        }
        nccl_id128 = &own_id;
    }
    if ((world > 1 || d->force_comm) && !ops) {  // RCCL: world communicator, then one per process row / column
        if (!rccl().ok) {
            madqp_dist_destroy(d);
            return madqp_fail(ctx, MADQP_ERR_STATE, "RCCL is not available in this process (librccl.so)");
        }
        ncclUniqueId id;
        memcpy(&id, nccl_id128, sizeof(id));
        ncclComm_t w = nullptr, row = nullptr, col = nullptr;
        ncclResult_t nr = rccl().CommInitRank(&w, world, id, rank);
        if (nr == ncclSuccess) {
            d->nccl[GRP_WORLD] = w;
            nr = rccl().CommSplit(w, d->p, d->q, &row, nullptr);  // same p: rank inside = q
        }
        if (nr == ncclSuccess) {
            d->nccl[GRP_ROW] = row;
            nr = rccl().CommSplit(w, d->q, d->p, &col, nullptr);  // same q: rank inside = p
        }
        if (nr == ncclSuccess) d->nccl[GRP_COL] = col;
        if (nr != ncclSuccess) {
            madqp_fail(ctx, MADQP_ERR_HIP, "RCCL communicator set-up failed: %s",
                       rccl().GetErrorString ? rccl().GetErrorString(nr) : "?");
            madqp_dist_destroy(d);
            return MADQP_ERR_HIP;
        }
    }
    // the buffers: a COLLECTIVE decision (every rank fits and every allocation succeeded, or every rank refuses with
    // MADQP_ERR_ALLOC) -- taken after the communicators exist, so that no rank waits in them for one that has left
    r = distcore::allocate(d);
    if (r) {
        madqp_fail(ctx, r, "madqp_dist_create: %s", d->err);
        madqp_dist_destroy(d);
        return r;
    }
    *out = d;
    return MADQP_OK;
}

extern "C" int32_t madqp_dist_layout(madqp_dist* d, int64_t* out8) {
    if (!d || !out8) return MADQP_ERR_ARG;
    const int64_t v[8] = {d->p, d->q, d->mt, d->nt, d->mloc, d->nloc, d->ld, d->ncp};
    memcpy(out8, v, sizeof(v));
    return MADQP_OK;
}

extern "C" int32_t madqp_dist_matrix(madqp_dist* d, double** K, int64_t* ld) {
    if (!d || !K || !ld) return MADQP_ERR_ARG;
    *K = d->K;
    *ld = d->ld;
    return MADQP_OK;
}

extern "C" int32_t madqp_dist_factor(madqp_dist* d, int32_t* info_host) {
    if (!d || !info_host) return MADQP_ERR_ARG;
    // on any error distcore::factor has already joined both internal streams into the caller's (dop_join)
    const int32_t r = distcore::factor(d, info_host);
    d->dev->ctx->gemm_cap_slots = 0;
    if (d->dev->two_streams) d->dev->ctx->stream = d->dev->main;
    return r;
}

extern "C" int32_t madqp_dist_solve(madqp_dist* d, double* rhs) {
    if (!d) return MADQP_ERR_ARG;
    ARG_TRY(d->dev->ctx, rhs || d->n == 0);
    if (!d->factored)  // a factorisation that ended in an error leaves no factor (info > 0 is not an error: unit pivots)
        return madqp_fail(d->dev->ctx, MADQP_ERR_STATE, "madqp_dist_solve: no completed factorisation");
    return distcore::solve(d, rhs);
}

// who carries the collectives, as the LIBRARY sees it: out8 = (backend: 0 = one rank, none needed; 1 = RCCL; 2 = the
// caller's madqp_comm_ops | ranks in the world communicator (RCCL: ncclCommCount) | ranks in my process-row communicator |
// ranks in my process-column communicator | my rank in the world communicator (ncclCommUserRank) | workgroup slots the
// bulk updates leave to the collectives' kernels | internal streams in use (0 / 2) | 0)
extern "C" int32_t madqp_dist_comm_info(madqp_dist* d, int64_t* out8) {
    if (!d || !out8) return MADQP_ERR_ARG;
    int64_t v[8] = {0, d->world, d->Q, d->P, d->rank, d->dev->free_slots, d->dev->two_streams ? 2 : 0, 0};
    if (d->host) {
        v[0] = 2;
    } else if (d->nccl[GRP_WORLD]) {
        v[0] = 1;
        for (int g = 0; g < 3; ++g) {
            int c = -1;
            if (d->nccl[g] && rccl().CommCount && rccl().CommCount((ncclComm_t)d->nccl[g], &c) == ncclSuccess) v[1 + g] = c;
        }
        int ur = -1;
        if (rccl().CommUserRank && rccl().CommUserRank((ncclComm_t)d->nccl[GRP_WORLD], &ur) == ncclSuccess) v[4] = ur;
    }
    memcpy(out8, v, sizeof(v));
    return MADQP_OK;
}

// device bytes this handle holds: out8 = (total | local matrix K | stored row operands XW | stored column operands YW |
// solve bands | broadcast images, staging and vectors | levels of the operand staircases | 0)
extern "C" int32_t madqp_dist_memory(madqp_dist* d, int64_t* out8) {
    if (!d || !out8) return MADQP_ERR_ARG;
    const int64_t v[8] = {d->bytes_total, d->bytes_K, 8 * d->xw_count, 8 * d->yw_count, d->bytes_band, d->bytes_stage,
                          d->nlev, 0};
    memcpy(out8, v, sizeof(v));
    return MADQP_OK;
}

extern "C" int32_t madqp_dist_bytes_sent(madqp_dist* d, int64_t* bytes_host) {
    if (!d || !bytes_host) return MADQP_ERR_ARG;
    *bytes_host = d->bytes_sent;
    return MADQP_OK;
}

// =====================================================================================================
// madqp_dkkt_*: the condensed KKT system K = H + Sigma_x + A' Theta A (kkt.hip) with K distributed over the grid.
//
// Same plugin methods as madqp_kkt_* (build_kkt! / factorize! / solve! / mul! / jtprod! and the model callbacks;
// src/KKT/normalkkt.jl:162-219, src/solver.jl:166-169,338-340), same algebra.  What a rank holds:
//   Hloc  its tiles of H, laid out like the local K (lower tiles; diagonal tiles complete, both triangles)
//   A_I   the columns of A that belong to its tile ROWS   (m16 x ld, row k of A contiguous, zero padded)
//   A_J   the columns of A that belong to its tile COLUMNS (m16 x ncp)
// -- 1/(PQ) of H and of K each and (1/P + 1/Q) of A.  (The factorisation itself keeps more: the operands of its lazy
// updates, this rank's tile rows and tile columns of L -- (1/P + 1/Q) n^2/2 doubles as staircases, dist_core.inc; the
// factor IS replicated Q-fold along process rows and P-fold along process columns.  madqp_dist_memory has the bytes.)
// Assembly needs no communication: K_loc = Hloc + A_I' (Theta A_J) is one masked MFMA GEMM over the local lower tiles.
// Iterates, right-hand sides and every scalar are replicated (n + m doubles); products with A, A' and H are summed over
// the ranks with ONE all-reduce each (every lower tile of H and every column block of A has exactly one owner, the
// diagonal-tile owner of that block), so all ranks see bitwise identical vectors and take identical branches.
#define TPB 256
#define MADQP_MAX_BLOCKS 1024
namespace {
#define MQ_KERNEL __global__ __launch_bounds__(TPB) void
#define MQ_BLOCK blockIdx.x
#define GRID_STRIDE(i, len) \
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < (len); i += (int64_t)gridDim.x * TPB)
inline int dk_grid(int64_t len) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((len + TPB - 1) / TPB, MADQP_MAX_BLOCKS));
}
#include "kkt_kernels.inc"

// S[k, :] = theta[k] * A[k, :] (root != 0: sqrt(theta[k])) for k < m, zero rows up to m16
__global__ __launch_bounds__(256) void dk_scale_rows_kernel(int64_t m, int64_t cols, const double* __restrict__ A, int64_t lda,
                                                            const double* __restrict__ theta, double* __restrict__ S,
                                                            int64_t lds, int root) {
    const int64_t k = blockIdx.y;
    const double t = (k < m) ? (root ? sqrt(theta[k]) : theta[k]) : 0.0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cols; c += (int64_t)gridDim.x * 256)
        S[k * lds + c] = (k < m) ? t * A[k * lda + c] : 0.0;
}
// K_loc[diag of local diagonal tile] += dvec[global index]
__global__ __launch_bounds__(256) void dk_add_diag_kernel(int64_t ntile, const int64_t* __restrict__ tiles, int64_t nb,
                                                          int64_t n, const double* __restrict__ dvec,
                                                          double* __restrict__ K, int64_t ld) {
    const int64_t t = blockIdx.y;
    if (t >= ntile) return;
    const int64_t li = tiles[3 * t], lj = tiles[3 * t + 1], I = tiles[3 * t + 2];
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nb; r += (int64_t)gridDim.x * 256) {
        const int64_t g = I * nb + r;
        if (g < n) K[(li * nb + r) + (lj * nb + r) * ld] += dvec[g];
    }
}
// local-row order <-> global order: out_loc[li*nb + r] = x[(li*R + res)*nb + r]
__global__ __launch_bounds__(256) void dk_gather_kernel(int64_t cnt, int64_t nb, int64_t R, int64_t res, int64_t n,
                                                        const double* __restrict__ x, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
        const int64_t g = ((i / nb) * R + res) * nb + i % nb;
        out[i] = (g < n) ? x[g] : 0.0;
    }
}
__global__ __launch_bounds__(256) void dk_scatter_add_kernel(int64_t cnt, int64_t nb, int64_t R, int64_t res, int64_t n,
                                                             const double* __restrict__ loc, double alpha,
                                                             double* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
        const int64_t g = ((i / nb) * R + res) * nb + i % nb;
        if (g < n) y[g] += alpha * loc[i];
    }
}
// w = alpha * g + beta * w  (beta == 0: w is not read)
__global__ __launch_bounds__(256) void dk_axpby_kernel(int64_t n, double alpha, const double* __restrict__ g, double beta,
                                                       double* __restrict__ w) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        w[i] = (beta == 0.0) ? alpha * g[i] : alpha * g[i] + beta * w[i];
}
}  // namespace

struct madqp_dkkt {
    madqp_dist* d;
    madqp_ctx* ctx;
    int64_t nx, m, ns, m16;
    const double* H;  // local tiles (ldh) or nullptr
    int64_t ldh;
    const double *AI, *AJ;
    int64_t ldai, ldaj;
    double* SJ;          // Theta * A_J (m16 x ncp), owned
    int64_t* d_ind_ineq; // ns
    int64_t* d_slot;     // m
    int64_t* d_diag;     // local diagonal tiles: (li, lj, I) triples
    int64_t ndiag;
    std::vector<int64_t> own_li;  // local tile rows whose A / H column block this rank owns for the mat-vecs
    double *theta, *t, *u;        // m
    double *gn, *gm;              // n + m contiguous: all-reduce buffer
    double *xr, *yr, *yc;         // local-order scratch: mloc, mloc, nloc
};

#define DKL(kern, len, ...)                                                                       \
    do {                                                                                          \
        hipLaunchKernelGGL(kern, dim3(dk_grid(len)), dim3(TPB), 0, ctx->stream, __VA_ARGS__);     \
        LAUNCH_CHECK(ctx);                                                                        \
    } while (0)

extern "C" int32_t madqp_dkkt_destroy(madqp_dkkt* k) {
    if (!k) return MADQP_OK;
    (void)hipStreamSynchronize(k->ctx->stream);
    void* ptrs[] = {k->SJ, k->d_ind_ineq, k->d_slot, k->d_diag, k->theta, k->t, k->u, k->gn, k->xr, k->yr, k->yc};
    for (void* ptr : ptrs)
        if (ptr) (void)hipFree(ptr);
    delete k;
    return MADQP_OK;
}

extern "C" int32_t madqp_dkkt_create(madqp_dist* d, int64_t nx, int64_t m, int64_t ns, const int64_t* ind_ineq_host,
                                     const double* Hloc, int64_t ldh, const double* A_I, int64_t ld_ai,
                                     const double* A_J, int64_t ld_aj, madqp_dkkt** out) {
    if (!d || !out) return MADQP_ERR_ARG;
    madqp_ctx* ctx = d->dev->ctx;
    *out = nullptr;
    ARG_TRY(ctx, nx == d->n && m >= 0 && ns >= 0 && ns <= m && (ns == 0 || ind_ineq_host));
    ARG_TRY(ctx, !Hloc || ldh >= d->ld);
    ARG_TRY(ctx, m == 0 || nx == 0 || (A_I && ld_ai >= d->ld && A_J && ld_aj >= d->ncp));
    std::vector<int64_t> slot((size_t)std::max<int64_t>(m, 1), -1);
    for (int64_t k = 0; k < ns; ++k) {
        const int64_t r = ind_ineq_host[k];
        ARG_TRY(ctx, r >= 0 && r < m && slot[r] < 0 && (k == 0 || ind_ineq_host[k - 1] < r));
        slot[r] = k;
    }
    madqp_dkkt* k = new (std::nothrow) madqp_dkkt();
    if (!k) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    k->d = d;
    k->ctx = ctx;
    k->nx = nx;
    k->m = m;
    k->ns = ns;
    k->m16 = (m + 15) / 16 * 16;
    k->H = Hloc;
    k->ldh = ldh;
    k->AI = A_I;
    k->ldai = ld_ai;
    k->AJ = A_J;
    k->ldaj = ld_aj;
    k->SJ = nullptr;
    k->d_ind_ineq = k->d_slot = k->d_diag = nullptr;
    k->theta = k->t = k->u = k->gn = k->gm = k->xr = k->yr = k->yc = nullptr;
    std::vector<int64_t> diag;
    for (int64_t li = 0; li < d->mt; ++li) {
        const int64_t I = li * d->P + d->p;
        if (I % d->Q == d->q) {
            diag.insert(diag.end(), {li, I / d->Q, I});
            k->own_li.push_back(li);
        }
    }
    k->ndiag = (int64_t)diag.size() / 3;
    const size_t mb = (size_t)std::max<int64_t>(m, 1) * sizeof(double);
    hipError_t e = hipMalloc(&k->SJ, (size_t)std::max<int64_t>(k->m16 * d->ncp, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&k->d_ind_ineq, (size_t)std::max<int64_t>(ns, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->d_slot, (size_t)std::max<int64_t>(m, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->d_diag, (size_t)std::max<int64_t>(3 * k->ndiag, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&k->theta, mb);
    if (e == hipSuccess) e = hipMalloc(&k->t, mb);
    if (e == hipSuccess) e = hipMalloc(&k->u, mb);
    if (e == hipSuccess) e = hipMalloc(&k->gn, (size_t)std::max<int64_t>(nx + m, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&k->xr, (size_t)d->ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&k->yr, (size_t)d->ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&k->yc, (size_t)d->ncp * sizeof(double));
    if (e == hipSuccess && ns) e = hipMemcpy(k->d_ind_ineq, ind_ineq_host, ns * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && m) e = hipMemcpy(k->d_slot, slot.data(), m * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && k->ndiag)
        e = hipMemcpy(k->d_diag, diag.data(), diag.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        madqp_dkkt_destroy(k);
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_dkkt_create: %s", hipGetErrorString(e));
    }
    k->gm = k->gn + nx;
    *out = k;
    return MADQP_OK;
}

static int32_t dk_check(madqp_dkkt* k, const madqp_state* st) {
    if (!k) return MADQP_ERR_ARG;
    ARG_TRY(k->ctx, st && st->n == k->nx + k->ns && st->m == k->m);
    return MADQP_OK;
}

// partial products of the blocks this rank owns, accumulated into the all-reduce buffers (which the caller zeroed):
//   gn += alpha A' u (u: m),   gm += alpha A x (x: global order),   gn += alpha H x
static int32_t dk_At_partial(madqp_dkkt* k, double alpha, const double* u) {
    madqp_dist* d = k->d;
    if (!k->m) return MADQP_OK;
    if (d->P == 1 && d->Q == 1)  // one rank owns every block, local order = global order: one product
        return madqp_gemv_impl(k->ctx, 1, k->m, d->n, alpha, k->AI, k->ldai, u, 1.0, k->gn, MADQP_PROF_GEMV);
    for (int64_t li : k->own_li) {
        const int64_t I = li * d->P + d->p, w = distcore::tsize(d, I);
        int32_t r = madqp_gemv_impl(k->ctx, 1, k->m, w, alpha, k->AI + li * d->nb, k->ldai, u, 1.0, k->gn + I * d->nb,
                                    MADQP_PROF_GEMV);
        if (r) return r;
    }
    return MADQP_OK;
}
static int32_t dk_A_partial(madqp_dkkt* k, double alpha, const double* x) {
    madqp_dist* d = k->d;
    if (!k->m) return MADQP_OK;
    if (d->P == 1 && d->Q == 1)
        return madqp_gemv_impl(k->ctx, 0, k->m, d->n, alpha, k->AI, k->ldai, x, 1.0, k->gm, MADQP_PROF_GEMV);
    for (int64_t li : k->own_li) {
        const int64_t I = li * d->P + d->p, w = distcore::tsize(d, I);
        int32_t r = madqp_gemv_impl(k->ctx, 0, k->m, w, alpha, k->AI + li * d->nb, k->ldai, x + I * d->nb, 1.0, k->gm,
                                    MADQP_PROF_GEMV);
        if (r) return r;
    }
    return MADQP_OK;
}
static int32_t dk_H_partial(madqp_dkkt* k, double alpha, const double* x) {
    madqp_dist* d = k->d;
    madqp_ctx* ctx = k->ctx;
    if (!k->H || d->mloc == 0 || d->nloc == 0) return MADQP_OK;
    const int64_t nb = d->nb;
    // one rank: local order = global order, x and the accumulator are whole vectors -- one pass over the triangle
    // (gemv.hip) instead of two over every tile column (8 ms per iteration at n = 50 000).  The contract of
    // madqp_dkkt_create hands over the tiles I >= J of column-major H (element (i, j) at j*ldh + i; the tiles above
    // the diagonal may hold anything -- the Julia glue's COO map never fills them): in the row-major view of the
    // mat-vec kernels that is the side c >= r, so it is madqp_symv_upper, which reads nothing else (the
    // first version called madqp_symv_lower here and read exactly the tiles a rank is NOT required to hold).
    if (d->world == 1 && !d->force_comm && d->n >= 2048 && (((uintptr_t)k->H) & 15) == 0 && k->ldh % 2 == 0)
        return madqp_symv_upper(ctx, d->n, alpha, k->H, k->ldh, x, 1.0, k->gn, MADQP_PROF_GEMV);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        DKL(dk_gather_kernel, d->mloc, d->mloc, nb, (int64_t)d->P, (int64_t)d->p, d->n, x, k->xr);
        HIP_TRY(ctx, hipMemsetAsync(k->yr, 0, (size_t)d->mloc * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(k->yc, 0, (size_t)d->nloc * sizeof(double), ctx->stream));
    }
    for (int64_t lj = 0; lj < d->nt; ++lj) {
        const int64_t J = lj * d->Q + d->q, w = distcore::tsize(d, J);
        const int64_t li0 = (J > d->p) ? (J - d->p + d->P - 1) / d->P : 0;  // first local tile row with I >= J
        if (li0 >= d->mt) continue;
        const double* Pn = k->H + li0 * nb + lj * nb * k->ldh;
        const int64_t rows = d->mloc - li0 * nb;
        int32_t r = madqp_gemv_impl(ctx, 1, w, rows, 1.0, Pn, k->ldh, x + J * nb, 1.0, k->yr + li0 * nb, MADQP_PROF_GEMV);
        if (r) return r;
        const bool has_diag = (li0 * d->P + d->p == J);  // the diagonal tile is the first one of the panel
        const int64_t skip = has_diag ? nb : 0;          // its transpose part is already in the product above
        if (rows - skip > 0) {
            r = madqp_gemv_impl(ctx, 0, w, rows - skip, 1.0, Pn + skip, k->ldh, k->xr + li0 * nb + skip, 1.0,
                                k->yc + lj * nb, MADQP_PROF_GEMV);
            if (r) return r;
        }
    }
    ProfScope ps(ctx, MADQP_PROF_VEC);
    DKL(dk_scatter_add_kernel, d->mloc, d->mloc, nb, (int64_t)d->P, (int64_t)d->p, d->n, k->yr, alpha, k->gn);
    DKL(dk_scatter_add_kernel, d->nloc, d->nloc, nb, (int64_t)d->Q, (int64_t)d->q, d->n, k->yc, alpha, k->gn);
    return MADQP_OK;
}
static int32_t dk_zero(madqp_dkkt* k, double* ptr, int64_t count) {
    if (count > 0) HIP_TRY(k->ctx, hipMemsetAsync(ptr, 0, (size_t)count * sizeof(double), k->ctx->stream));
    return MADQP_OK;
}

// MadNLP.build_kkt! (src/KKT/normalkkt.jl:166-180): Theta, then K_loc = H_loc + Sigma_x + A_I' (Theta A_J)
extern "C" int32_t madqp_dkkt_build(madqp_dkkt* k, const madqp_state* st) {
    int32_t r = dk_check(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    madqp_dist* d = k->d;
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (k->m) DKL(theta_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, st->du_diag, k->theta);
        if (k->m16 && d->nloc) {
            const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((d->ncp + 255) / 256, 64));
            // one rank: A_I and A_J are the same columns, so K = H + S'S with ONE operand stream S = sqrt(Theta) A
            // (what the one-GPU assembly does, gemm_f64.hip); several ranks: K_loc = H_loc + A_I' (Theta A_J)
            hipLaunchKernelGGL(dk_scale_rows_kernel, dim3(gx, (unsigned)k->m16), dim3(256), 0, ctx->stream, k->m, d->ncp,
                               k->AJ, k->ldaj, k->theta, k->SJ, d->ncp, (d->world == 1 && !d->force_comm) ? 1 : 0);
            LAUNCH_CHECK(ctx);
        }
    }
    if (k->m16 == 0 && d->mloc && d->nloc) {  // no constraints: K = H (+ diagonal)
        if (k->H)
            HIP_TRY(ctx, hipMemcpy2DAsync(d->K, (size_t)d->ld * sizeof(double), k->H, (size_t)k->ldh * sizeof(double),
                                          (size_t)d->mloc * sizeof(double), (size_t)d->nloc, hipMemcpyDeviceToDevice,
                                          ctx->stream));
        else
            HIP_TRY(ctx, hipMemset2DAsync(d->K, (size_t)d->ld * sizeof(double), 0, (size_t)d->mloc * sizeof(double),
                                          (size_t)d->nloc, ctx->stream));
    } else {
        d->dev->prof_cls = MADQP_PROF_SYRK;
        const bool one = (d->world == 1 && !d->force_comm);  // (ld == ncp on a 1 x 1 grid)
        r = distcore::assemble_product(d, k->H, k->ldh, one ? k->SJ : k->AI, one ? d->ncp : k->ldai, k->SJ, d->ncp, k->m16);
        d->dev->prof_cls = MADQP_PROF_POTRF_GEMM;
        if (r) return r;
    }
    if (k->ndiag) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(dk_add_diag_kernel, dim3((unsigned)std::min<int64_t>((d->nb + 255) / 256, 64), (unsigned)k->ndiag),
                           dim3(256), 0, ctx->stream, k->ndiag, k->d_diag, d->nb, d->n, st->pr_diag, d->K, d->ld);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_dkkt_factorize(madqp_dkkt* k, int32_t* info_host) {
    if (!k) return MADQP_ERR_ARG;
    return madqp_dist_factor(k->d, info_host);
}

// MadNLP.solve!(kkt, w) (src/KKT/normalkkt.jl:182-205) with the condensed algebra of kkt.hip
extern "C" int32_t madqp_dkkt_solve(madqp_dkkt* k, const madqp_state* st, double* w) {
    int32_t r = dk_check(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    madqp_dist* d = k->d;
    ARG_TRY(ctx, w != nullptr);
    double* wx = w;
    double* wy = w + st->n;
    if ((r = madqp_reduce_rhs(ctx, st, w))) return r;
    if (k->m) {
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            DKL(condense_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, k->theta, wx, wy, k->t, k->u);
        }
        if ((r = dk_zero(k, k->gn, k->nx)) || (r = dk_At_partial(k, 1.0, k->u))) return r;
        if ((r = distcore::comm_allreduce(d, k->gn, k->nx, GRP_WORLD))) return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        DKL(dk_axpby_kernel, k->nx, k->nx, 1.0, k->gn, 1.0, wx);  // rhs_x = r1_x + A' (theta t)
    }
    if ((r = madqp_dist_solve(d, wx))) return r;
    if (k->m) {
        if ((r = dk_zero(k, k->gm, k->m)) || (r = dk_A_partial(k, 1.0, wx))) return r;
        if ((r = distcore::comm_allreduce(d, k->gm, k->m, GRP_WORLD))) return r;
        ProfScope ps(ctx, MADQP_PROF_VEC);
        DKL(decondense_kernel, k->m, k->m, k->nx, k->d_slot, st->pr_diag, k->theta, k->t, k->gm, wx, wy);
    }
    return madqp_finish_aug_solve(ctx, st, w);
}

// MadNLP.jtprod!(out, kkt, y) (src/KKT/normalkkt.jl:162-164): out(n) = [A' y ; -y[ind_ineq]]
extern "C" int32_t madqp_dkkt_jtprod(madqp_dkkt* k, double* out, const double* y) {
    if (!k) return MADQP_ERR_ARG;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, (out || k->nx + k->ns == 0) && (y || k->m == 0));
    int32_t r;
    if ((r = dk_zero(k, k->gn, k->nx)) || (r = dk_At_partial(k, 1.0, y))) return r;
    if ((r = distcore::comm_allreduce(k->d, k->gn, k->nx, GRP_WORLD))) return r;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (k->nx) DKL(dk_axpby_kernel, k->nx, k->nx, 1.0, k->gn, 0.0, out);
    if (k->ns) DKL(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, y, out + k->nx, 1.0, 0.0);
    return MADQP_OK;
}

// MadNLP.mul!(w, kkt, v, alpha, beta) (src/KKT/normalkkt.jl:207-219) with H for a QP
extern "C" int32_t madqp_dkkt_mul(madqp_dkkt* k, const madqp_state* st, double* w, const double* v, double alpha,
                                  double beta) {
    int32_t r = dk_check(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, w && v);
    const int64_t nx = k->nx, n = st->n;
    // gn = A' vy + H vx, gm = A vx: one all-reduce of n_x + m doubles
    if ((r = dk_zero(k, k->gn, nx + k->m)) || (r = dk_At_partial(k, 1.0, v + n)) || (r = dk_H_partial(k, 1.0, v)) ||
        (r = dk_A_partial(k, 1.0, v)))
        return r;
    if ((r = distcore::comm_allreduce(k->d, k->gn, nx + k->m, GRP_WORLD))) return r;
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (nx) DKL(dk_axpby_kernel, nx, nx, alpha, k->gn, beta, w);
        if (k->ns) DKL(jt_slack_kernel, k->ns, k->ns, k->d_ind_ineq, v + n, w + nx, alpha, beta);
        if (k->m) DKL(mul_rows_kernel, k->m, k->m, k->d_slot, k->gm, v + nx, w + n, alpha, beta);
    }
    return madqp_kktmul(ctx, st, w, v, alpha, beta);
}

// model callbacks (src/solver.jl:166-169,338-340): f = [H x + q ; 0], c = A x - s - rhs, obj = c0 + q'x + x'Hx/2
extern "C" int32_t madqp_dkkt_eval(madqp_dkkt* k, const madqp_state* st, const double* q, const double* rhs, double c0,
                                   double* obj_host) {
    int32_t r = dk_check(k, st);
    if (r) return r;
    madqp_ctx* ctx = k->ctx;
    ARG_TRY(ctx, obj_host && (k->nx == 0 || q) && (k->m == 0 || rhs));
    const int64_t nx = k->nx, n = st->n;
    if ((r = dk_zero(k, k->gn, nx + k->m)) || (r = dk_H_partial(k, 1.0, st->x)) || (r = dk_A_partial(k, 1.0, st->x)))
        return r;
    if ((r = distcore::comm_allreduce(k->d, k->gn, nx + k->m, GRP_WORLD))) return r;
    double sums[2] = {0.0, 0.0};
    if (n) {
        const int nb = dk_grid(n);
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (k->H && nx) DKL(dk_axpby_kernel, nx, nx, 1.0, k->gn, 0.0, st->f);
        hipLaunchKernelGGL(eval_grad_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, n, nx, (k->H && nx) ? 1 : 0, q, st->x,
                           st->f, ctx->d_part);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(sum2_final_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part, nb, ctx->d_res);
        LAUNCH_CHECK(ctx);
    }
    if (k->m) {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        DKL(dk_axpby_kernel, k->m, k->m, 1.0, k->gm, 0.0, st->c);
        DKL(eval_cons_kernel, k->m, k->m, k->d_slot, st->x + nx, rhs, st->c);
    }
    if (n) {
        if ((r = madqp_read_results(ctx, 2, sums))) return r;
    } else if ((r = madqp_ctx_sync(ctx))) {
        return r;
    }
    *obj_host = c0 + sums[0] + 0.5 * sums[1];
    return MADQP_OK;
}
