// madqp_dist_*: one dense KKT matrix over a P x Q grid of GPUs, one process per GPU (BASELINE
// configs[4]) -- the HIP / RCCL instantiation of dist_core.inc (schedule and layout are described there).
//
// Rank-local kernels: the MFMA GEMM core (gemm_f64.hip) for the panel solves and the trailing updates, the diagonal
// tile through the blocked Cholesky of chol.hip (factor_begin / factor_panel / panel_pack on an order-nb object), the
// tile solves through the sweep kernels of chol.hip, GEMVs of gemv.hip.  Collectives: RCCL (ncclBroadcast / ncclReduce
// / ncclAllReduce on the world communicator and on row / column communicators made with ncclCommSplit), resolved at
// run time from the RCCL the process already has (torch's) -- or, when the caller hands in `madqp_comm_ops`, its
// host-staged callbacks (multi-rank rehearsals on one GPU over gloo, where RCCL refuses two ranks per device).
// The panel phase of step k+1 runs on its own stream beside the trailing update of step k.
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
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char* names[] = {getenv("MADQP_RCCL_LIB"), "librccl.so", "librccl.so.1"};
    for (const char* nm : names) {
        if (!nm) continue;
        r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // the copy the process already uses (torch's)
        if (!r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return r;
#define RSYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name)
    RSYM(GetUniqueId, "ncclGetUniqueId");
    RSYM(CommInitRank, "ncclCommInitRank");
    RSYM(CommSplit, "ncclCommSplit");
    RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(Broadcast, "ncclBroadcast");
    RSYM(Reduce, "ncclReduce");
    RSYM(AllReduce, "ncclAllReduce");
    RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommSplit && r.CommDestroy && r.Broadcast && r.Reduce && r.AllReduce;
    return r;
}

struct Dev {
    madqp_ctx* ctx;
    hipStream_t main, sP, sU;  // caller's stream, panel stream, update stream
    hipEvent_t ev[3];
    madqp_chol* chol_nb;    // order nb: diagonal tiles
    madqp_chol* chol_last;  // order of the (partial) last tile, or nullptr
    int64_t nb, wlast;
    int32_t* ctl;           // 4 ints: tile sweeps
    bool two_streams;
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
int32_t dop_gemm(Dev* dev, double* C, int64_t ldc, const double* X, int64_t ldx, const double* Y, int64_t ldy, int64_t M,
                 int64_t N, int64_t K, double alpha, double beta, int64_t Mread, int64_t Nread, const int64_t* row0) {
    GemmArgs g{};
    g.X = X;
    g.ldx = ldx;
    g.Y = Y;
    g.ldy = ldy;
    g.C = C;
    g.ldc = ldc;
    g.Cin = (beta != 0.0) ? C : nullptr;
    g.ldcin = ldc;
    g.alpha = alpha;
    g.beta = beta;
    g.M = M;
    g.N = N;
    g.K = K;
    g.Mread = std::max(M, std::min(Mread, (M + 127) / 128 * 128));
    g.Nread = std::max(N, std::min(Nread, (N + 127) / 128 * 128));
    g.tile_row0 = row0;
    return madqp_gemm_tn(dev->ctx, g, MADQP_PROF_POTRF_GEMM);
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

// X (rows x w, leading dimension ldx) <- X L^-T with the inverse 128-blocks W of L's diagonal (recursive, MFMA)
int32_t trsm_range(Dev* dev, double* X, int64_t ldx, int64_t rows, int64_t rread, const double* L, int64_t ldl,
                   const double* W, int64_t j0, int64_t wd) {
    if (wd <= 128) {
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
int32_t dop_tile_solve(Dev* dev, int32_t trans, const double* L, int64_t ld, const double* W, double* v, int64_t w,
                       double* scratch) {
    return madqp_trsv_tile(dev->ctx, trans, L, ld, W, v, w, scratch, dev->ctl);
}

// ---- streams: 0 begin (panel stream), 1 panel -> update, 2 update -> panel, 3 back to update, 4 join
int32_t dop_phase(Dev* dev, int code) {
    madqp_ctx* ctx = dev->ctx;
    if (!dev->two_streams) return 0;
    switch (code) {
        case 0:
            HIP_TRY(ctx, hipEventRecord(dev->ev[0], dev->main));
            HIP_TRY(ctx, hipStreamWaitEvent(dev->sP, dev->ev[0], 0));
            HIP_TRY(ctx, hipStreamWaitEvent(dev->sU, dev->ev[0], 0));
            ctx->stream = dev->sP;
            break;
        case 1:
            HIP_TRY(ctx, hipEventRecord(dev->ev[1], dev->sP));
            HIP_TRY(ctx, hipStreamWaitEvent(dev->sU, dev->ev[1], 0));
            ctx->stream = dev->sU;
            break;
        case 2:
            HIP_TRY(ctx, hipEventRecord(dev->ev[2], dev->sU));
            HIP_TRY(ctx, hipStreamWaitEvent(dev->sP, dev->ev[2], 0));
            ctx->stream = dev->sP;
            break;
        case 3:
            ctx->stream = dev->sU;
            break;
        default:
            HIP_TRY(ctx, hipEventRecord(dev->ev[1], dev->sP));
            HIP_TRY(ctx, hipEventRecord(dev->ev[2], dev->sU));
            HIP_TRY(ctx, hipStreamWaitEvent(dev->main, dev->ev[1], 0));
            HIP_TRY(ctx, hipStreamWaitEvent(dev->main, dev->ev[2], 0));
            ctx->stream = dev->main;
            break;
    }
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
}  // namespace

#include "dist_core.inc"

namespace {
void dev_destroy(Dev* dev) {
    if (!dev) return;
    if (dev->chol_nb) madqp_chol_destroy(dev->chol_nb);
    if (dev->chol_last) madqp_chol_destroy(dev->chol_last);
    if (dev->ctl) (void)hipFree(dev->ctl);
    for (auto& e : dev->ev)
        if (e) (void)hipEventDestroy(e);
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
    ARG_TRY(ctx, !ops || (ops->bcast && ops->reduce_sum && ops->allreduce_sum));
    Dev* dev = new (std::nothrow) Dev();
    if (!dev) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    memset(dev, 0, sizeof(*dev));
    dev->ctx = ctx;
    dev->main = ctx->stream;
    dev->nb = nb;
    const int64_t T = (n + nb - 1) / nb;
    dev->wlast = T ? n - (T - 1) * nb : 0;
    static const int two = getenv("MADQP_DIST_STREAMS") ? atoi(getenv("MADQP_DIST_STREAMS")) : 1;
    dev->two_streams = two != 0 && !ops;  // host-staged collectives synchronise anyway
    hipError_t e = hipMalloc(&dev->ctl, 4 * sizeof(int32_t));
    if (e == hipSuccess && dev->two_streams) {
        e = hipStreamCreateWithFlags(&dev->sP, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&dev->sU, hipStreamNonBlocking);
        for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&dev->ev[i], hipEventDisableTiming);
    }
    int32_t r = (e == hipSuccess) ? MADQP_OK : madqp_fail(ctx, MADQP_ERR_HIP, "madqp_dist_create: %s", hipGetErrorString(e));
    if (!r && n > 0) r = madqp_chol_create(ctx, std::min(nb, n), &dev->chol_nb);
    if (!r && dev->wlast && dev->wlast != std::min(nb, n)) r = madqp_chol_create(ctx, dev->wlast, &dev->chol_last);
    if (r) {
        dev_destroy(dev);
        return r;
    }
    if (n > 0 && n < nb) dev->nb = n;  // a single partial tile: chol_nb has order n
    madqp_dist* d = nullptr;
    r = distcore::create(dev, rank, world, P, Q, n, nb, ops, &d);
    if (r) {
        if (d) madqp_fail(ctx, r, "madqp_dist_create: %s", d->err);
        distcore::destroy(d);
        dev_destroy(dev);
        return r == MADQP_ERR_ARG ? madqp_fail(ctx, r, "madqp_dist_create: bad grid / tile size") : r;
    }
    if (world > 1 && !ops) {  // RCCL: world communicator, then one communicator per process row and per process column
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
    int32_t r = distcore::factor(d, info_host);
    if (r && d->dev->two_streams) d->dev->ctx->stream = d->dev->main;  // never leave the context on an internal stream
    return r;
}

extern "C" int32_t madqp_dist_solve(madqp_dist* d, double* rhs) {
    if (!d) return MADQP_ERR_ARG;
    ARG_TRY(d->dev->ctx, rhs || d->n == 0);
    return distcore::solve(d, rhs);
}

extern "C" int32_t madqp_dist_bytes_sent(madqp_dist* d, int64_t* bytes_host) {
    if (!d || !bytes_host) return MADQP_ERR_ARG;
    *bytes_host = d->bytes_sent;
    return MADQP_OK;
}
