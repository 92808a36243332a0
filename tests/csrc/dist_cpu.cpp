// TEST INFRASTRUCTURE -- not part of the product, never loaded by madqp_jl_amd/.
//
// The distributed schedule of madqp_jl_amd/csrc/dist_core.inc (2-D block-cyclic Cholesky + sweeps) compiled with
// plain C++ loops in place of the rank-local HIP kernels and with the caller's host-staged collectives, so that the
// CPU suite (`-m "not gpu"`) can run the REAL orchestration with world_size 2..6 over gloo (tests/test_dist2d.py).
// "Device" memory is host memory here.  Built by tests/csrc/Makefile into tests/_build/libmadqp_dist_cpuref.so.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/madqp.h"

#include <vector>

// the schedule recorder: every dop_mark() of dist_core.inc and which "stream" it was issued on, in program order
struct Dev {
    int stream = 0;                 // 0 = kernel stream U, 1 = communication stream P (dop_stream)
    std::vector<int64_t> trace;     // triples (mark, k, stream) and (-1, from, to) for dop_link
};

static void* dop_alloc(Dev*, size_t bytes) { return calloc(1, bytes); }
static void dop_free(Dev*, void* p) { free(p); }
// free "device" memory: unknown (-1) -- unless the test names a figure (tests/test_dist2d.py: the refusal of a problem
// that does not fit must come from distcore::create's own sum, before anything is allocated)
static int64_t dop_mem_free(Dev*) {
    const char* e = getenv("MADQP_TEST_MEM_FREE");
    return e ? atoll(e) : -1;
}
static int32_t dop_sync(Dev*) { return 0; }
static int32_t dop_h2d(Dev*, void* dst, const void* src, size_t bytes) {
    memcpy(dst, src, bytes);
    return 0;
}
static int32_t dop_d2h(Dev*, void* dst, const void* src, size_t bytes) {
    memcpy(dst, src, bytes);
    return 0;
}
static int32_t dop_zero(Dev*, double* p, int64_t count) {
    if (count > 0) memset(p, 0, (size_t)count * sizeof(double));
    return 0;
}
static int32_t dop_copy(Dev*, double* dst, const double* src, int64_t count) {
    if (count > 0) memcpy(dst, src, (size_t)count * sizeof(double));
    return 0;
}
static int32_t dop_copy2d(Dev*, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) {
    for (int64_t c = 0; c < cols; ++c)
        if (rows > 0) memcpy(dst + c * ldd, src + c * lds, (size_t)rows * sizeof(double));
    return 0;
}
static int32_t dop_info_store(Dev*, const double* info, double* hdr) {
    hdr[0] = *info;
    hdr[1] = 0.0;
    return 0;
}
static int32_t dop_info_merge(Dev*, double* info, const double* hdr) {
    if (*info == 0.0 && hdr[0] != 0.0) *info = hdr[0];
    return 0;
}
static int32_t dop_vsub(Dev*, double* a, const double* b, int64_t n) {
    for (int64_t i = 0; i < n; ++i) a[i] -= b[i];
    return 0;
}
// C = beta C + alpha X Y' on the 128-tiles (ti, tj) with ti >= row0[tj]; the padded rows / columns that the HIP kernel
// may READ (Mread, Nread) must be inside the buffers: touch them so that a wrong extent shows up under a sanitizer
static int32_t dop_gemm(Dev*, double* C, int64_t ldc, const double* Cin, int64_t ldcin, const double* X, int64_t ldx,
                        const double* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, double alpha, double beta,
                        int64_t Mread, int64_t Nread, const int64_t* row0) {
    volatile double sink = 0.0;
    if (K > 0 && Mread > 0) sink = X[(Mread - 1) + (K - 1) * ldx];
    if (K > 0 && Nread > 0) sink = Y[(Nread - 1) + (K - 1) * ldy];
    (void)sink;
    for (int64_t j = 0; j < N; ++j) {
        const int64_t i0 = row0 ? row0[j / 128] * 128 : 0;
        for (int64_t i = i0; i < M; ++i) {
            double s = 0.0;
            for (int64_t k = 0; k < K; ++k) s += X[i + k * ldx] * Y[j + k * ldy];
            C[i + j * ldc] = (beta != 0.0 && Cin ? beta * Cin[i + j * ldcin] : 0.0) + alpha * s;
        }
    }
    return 0;
}
static int32_t dop_potrf_tile(Dev*, double* T, int64_t ld, int64_t w, double* buf, int64_t col0, double* info) {
    double first = 0.0;
    for (int64_t j = 0; j < w; ++j) {
        double d = T[j + j * ld];
        for (int64_t k = 0; k < j; ++k) d -= T[j + k * ld] * T[j + k * ld];
        if (!(d > 0.0)) {  // as the device kernel: record the first failing column, go on with a unit pivot
            if (first == 0.0) first = (double)(col0 + j + 1);
            d = 1.0;
        }
        const double l = std::sqrt(d);
        T[j + j * ld] = l;
        for (int64_t i = j + 1; i < w; ++i) {
            double v = T[i + j * ld];
            for (int64_t k = 0; k < j; ++k) v -= T[i + k * ld] * T[j + k * ld];
            T[i + j * ld] = v / l;
        }
    }
    const int64_t nblk = (w + 127) / 128;
    buf[0] = first;
    buf[1] = 0.0;
    memset(buf + 2, 0, (size_t)(nblk * 2 * 128 * 128) * sizeof(double));  // inverse blocks: not used by these loops
    double* L = buf + 2 + nblk * 2 * 128 * 128;
    for (int64_t j = 0; j < w; ++j)
        for (int64_t i = 0; i < w; ++i) L[i + j * w] = T[i + j * ld];
    if (*info == 0.0 && first != 0.0) *info = first;
    return 0;
}
static int32_t dop_trsm(Dev*, double* X, int64_t ldx, int64_t rows, int64_t, const double* L, int64_t ldl, const double*,
                        int64_t w);
// P == 1: tile factorisation + the solve of the rows below, as one step (the HIP build runs chol.hip's panel code here)
static int32_t dop_panel_local(Dev* dev, double* K, int64_t ld, int64_t lc, int64_t k, int64_t nb, int64_t w, double* buf,
                               double* info) {
    double* T = K + k * nb + lc * nb * ld;
    int32_t r = dop_potrf_tile(dev, T, ld, w, buf, k * nb, info);
    if (r) return r;
    // rows below the tile: everything this rank holds (P == 1: all rows of the matrix) -- the caller's local row count
    // is not known here, so the extent comes from the leading dimension's owner: dist_core passes ld >= rows; the
    // padded rows are zero and stay zero
    const int64_t nblk = (w + 127) / 128;
    const double* L = buf + 2 + nblk * 2 * 128 * 128;
    const int64_t below = ld - (k * nb + w);
    return below > 0 ? dop_trsm(dev, T + w, ld, below, below, L, w, nullptr, w) : 0;
}
static int32_t dop_trsm(Dev*, double* X, int64_t ldx, int64_t rows, int64_t, const double* L, int64_t ldl, const double*,
                        int64_t w) {
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < w; ++j) {
            double v = X[i + j * ldx];
            for (int64_t k = 0; k < j; ++k) v -= X[i + k * ldx] * L[j + k * ldl];
            X[i + j * ldx] = v / L[j + j * ldl];
        }
    return 0;
}
// madqp_gemv semantics: A has `rows` rows of length `cols`, row r at A + r*lda
static int32_t dop_gemv(Dev*, int32_t trans, int64_t rows, int64_t cols, double alpha, const double* A, int64_t lda,
                        const double* x, double beta, double* y) {
    if (!trans) {
        for (int64_t r = 0; r < rows; ++r) {
            double s = 0.0;
            for (int64_t c = 0; c < cols; ++c) s += A[r * lda + c] * x[c];
            y[r] = (beta != 0.0 ? beta * y[r] : 0.0) + alpha * s;
        }
    } else {
        for (int64_t c = 0; c < cols; ++c) {
            double s = 0.0;
            for (int64_t r = 0; r < rows; ++r) s += A[r * lda + c] * x[r];
            y[c] = (beta != 0.0 ? beta * y[c] : 0.0) + alpha * s;
        }
    }
    return 0;
}
static int32_t dop_tile_solve(Dev*, int32_t trans, const double* L, int64_t ld, const double*, double* v, int64_t w,
                              double*);
static madqp_dist* g_solve_local_of = nullptr;  // (the CPU stand-in needs the matrix: set by madqp_distcpu_solve)
static int32_t dop_solve_local(Dev* dev, double* rhs);
static int32_t dop_tile_solve(Dev*, int32_t trans, const double* L, int64_t ld, const double*, double* v, int64_t w,
                              double*) {
    if (!trans) {
        for (int64_t i = 0; i < w; ++i) {
            double s = v[i];
            for (int64_t k = 0; k < i; ++k) s -= L[i + k * ld] * v[k];
            v[i] = s / L[i + i * ld];
        }
    } else {
        for (int64_t i = w - 1; i >= 0; --i) {
            double s = v[i];
            for (int64_t k = i + 1; k < w; ++k) s -= L[k + i * ld] * v[k];
            v[i] = s / L[i + i * ld];
        }
    }
    return 0;
}
static int32_t dop_stream(Dev* dev, int which) {
    dev->stream = which;
    return 0;
}
static int32_t dop_link(Dev* dev, int from, int to) {
    dev->trace.insert(dev->trace.end(), {-1, from, to});
    return 0;
}
static int32_t dop_fork(Dev* dev) {
    dev->stream = 0;
    dev->trace.clear();
    return 0;
}
static int32_t dop_join(Dev* dev) {
    dev->stream = 0;
    return 0;
}
static void dop_mark(Dev* dev, int code, int64_t k) { dev->trace.insert(dev->trace.end(), {code, k, dev->stream}); }
static int32_t dop_bulk(Dev*, int) { return 0; }
static int32_t dop_gather_tiles(Dev*, double* dst, int64_t ldd, int64_t dstep, const double* src, int64_t lds,
                                int64_t sstep, int64_t count, int64_t nb, int64_t w, int64_t limit, int64_t lstep,
                                int zero_pad) {
    for (int64_t t = 0; t < count; ++t) {
        int64_t rows = limit - t * lstep;
        rows = rows < 0 ? 0 : (rows > nb ? nb : rows);
        for (int64_t c = 0; c < w; ++c)
            for (int64_t r = 0; r < (zero_pad ? nb : rows); ++r)
                dst[t * dstep + r + c * ldd] = (r < rows) ? src[t * sstep + r + c * lds] : 0.0;
    }
    return 0;
}
static int32_t dop_nccl_bcast(Dev*, void*, double*, int64_t, int) { return MADQP_ERR_STATE; }
static int32_t dop_nccl_reduce(Dev*, void*, double*, int64_t, int) { return MADQP_ERR_STATE; }
static int32_t dop_nccl_allreduce(Dev*, void*, double*, int64_t) { return MADQP_ERR_STATE; }
static int32_t dop_nccl_send(Dev*, void*, const double*, int64_t, int) { return MADQP_ERR_STATE; }
static int32_t dop_nccl_recv(Dev*, void*, double*, int64_t, int) { return MADQP_ERR_STATE; }
static int32_t dop_nccl_group(Dev*, int) { return MADQP_ERR_STATE; }
static int32_t dop_group_pack(Dev*, double* out, const double* loc, int64_t I0, int64_t wg, int64_t nb, int64_t R,
                              int64_t r) {
    for (int64_t i = 0; i < wg; ++i) {
        const int64_t I = I0 + i / nb;
        out[i] = (I % R == r) ? loc[(I / R) * nb + i % nb] : 0.0;
    }
    return 0;
}

#include "../../madqp_jl_amd/csrc/dist_core.inc"

static int32_t dop_solve_local(Dev* dev, double* rhs) {  // one rank: forward and backward substitution over the whole factor
    madqp_dist* d = g_solve_local_of;
    int32_t r = dop_tile_solve(dev, 0, d->K, d->ld, nullptr, rhs, d->n, nullptr);
    return r ? r : dop_tile_solve(dev, 1, d->K, d->ld, nullptr, rhs, d->n, nullptr);
}

static char g_last_error[256] = "";
extern "C" {
const char* madqp_distcpu_last_error() { return g_last_error; }
// (total, K, XW, YW, bands, staging, levels, 0) -- madqp_dist_memory of the product
int32_t madqp_distcpu_memory(madqp_dist* d, int64_t* out8) {
    const int64_t v[8] = {d->bytes_total, d->bytes_K, 8 * d->xw_count, 8 * d->yw_count, d->bytes_band, d->bytes_stage,
                          d->nlev, 0};
    memcpy(out8, v, sizeof(v));
    return 0;
}
int32_t madqp_distcpu_create(int32_t rank, int32_t world, int32_t P, int32_t Q, int64_t n, int64_t nb,
                             const madqp_comm_ops* ops, madqp_dist** out) {
    if (!out || (world > 1 && !ops)) return MADQP_ERR_ARG;
    Dev* dev = new Dev();
    int32_t r = distcore::create(dev, rank, world, P, Q, n, nb, ops, 0, out);
    if (!r) r = distcore::allocate(*out);  // collective: every rank fits, or every rank refuses
    if (r) {
        if (*out) snprintf(g_last_error, sizeof(g_last_error), "%s", (*out)->err);
        distcore::destroy(*out);
        *out = nullptr;
        delete dev;
    }
    return r;
}
int32_t madqp_distcpu_destroy(madqp_dist* d) {
    if (!d) return 0;
    Dev* dev = d->dev;
    distcore::destroy(d);
    delete dev;
    return 0;
}
int32_t madqp_distcpu_layout(madqp_dist* d, int64_t* out8) {
    const int64_t v[8] = {d->p, d->q, d->mt, d->nt, d->mloc, d->nloc, d->ld, d->ncp};
    memcpy(out8, v, sizeof(v));
    return 0;
}
int32_t madqp_distcpu_matrix(madqp_dist* d, double** K, int64_t* ld) {
    *K = d->K;
    *ld = d->ld;
    return 0;
}
int32_t madqp_distcpu_factor(madqp_dist* d, int32_t* info) { return distcore::factor(d, info); }
int32_t madqp_distcpu_solve(madqp_dist* d, double* rhs) {
    g_solve_local_of = d;
    return distcore::solve(d, rhs);
}
int32_t madqp_distcpu_bytes_sent(madqp_dist* d, int64_t* b) {
    *b = d->bytes_sent;
    return 0;
}
// the recorded schedule of the last factorisation: returns the number of int64 entries (triples), copies up to cap
int64_t madqp_distcpu_trace(madqp_dist* d, int64_t* out, int64_t cap) {
    const std::vector<int64_t>& t = d->dev->trace;
    for (int64_t i = 0; i < (int64_t)t.size() && i < cap; ++i) out[i] = t[(size_t)i];
    return (int64_t)t.size();
}
}
