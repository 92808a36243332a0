// fp64 MFMA GEMM core for gfx950:  C[i,j] = alpha * sum_k X[i,k] Y[j,k] + beta * Cin[i,j] (+ dvec on the diagonal)
//
// One kernel serves the three dense contractions of the hot path:
//   * condensed-KKT assembly  K = H + Sigma_x + A' Theta A          (madqp_syrk_assemble)
//   * left-looking Cholesky panel update  C -= L[:, :k] L[J, :k]'   (chol.hip)
//   * panel times inverse diagonal block  L[:, J] = C[:, J] W'      (chol.hip)
//
// Both operands are "k-major": for a fixed k the M (resp. N) entries are contiguous in memory.
// That is the natural layout of column-major L (column k contiguous) and of row-major A
// (row k contiguous), so every tile row is one coalesced 1 KiB wave load and both LDS images
// are [k][index] with the index fastest.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (4 waves = 2 x 2, one per SIMD), each wave
// owns 64 x 64 = 4 x 4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  K is consumed in
// stages of 16, LDS double buffered, one barrier per stage.  Interior tiles stage with LDS-DMA
// (global_load_lds_dwordx4: one wave instruction = one 1 KiB tile row, no VGPR round trip, no
// ds_write); edge tiles stage through registers with bounds checks and zero fill.  Per stage a
// wave issues 64 MFMAs and 32 ds_read_b64.  On gfx950 the fp64 MFMA shares the SIMD's issue with
// every other vector instruction (measured: each v_fma between two MFMAs costs ~9 cycles of
// matrix pipe, tools/mfma_probe.hip), so the loop is written to keep VALU work out of it.
// LDS row stride 144 doubles (= 16 mod 32) makes the two k-rows of a 32-lane ds_read_b64 group
// hit disjoint banks.
//
// MFMA operand roles (v_mfma_f64_16x16x4_f64: D[r][c] = sum_k A[r][k] B[k][c]; lane l feeds
// A[l&15][l>>4] and B[l>>4][l&15]; lane l holds D[(l>>4) + 4v][l&15], v = 0..3):
//   A-operand <- Y fragment (r = column index j of C),  B-operand <- X fragment (c = row index i)
// so the 16 lanes l&15 of an accumulator register walk 16 consecutive rows i of column-major C
// and each store instruction writes four 128-byte segments.
//
// Tile order: the host builds a table of the active tiles (all, or the lower triangle) in
// 8 x 8-tile patches; workgroup ids are remapped so that each XCD (ids equal mod 8 share an
// XCD and its L2) walks one contiguous chunk of the table: the 64 tiles resident on an XCD
// share 8 X panels and 8 Y panels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>

#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {
#include "gemm_core.inc"

struct KArgs {
    GemmArgs g;
    const int32_t* table;
    int32_t ntiles;
    int32_t fast_ok;  // pointers / leading dimensions allow 16-byte loads
    int32_t xcd_remap;
    int32_t batch_xcd;  // batch: problems dealt to XCDs whole (see gemm_tn_f64_kernel)
    GemmBatch batch;  // B <= 1: single problem
    // split-K (launches with fewer tiles than resident workgroups): blockIdx.y = split, each split
    // covers kchunk of K and stores its raw 128 x 128 partial into work[(split * ntiles + t)]
    int32_t ksplit;
    int64_t kchunk;
    double* work;
#ifdef MADQP_STAMPS
    unsigned long long* stamps;  // diagnostic build only (tools/gemm_probe): per-workgroup clocks
#endif
};

// wave-uniform operand offsets of a split-K chunk / of problem blockIdx.y of a batch; false: nothing to do
__device__ __forceinline__ bool gemm_select(const KArgs& ka, GemmArgs& g, int64_t b_batch = -1) {
    if (ka.ksplit > 1) {
        const int64_t k0 = (int64_t)blockIdx.y * ka.kchunk;
        g.X += k0 * g.ldx;
        g.Y += k0 * g.ldy;
        g.K = (g.K - k0 < ka.kchunk) ? (g.K - k0) : ka.kchunk;
    } else if (ka.batch.B > 1 || ka.batch.skip) {  // wave-uniform pointer offsets of problem blockIdx.y
        // (a batch of ONE with a skip list is still a batch: round 3 ignored the list then, and the masked retry rounds of
        // the batched engine re-assembled K over the factor of a problem that had not failed -- unnoticed while the sweeps
        // read only the inverse images and the re-computed panels, fatal once the panel solve reads L_kk itself)
        const int64_t b = b_batch >= 0 ? b_batch : (int64_t)blockIdx.y;
        if (ka.batch.skip && ka.batch.skip[b] != 0) return false;
        g.X += b * ka.batch.sX;
        g.Y += b * ka.batch.sY;
        g.C += b * ka.batch.sC;
        if (g.Cin) g.Cin += b * ka.batch.sCin;
        if (g.dvec) g.dvec += b * ka.batch.sD;
    }
    return true;
}

// one 128 x 128 output tile (entry t of the tile table) by the calling workgroup
__device__ __forceinline__ void gemm_tile(const KArgs& ka, const GemmArgs& g, int t, int bid, double* lds) {
    const int tid = threadIdx.x;
    const int32_t packed = ka.table[t];
    const int64_t i0 = (int64_t)(packed >> 16) * BM;
    const int64_t j0 = (int64_t)(packed & 0xFFFF) * BN;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, lives in an SGPR
    const int wi = wave & 1, wj = wave >> 1;
#ifdef MADQP_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

    const bool interior = ka.fast_ok && (i0 + BM <= (g.Mread ? g.Mread : g.M)) &&
                          (j0 + BN <= (g.Nread ? g.Nread : g.N)) && (g.K % BK == 0);
    if (interior)
        mainloop_dma(g, i0, j0, lds, wi, wj, lane, wave, acc);
    else
        mainloop_staged(g, i0, j0, lds, tid, wi, wj, lane, acc);

#ifdef MADQP_STAMPS
    if (ka.stamps && tid == 0) {
        ka.stamps[2 * (size_t)bid] = __builtin_amdgcn_s_memtime() - st_t0;
        ka.stamps[2 * (size_t)bid + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
    const int lo = lane & 15, hi = lane >> 4;
    if (ka.ksplit > 1) {  // raw partial tile, column-major 128 x 128; the epilogue runs in splitk_reduce_kernel
        double* W = ka.work + ((int64_t)blockIdx.y * ka.ntiles + t) * (BM * BN);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    W[(wj * 64 + tj * 16 + hi + 4 * v) * BM + wi * 64 + ti * 16 + lo] = acc[ti][tj][v];
        return;
    }
    // out = alpha*acc + beta*Cin (+ dvec on the diagonal).  The addend is fetched 16 elements at a time from clamped
    // (always valid) addresses BEFORE the guarded stores: as one guarded read-modify-write per element the compiler
    // emits load, s_waitcnt vmcnt(0), store -- 64 dependent round trips, 40-45 us per tile, which is what held the
    // K = 128 .. 1024 update launches of the factorisation at 15-59 TFLOP/s.  Same arithmetic, same results.
    const bool has_cin = (g.Cin != nullptr);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int64_t gi = i0 + wi * 64 + ti * 16 + lo;
        double cin[4][4];
        if (has_cin) {
            const int64_t gic = gi < g.M ? gi : g.M - 1;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t gj = j0 + wj * 64 + tj * 16 + hi + 4 * v;
                    cin[tj][v] = g.Cin[gic + (gj < g.N ? gj : g.N - 1) * g.ldcin];
                }
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t gj = j0 + wj * 64 + tj * 16 + hi + 4 * v;
                if (gi < g.M && gj < g.N && (!g.lower_only || gi + g.diag_off >= gj)) {
                    double val = g.alpha * acc[ti][tj][v];
                    if (has_cin) val += g.beta * cin[tj][v];
                    g.C[gi + gj * g.ldc] = val;
                }
            }
        }
    }
    // + dvec where global row == global column: a second pass over the (at most 128) diagonal elements of the tile,
    // after the stores above are visible to the workgroup -- val + dvec[j] either way
    if (g.dvec) {
        __syncthreads();
        if (tid < BN) {
            const int64_t gj = j0 + tid, gi = gj - g.diag_off;
            if (gj < g.N && gi >= i0 && gi < i0 + BM && gi < g.M) g.C[gi + gj * g.ldc] += g.dvec[gj];
        }
    }
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_f64_kernel(KArgs ka) {
    __shared__ __attribute__((aligned(16))) double lds[4 * TILE_DOUBLES];
    if (ka.batch.list) {  // compacted batch (GemmBatch::list): the problems of this slot, one after the other
        const int cnt = __builtin_amdgcn_readfirstlane(*ka.batch.count);
        const int bid = blockIdx.x, T = ka.ntiles;
        const int xcd = bid & 7, q = T >> 3, r = T & 7;
        const int t = ka.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
        for (int pb = blockIdx.y; pb < cnt; pb += gridDim.y) {
            const int64_t b = ka.batch.list[pb];
            GemmArgs g = ka.g;
            g.X += b * ka.batch.sX;
            g.Y += b * ka.batch.sY;
            g.C += b * ka.batch.sC;
            if (g.Cin) g.Cin += b * ka.batch.sCin;
            if (g.dvec) g.dvec += b * ka.batch.sD;
            gemm_tile(ka, g, t, bid, lds);
            __syncthreads();  // the LDS image is reused
        }
        return;
    }
    GemmArgs g = ka.g;
    if (ka.batch_xcd) {
        // A batch of small problems (grid = tiles x problems, dispatched tile index first): in launch order the ~10 tiles
        // of ONE problem land on eight different XCDs and each fetches its two operand blocks from HBM -- 5 MB per
        // problem where the operand is 1 MB.  Re-deal: XCD x (workgroups equal mod 8) takes problems x, x + 8, ..
        // whole, their tiles on workgroups that start together, so an operand block is fetched once per XCD.
        const int T = ka.ntiles;
        const int lin = (int)blockIdx.y * T + (int)blockIdx.x, slot = lin >> 3;
        const int64_t b = (int64_t)(slot / T) * 8 + (lin & 7);
        if (!gemm_select(ka, g, b)) return;
        gemm_tile(ka, g, slot % T, (int)blockIdx.x, lds);
        return;
    }
    if (!gemm_select(ka, g)) return;
    // Tile selection: the table is cut into 8 contiguous chunks, one per XCD (workgroup ids equal
    // mod 8 share an XCD under the observed round-robin dispatch; claiming tiles by the real
    // HW_REG_XCC_ID gave the same traffic and time, so the static map is kept).  Speed only: any
    // placement computes the same result.
    const int bid = blockIdx.x, T = ka.ntiles;
    const int xcd = bid & 7, q = T >> 3, r = T & 7;
    const int t = ka.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
    gemm_tile(ka, g, t, bid, lds);
}

// The same tiles by a FIXED number of workgroups (gridDim.x, a multiple of 8, fewer than the chip holds): each draws
// tiles of its XCD's chunk of the table from that XCD's ticket counter until the chunk is used up.  Which workgroup
// computes a tile does not enter the result.  The eight counters are zeroed on the stream before the launch.
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_f64_persistent_kernel(KArgs ka, unsigned long long* tickets) {
    __shared__ __attribute__((aligned(16))) double lds[4 * TILE_DOUBLES];
    __shared__ int next_tile;
    const GemmArgs g = ka.g;
    const int bid = blockIdx.x, T = ka.ntiles;
    const int xcd = bid & 7, q = T >> 3, r = T & 7;
    const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int chunk = q + (xcd < r ? 1 : 0);
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned long long tix = atomicAdd(tickets + xcd, 1ull);
            next_tile = tix < (unsigned long long)chunk ? chunk0 + (int)tix : -1;
        }
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(next_tile);
        if (t < 0) break;
        gemm_tile(ka, g, t, bid, lds);
        __syncthreads();  // the LDS image and next_tile are reused
    }
}

// C tile = epilogue(sum over the splits, in split order) -- deterministic two-pass split-K
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g, const int32_t* __restrict__ table,
                                                            int32_t ntiles, int32_t ksplit,
                                                            const double* __restrict__ work) {
    const int t = blockIdx.x;
    const int32_t packed = table[t];
    const int64_t i0 = (int64_t)(packed >> 16) * BM, j0 = (int64_t)(packed & 0xFFFF) * BN;
    const int li = threadIdx.x & (BM - 1);
    const int64_t gi = i0 + li;
    // eight columns per pass with all their loads issued before the first store, every load unconditional (clamped
    // addresses; the partial tiles are whole): a load under a condition is followed by its own s_waitcnt -- one
    // dependent round trip per element made this kernel 60-80 us for a hundred tiles
    constexpr int U = 8;
    const bool has_cin = (g.Cin != nullptr), has_dvec = (g.dvec != nullptr);
    const int64_t gic = gi < g.M ? gi : g.M - 1;
    for (int lj0 = threadIdx.x >> 7; lj0 < BN; lj0 += 2 * U) {  // lj0 + 2u <= 127
        double sum[U], cin[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t gj = j0 + lj0 + 2 * u, gjc = gj < g.N ? gj : g.N - 1;
            sum[u] = 0.0;
            cin[u] = has_cin ? g.Cin[gic + gjc * g.ldcin] : 0.0;
            dv[u] = has_dvec ? g.dvec[gjc] : 0.0;
        }
        for (int sp = 0; sp < ksplit; ++sp) {  // split order: the same result on every run
            const double* w = work + ((int64_t)sp * ntiles + t) * (BM * BN) + li;
#pragma unroll
            for (int u = 0; u < U; ++u) sum[u] += w[(lj0 + 2 * u) * BM];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t gj = j0 + lj0 + 2 * u;
            if (!(gi < g.M && gj < g.N) || (g.lower_only && gi + g.diag_off < gj)) continue;
            double val = g.alpha * sum[u];
            if (has_cin) val += g.beta * cin[u];
            if (has_dvec && gi + g.diag_off == gj) val += dv[u];
            g.C[gi + gj * g.ldc] = val;
        }
    }
}

struct TableKey {
    int64_t tm, tn, lower, doff;
    uint64_t cols_hash;  // 0: all tile columns
    bool operator<(const TableKey& o) const {
        if (tm != o.tm) return tm < o.tm;
        if (tn != o.tn) return tn < o.tn;
        if (lower != o.lower) return lower < o.lower;
        if (doff != o.doff) return doff < o.doff;
        return cols_hash < o.cols_hash;
    }
};
struct TableVal {
    int32_t* d;
    int32_t n;
    std::vector<int64_t> mask;  // the column ranges / per-column first rows the table was built for: the key holds only
                                // their hash, a hit is a hit only if these agree (the distributed path makes hundreds
                                // of masks per context)
};
// per-context cache of tile tables.  A context is used by one host thread at a time, different
// contexts may live on different threads (batch sharding): the outer map is guarded by a mutex,
// the inner map belongs to its context's thread (std::map nodes are address stable).
std::mutex& table_mutex() {
    static std::mutex m;
    return m;
}
std::map<madqp_ctx*, std::map<TableKey, TableVal>>& table_cache() {
    static std::map<madqp_ctx*, std::map<TableKey, TableVal>> c;
    return c;
}
std::map<TableKey, TableVal>& tables_of(madqp_ctx* ctx) {
    std::lock_guard<std::mutex> lock(table_mutex());
    return table_cache()[ctx];
}
}  // namespace

void madqp_gemm_release_tables(madqp_ctx* ctx) {
    std::map<TableKey, TableVal> mine;
    {
        std::lock_guard<std::mutex> lock(table_mutex());
        auto& all = table_cache();
        auto it = all.find(ctx);
        if (it == all.end()) return;
        mine.swap(it->second);
        all.erase(it);
    }
    for (auto& kv : mine) (void)hipFree(kv.second.d);
}

int32_t madqp_gemm_tn(madqp_ctx* ctx, const GemmArgs& a, int prof_cls, const int64_t* cols, int64_t ncols,
                      const GemmBatch* batch) {
    ARG_TRY(ctx, a.M >= 0 && a.N >= 0 && a.K >= 0 && a.X && a.Y && a.C);
    if (a.M == 0 || a.N == 0) return MADQP_OK;
    const int64_t tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    ARG_TRY(ctx, tiles_m < 65536 && tiles_n < 32768);
    uint64_t cols_hash = 0;
    std::vector<char> col_on;
    if (cols) {
        col_on.assign((size_t)tiles_n, 0);
        cols_hash = 1469598103934665603ull;  // FNV-1a over the range list
        for (int64_t r = 0; r < 2 * ncols; ++r) {
            ARG_TRY(ctx, cols[r] >= 0 && cols[r] <= a.N && (cols[r] % BN == 0 || cols[r] == a.N));
            cols_hash = (cols_hash ^ (uint64_t)cols[r]) * 1099511628211ull;
        }
        if (cols_hash == 0) cols_hash = 1;
        for (int64_t r = 0; r < ncols; ++r)
            for (int64_t t = cols[2 * r] / BN; t < (cols[2 * r + 1] + BN - 1) / BN; ++t) col_on[(size_t)t] = 1;
    }
    if (a.tile_row0) {  // per-column first active tile row: part of the table's identity
        ARG_TRY(ctx, !cols && !a.lower_only);
        cols_hash = 1469598103934665603ull ^ 0x9E3779B97F4A7C15ull;
        for (int64_t t = 0; t < tiles_n; ++t) cols_hash = (cols_hash ^ (uint64_t)a.tile_row0[t]) * 1099511628211ull;
        if (cols_hash == 0) cols_hash = 1;
    }
    // diag_off in tile units must be exact for the tile-skip test used when building the table
    TableKey key{tiles_m, tiles_n,
                 (a.lower_only ? 1 : 0) | ((a.M % BM) != 0 ? 2 : 0) | ((a.N % BN) != 0 ? 4 : 0),
                 a.lower_only ? a.diag_off : 0, cols_hash};
    std::vector<int64_t> mask;
    if (cols) mask.assign(cols, cols + 2 * ncols);
    if (a.tile_row0) mask.assign(a.tile_row0, a.tile_row0 + tiles_n);
    auto& cache = tables_of(ctx);
    auto it = cache.find(key);
    while (it != cache.end() && it->second.mask != mask) {  // same hash, another mask: walk to a free or matching key
        key.cols_hash += 0x9E3779B97F4A7C15ull;
        if (key.cols_hash == 0) key.cols_hash = 1;
        it = cache.find(key);
    }
    if (it == cache.end()) {
        std::vector<int32_t> tab;
        tab.reserve((size_t)tiles_m * tiles_n);
        // patch of PM x PN tiles (tuning knobs for experiments: MADQP_GEMM_PATCH_M / _N)
        static const int64_t PM = getenv("MADQP_GEMM_PATCH_M") ? atoi(getenv("MADQP_GEMM_PATCH_M")) : 8;
        static const int64_t PN = getenv("MADQP_GEMM_PATCH_N") ? atoi(getenv("MADQP_GEMM_PATCH_N")) : 8;
        // Edge tiles (partial last tile row / column, or K not a multiple of 16) run the slower
        // register-staged loop: they go FIRST so that they overlap with the bulk instead of forming
        // the tail of the launch.
        const bool m_edge = (a.M % BM) != 0, n_edge = (a.N % BN) != 0;
        auto active = [&](int64_t tm, int64_t tn) {
            if (cols && !col_on[(size_t)tn]) return false;
            if (a.tile_row0 && tm < a.tile_row0[tn]) return false;
            return !(a.lower_only && (tm * BM + BM - 1 + a.diag_off < tn * BN));
        };
        auto is_edge = [&](int64_t tm, int64_t tn) {
            return (m_edge && tm == tiles_m - 1) || (n_edge && tn == tiles_n - 1);
        };
        if (m_edge)
            for (int64_t tn = 0; tn < tiles_n; ++tn)
                if (active(tiles_m - 1, tn)) tab.push_back((int32_t)(((tiles_m - 1) << 16) | tn));
        if (n_edge)
            for (int64_t tm = 0; tm < tiles_m - (m_edge ? 1 : 0); ++tm)
                if (active(tm, tiles_n - 1)) tab.push_back((int32_t)((tm << 16) | (tiles_n - 1)));
        for (int64_t pm = 0; pm < tiles_m; pm += PM)
            for (int64_t pn = 0; pn < tiles_n; pn += PN)
                for (int64_t tn = pn; tn < pn + PN && tn < tiles_n; ++tn)
                    for (int64_t tm = pm; tm < pm + PM && tm < tiles_m; ++tm) {
                        if (!active(tm, tn) || is_edge(tm, tn)) continue;
                        tab.push_back((int32_t)((tm << 16) | tn));
                    }
        TableVal v{nullptr, (int32_t)tab.size(), mask};
        if (!tab.empty()) {
            HIP_TRY(ctx, hipMalloc(&v.d, tab.size() * sizeof(int32_t)));
            HIP_TRY(ctx, hipMemcpyAsync(v.d, tab.data(), tab.size() * sizeof(int32_t),
                                        hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        it = cache.emplace(key, v).first;
    }
    if (it->second.n == 0) return MADQP_OK;
    KArgs ka;
    ka.g = a;
    ka.table = it->second.d;
    ka.ntiles = it->second.n;
    static const int xcd_remap = getenv("MADQP_GEMM_XCD") ? atoi(getenv("MADQP_GEMM_XCD")) : 1;
    ka.xcd_remap = xcd_remap;
    ka.batch = batch ? *batch : GemmBatch{1, 0, 0, 0, 0, 0, nullptr};
    unsigned gy = (unsigned)std::max<int64_t>(1, ka.batch.B);
    ARG_TRY(ctx, gy <= 65535);
    static const int batch_xcd = getenv("MADQP_GEMM_BATCH_XCD") ? atoi(getenv("MADQP_GEMM_BATCH_XCD")) : 1;
    ka.batch_xcd = (batch_xcd && batch && !ka.batch.list && ka.batch.B >= 8 && ka.batch.B % 8 == 0) ? 1 : 0;
    // Split-K: a launch with far fewer tiles than resident workgroups leaves most of the chip idle while
    // each tile walks all of K alone (5k-20k matrices, the last panels of a large one).  Cut K into up
    // to 16 chunks of >= 256, one workgroup per (tile, chunk), partials summed in chunk order by a second
    // small kernel: same result on every run.
    static const int split_on = getenv("MADQP_GEMM_SPLITK") ? atoi(getenv("MADQP_GEMM_SPLITK")) : 1;
    ka.ksplit = 1;
    ka.kchunk = a.K;
    ka.work = nullptr;
    int64_t S_few = 0;
    if (split_on && !batch && (int64_t)ka.ntiles * 2 > ctx->gemm_slots && (int64_t)ka.ntiles < 8 * ctx->gemm_slots &&
        a.K >= 4096) {
        // A launch of a few rounds of LONG tiles (the wide updates of the lazy distributed schedule, whose tile-column
        // width is the grid's tile and cannot be tuned to fill the rounds as chol.hip tunes its panels: 560 tiles of
        // K = 38 000 take two rounds of 11 ms, the second one a tenth full, and whoever shares a CU with a finished
        // workgroup runs on alone): cut K into S chunks -- more, shorter rounds.  S minimises the model
        //   rounds(S) x (tile time of K/S + fixed cost of a tile) + pass over the S x tiles partial tiles
        // (microseconds: 0.216 per unit of K, 10 per tile, 0.05 per partial tile).  With short tiles (n_x = 5 000:
        // K = 2 000) the model and the measurement agree that it does not pay; those launches are left alone.
        auto cost = [&](int64_t S) {  // (an extra term for the drain of the last round was tried: 0.5 .. 2 tile times cost 0 .. 5 ms)
            const double rounds = std::ceil((double)ka.ntiles * (double)S / (double)ctx->gemm_slots);
            return rounds * (0.216 * (double)a.K / (double)S + 10.0) + 0.05 * (double)S * (double)ka.ntiles;
        };
        double best = cost(1) * 0.97;
        for (int64_t S = 2; S <= 16 && a.K / S >= 2048; ++S)
            if (cost(S) < best) {
                best = cost(S);
                S_few = S;
            }
    }
    if (split_on && !batch && (((int64_t)ka.ntiles * 2 <= ctx->gemm_slots && a.K >= 512) || S_few)) {
        int64_t S = S_few ? S_few : std::min<int64_t>(std::min<int64_t>(ctx->gemm_slots / ka.ntiles, a.K / 256), 16);
        if (S >= 2) {
            const int64_t chunk = ((a.K + S - 1) / S + BK - 1) / BK * BK;
            S = (a.K + chunk - 1) / chunk;
            if (S >= 2) {
                const size_t bytes = (size_t)S * ka.ntiles * BM * BN * sizeof(double);
                int32_t r = madqp_work_reserve(ctx, bytes);
                if (r) return r;
                ka.ksplit = (int32_t)S;
                ka.kchunk = chunk;
                ka.work = ctx->d_work;
                gy = (unsigned)S;
            }
        }
    }
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    ka.fast_ok = al16(a.X) && al16(a.Y) && (a.ldx % 2 == 0) && (a.ldy % 2 == 0);
#ifdef MADQP_STAMPS
    extern unsigned long long* madqp_stamp_buffer;
    ka.stamps = madqp_stamp_buffer;
#endif
    // Tail split (round 4): a launch of one to three rounds whose LAST round is partly empty -- the assembly of a
    // mid-size matrix: 820 tiles on 512 slots are 1.6 rounds that take the time of 2 -- runs its whole rounds as they are
    // and cuts only the tiles of the last round into S chunks of K (more, shorter pieces that fill the chip), summed in
    // chunk order by splitk_reduce_kernel like every split launch: same result on every run.  S from the same cost model
    // as above (microseconds: 0.216 per unit of K and 10 per piece, 0.05 per partial tile of the second pass).
    int64_t tail_S = 0, tail_n = 0;
    static const int tail_on = getenv("MADQP_GEMM_TAILSPLIT") ? atoi(getenv("MADQP_GEMM_TAILSPLIT")) : 1;
    if (split_on && tail_on && !batch && ka.ksplit == 1 && a.K >= 1024 && ctx->gemm_cap_slots == 0 &&
        (int64_t)ka.ntiles > ctx->gemm_slots && (int64_t)ka.ntiles < 4 * ctx->gemm_slots) {
        const int64_t slots = ctx->gemm_slots, tl = (int64_t)ka.ntiles % slots;
        if (tl > 0 && 10 * tl < 8 * slots) {
            const double tile = 0.216 * (double)a.K + 10.0;
            double best = tile * 0.93;  // (the last round as it is; worth it only with a clear gain)
            for (int64_t S = 2; S <= 8 && a.K / S >= 256; ++S) {
                const double rounds = std::ceil((double)tl * (double)S / (double)slots);
                const double c = rounds * (0.216 * (double)a.K / (double)S + 10.0) + 0.05 * (double)S * (double)tl;
                if (c < best) {
                    best = c;
                    tail_S = S;
                }
            }
            if (tail_S) tail_n = tl;
        }
    }
    // Long launches are cut into segments of 64 rounds of resident workgroups.  Equal-cost tiles
    // that start together sweep K in lockstep and share their operand panels through the XCD's
    // L2; over many rounds that lockstep diffuses away (measured at n = 50000, K = 20480: 0.93-1.3 TB
    // fetched by one assembly launch, 0.75 TB when re-synchronised every 64 rounds, floor 0.68 TB).
    // The last round of a segment finishes almost simultaneously: +0.25 % time.
    static const int seg_rounds = getenv("MADQP_GEMM_SEG_ROUNDS") ? atoi(getenv("MADQP_GEMM_SEG_ROUNDS")) : 64;
    const int64_t seg = seg_rounds > 0 ? (int64_t)seg_rounds * ctx->gemm_slots : (int64_t)ka.ntiles;
    const int32_t* table0 = ka.table;
    const int32_t total = ka.ntiles - (int32_t)tail_n;  // (tiles launched whole)
    ProfScope ps(ctx, prof_cls);
    // capped launch (ctx->gemm_cap_slots, set by dist.hip around a bulk trailing update): a persistent grid that leaves
    // workgroup slots free for the kernels of other streams
    const int64_t capped = ctx->gemm_cap_slots > 0 ? std::max<int64_t>(8, (ctx->gemm_slots - ctx->gemm_cap_slots) / 8 * 8) : 0;
    if (capped && ka.ksplit == 1 && !batch && total > capped) {
        if (!ctx->d_tickets) HIP_TRY(ctx, hipMalloc(&ctx->d_tickets, 8 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_tickets, 0, 8 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(gemm_tn_f64_persistent_kernel, dim3((unsigned)capped), dim3(NTHREADS), 0, ctx->stream, ka,
                           ctx->d_tickets);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    for (int64_t off = 0; off < total; off += seg) {
        int64_t cnt = std::min<int64_t>(seg, total - off);
        if (total - off - cnt < seg / 4) cnt = total - off;  // no tiny last segment
        ka.table = table0 + off;
        ka.ntiles = (int32_t)cnt;
        hipLaunchKernelGGL(gemm_tn_f64_kernel, dim3(ka.ntiles, gy), dim3(NTHREADS), 0, ctx->stream, ka);
        LAUNCH_CHECK(ctx);
        if (cnt == total - off) break;
    }
    if (tail_n) {  // the tiles of the last round, K-split
        const int64_t chunk = ((a.K + tail_S - 1) / tail_S + BK - 1) / BK * BK;
        const int64_t S = (a.K + chunk - 1) / chunk;
        const size_t bytes = (size_t)S * tail_n * BM * BN * sizeof(double);
        int32_t r = madqp_work_reserve(ctx, bytes);
        if (r) return r;
        ka.table = table0 + total;
        ka.ntiles = (int32_t)tail_n;
        ka.ksplit = (int32_t)S;
        ka.kchunk = chunk;
        ka.work = ctx->d_work;
        hipLaunchKernelGGL(gemm_tn_f64_kernel, dim3(ka.ntiles, (unsigned)S), dim3(NTHREADS), 0, ctx->stream, ka);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)tail_n), dim3(256), 0, ctx->stream, a, table0 + total,
                           (int32_t)tail_n, (int32_t)S, ka.work);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    if (ka.ksplit > 1) {  // (a split launch is always a single segment: few tiles)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(total), dim3(256), 0, ctx->stream, a, table0, total, ka.ksplit,
                           ka.work);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}

namespace {
// S[k, :] = sqrt(w[k]) * B[k, :], zero padded to npad columns and kpad rows: one streaming pass
// (16 n kdim bytes).  The assembly then is the plain product S'S with BOTH operands read from this
// library-owned, 128-aligned image: no per-k scaling inside the MFMA loop (every VALU instruction
// there costs matrix-pipe cycles), a single operand stream, and no partial tiles.
__global__ __launch_bounds__(256) void scale_rows_kernel(int64_t n, int64_t npad, int64_t kdim,
                                                         const double* __restrict__ B, int64_t ldb,
                                                         const double* __restrict__ w,
                                                         double* __restrict__ out, int vec) {
    const int64_t k = blockIdx.y;
    double* dst = out + k * npad;
    if (k >= kdim) {  // zero rows up to a multiple of 16
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256)
            dst[i] = 0.0;
        return;
    }
    const double wk = sqrt(w[k]);
    const double* src = B + k * ldb;
    if (vec) {
        const int64_t pairs = npad >> 1, full = n >> 1;
        for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (int64_t)gridDim.x * 256) {
            double2_t v = {0.0, 0.0};
            if (p < full) {
                v = *reinterpret_cast<const double2_t*>(src + 2 * p);
                v.x *= wk;
                v.y *= wk;
            } else if (2 * p < n) {
                v.x = src[2 * p] * wk;
            }
            *reinterpret_cast<double2_t*>(dst + 2 * p) = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256)
            dst[i] = (i < n) ? src[i] * wk : 0.0;
    }
}
}  // namespace

// Lower triangle of C = base + diag(dvec) + B' diag(w) B, restricted to the column ranges
// [ranges[2r], ranges[2r+1]) (rows >= the range start); ranges == nullptr: all columns.
static int32_t syrk_assemble_impl(madqp_ctx* ctx, int64_t n, int64_t kdim, const double* B, int64_t ldb,
                                  const double* w, const double* base, int64_t ldbase,
                                  const double* dvec, double* C, int64_t ldc, int64_t nranges,
                                  const int64_t* ranges) {
    ARG_TRY(ctx, n >= 0 && kdim >= 0 && C && ldc >= n && (kdim == 0 || (B && ldb >= n)));
    ARG_TRY(ctx, !base || ldbase >= n);
    const double* X = B ? B : C;  // K == 0: never dereferenced
    int64_t ldx = ldb, K = kdim, npad = 0;
    if (w && kdim > 0 && n > 0) {
        npad = (n + BM - 1) / BM * BM;
        const int64_t kpad = (kdim + BK - 1) / BK * BK;
        const size_t bytes = (size_t)kpad * npad * sizeof(double);
        if (bytes > ctx->scaled_bytes) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_scaled) HIP_TRY(ctx, hipFree(ctx->d_scaled));
            ctx->d_scaled = nullptr;
            ctx->scaled_bytes = 0;
            hipError_t e = hipMalloc(&ctx->d_scaled, bytes);
            if (e != hipSuccess)
                return madqp_fail(ctx, MADQP_ERR_ALLOC, "scaled operand hipMalloc(%zu): %s", bytes,
                                  hipGetErrorString(e));
            ctx->scaled_bytes = bytes;
        }
        const int vec = ((((uintptr_t)B) & 15) == 0) && (ldb % 2 == 0);
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((npad / 2 + 255) / 256, 64));
            hipLaunchKernelGGL(scale_rows_kernel, dim3(gx, (unsigned)kpad), dim3(256), 0, ctx->stream, n,
                               npad, kdim, B, ldb, w, ctx->d_scaled, vec);
            LAUNCH_CHECK(ctx);
        }
        X = ctx->d_scaled;
        ldx = npad;
        K = kpad;
    }
    if (ranges) {
        bool any = false;
        for (int64_t r = 0; r < nranges; ++r) {
            ARG_TRY(ctx, 0 <= ranges[2 * r] && ranges[2 * r] <= ranges[2 * r + 1] && ranges[2 * r + 1] <= n &&
                             ranges[2 * r] % BM == 0);
            any = any || ranges[2 * r + 1] > ranges[2 * r];
        }
        if (!any) return MADQP_OK;
    }
    GemmArgs g{};
    g.X = X;
    g.ldx = ldx;
    g.Y = X;
    g.ldy = ldx;
    g.K = K;
    if (npad) g.Mread = g.Nread = npad;
    g.C = C;
    g.ldc = ldc;
    g.Cin = base;
    g.ldcin = ldbase;
    g.dvec = dvec;
    g.alpha = 1.0;
    g.beta = 1.0;
    g.M = n;
    g.N = n;
    g.diag_off = 0;
    g.lower_only = 1;
    return madqp_gemm_tn(ctx, g, MADQP_PROF_SYRK, ranges, ranges ? nranges : 0);  // one launch, masked columns
}

int32_t madqp_syrk_assemble_ranges(madqp_ctx* ctx, int64_t n, int64_t kdim, const double* B, int64_t ldb,
                                   const double* w, const double* base, int64_t ldbase,
                                   const double* dvec, double* C, int64_t ldc, int64_t nranges,
                                   const int64_t* ranges) {
    return syrk_assemble_impl(ctx, n, kdim, B, ldb, w, base, ldbase, dvec, C, ldc, nranges, ranges);
}

extern "C" int32_t madqp_syrk_assemble(madqp_ctx* ctx, int64_t n, int64_t kdim, const double* B,
                                       int64_t ldb, const double* w, const double* base,
                                       int64_t ldbase, const double* dvec, double* C,
                                       int64_t ldc) {
    ARG_TRY(ctx, ctx != nullptr);
    return syrk_assemble_impl(ctx, n, kdim, B, ldb, w, base, ldbase, dvec, C, ldc, 0, nullptr);
}

// ------------------------------------------------------------------ hardware probe
namespace {
__global__ __launch_bounds__(256) void mfma_f64_probe_kernel(int iters, double* sink) {
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    // inline asm pins the accumulators in VGPRs: with the builtin the compiler parks them in AGPRs and the
    // loop measures its accvgpr copies (47 instead of 77.7 TFLOP/s; tools/mfma_probe.hip)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;  // keep the chain alive
}
}  // namespace

extern "C" int32_t madqp_probe_mfma_f64(madqp_ctx* ctx, int32_t iters, double* tflops_host) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, iters > 0 && tflops_host);
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 2;  // two 256-thread workgroups per CU
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, 64, ctx->d_res);
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, iters, ctx->d_res);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2048.0;
    *tflops_host = flops / (ms * 1e-3) * 1e-12;
    return MADQP_OK;
}
