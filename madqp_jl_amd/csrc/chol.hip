// Blocked fp64 Cholesky (lower, column-major, in place) and the triangular solves.
//
// Replaces the AbstractLinearSolver used behind NormalKKTSystem (MadNLP.LapackCPUSolver -> LAPACK
// dpotrf/dpotrs; reference call sites src/KKT/normalkkt.jl:99-101,196, src/linear_solver.jl:10-11).
//
// Factorisation, two levels of left-looking blocking so that >96 % of the flops run in wide GEMMs:
//   for each outer panel J (768..2560 columns, chosen to fill the last round of workgroups)
//     C[J0:n, J]  -= L[J0:n, 0:J0] * L[J, 0:J0]'            gemm core, N = |J|,  K = J0   (MFMA)
//     inside J, recursively: factor the first half, update the second half with it
//       C[h:n, h:w] -= L[h:n, 0:h] * L[h:w, 0:h]'           gemm core, N = K = w/2        (MFMA)
//     down to 128-column blocks jb:
//       L_jj = chol(C_jj),  W_jj = L_jj^-1                   one workgroup, block resident in LDS
//       L[jb+128:n, jb] = C[jb+128:n, jb] * W_jj'            gemm core, K = N = 128        (MFMA)
// Matrices up to n = 13 312 take a right-looking schedule instead (chol_mid_step_kernel below): two launches per
// 128-column block, the diagonal block factored inside the launch that updates trailing tiles -- which tiles, with
// which panels, a plan made once per handle decides (mid_plan.inc: updates are applied when there is room, two
// panels at a time, no later than due).
// The inverse diagonal blocks W are kept (two images, 2 x n x 128 doubles) and turn the diagonal
// solves of the two triangular sweeps into 128 x 128 mat-vecs; each sweep is ONE launch of
// ticket-ordered workgroups handing the solved blocks on through a sentinel-tagged vector, HBM bound
// (4 n^2 bytes); block rows longer than 64 tiles are streamed by several workgroups (SweepPlan).  Also here: the same factorisation for a batch of small matrices
// (madqp_chol_factor_batched) and its pieces for the multi-GPU panel loop (madqp_chol_factor_panel, ...).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <vector>

#include "common.h"
#include "mid_plan.inc"

typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {
constexpr int NB = 128;    // diagonal block
constexpr int NBO = 1024;  // outer panel
#ifndef P2_LDS_LD
#define P2_LDS_LD (NB + 1)
#endif
constexpr int LDS_LD = P2_LDS_LD;
constexpr int64_t WBLK = 2 * NB * NB;  // doubles per block in chol->winv: [Wcm | Wrm]

typedef double double4_t __attribute__((ext_vector_type(4)));
#include "gemm_core.inc"
constexpr int SB = 16;             // sub-block of the diagonal kernel (one MFMA tile)
constexpr int NSB = NB / SB;       // 8 sub-block columns
constexpr int WD_LD = SB + 1;      // row stride of the inverse diagonal sub-blocks

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(a) for a normal, positive pivot: hardware estimate (v_rsq_f64: 2^-24.2 measured) + two coupled
// Goldschmidt steps in fma form -- eight instructions, six deep, instead of the sqrt-and-divide sequence behind rsqrt().
// Max error 1.90 ulp over 4 M arguments (tools/scratch/rsq_probe.hip).  (One third-order step, y (1 + e + 3/2 e^2)
// with e = (1 - a y^2) / 2, is six instructions, four deep, 1.24 ulp -- and 0.05 us per 16 pivots, which is not worth
// moving every last bit of every factor for: the distributed path's hardest parity case then sits at 19 x instead of
// under 16 x the distance between two CPU runs.)
__device__ __forceinline__ double fast_rsqrt(double a) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
    return h + h;
}

// the 16 x 16 sub-block at (b, b) of S, both triangles from the stored lower one, in the accumulator layout
__device__ __forceinline__ double4_t diag16_load(const double* __restrict__ S, int b, int lane) {
    const int lo = lane & 15, hi = lane >> 4;
    double4_t A;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = hi + 4 * v;
        A[v] = (row >= lo) ? S[(b + lo) * LDS_LD + b + row] : S[(b + row) * LDS_LD + b + lo];
    }
    return A;
}
// One wave: Cholesky of the 16 x 16 sub-block at (b, b) of S and the inverse of its factor, in four RANK-4 steps on the
// MFMA unit.  The (symmetric) sub-block and W live in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l,
// register v: row (l>>4) + 4v, column l&15), so the four rows 4p .. 4p+3 of the block are ONE register (A[p]: lane
// group hi holds row 4p + hi) -- and one register per lane group is exactly what a k-slot of the instruction takes: the
// four scaled rows of a 4-row group are the A- and the B-operand of D -= sum_j l_j l_j' as they stand, ONE dependent
// MFMA per four pivots.  (Round 1 kept a row per lane and broadcast with 2 x 135 v_readlane: 5.0 us per sub-block;
// rounds 1-2 ran 16 rank-1 MFMA steps, 3.08 us: every pivot paid the round trip through a dependent MFMA; this form
// 2.84 us -- what remains is 16 x (read-lane -> rsqrt chain -> scale -> read-lane -> fma) on the vector ALU, ~175 ns
// each, tools/potf2_probe.)  The 4 x 4 elimination inside the group runs
// on the vector ALU, on all 64 lanes redundantly: the four rows (and the four rows of W) are first replicated to every
// lane group (one cross-lane read each), after which a pivot needs two read-lanes, the rsqrt chain and one fma per
// later row.  Every update is one fused multiply-add, in the order of the pivots; W = L^-1 follows from the same steps
// applied to the identity (W[k,:] *= 1/l_kk, W[i,:] -= l_ik W[k,:], i > k).  Writes L (lower) back to S and W to
// Wd[r*WD_LD + c] (zero above the diagonal).
// A: the symmetric sub-block in the accumulator layout (diag16_load, or the accumulators of the update that completed it)
__device__ __forceinline__ void diag16_factor_invert(double4_t A, double* __restrict__ S, int b, double* __restrict__ Wd,
                                                        int32_t* __restrict__ info, int32_t col0, int lane) {
    const int lo = lane & 15, hi = lane >> 4;
    double4_t W;
#pragma unroll
    for (int v = 0; v < 4; ++v) W[v] = (hi + 4 * v == lo) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int k0 = 4 * p;
        double R[4], V[4], l[4], w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // rows k0 + j of A and of W, in every lane group
            R[j] = __shfl(A[p], lo + 16 * j, 64);
            V[j] = __shfl(W[p], lo + 16 * j, 64);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            double akk = readlane_f64(R[j], k);
            if (__builtin_expect(!(akk > 0.0), 0)) {  // not positive definite (or NaN): record the first column, go on with a unit pivot
                if (lane == 0) atomicCAS(info, 0, col0 + b + k + 1);
                akk = 1.0;
                R[j] = (lo == k) ? 1.0 : R[j];
            }
            const double rd = fast_rsqrt(akk);
            // row k scaled: column k of it is a_kk / sqrt(a_kk) = l_kk itself (the row holds a_kk there), exact zeros above
            // the diagonal.  (The pivot steps are bound by the number of vector instructions one wave issues in order,
            // ~48 per pivot, not by the latency of any of them: every select and masked store here is time on the chain.)
            const double lk = (lo >= k) ? R[j] * rd : 0.0;
            l[j] = lk;
            w[j] = V[j] * rd;  // row k of W, scaled
#pragma unroll
            for (int j2 = j + 1; j2 < 4; ++j2) {  // the later rows of the group: A[i,:] -= l_ik l_k', W[i,:] -= l_ik w_k
                const double s = readlane_f64(lk, k0 + j2);
                R[j2] = __builtin_fma(-s, lk, R[j2]);
                V[j2] = __builtin_fma(-s, w[j], V[j2]);
            }
        }
        const double lsel = (hi == 0) ? l[0] : (hi == 1) ? l[1] : (hi == 2) ? l[2] : l[3];
        const double wsel = (hi == 0) ? w[0] : (hi == 1) ? w[1] : (hi == 2) ? w[2] : w[3];
        if (lo >= k0 + hi) S[(b + k0 + hi) * LDS_LD + b + lo] = lsel;  // rows k0 .. k0+3 of L, one lane group each
        A = __builtin_amdgcn_mfma_f64_16x16x4f64(-lsel, lsel, A, 0, 0, 0);
        const double aw = (lo >= k0 + 4) ? lsel : 0.0;  // rows below the group; its own rows were done above
        W = __builtin_amdgcn_mfma_f64_16x16x4f64(-aw, wsel, W, 0, 0, 0);
        W[p] = wsel;  // rows k0 .. k0+3 of W are final
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) Wd[(hi + 4 * v) * WD_LD + lo] = W[v];
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
#ifdef MADQP_POTF2_STAMPS
__device__ unsigned long long madqp_potf2_stamps[64];  // diagnostic build only (tools/potf2_probe.cpp)
#define P2_STAMP(i)                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && MODE != 2) madqp_potf2_stamps[i] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define P2_STAMP(i)
#endif
struct Potf2Batch {  // problem blockIdx.x: pointer strides (doubles / ints); skip[b] != 0: leave untouched
    int64_t sA, sW, sInfo;
    const int32_t* skip;
    const int32_t* list = nullptr;   // compacted form (GemmBatch::list): slot blockIdx.x works off list[x], list[x + grid], ..
    const int32_t* count = nullptr;
};
#ifndef P2_QUIET_WAVE
#define P2_QUIET_WAVE 4
#endif
constexpr int P2_QUIET = P2_QUIET_WAVE;  // -1: every helper wave works in the trailing phase
// 512 threads: wave 0 runs the chain of 16 x 16 diagonal factorisations, waves 1..7 everything else
constexpr int P2_S_DOUBLES = NB * LDS_LD;         // S[c*LDS_LD + r] = element (r, c)
constexpr int P2_WD_DOUBLES = NSB * SB * WD_LD;   // inverse diagonal sub-blocks
// The helper waves' tile products of step J, NHE waves sharing them round-robin: waves[J][hr] = up to eight entries of
// 8 bits, 0xFF ends the list; bit 6: 0 = trailing tile (K = bits 5..3, I = bits 2..0), 1 = inverse tile (I, J').
// Drawn at compile time: enumerating the 34 tiles of a step in every wave (loop, modulo, three branches per tile) cost
// more instruction issue and fetch than the products themselves -- the phase got SLOWER with more helper waves.
// MODE 0: trailing tiles and inverse tiles (factor + invert in one kernel); 1: trailing tiles only (factor); 2: inverse tiles
// only (the inverse of an already factored block)
template <int NHE, int MODE>
struct P2Lists {
    static_assert(NHE >= 5, "at most 34 tiles per step: seven entries and the end mark per wave");
    unsigned long long waves[NB / 16][NHE];
    constexpr P2Lists() : waves{} {
        constexpr int N = NB / 16;
        for (int J = 0; J < N; ++J) {
            int cnt[NHE] = {};
            for (int h = 0; h < NHE; ++h) waves[J][h] = ~0ull;
            int c = 0;
            auto put = [&](int e) {
                const int h = c % NHE;
                waves[J][h] = (waves[J][h] & ~(0xFFull << (8 * cnt[h]))) | ((unsigned long long)e << (8 * cnt[h]));
                ++cnt[h];
                ++c;
            };
            if (MODE != 2)
                for (int K = J + 1; K < N; ++K)
                    for (int I = K; I < N; ++I)
                        if (!(K == J + 1 && I == J + 1)) put(K * 8 + I);
            if (MODE != 1)
                for (int I = J + 1; I < N; ++I)
                    for (int Jp = 0; Jp <= J; ++Jp) put(64 + I * 8 + Jp);
        }
    }
};
template <int NHE, int MODE>
__device__ __constant__ const P2Lists<NHE, MODE> p2_lists{};

// LDS reads of a 16 x 16 tile product issued TOGETHER (inline asm: left to itself the compiler puts every operand pair
// next to the MFMA that consumes it -- read, wait, MFMA, four times in a row, five LDS round trips per tile product:
// 0.76 us per product on a helper wave where the four dependent MFMAs take 0.11).
template <int OFF>
__device__ __forceinline__ double p2_read(unsigned addr) {
    double d;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
    return d;
}
template <int STRIDE>
__device__ __forceinline__ void p2_read4(unsigned addr, double (&v)[4]) {
    v[0] = p2_read<0>(addr);
    v[1] = p2_read<STRIDE>(addr);
    v[2] = p2_read<2 * STRIDE>(addr);
    v[3] = p2_read<3 * STRIDE>(addr);
}
__device__ __forceinline__ void p2_wait(double (&a)[4], double (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 :
                 : "memory");
}
__device__ __forceinline__ void p2_wait(double (&a)[4], double (&b)[4], double (&c)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 :
                 : "memory");
}
__device__ __forceinline__ unsigned p2_lds(const double* p) { return (unsigned)(uintptr_t)(lds_ptr_t)p; }
// the whole workgroup (NT threads: 512, or 1024 in probe builds) calls this; S and Wd are its LDS work areas
// MODE (round 4): 0 = factor and invert (as rounds 1-3); 1 = factor only -- L_jj and the inverses of its eight 16 x 16
// diagonal sub-blocks, which is all the panel solve (panel_sub16_kernel) needs: the serial spine of a factorisation then
// carries neither the inverse's tile products nor the stores of two 128 x 128 images; 2 = the inverse of an already
// factored block from L_jj and those sub-block inverses -- every block of a factorisation in ONE launch, off the chain
// (potf2_invert_kernel), before anything reads the images (the sweeps).
template <int NT, int MODE = 0>
__device__ __forceinline__ void potf2_inv_body(double* __restrict__ A, int64_t lda, int nb,
                                               double* __restrict__ Wcm, double* __restrict__ Wrm,
                                               int32_t* __restrict__ info, int32_t col0,
                                               double* __restrict__ S, double* __restrict__ Wd,
                                               bool tile_in_lds = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    P2_STAMP(0);
    if (!tile_in_lds) {  // lower triangle -> LDS: row r = tid & 127, columns (tid >> 7) + CG*i; all loads of a thread in flight at
        // once (one HBM / L2 latency; the kernel runs alone on its CU, registers are free)
        constexpr int CG = NT / NB;
        const int r = tid & (NB - 1), c0 = tid >> 7;
        double v[NB / CG];
#pragma unroll
        for (int i = 0; i < NB / CG; ++i) {
            const int c = c0 + CG * i;
            double x = 0.0;
            if (r < nb && c < nb) {
                if (r >= c) x = A[r + (int64_t)c * lda];
            } else if (r == c) {
                x = 1.0;  // identity padding keeps the padded block positive definite
            }
            v[i] = x;
        }
#pragma unroll
        for (int i = 0; i < NB / CG; ++i) S[(c0 + CG * i) * LDS_LD + r] = v[i];
    }
    if (MODE == 2) {  // the sub-block inverses the factor-only kernel left on the diagonal of the column-major image
        // (rows at or beyond the order of a short block were never written by that kernel: identity padding, as MODE 0 has)
        for (int e = tid; e < NSB * SB * SB; e += NT) {
            const int J = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            const double wv = Wcm[(SB * J + cc) * NB + SB * J + rr];
            Wd[(J * SB + rr) * WD_LD + cc] = (SB * J + rr < nb) ? ((cc <= rr) ? wv : 0.0) : ((rr == cc) ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    P2_STAMP(1);
    // one 16 x 16 trailing tile (K, I) of step J: A_IK -= L_IJ L_KJ'
    auto trailing_tile = [&](int J, int K, int I) {
        const int b = J * SB;
        double* cp = S + (SB * K + lo) * LDS_LD + SB * I + hi;
        double c[4], av[4], bv[4];
        p2_read4<4 * 8>(p2_lds(cp), c);
        p2_read4<4 * LDS_LD * 8>(p2_lds(S + (b + hi) * LDS_LD + SB * I + lo), av);  // L_IJ[lo][k]
        p2_read4<4 * LDS_LD * 8>(p2_lds(S + (b + hi) * LDS_LD + SB * K + lo), bv);  // L_KJ[lo][k]
        p2_wait(c, av, bv);
        double4_t acc = {c[0], c[1], c[2], c[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) cp[4 * v] = acc[v];
    };
    // W = L^-1 is built inside the same eight steps by the helper waves (wave 0's 16-step chain is the critical
    // path of a step): T_IJ' = sum_{K=J'}^{I-1} L_IK W_KJ' accumulates in the unused upper triangle (W(r, c) lives
    // at S[r*LDS_LD + c], r > c) -- step J adds L_IJ W_JJ' for every I > J, J' <= J -- and row I is finished as
    // W_IJ' = -W_II T_IJ' at the start of step I, once step I-1 has produced W_II.  The f64 accumulator layout
    // (row = (lane>>4) + 4v) is the B-operand layout (k = 4s + (lane>>4)), so T feeds the second product from
    // registers.  (As a separate phase after the factorisation this cost 8.4 us of the block's 58.)
    auto t_update = [&](int J, int I, int Jp) {  // T_IJ' += L_IJ W_JJ'
        double* tp = S + (SB * I + hi) * LDS_LD + SB * Jp + lo;
        double c[4], av[4], bv[4];
        p2_read4<4 * LDS_LD * 8>(p2_lds(tp), c);
        p2_read4<4 * LDS_LD * 8>(p2_lds(S + (SB * J + hi) * LDS_LD + SB * I + lo), av);  // L_IJ[lo][k]
        if (Jp == J)
            p2_read4<4 * WD_LD * 8>(p2_lds(Wd + (J * SB + hi) * WD_LD + lo), bv);  // W_JJ[k][lo]
        else
            p2_read4<4 * LDS_LD * 8>(p2_lds(S + (SB * J + hi) * LDS_LD + SB * Jp + lo), bv);  // W_JJ'[k][lo]
        p2_wait(c, av, bv);
        double4_t T = {c[0], c[1], c[2], c[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) T = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], T, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) tp[4 * v * LDS_LD] = T[v];
    };
    auto w_finish = [&](int I, int Jp) {  // W_IJ' = -W_II T_IJ'
        double4_t T, R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int v = 0; v < 4; ++v) T[v] = S[(SB * I + hi + 4 * v) * LDS_LD + SB * Jp + lo];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double av = Wd[(I * SB + lo) * WD_LD + 4 * s + hi];  // W_II[lo][4s+hi]
            R = __builtin_amdgcn_mfma_f64_16x16x4f64(av, T[s], R, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) S[(SB * I + hi + 4 * v) * LDS_LD + SB * Jp + lo] = -R[v];
    };
    constexpr int NWV = NT / 64, NH = NWV - 1;  // waves; helper waves 1..NH
    if (MODE != 2 && wave == 0) diag16_factor_invert(diag16_load(S, 0, lane), S, 0, Wd, info, col0, lane);
    __syncthreads();
    P2_STAMP(2);
    for (int J = 0; J < NSB; ++J) {
        const int b = J * SB;
        double* WdJ = Wd + J * SB * WD_LD;
        // panel L_IJ = A_IJ * W_JJ' (I > J) and the rows W_JJ' (J' < J) of the inverse: 7 tile products
        for (int o = wave; o < NSB - 1; o += NWV) {
            if (o < J) {
                if (MODE != 1) w_finish(J, o);
                continue;
            }
            if (MODE == 2) continue;  // (the panel tiles of L are final)
            const int I = o + 1;
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
        P2_STAMP(3 + 3 * J);
        // trailing update with look-ahead: wave 0 updates the next diagonal tile and factors it at once
        // (the 16-step chain of diag16_factor_invert is the critical path) while the helper waves update the
        // other tiles of this step and the T tiles of the inverse
        if (J + 1 < NSB) {
            if (wave == 0) {
                if (MODE != 2) {
                    trailing_tile(J, J + 1, J + 1);
                    // (keeping this update in registers as the input of the pivot chain, instead of the round trip through
                    // S, changed nothing: 2.12-2.20 us per step either way)
                    diag16_factor_invert(diag16_load(S, b + SB, lane), S, b + SB, WdJ + SB * WD_LD, info, col0, lane);
                }
            } else if (P2_QUIET < 0 || (wave & 3) != 0) {
                // (waves 4, 8, .. share their SIMD with wave 0 and sit this phase out: the fp64 MFMAs of a helper
                // there hold up every vector instruction of the pivot chain)
                constexpr int NHE = (P2_QUIET < 0) ? NH : NH - (NWV - 1) / 4;
                const int hr = __builtin_amdgcn_readfirstlane((P2_QUIET >= 0) ? wave - 1 - (wave >> 2) : wave - 1);  // 0 .. NHE-1
                unsigned long long lst = p2_lists<NHE, MODE>.waves[J][hr];  // this wave's tiles of the step (P2Lists)
                while ((lst & 0xFF) != 0xFF) {
                    const int e = (int)(lst & 0xFF);
                    lst = (lst >> 8) | (0xFFull << 56);
                    if (e & 64)
                        t_update(J, (e >> 3) & 7, e & 7);
                    else
                        trailing_tile(J, (e >> 3) & 7, e & 7);
                }
                // the 16 columns of L that step J completed go to global memory now, beside wave 0's pivot chain
                // (round 4: as one pass after the last step the store of the factor was 1.5 us at the END of the chain)
                if (MODE != 2) {
                    for (int c = b + hr; c < b + SB && c < nb; c += NHE) {
                        const double* sc = S + c * LDS_LD;
                        double* gc = A + (int64_t)c * lda;
                        if (lane >= c && lane < nb) gc[lane] = sc[lane];
                        if (lane + 64 >= c && lane + 64 < nb) gc[lane + 64] = sc[lane + 64];
                    }
                }
            }
        }
        __syncthreads();
        P2_STAMP(4 + 3 * J);
    }
    // the last 16 columns of the factor -> global (the others went out step by step)
    if (MODE != 2) {
        const int r = (NSB - 1) * SB + (tid & 15), c = (NSB - 1) * SB + (tid >> 4);
        if (tid < SB * SB && c <= r && r < nb) A[r + (int64_t)c * lda] = S[c * LDS_LD + r];
    }
    P2_STAMP(26);
    P2_STAMP(27);
    // only the lower triangles are written: the images are zero filled once when they are allocated
    if (MODE == 1) {  // the eight 16 x 16 diagonal inverses, at their places in both images
        for (int e = tid; e < NSB * SB * SB; e += NT) {
            const int J = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            const int gr = SB * J + rr, gc = SB * J + cc;
            if (cc <= rr && gr < nb) {
                const double wv = Wd[(J * SB + rr) * WD_LD + cc];
                Wcm[gc * NB + gr] = wv;
                Wrm[gr * NB + gc] = wv;
            }
        }
    } else {
        constexpr int CG = NT / NB;
        const int i = tid & (NB - 1), j0 = tid >> 7;
        auto W_at = [&](int r, int c) {
            return (r / SB == c / SB) ? Wd[r * WD_LD + c % SB] : S[r * LDS_LD + c];
        };
        if (i < nb) {
#pragma unroll 8
            for (int q = 0; q < NB / CG; ++q) {
                const int j = j0 + CG * q;
                if (j <= i) Wcm[j * NB + i] = W_at(i, j);            // column-major: r = i (fast), c = j
                if (j >= i && j < nb) Wrm[j * NB + i] = W_at(j, i);  // row-major:    c = i (fast), r = j
            }
        }
    }
    P2_STAMP(28);
}

#ifndef P2_KTHREADS
#define P2_KTHREADS 512  // (1024: loads and stores of the block 2.7 us faster, the steps the same; no gain in the applications)
#endif
template <int MODE>
__global__ __launch_bounds__(P2_KTHREADS) void potf2_inv_kernel(double* __restrict__ A, int64_t lda, int nb,
                                                        double* __restrict__ Wcm,
                                                        double* __restrict__ Wrm,
                                                        int32_t* __restrict__ info, int32_t col0,
                                                        Potf2Batch bt) {
    __shared__ double S[P2_S_DOUBLES];
    __shared__ double Wd[P2_WD_DOUBLES];
    if (bt.list) {
        const int cnt = *bt.count;
        for (int pb = blockIdx.x; pb < cnt; pb += gridDim.x) {
            const int64_t b = bt.list[pb];
            potf2_inv_body<P2_KTHREADS, MODE>(A + b * bt.sA, lda, nb, Wcm + b * bt.sW, Wrm + b * bt.sW, info + b * bt.sInfo, col0, S, Wd);
            __syncthreads();
        }
        return;
    }
    if (gridDim.x > 1 || bt.skip) {
        const int64_t b = blockIdx.x;
        if (bt.skip && bt.skip[b] != 0) return;
        A += b * bt.sA;
        Wcm += b * bt.sW;
        Wrm += b * bt.sW;
        info += b * bt.sInfo;
    }
    potf2_inv_body<P2_KTHREADS, MODE>(A, lda, nb, Wcm, Wrm, info, col0, S, Wd);
}
// The inverse images of the diagonal blocks j0/128 .. of a factored matrix, one workgroup per block, in ONE launch: what
// the factor-only diagonal kernel (MODE 1) leaves out of the serial spine.  A: the matrix (block b at (j0 + 128 b) (lda + 1)).
__global__ __launch_bounds__(P2_KTHREADS) void potf2_invert_kernel(double* __restrict__ A, int64_t lda, int64_t n, int64_t j0,
                                                                   double* __restrict__ winv) {
    __shared__ double S[P2_S_DOUBLES];
    __shared__ double Wd[P2_WD_DOUBLES];
    const int64_t jb = j0 + (int64_t)blockIdx.x * NB;
    const int nb = (int)((n - jb < NB) ? (n - jb) : NB);
    double* Wcm = winv + (jb / NB) * WBLK;
    potf2_inv_body<P2_KTHREADS, 2>(A + jb + jb * lda, lda, nb, Wcm, Wcm + NB * NB, nullptr, 0, S, Wd);
}

// The off-diagonal 16 x 16 sub-blocks of both sweep images of the diagonal blocks j0/128 .. of a factored matrix, one
// workgroup per block, ONE launch per factorisation (see "the diagonal step of a sweep" below): image 1 (column-major)
// gets Lt_IJ = L_IJ W_JJ at (16 I.., 16 J..), image 2 gets Lh_IJ = W_II L_IJ transposed, i.e. at [(16 I + k) NB + 16 J + c]
// -- the places the 128 x 128 inverse and its transpose used to fill; the diagonals (W_JJ, left by the factor-only
// diagonal kernels) stay.  normalise = 0: the sub-blocks of L themselves (plain block substitution, MADQP_SWEEP_DIAG=sub16).
__global__ __launch_bounds__(256) void sweep_image_kernel(const double* __restrict__ A, int64_t lda, int64_t n, int64_t j0,
                                                          double* __restrict__ winv, int normalise) {
    __shared__ double S[P2_S_DOUBLES];   // S[c*LDS_LD + r] = L(r, c), zero beyond the order of a short block
    __shared__ double Wd[P2_WD_DOUBLES]; // W_JJ[r][c] at (J*SB + r)*WD_LD + c
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const int64_t jb = j0 + (int64_t)blockIdx.x * NB;
    const int nb = (int)((n - jb < NB) ? (n - jb) : NB);
    double* Wcm = winv + (jb / NB) * WBLK;
    double* Wrm = Wcm + NB * NB;
    const double* Ab = A + jb + jb * lda;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e & (NB - 1), c = e >> 7;
        const int rr = r < nb ? r : nb - 1, cc = c < nb ? c : nb - 1;  // (unconditional loads from addresses that exist)
        const double v = Ab[rr + (int64_t)cc * lda];
        S[c * LDS_LD + r] = (r < nb && c <= r) ? v : 0.0;
    }
    for (int e = tid; e < NSB * SB * SB; e += 256) {
        const int J = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
        const double v = Wcm[(SB * J + cc) * NB + SB * J + rr];
        Wd[(J * SB + rr) * WD_LD + cc] = (cc <= rr) ? v : 0.0;
    }
    __syncthreads();
    for (int p = wave; p < NSB * (NSB - 1) / 2; p += 4) {
        int I = 1, q = p;  // pair p -> (I, J), I > J
        while (q >= I) {
            q -= I;
            ++I;
        }
        const int J = q;
        double4_t P = {0.0, 0.0, 0.0, 0.0}, Q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = 4 * s4 + hi;
            // P = L_IJ W_JJ: A[m = lo][k] = L_IJ[lo][k], B[k][n = lo] = W_JJ[k][lo]
            const double pa = S[(SB * J + k) * LDS_LD + SB * I + lo];
            const double pb = normalise ? Wd[(J * SB + k) * WD_LD + lo] : (k == lo ? 1.0 : 0.0);
            P = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, pb, P, 0, 0, 0);
            // Q = W_II L_IJ: A[m = lo][k] = W_II[lo][k], B[k][n = lo] = L_IJ[k][lo]
            const double qa = normalise ? Wd[(I * SB + lo) * WD_LD + k] : (k == lo ? 1.0 : 0.0);
            const double qb = S[(SB * J + lo) * LDS_LD + SB * I + k];
            Q = __builtin_amdgcn_mfma_f64_16x16x4f64(qa, qb, Q, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {  // D[m = hi + 4v][n = lo]
            Wcm[(SB * J + lo) * NB + SB * I + hi + 4 * v] = P[v];
            Wrm[(SB * I + hi + 4 * v) * NB + SB * J + lo] = Q[v];
        }
    }
}

// ---- triangular sweeps, one launch each ---------------------------------------------------------
// A sweep over the 128-row blocks is a chain: block r needs the solutions of all blocks before it.
// Instead of one launch per block (391 launches per sweep at n = 50 000, each with its own ramp and
// tail) ONE launch runs a workgroup per block; workgroup r streams its own tiles L[r, j] (forward) /
// L[j, r] (backward) as the x_j appear and then solves its diagonal block with the stored inverse.
//   * order: a workgroup draws its block index from a ticket counter, so it only ever waits for
//     blocks drawn earlier, which are resident or finished -- the grid always drains, whatever the
//     dispatch order and however many workgroups fit on the chip;
//   * hand-off: x_j is published as 128 naturally aligned 8-byte write-through (sc1) stores into a
//     vector pre-filled with a NaN sentinel; consumers poll the values themselves with sc1 loads
//     (data = tag: no flag, no fence; MI355X_MICROARCH.md "handoff-1to1", ~1 us per hop);
//   * streaming: 16 waves x (2 rows per lane) x (4 columns per wave) = half a tile per stage, two
//     stages in flight in registers while the wave waits for the next x_j; the inverse diagonal
//     block sits in registers from the start, so the critical path per block is
//     poll -> 8 fma -> cross-wave sum -> 128 x 128 product -> publish.
// Summation order is fixed by the thread mapping: results do not depend on timing.
constexpr unsigned long long SWEEP_SENTINEL = 0x7FF8A5A57FF8A5A5ull;  // a NaN no arithmetic produces
constexpr int SWEEP_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ unsigned long long ld_sc1_u64(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_f64(double* p, double v) {
    unsigned long long u = __double_as_longlong(v);
    if (v != v) u = 0x7FF8000000000000ull;  // never publish the sentinel pattern
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0: wait until the 128 entries of block j of `x` have been published, copy them to LDS
// (entries at or beyond n are zero).  Returns with the error flag set if the producer never shows up.
__device__ __forceinline__ void sweep_poll_block(const double* __restrict__ x, int64_t j, int64_t n,
                                                 double* __restrict__ dst, double* __restrict__ err,
                                                 int lane) {
    const int64_t i0 = j * NB + 2 * lane;
    unsigned long long u0 = 0, u1 = 0;
    int spins = 0;
    bool wait0 = i0 < n, wait1 = i0 + 1 < n;
    while (wait0 || wait1) {
        if (wait0) {
            u0 = ld_sc1_u64(x + i0);
            wait0 = (u0 == SWEEP_SENTINEL);
        }
        if (wait1) {
            u1 = ld_sc1_u64(x + i0 + 1);
            wait1 = (u1 == SWEEP_SENTINEL);
        }
        if ((wait0 || wait1) && ++spins > SWEEP_SPIN_LIMIT) {
            *err = 1.0;  // the context's fault word: read back with the next scalar result (madqp_read_results)
            u0 = u1 = 0x7FF8000000000000ull;
            break;
        }
    }
    dst[2 * lane] = (i0 < n) ? __longlong_as_double(u0) : 0.0;
    dst[2 * lane + 1] = (i0 + 1 < n) ? __longlong_as_double(u1) : 0.0;
}

// A long block row is streamed by several workgroups (madqp_chol_create builds the job list): job (r, c) takes the
// tiles c*chunk .. min(r, (c+1)*chunk) - 1 of block r (counted from the far end of the dependency chain); the job
// holding the tile next to the diagonal owns the block, the others publish their 128 partial sums to
// part[(r*maxc + c)*128 ..] the way solution blocks are published.  jobs == nullptr: one job per block.
struct SweepPlan {
    const int32_t* jobs;  // ticket -> r | c << 20, ordered by (r, c): a job waits only for earlier tickets
    double* part;
    int32_t chunk, maxc;
};

// one published value (sentinel until its producer has stored it)
__device__ __forceinline__ double sweep_poll_one(const double* p, double* __restrict__ err) {
    unsigned long long u = ld_sc1_u64(p);
    int spins = 0;
    while (u == SWEEP_SENTINEL) {
        if (++spins > SWEEP_SPIN_LIMIT) {
            *err = 1.0;
            return __longlong_as_double(0x7FF8000000000000ull);
        }
        u = ld_sc1_u64(p);
    }
    return __longlong_as_double(u);
}

// The partial sums of a block's other jobs for the lane's two entries, added in job order.  Up to four jobs' words are
// requested at once: one after the other, each an L2 round trip of ~1 us, they were half of a hop at n = 50 000 (six
// jobs per block row: 6.2 us per hop where n = 5 000, one job per row, takes 3.2).
__device__ __forceinline__ void sweep_far2(const double* __restrict__ pp, int c, int lane, double* __restrict__ err,
                                           double& f0, double& f1) {
    f0 = f1 = 0.0;
    for (int c0 = 0; c0 < c; c0 += 4) {
        unsigned long long u[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < c) {
                u[k][0] = ld_sc1_u64(pp + (int64_t)(c0 + k) * NB + lane);
                u[k][1] = ld_sc1_u64(pp + (int64_t)(c0 + k) * NB + 64 + lane);
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < c) {
                const double a = (u[k][0] == SWEEP_SENTINEL) ? sweep_poll_one(pp + (int64_t)(c0 + k) * NB + lane, err)
                                                             : __longlong_as_double(u[k][0]);
                const double b = (u[k][1] == SWEEP_SENTINEL) ? sweep_poll_one(pp + (int64_t)(c0 + k) * NB + 64 + lane, err)
                                                             : __longlong_as_double(u[k][1]);
                f0 += a;
                f1 += b;
            }
    }
}

// half a 128 x 128 column-major tile: rows 2*lane, 2*lane+1 and the 4 columns of this wave
__device__ __forceinline__ void sweep_load_half(const double* __restrict__ base, int64_t ld, bool ok0,
                                                bool ok1, bool vec, double2_t (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double* p = base + (int64_t)q * ld;
        if (vec && ok1) {
            dst[q] = *reinterpret_cast<const double2_t*>(p);
        } else {
            dst[q].x = ok0 ? p[0] : 0.0;
            dst[q].y = ok1 ? p[1] : 0.0;
        }
    }
}

// One product with a 128 x 128 block image held in registers (w0, w1 = the two column halves of this thread's two
// rows): the 128 results, summed over the 16 waves in wave order, are returned to the threads tid < 128 (others: 0).
// Ends with the partial sums in `red`: the caller synchronises before `red` is written again.
__device__ __forceinline__ double sweep_block_product(const double2_t (&w0)[4], const double2_t (&w1)[4],
                                                      const double* __restrict__ vs, double (*red)[NB], int tid, int lane,
                                                      int wave) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double v0 = vs[wave * 4 + q], v1 = vs[64 + wave * 4 + q];
        a0 = __builtin_fma(w0[q].x, v0, a0);
        a1 = __builtin_fma(w0[q].y, v0, a1);
        a0 = __builtin_fma(w1[q].x, v1, a0);
        a1 = __builtin_fma(w1[q].y, v1, a1);
    }
    red[wave][2 * lane] = a0;
    red[wave][2 * lane + 1] = a1;
    __syncthreads();
    double sum = 0.0;
    if (tid < NB) {
        sum = red[0][tid];
#pragma unroll
        for (int q = 1; q < 16; ++q) sum += red[q][tid];
    }
    return sum;
}
// ---- the diagonal step of a sweep by UNIT BLOCK SUBSTITUTION over the 16 x 16 sub-blocks (round 5) -----------------------
// A product with the stored 128 x 128 inverse leaves a residual of cond(L_rr) eps where substitution leaves eps, and the
// interior-point iterates see the residual: on ill-conditioned problems the per-iteration traces sat up to 80 x the CPU
// noise floor from LAPACK's (profiles/r04_parity_ratios_default.json; tools/numerics/blockchol_emul.py: it is the product
// with ANY stored 128-inverse, while block substitution over 16 x 16 sub-blocks is statistically as good as dtrsv).  The
// sweeps therefore substitute: with W_JJ = (L_rr)_JJ^-1 (16 x 16, left on the images' diagonals by the factor-only
// diagonal kernel) and the off-diagonal sub-blocks normalised ONCE per factorisation (sweep_image_kernel),
//     forward   Lt_IJ = L_IJ W_JJ :   u_I = v_I - sum_{J<I} Lt_IJ u_J,    z_I = W_II u_I
//     backward  Lh_IJ = W_II L_IJ :   w_J = v_J - sum_{I>J} Lh_IJ' w_I,   x_J = W_JJ' w_J
// (the same error bound as z_I = W_II (v_I - sum L_IJ z_J): the 16 x 16 inverses enter once per term either way), so
// that ONE broadcast of the 16 values u_J serves both the rows of sub-block J (-> z_J) and every row below (-> update):
// eight steps of 32 read-lanes + 32 fused multiply-adds on ONE wave, no barrier, no LDS round trip on the chain.
// DIAG = 2 keeps plain block substitution (images hold L_IJ itself; a second broadcast per step) for A/B runs,
// DIAG = 0 the product with the 128 x 128 inverse (MADQP_SWEEP_DIAG=inv, and whenever the images hold full inverses).
// The image of a block sits in LDS packed by columns: forward column c (sub-block J) holds rows 16 J .. 127, backward
// column q (sub-block I) rows 0 .. 16 I + 15; a lane reads consecutive rows of one column: conflict free.
constexpr int XIMG_DOUBLES = 16 * (8 * NB - 16 * 28);                                               // 9 216
__host__ __device__ constexpr int ximg_f_col(int J) { return 16 * (NB * J - 8 * J * (J - 1)); }      // first column of sub-block J
__host__ __device__ constexpr int ximg_b_col(int I) { return NB * I * (I + 1); }
static_assert(ximg_f_col(8) == XIMG_DOUBLES && ximg_b_col(8) == XIMG_DOUBLES, "packed image size");

// the column-major 128 x 128 image G (global) -> packed LDS image; all 1024 threads, 16 loads each in flight at once
// (load and store are separate calls so that the first tile loads of the sweep can be issued between them)
__device__ __forceinline__ void ximg_load(const double* __restrict__ G, int tid, double (&v)[16]) {
    const int r = tid & (NB - 1), cg = tid >> 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = G[(cg + 8 * i) * NB + r];
}
template <bool BWD>
__device__ __forceinline__ void ximg_store(const double (&v)[16], double* __restrict__ X, int tid) {
    const int r = tid & (NB - 1), cg = tid >> 7;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int B = i >> 1, ck = cg + 8 * (i & 1);  // sub-block and column inside it
        if (!BWD) {
            if (r >= 16 * B) X[ximg_f_col(B) + ck * (NB - 16 * B) + r - 16 * B] = v[i];
        } else {
            if (r < 16 * (B + 1)) X[ximg_b_col(B) + ck * 16 * (B + 1) + r] = v[i];
        }
    }
}

// Eight of the 16 values of sub-block g (lanes 16 g + 8 h .. + 7 of `src`) to every lane, as scalar operands: 16 read-lanes.
// (Through LDS instead -- one store of all 64 lanes, four 16-byte reads of one address each -- measured the same time
// per hop and costs registers: dropped.)
__device__ __forceinline__ void sweep_bcast8(double src, int g, int h, double (&b)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = readlane_f64(src, 16 * g + 8 * h + k);
}
// t0 = sum_k x0[k] b[k], t1 = sum_k x1[k] b[k] over the 16 columns of one sub-block (two partial sums each, fixed order);
// column k of the packed image at X + k * len (+ 64 for the second row of the lane); HAS0 / HAS1: which of the lane's two
// rows take part (compile time)
template <int DIAG, bool HAS0, bool HAS1>
__device__ __forceinline__ void sweep_dot16(const double* __restrict__ X, int len, double src, int g, int lane,
                                            double& t0, double& t1) {
    double t0a = 0.0, t0b = 0.0, t1a = 0.0, t1b = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double b[8], x0[8], x1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (HAS0) x0[k] = X[(8 * h + k) * len + lane];
            if (HAS1) x1[k] = X[(8 * h + k) * len + 64 + lane];
        }
        sweep_bcast8(src, g, h, b);
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            if (HAS0) {
                t0a = __builtin_fma(x0[k], b[k], t0a);
                t0b = __builtin_fma(x0[k + 1], b[k + 1], t0b);
            }
            if (HAS1) {
                t1a = __builtin_fma(x1[k], b[k], t1a);
                t1b = __builtin_fma(x1[k + 1], b[k + 1], t1b);
            }
        }
    }
    t0 = t0a + t0b;
    t1 = t1a + t1b;
}
// wave 0: the 128 values of the diagonal step; lane l holds rows l (u0) and 64 + l (u1) on entry (v) and on return (z)
template <int DIAG>
__device__ __forceinline__ double2_t sweep_diag_fwd(const double* __restrict__ X, int lane, double u0, double u1) {
    // (inlined, behind a compiler barrier: as a call the callee saves 48 registers to scratch on the chain; without the
    // barrier the image reads drift up into the streaming loop and its registers spill)
    asm volatile("" ::: "memory");
    const int I0 = lane >> 4;  // sub-block of row `lane`; row 64 + lane sits in sub-block 4 + I0
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const int len = NB - 16 * J;
        const double* Xj = X + ximg_f_col(J) - 16 * J;  // (lanes above sub-block J read a neighbour's entries: discarded)
        const double src = (J < 4) ? u0 : u1;
        double t0 = 0.0, t1 = 0.0;
        if (J < 4)
            sweep_dot16<DIAG, true, true>(Xj, len, src, J & 3, lane, t0, t1);
        else
            sweep_dot16<DIAG, false, true>(Xj, len, src, J & 3, lane, t0, t1);
        if (DIAG == 2) {  // plain block substitution: the rows below take z_J, not u_J
            const double zsrc = (J < 4) ? t0 : t1;
            double s0 = 0.0, s1 = 0.0;
            if (J < 4)
                sweep_dot16<DIAG, true, true>(Xj, len, zsrc, J & 3, lane, s0, s1);
            else
                sweep_dot16<DIAG, false, true>(Xj, len, zsrc, J & 3, lane, s0, s1);
            if (J < 4) t0 = (I0 == J) ? t0 : s0;
            t1 = (4 + I0 == J) ? t1 : s1;
        }
        if (J < 4) {
            u0 = (I0 == J) ? t0 : (I0 > J) ? u0 - t0 : u0;
            u1 = u1 - t1;
        } else {
            u1 = (4 + I0 == J) ? t1 : (4 + I0 > J) ? u1 - t1 : u1;
        }
    }
    return double2_t{u0, u1};
}
template <int DIAG>
__device__ __forceinline__ double2_t sweep_diag_bwd(const double* __restrict__ X, int lane, double u0, double u1) {
    // (inlined, behind a compiler barrier: as a call the callee saves 48 registers to scratch on the chain; without the
    // barrier the image reads drift up into the streaming loop and its registers spill)
    asm volatile("" ::: "memory");
    const int J0 = lane >> 4;  // sub-block of entry `lane`; entry 64 + lane: 4 + J0
#pragma unroll
    for (int I = 7; I >= 0; --I) {
        const int len = 16 * (I + 1);
        const double* Xi = X + ximg_b_col(I);  // (entries beyond sub-block I read a neighbour's column: discarded)
        const double src = (I < 4) ? u0 : u1;
        double t0 = 0.0, t1 = 0.0;
        if (I >= 4)
            sweep_dot16<DIAG, true, true>(Xi, len, src, I & 3, lane, t0, t1);
        else
            sweep_dot16<DIAG, true, false>(Xi, len, src, I & 3, lane, t0, t1);
        if (DIAG == 2) {
            const double zsrc = (I < 4) ? t0 : t1;
            double s0 = 0.0, s1 = 0.0;
            if (I >= 4)
                sweep_dot16<DIAG, true, true>(Xi, len, zsrc, I & 3, lane, s0, s1);
            else
                sweep_dot16<DIAG, true, false>(Xi, len, zsrc, I & 3, lane, s0, s1);
            t0 = (J0 == I) ? t0 : s0;
            if (I >= 4) t1 = (4 + J0 == I) ? t1 : s1;
        }
        if (I >= 4) {
            u1 = (4 + J0 == I) ? t1 : (4 + J0 < I) ? u1 - t1 : u1;
            u0 = u0 - t0;
        } else {
            u0 = (J0 == I) ? t0 : (J0 < I) ? u0 - t0 : u0;
        }
    }
    return double2_t{u0, u1};
}

__global__ __launch_bounds__(256) void negate_kernel(double* __restrict__ v, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) v[i] = -v[i];
}

// REV (DIAG = 1 only): the same sweep on U = L' from the far end -- the backward substitution x = L^-T y as sums over the
// ROWS of U's blocks: block r of the chain is block nblk-1-r of the matrix, its sources lie to the right of the diagonal,
// the diagonal step is the backward one on the block's second image.  The mid-size factorisation leaves U in a buffer of
// its own (lower_to_upper_kernel); two accumulators per thread where the transposed product on L (trsv_bwd_sweep_kernel)
// needs sixteen and a column reduction through LDS per block: 3.3 us per hop instead of 4.3.  out: optional second copy
// of the solution (plain stores; the polled copy y may then be scratch memory).
template <int DIAG, bool REV = false>
__global__ __launch_bounds__(1024) void trsv_fwd_sweep_kernel(const double* __restrict__ L, int64_t lda,
                                                              const double* __restrict__ winv,
                                                              const double* __restrict__ b,
                                                              double* __restrict__ y, int64_t n,
                                                              int32_t* __restrict__ ctl, double* __restrict__ fault,
                                                              int vec, SweepPlan plan, double* __restrict__ out) {
    static_assert(!REV || DIAG == 1, "the sweep on U runs the substituting diagonal step");
    __shared__ double xs[2][NB];
    __shared__ double red[16][NB];
    __shared__ double vs[NB];
    __shared__ double ximg[DIAG ? XIMG_DOUBLES : 1];
    __shared__ int s_r;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_r = atomicAdd(&ctl[REV ? 2 : 1], 1);
    __syncthreads();
    int r = s_r, c = 0, j0 = 0, j1 = r;
    if (plan.jobs) {
        const int32_t job = plan.jobs[r];
        r = job & 0xFFFFF;
        c = job >> 20;
        j0 = c * plan.chunk;
        j1 = (j0 + plan.chunk < r) ? j0 + plan.chunk : r;
    }
    const bool owner = (j1 == r);
    const int rl = r;  // the block's place in the chain (partial sums are filed under it)
    const int nblk = (int)((n + NB - 1) / NB);
    if (REV) r = nblk - 1 - r;  // from here on r is the block of the matrix
    const int64_t row0 = (int64_t)r * NB;
    const int w = (int)((n - row0 < NB) ? (n - row0) : NB);
    const bool ok0 = row0 + 2 * lane < n, ok1 = row0 + 2 * lane + 1 < n;
    // DIAG = 0: the inverse diagonal block (column-major image, zero padded, 16-byte aligned) in registers;
    // else: this block's forward image (16 x 16 inverses on the diagonal, normalised sub-blocks below) packed into LDS
    double2_t W0[4], W1[4];
    double xv[16];
    if (owner) {
        const double* Wcm = winv + (int64_t)r * WBLK + (REV ? NB * NB : 0);
        if (DIAG == 0) {
            sweep_load_half(Wcm + 2 * lane + (int64_t)(wave * 4) * NB, NB, true, true, true, W0);
            sweep_load_half(Wcm + 2 * lane + (int64_t)(64 + wave * 4) * NB, NB, true, true, true, W1);
        } else {
            ximg_load(Wcm, tid, xv);
        }
    }
    double a0 = 0.0, a1 = 0.0;
    const double* Lr = L + row0 + 2 * lane + (int64_t)(wave * 4) * lda;  // this thread's corner of tile (r, 0)
    auto col = [&](int j) { return (int64_t)(REV ? nblk - 1 - j : j) * NB * lda; };  // tile column of source j of the chain
    double2_t A[4], B[4];
    if (j1 > j0) sweep_load_half(Lr + col(j0), lda, ok0, ok1, vec, A);
    if (DIAG != 0 && owner) ximg_store<REV>(xv, ximg, tid);  // (visible to wave 0 after any later barrier)
    // the right-hand side of this block, fetched NOW: read behind the last barrier it is an L2 round trip on every hop
    double b0 = 0.0, b1 = 0.0;
    if (owner) {
        const int i0 = DIAG != 0 ? lane : (tid & (NB - 1)), i1 = 64 + lane;
        b0 = i0 < w ? b[row0 + i0] : 0.0;
        if (DIAG != 0) b1 = i1 < w ? b[row0 + i1] : 0.0;
    }
    for (int j = j0; j < j1; ++j) {
        const double* Tj = Lr + col(j);
        sweep_load_half(Tj + 64 * lda, lda, ok0, ok1, vec, B);
        if (wave == 0) sweep_poll_block(y, REV ? nblk - 1 - j : j, n, xs[j & 1], fault, lane);
        __syncthreads();
        const double* xj = xs[j & 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double xv = xj[wave * 4 + q];
            a0 = __builtin_fma(A[q].x, xv, a0);
            a1 = __builtin_fma(A[q].y, xv, a1);
        }
        if (j + 1 < j1) sweep_load_half(Lr + col(j + 1), lda, ok0, ok1, vec, A);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double xv = xj[64 + wave * 4 + q];
            a0 = __builtin_fma(B[q].x, xv, a0);
            a1 = __builtin_fma(B[q].y, xv, a1);
        }
    }
    red[wave][2 * lane] = a0;
    red[wave][2 * lane + 1] = a1;
    __syncthreads();
    if (DIAG == 0 || !owner) {
        if (tid < NB) {
            double sum = red[0][tid];
#pragma unroll
            for (int q = 1; q < 16; ++q) sum += red[q][tid];
            if (!owner) {
                st_sc1_f64(plan.part + ((int64_t)rl * plan.maxc + c) * NB + tid, sum);
            } else {
                double far = 0.0;  // the partial sums of this block's other jobs, in column order
                for (int cc = 0; cc < c; ++cc)
                    far += sweep_poll_one(plan.part + ((int64_t)rl * plan.maxc + cc) * NB + tid, fault);
                vs[tid] = (tid < w) ? (b0 - (far + sum)) : 0.0;
            }
        }
        if (!owner) return;
        __syncthreads();
        const double z0 = sweep_block_product(W0, W1, vs, red, tid, lane, wave);  // z = W v
        if (tid < w) st_sc1_f64(y + row0 + tid, z0);
        if (out && tid < w) out[row0 + tid] = z0;
        return;
    }
    if (wave != 0) return;  // (the image was complete before the first barrier of this workgroup)
    // wave 0 alone from here: the sums over the 16 waves for its two rows (the same order as above), no second barrier
    double s0 = red[0][lane], s1 = red[0][64 + lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        s0 += red[q][lane];
        s1 += red[q][64 + lane];
    }
    double far0, far1;  // the partial sums of this block's other jobs, in column order
    sweep_far2(plan.part + (int64_t)rl * plan.maxc * NB, c, lane, fault, far0, far1);
    const double u0 = lane < w ? b0 - (far0 + s0) : 0.0, u1 = 64 + lane < w ? b1 - (far1 + s1) : 0.0;
    const double2_t z = REV ? sweep_diag_bwd<DIAG>(ximg, lane, u0, u1) : sweep_diag_fwd<DIAG>(ximg, lane, u0, u1);
    if (lane < w) st_sc1_f64(y + row0 + lane, z.x);
    if (64 + lane < w) st_sc1_f64(y + row0 + 64 + lane, z.y);
    if (out) {
        if (lane < w) out[row0 + lane] = z.x;
        if (64 + lane < w) out[row0 + 64 + lane] = z.y;
    }
}

// U = L' for the tiles strictly below the diagonal blocks (64 x 64 pieces through LDS; rows of L at or beyond n give zeros:
// the sweep on U multiplies those columns with zeros and must not meet a NaN there)
__global__ __launch_bounds__(256) void lower_to_upper_kernel(const double* __restrict__ A, int64_t lda, int64_t n,
                                                            double* __restrict__ U, int64_t ldu) {
    const int i = blockIdx.x, j = blockIdx.y;  // 64-blocks: source rows i, source columns j
    if ((i >> 1) <= (j >> 1)) return;
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r = (int64_t)i * 64 + tx;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = ty + 4 * q;
        tile[c][tx] = r < n ? A[r + ((int64_t)j * 64 + c) * lda] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int rr = ty + 4 * q;  // source row inside the piece = destination column
        U[(int64_t)j * 64 + tx + ((int64_t)i * 64 + rr) * ldu] = tile[tx][rr];
    }
}
// the scratch vectors of a solve <- sentinel, the sweeps' ticket counters <- 0: one launch per solve instead of three fills
__global__ __launch_bounds__(256) void sweep_prep_kernel(unsigned long long* __restrict__ v, int64_t len, int32_t* __restrict__ ctl) {
    if (blockIdx.x == 0 && threadIdx.x < 3) ctl[1 + threadIdx.x] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) v[i] = SWEEP_SENTINEL;
}

template <int DIAG>
__global__ __launch_bounds__(1024) void trsv_bwd_sweep_kernel(const double* __restrict__ L, int64_t lda,
                                                              const double* __restrict__ winv,
                                                              const double* __restrict__ y,
                                                              double* __restrict__ x, int64_t n,
                                                              int32_t* __restrict__ ctl, double* __restrict__ fault,
                                                              int vec, SweepPlan plan) {
    __shared__ double xs[2][NB];
    __shared__ double red[DIAG ? 1 : 16][NB];
    __shared__ double vs[NB];
    __shared__ double colred[NB][65];  // column-sum staging, padded against bank conflicts
    __shared__ double ximg[DIAG ? XIMG_DOUBLES : 1];
    __shared__ int s_r;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = (int)((n + NB - 1) / NB);
    if (tid == 0) s_r = atomicAdd(&ctl[2], 1);
    __syncthreads();
    // block nblk-1-rr has rr tiles below its diagonal block; the job takes tiles k0 .. k1-1 counted from the last row
    int rr = s_r, c = 0, k0 = 0, k1 = rr;
    if (plan.jobs) {
        const int32_t job = plan.jobs[rr];
        rr = job & 0xFFFFF;
        c = job >> 20;
        k0 = c * plan.chunk;
        k1 = (k0 + plan.chunk < rr) ? k0 + plan.chunk : rr;
    }
    const bool owner = (k1 == rr);
    const int r = nblk - 1 - rr;
    const int64_t col0 = (int64_t)r * NB;
    const int w = (int)((n - col0 < NB) ? (n - col0) : NB);
    // DIAG = 0: W' v through the row-major image (Wrm[c + r*NB] = W(r, c): output index c is the fast one), in registers;
    // else: this block's backward image packed into LDS
    double2_t W0[4], W1[4];
    double xv[16];
    if (owner) {
        const double* Wrm = winv + (int64_t)r * WBLK + NB * NB;
        if (DIAG == 0) {
            sweep_load_half(Wrm + 2 * lane + (int64_t)(wave * 4) * NB, NB, true, true, true, W0);
            sweep_load_half(Wrm + 2 * lane + (int64_t)(64 + wave * 4) * NB, NB, true, true, true, W1);
        } else {
            ximg_load(Wrm, tid, xv);
        }
    }
    // this thread's columns: col0 + h*64 + wave*4 + q; rows 2*lane, 2*lane+1 of the row block j
    double acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[h][q] = 0.0;
    const double* Lc = L + 2 * lane + (col0 + wave * 4) * lda;  // + row block offset
    double2_t A[4], B[4];
    const int jhi = nblk - 1 - k0, jlo = nblk - 1 - k1;  // row blocks jhi, jhi-1, .., jlo+1
    auto rows_ok = [&](int j, bool& o0, bool& o1) {
        const int64_t i0 = (int64_t)j * NB + 2 * lane;
        o0 = i0 < n;
        o1 = i0 + 1 < n;
    };
    // (the columns of block r all exist whenever it has tiles: only the last block can be short)
    if (jhi > jlo) {
        bool o0, o1;
        rows_ok(jhi, o0, o1);
        sweep_load_half(Lc + (int64_t)jhi * NB, lda, o0, o1, vec, A);
    }
    if (DIAG != 0 && owner) ximg_store<true>(xv, ximg, tid);
    double y0 = 0.0, y1 = 0.0;  // this block of the right-hand side, fetched now (see the forward kernel)
    if (owner) {
        const int i0 = DIAG != 0 ? lane : (tid & (NB - 1)), i1 = 64 + lane;
        y0 = i0 < w ? y[col0 + i0] : 0.0;
        if (DIAG != 0) y1 = i1 < w ? y[col0 + i1] : 0.0;
    }
    for (int j = jhi; j > jlo; --j) {
        bool o0, o1;
        rows_ok(j, o0, o1);
        const double* Tj = Lc + (int64_t)j * NB;
        sweep_load_half(Tj + 64 * lda, lda, o0, o1, vec, B);
        if (wave == 0) sweep_poll_block(x, j, n, xs[j & 1], fault, lane);
        __syncthreads();
        const double x0 = xs[j & 1][2 * lane], x1 = xs[j & 1][2 * lane + 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][q] = __builtin_fma(A[q].y, x1, __builtin_fma(A[q].x, x0, acc[0][q]));
        if (j - 1 > jlo) {
            bool p0, p1;
            rows_ok(j - 1, p0, p1);
            sweep_load_half(Tj - NB, lda, p0, p1, vec, A);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[1][q] = __builtin_fma(B[q].y, x1, __builtin_fma(B[q].x, x0, acc[1][q]));
    }
    // column sums over the 64 lanes, then v = y_r - sums.  Through LDS (a 64-lane shuffle butterfly per column put
    // 48 dependent cross-lane operations on the hand-off chain: the backward hop was 6.1 us against 2.6 us forward):
    // every thread parks its 8 partial sums, then 8 threads per column add 8 lanes each and finish with three
    // steps inside their group of 8 lanes.  Fixed order: the same result on every run.
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) colred[h * 64 + wave * 4 + q][lane] = acc[h][q];
    __syncthreads();
    {
        const int cl = tid >> 3, part = tid & 7;
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += colred[cl][part * 8 + u];
        t += __shfl_down(t, 4, 8);
        t += __shfl_down(t, 2, 8);
        t += __shfl_down(t, 1, 8);
        if (part == 0) vs[cl] = t;
    }
    __syncthreads();
    if (!owner) {
        if (tid < NB) st_sc1_f64(plan.part + ((int64_t)rr * plan.maxc + c) * NB + tid, vs[tid]);
        return;
    }
    if (DIAG == 0) {
        if (tid < NB) {
            double far = 0.0;
            for (int cc = 0; cc < c; ++cc)
                far += sweep_poll_one(plan.part + ((int64_t)rr * plan.maxc + cc) * NB + tid, fault);
            vs[tid] = (tid < w) ? (y0 - (far + vs[tid])) : 0.0;
        }
        __syncthreads();
        const double x0 = sweep_block_product(W0, W1, vs, red, tid, lane, wave);  // x = W' v
        if (tid < w) st_sc1_f64(x + col0 + tid, x0);
        return;
    }
    if (wave != 0) return;  // wave 0 alone from here, no further barrier
    double far0, far1;
    sweep_far2(plan.part + (int64_t)rr * plan.maxc * NB, c, lane, fault, far0, far1);
    const double u0 = lane < w ? y0 - (far0 + vs[lane]) : 0.0, u1 = 64 + lane < w ? y1 - (far1 + vs[64 + lane]) : 0.0;
    const double2_t z = sweep_diag_bwd<DIAG>(ximg, lane, u0, u1);
    if (lane < w) st_sc1_f64(x + col0 + lane, z.x);
    if (64 + lane < w) st_sc1_f64(x + col0 + 64 + lane, z.y);
}

// ---- panel times inverse for ONE 128-column block: L[rows, :] = C[rows, :] W'  (W = inverse of the block's factor) ----
// The GEMM kernel does this product with a 128 x 128 tile per workgroup: at most n/128 workgroups, one round, 14 us of
// MFMA per tile behind a staged prologue -- 23 us per launch at n = 5 000, forty times per factorisation.  Here a workgroup
// takes 32 ROWS (four times as many workgroups; in place is safe: it reads only the rows it writes), every operand goes
// straight from L2 to registers in the MFMA operand layout with all loads of a wave in flight at once, and the zero half
// of the lower-triangular W is skipped: column tile jt (16 columns) meets nonzeros of W only in k < 16 (jt + 1), so a
// wave that takes the tiles w and 7 - w runs 36 of the 64 k-steps.  D[j][i] = sum_k W(j, k) C(i, k): A-operand <- W
// (Wcm[j + k*128]), B-operand <- C rows (C[i + k*ld]); lane l holds D[(l>>4) + 4v][l & 15] -> stores walk i.
__global__ __launch_bounds__(256) void panel_inv_kernel(double* __restrict__ C, int64_t ld, const double* __restrict__ Wcm,
                                                        int64_t rows) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * 32;  // (rows up to the padded order exist: the caller's leading dimension)
    const int jt0 = wave, jt1 = 7 - wave;          // this wave's column tiles; jt0 < jt1
    const int ns0 = 4 * (jt0 + 1), ns1 = 4 * (jt1 + 1);  // k-steps (of 4) that meet nonzeros of W
    const double* Cp = C + i0 + lo + (int64_t)hi * ld;
    const double* W0 = Wcm + 16 * jt0 + lo + hi * NB;
    const double* W1 = Wcm + 16 * jt1 + lo + hi * NB;
    double b[32][2], a1[32], a0[16];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q < ns1) {  // wave-uniform
            b[q][0] = Cp[(int64_t)(4 * q) * ld];
            b[q][1] = Cp[(int64_t)(4 * q) * ld + 16];
            a1[q] = W1[4 * q * NB];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (q < ns0) a0[q] = W0[4 * q * NB];
    double4_t d0[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, d1[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q < ns1) {
            d1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b[q][0], d1[0], 0, 0, 0);
            d1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b[q][1], d1[1], 0, 0, 0);
        }
        if (q < 16 && q < ns0) {
            d0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b[q][0], d0[0], 0, 0, 0);
            d0[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b[q][1], d0[1], 0, 0, 0);
        }
    }
    // every load of the workgroup has been consumed before the first store (all four waves read all 32 rows)
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int64_t gi = i0 + 16 * it + lo;
        if (gi < rows) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                C[gi + (int64_t)(16 * jt0 + hi + 4 * v) * ld] = d0[it][v];
                C[gi + (int64_t)(16 * jt1 + hi + 4 * v) * ld] = d1[it][v];
            }
        }
    }
}

// ---- panel solve for ONE 128-column block by BLOCK SUBSTITUTION: L[rows, blk] = C[rows, blk] L_kk^-T --------------------
// Multiplying the panel by the explicit inverse of the 128 x 128 diagonal block (panel_inv_kernel above, rounds 1-3) costs
// cond(L_kk) in the backward error of the factorisation where LAPACK's dtrsm substitutes (tools/numerics/
// blockchol_emul.py: on condensed LPs the device's distance from LAPACK has a tail of tens of times the distance between
// two CPU executions, and neither a correctly rounded inverse nor a Newton-Schulz step removes it -- it is the product
// with ANY stored 128-inverse).  Here the solve runs over the eight 16-column sub-blocks, and only the inverses of the
// 16 x 16 DIAGONAL sub-blocks are multiplied with (statistically indistinguishable from scalar substitution in the same
// emulation).  With Z = X' (128 x rows):
//     Z_J = W_JJ (C'_J - sum_{I<J} L_JI Z_I),      W_JJ = (L_kk)_JJ^-1 = the diagonal 16 x 16 block of the stored inverse
// on the MFMA unit, entirely in registers: D[m][n] += A[m][k] B[k][n] with m = row inside sub-block J, n = one of 16
// matrix rows, k = index inside sub-block I; the accumulator layout (lane l, register v: m = (l>>4) + 4v, n = l&15) IS the
// B-operand layout of k-step v, so a finished Z_I feeds the updates of the later sub-blocks as it stands.  A-operands
// (tiles of L_kk, W_JJ) come straight from L2, two steps ahead.  The same 144 MFMAs per 16 rows as the inverse product
// with its zero half skipped.  A wave owns 16 rows; in place is safe (it reads only the rows it writes, all up front).
struct PanelBatch {  // problem blockIdx.y: pointer strides (doubles); skip[b] != 0: leave untouched
    int64_t sC, sL, sW;
    const int32_t* skip;
    const int32_t* list = nullptr;   // compacted form (GemmBatch::list)
    const int32_t* count = nullptr;
};
template <int J>
__device__ __forceinline__ void ps16_load(const double* __restrict__ Lq, int64_t ldl, const double* __restrict__ Wq,
                                          double (&aw)[4], double (&al)[7][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) aw[s] = -Wq[16 * J + (16 * J + 4 * s) * NB];  // -W_JJ: Z_J = (-W_JJ) u_J with u = -(C' - ..)
#pragma unroll
    for (int d = 0; d < 7; ++d)
        if (J + 1 + d < 8) {
#pragma unroll
            for (int s = 0; s < 4; ++s) al[d][s] = Lq[16 * (J + 1 + d) + (int64_t)(16 * J + 4 * s) * ldl];
        }
}
template <int J>
__device__ __forceinline__ void ps16_step(double4_t (&z)[8], const double (&aw)[4], const double (&al)[7][4]) {
    double4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) t = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[s], z[J][s], t, 0, 0, 0);
    z[J] = t;  // Z_J
#pragma unroll
    for (int d = 0; d < 7; ++d)
        if (J + 1 + d < 8) {
#pragma unroll
            for (int s = 0; s < 4; ++s)  // u_J2 += L_{J2,J} Z_J
                z[J + 1 + d] = __builtin_amdgcn_mfma_f64_16x16x4f64(al[d][s], t[s], z[J + 1 + d], 0, 0, 0);
        }
}
__device__ __forceinline__ void panel_sub16_body(double* __restrict__ C, int64_t ld, const double* __restrict__ Lkk,
                                                 int64_t ldl, const double* __restrict__ Wcm, int64_t rows, int64_t rows_read) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * 64 + wave * 16;
    if (i0 >= rows) return;  // (no barrier in this kernel)
    const int64_t gi = i0 + lo;
    const int64_t gir = gi < rows_read ? gi : rows_read - 1;  // rows that may be read: up to the padded order when it exists
    double* Cp = C + gir + (int64_t)hi * ld;
    const double* Lq = Lkk + lo + (int64_t)hi * ldl;
    const double* Wq = Wcm + lo + hi * NB;
    double aw[3][4], al[3][7][4];
    ps16_load<0>(Lq, ldl, Wq, aw[0], al[0]);
    ps16_load<1>(Lq, ldl, Wq, aw[1], al[1]);
    double4_t z[8];  // u_J = -(C'_J - ..) until step J, Z_J afterwards: element (16J + hi + 4v, row gi)
#pragma unroll
    for (int J = 0; J < 8; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) z[J][v] = -Cp[(int64_t)(16 * J + 4 * v) * ld];
    ps16_load<2>(Lq, ldl, Wq, aw[2], al[2]);
    ps16_step<0>(z, aw[0], al[0]);
    ps16_load<3>(Lq, ldl, Wq, aw[0], al[0]);
    ps16_step<1>(z, aw[1], al[1]);
    ps16_load<4>(Lq, ldl, Wq, aw[1], al[1]);
    ps16_step<2>(z, aw[2], al[2]);
    ps16_load<5>(Lq, ldl, Wq, aw[2], al[2]);
    ps16_step<3>(z, aw[0], al[0]);
    ps16_load<6>(Lq, ldl, Wq, aw[0], al[0]);
    ps16_step<4>(z, aw[1], al[1]);
    ps16_load<7>(Lq, ldl, Wq, aw[1], al[1]);
    ps16_step<5>(z, aw[2], al[2]);
    ps16_step<6>(z, aw[0], al[0]);
    ps16_step<7>(z, aw[1], al[1]);
    if (gi < rows) {
        double* Co = C + gi + (int64_t)hi * ld;
#pragma unroll
        for (int J = 0; J < 8; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) Co[(int64_t)(16 * J + 4 * v) * ld] = z[J][v];
    }
}

__global__ __launch_bounds__(256) void panel_sub16_kernel(double* __restrict__ C, int64_t ld, const double* __restrict__ Lkk,
                                                          int64_t ldl, const double* __restrict__ Wcm, int64_t rows,
                                                          int64_t rows_read, PanelBatch bt) {
    if (bt.list) {  // compacted batch: the problems of slot blockIdx.y, one after the other (no barrier in the body)
        const int cnt = *bt.count;
        for (int pb = blockIdx.y; pb < cnt; pb += gridDim.y) {
            const int64_t b = bt.list[pb];
            panel_sub16_body(C + b * bt.sC, ld, Lkk + b * bt.sL, ldl, Wcm + b * bt.sW, rows, rows_read);
        }
        return;
    }
    if (gridDim.y > 1 || bt.skip) {
        const int64_t b = blockIdx.y;
        if (bt.skip && bt.skip[b] != 0) return;
        C += b * bt.sC;
        Lkk += b * bt.sL;
        Wcm += b * bt.sW;
    }
    panel_sub16_body(C, ld, Lkk, ldl, Wcm, rows, rows_read);
}
// L[rows below, block] = C[..] L_kk^-T for the 128-column block whose factored diagonal block is Lkk (ldl) and whose inverse
// image is Wcm; `rows_read`: rows of C that exist in memory (>= rows); B > 1: the same for B problems at fixed strides
static int32_t panel_solve_sub16(madqp_ctx* ctx, double* C, int64_t ld, const double* Lkk, int64_t ldl, const double* Wcm,
                                 int64_t rows, int64_t rows_read, int64_t B = 1, PanelBatch bt = PanelBatch{0, 0, 0, nullptr}) {
    if (rows <= 0) return MADQP_OK;
    ARG_TRY(ctx, B >= 1 && B <= 65535 && rows_read >= rows);
    ProfScope ps(ctx, MADQP_PROF_POTRF_TRSM);
    hipLaunchKernelGGL(panel_sub16_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)B), dim3(256), 0, ctx->stream, C, ld,
                       Lkk, ldl, Wcm, rows, rows_read, bt);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

// ---- mid-size factorisation: right-looking, two launches per 128-column block ---------------------------------
// Below n ~ 10 000 the left-looking schedule above is a chain of short dependent launches (per block: update with
// split-K, its reduction, the diagonal kernel, panel times inverse -- 120 us, of which the matrix pipes are busy a
// fraction).  Here step k is
//   chol_mid_step_kernel  every tile (i, j), k <= j <= i, receives the update of panel k-1: C_ij -= L_i,k-1 L_j,k-1'
//                         (K = 128 always, no split-K; a tile accumulates its updates in the order k = 0, 1, ..), and
//                         the workgroup holding tile (k, k) goes on to factor and invert it without leaving the CU:
//                         the updated tile passes from the accumulators to the LDS image of potf2_inv_body;
//   panel times inverse   L_ik = C_ik W_k' (the GEMM kernel, as in factor_block).
// Workgroups have 512 threads and the diagonal kernel's LDS (one per CU).  That is schedule MODE 0 with two = 0, the
// form of round 3 (still selectable); round 4 applies the SAME tile-panel products at other times: MODE 1 (default)
// takes the units of a step from a plan (see "which trailing columns a block step visits" further down), MODE 0
// with two = 1 visits every second tile column with two panels.  Every tile still accumulates its panels in the
// order 0, 1, .., so the factor has the same bits under all of them.  In MODE 0 with more tiles than CUs a workgroup
// runs two tiles at once: waves 0-3 and waves 4-7 each work like one workgroup of gemm_tn_f64_kernel on their own
// half of the LDS, in lockstep (equal K: both halves execute the same barriers).
struct MidArgs {
    double* A;
    int64_t lda, n;
    int32_t nblk, k;
    int32_t pack;  // tiles per workgroup (1 or 2); the diagonal tile always has its workgroup to itself
    int32_t npair;  // pack == 2: workgroups 1 .. npair take two tiles, the ones behind them one (see chol_factor_enqueue)
    int32_t lite;   // the diagonal workgroup factors only (MODE 1); the inverse images follow in one launch at the end
    int32_t dsyrk;  // the diagonal workgroup updates its tile with mid_diag_syrk (all eight waves, operand fetched at once)
    int32_t two;    // two panels per trailing pass (see chol_factor_enqueue): workgroup 1 = the next diagonal tile (one panel)
    const uint32_t* plan;  // the units of this step (mid_plan_build, mid_plan.inc) or nullptr; one tile per workgroup then
    double* winv;
    int32_t* info;
};
#ifdef MADQP_MID_STAMPS
__device__ unsigned long long madqp_mid_stamps[64][24];  // diagnostic build only (tools/mid_probe.cpp)
#define MID_STAMP(slot)                                                                                              \
    do {                                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x <= 2)                                                                     \
            madqp_mid_stamps[a.k & 63][8 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memrealtime();                  \
    } while (0)
#else
#define MID_STAMP(slot)
#endif
constexpr int MID_THREADS = 512;

// ---- the diagonal workgroup's own update (round 4): S <- C_kk - R R' with R = L[block row k, panel k-1] ------------------
// Through the GEMM main loop (one 128 x 128 tile on four waves, K = 128 in eight stages with ONE stage of prefetch) this
// took 18-22 us of the ~56 us chain of a block step (tools/mid_probe): a single workgroup has nobody to hide the latency of
// its eight dependent stage loads behind, and it computes all 64 sub-tiles where potf2 reads 36.  Here the whole operand
// (128 KB: both operands of a SYRK are the same block row) is fetched at once -- every load of the workgroup in flight
// together, one latency -- into a k-major LDS image, all eight waves take the 36 lower 16 x 16 sub-tiles (a row pair
// I, 7-I per two waves: nine tiles sharing their fragments), and the result goes straight to the LDS image of potf2.
// D'[m][n] = sum_k R(16 J + m, k) R(16 I + n, k): A-operand <- fragment of block row J, B-operand <- block row I, so lane l holds
// element (row 16 I + (l & 15), column 16 J + (l >> 4) + 4 v): rows are the fast index of every global and LDS access.
constexpr int XS_LD = 144;  // k-row stride of the operand image (= 16 mod 32: the two k-rows of a half-wave read hit disjoint banks)
static_assert(NB * XS_LD <= P2_S_DOUBLES + P2_WD_DOUBLES, "the operand image fits the diagonal kernel's LDS");
#ifdef MADQP_MID_STAMPS
#define DS_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (threadIdx.x == 0) madqp_mid_stamps[kstep & 63][(slot)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
#else
#define DS_STAMP(slot)
#endif
// zero_upper: the strictly upper sub-tiles are zeroed (the work area of the in-kernel inverse; the factor-only diagonal
// kernel never touches them)
__device__ __forceinline__ void mid_diag_syrk(const double* __restrict__ Ckk, const double* __restrict__ R, int64_t lda, int nb,
                                              double* __restrict__ smem, int kstep, bool zero_upper) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    // this wave's tiles: row pair (p, 7 - p) has 9 lower tiles, waves 2p / 2p + 1 take the first 5 / last 4 of them
    const int p = wave >> 1, first = (wave & 1) ? 5 : 0, cnt = (wave & 1) ? 4 : 5;
    int tI[5], tJ[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        int q = first + (u < cnt ? u : cnt - 1);  // (a short list repeats its last tile; not stored twice)
        // tiles of the pair in order: (p, 0..p), then (7-p, 0..7-p)
        tI[u] = (q <= p) ? p : 7 - p;
        tJ[u] = (q <= p) ? q : q - (p + 1);
    }
    // 1. the operand: 16 x 16-byte loads per thread, all in flight; pair index e = tid + 512 q: column c = e / 64, rows 2 (e % 64)
    double2_t rv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + MID_THREADS * q, c = e >> 6, i = 2 * (e & 63);
        rv[q] = *reinterpret_cast<const double2_t*>(R + i + (int64_t)c * lda);
    }
    // 2. the tile itself, negated (the products are added, the result is negated back): element (r = 16 I + lo, c = 16 J + hi + 4 v)
    double4_t acc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            int c = 16 * tJ[u] + hi + 4 * v;
            c = c < nb ? c : nb - 1;  // (columns beyond the order need not exist in the caller's buffer: any value will do)
            acc[u][v] = -Ckk[(16 * tI[u] + lo) + (int64_t)c * lda];
        }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + MID_THREADS * q, c = e >> 6, i = 2 * (e & 63);
        *reinterpret_cast<double2_t*>(smem + c * XS_LD + i) = rv[q];
    }
    __syncthreads();
    DS_STAMP(4);
    const double* xa[5];
    const double* xb[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        xa[u] = smem + hi * XS_LD + 16 * tJ[u] + lo;  // A[m = lo][k = hi + 4 s] = R(16 J + lo, 4 s + hi)
        xb[u] = smem + hi * XS_LD + 16 * tI[u] + lo;  // B[k = hi + 4 s][n = lo] = R(16 I + lo, 4 s + hi)
    }
#pragma unroll 2
    for (int s4 = 0; s4 < 32; ++s4) {
#pragma unroll
        for (int u = 0; u < 5; ++u)
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u][4 * s4 * XS_LD], xb[u][4 * s4 * XS_LD], acc[u], 0, 0, 0);
    }
    DS_STAMP(5);
    __syncthreads();  // every wave has read its fragments: the image becomes potf2's S[c * LDS_LD + r]
    DS_STAMP(6);
    // 3. lower tiles -> S (upper triangle of the diagonal sub-tiles zero, identity beyond the order of a short last block);
    //    the strictly upper sub-tiles, which nobody computes, are the inverse's work area and must be zero: every
    //    off-diagonal lower tile (I, J) zeroes its mirror image (J, I) -- all 28 of them, no index arithmetic
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        if (u >= cnt) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 16 * tI[u] + lo, c = 16 * tJ[u] + hi + 4 * v;
            double x = -acc[u][v];
            if (r >= nb || c >= nb) x = (r == c) ? 1.0 : 0.0;
            smem[c * LDS_LD + r] = (r >= c) ? x : 0.0;
            if (zero_upper && tI[u] != tJ[u]) smem[(16 * tI[u] + hi + 4 * v) * LDS_LD + 16 * tJ[u] + lo] = 0.0;  // element (16 J + lo, 16 I + hi + 4 v)
        }
    }
    __syncthreads();
}

// MODE 0: the tiles of the step in launch order (pack / npair / two); 1: planned units (a.plan).
template <int MODE>
__global__ __launch_bounds__(MID_THREADS) void chol_mid_step_kernel(MidArgs a) {
    __shared__ __attribute__((aligned(16))) double smem[P2_S_DOUBLES + P2_WD_DOUBLES];
    static_assert(P2_S_DOUBLES + P2_WD_DOUBLES >= 8 * TILE_DOUBLES, "two GEMM halves fit the diagonal kernel's LDS");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w4 = wave & 3;
    const int wi = w4 & 1, wj = w4 >> 1;
    const int lo = lane & 15, hi = lane >> 4;
    const int k = a.k, rem = a.nblk - k;
    const bool diag = (blockIdx.x == 0);
    MID_STAMP(0);
    int ip = 0, jp = 0, kpan = 1;  // tile (k + ip, k + jp); panels this tile receives (the last kpan ones before k)
    int pan0 = k - 1;              // ... starting with this one
    bool active;
    if (MODE != 0) {
        // Planned units (round 4, mid_plan_build): workgroup 1 + u takes unit u of this step's list -- one trailing tile
        // (row, column) and the panels it receives in one product; the other four waves keep its barriers company.
        // (Measured and dropped: two units per workgroup in the steps of more than one round of tiles -- 6.15 against
        // 5.77 ms per factorisation at n = 8 000.)
        active = (half == 0);
        if (!diag) {
            const uint32_t e = a.plan[(int)blockIdx.x - 1];
            ip = (int)(e & 255u) - k;
            jp = (int)((e >> 8) & 255u) - k;
            pan0 = (int)((e >> 16) & 255u);
            kpan = (int)(e >> 24);
        }
        kpan = __builtin_amdgcn_readfirstlane(kpan);
        pan0 = __builtin_amdgcn_readfirstlane(pan0);
    } else if (MODE == 0 && !a.two) {
        // tile t = i'(i'+1)/2 + j' (0 <= j' <= i' < rem) is (k + i', k + j'); t = 0, the diagonal tile, is workgroup 0
        const int ntiles = rem * (rem + 1) / 2;
        const int wg = (int)blockIdx.x - 1;  // 0-based among the workgroups of the trailing tiles
        const bool paired = a.pack == 2 && wg < a.npair;
        const int t = diag ? 0 : (paired || a.pack == 1 ? 1 + a.pack * wg + half : 1 + 2 * a.npair + (wg - a.npair));
        active = diag ? (half == 0) : (t < ntiles && (paired ? true : half == 0));
        if (active && !diag) {
            ip = (int)((sqrtf(1.0f + 8.0f * (float)t) - 1.0f) * 0.5f);
            while (ip * (ip + 1) / 2 > t) --ip;
            while ((ip + 1) * (ip + 2) / 2 <= t) ++ip;
            jp = t - ip * (ip + 1) / 2;
        }
    } else if (MODE == 0) {
        // Two panels per pass (round 4).  Step k touches only the tile COLUMNS of its own parity, k, k+2, k+4, ..: each
        // receives the two panels k-2 and k-1 in ONE product (K = 256; step 1: panel 0 alone), so a trailing tile is read
        // and written every second step -- half the HBM traffic that bound the first 17 steps.  Column k itself is in
        // that class, so the panel of this step is up to date; its diagonal tile is the diagonal workgroup's (one panel:
        // the tile got panel k-2 as "next diagonal tile" in step k-1), and workgroup 1 gives the NEXT diagonal tile
        // (k+1, k+1) panel k-1 alone, so that step k+1 again finds a tile that lacks one panel (one SYRK, K = 128).
        const bool has_exc = (k + 1 < a.nblk);
        const int wg = (int)blockIdx.x - 1 - (has_exc ? 1 : 0);  // 0-based among the workgroups of the class tiles
        int ncls = rem - 1;
        for (int c = 1; 2 * c < rem; ++c) ncls += rem - 2 * c;
        if (diag) {
            active = (half == 0);
        } else if (has_exc && blockIdx.x == 1) {  // the next diagonal tile, alone in its workgroup (its K differs)
            active = (half == 0);
            ip = jp = 1;
        } else {
            const bool paired = a.pack == 2 && wg < a.npair;
            const int u = paired || a.pack == 1 ? a.pack * wg + half : 2 * a.npair + (wg - a.npair);
            active = u < ncls && (paired ? true : half == 0);
            kpan = (k >= 2) ? 2 : 1;
            if (active) {
                if (u < rem - 1) {
                    jp = 0;
                    ip = 1 + u;
                } else {
                    int v = u - (rem - 1), c = 1;
                    while (v >= rem - 2 * c) {
                        v -= rem - 2 * c;
                        ++c;
                    }
                    jp = 2 * c;
                    ip = jp + v;
                }
            }
        }
        kpan = __builtin_amdgcn_readfirstlane(kpan);
        pan0 = k - kpan;
    }
    ip = __builtin_amdgcn_readfirstlane(ip);  // wave-uniform: the LDS-DMA rows are addressed from scalar registers
    jp = __builtin_amdgcn_readfirstlane(jp);
    const int64_t i0 = (int64_t)(k + ip) * NB, j0 = (int64_t)(k + jp) * NB;
    const int64_t npad = (int64_t)a.nblk * NB;
    double* lds = smem + half * (4 * TILE_DOUBLES);
    double4_t acc[4][4];

    const bool own_syrk = diag && k > 0 && a.dsyrk;  // (uniform over the workgroup)
    if (own_syrk) {
        mid_diag_syrk(a.A + i0 + i0 * a.lda, a.A + i0 + (int64_t)(k - 1) * NB * a.lda, a.lda,
                      (int)((a.n - i0 < NB) ? (a.n - i0) : NB), smem, k, /*zero_upper=*/!a.lite);
        MID_STAMP(1);
    } else if (k > 0) {
        GemmArgs g{};
        g.X = a.A + (int64_t)pan0 * NB * a.lda;  // kpan panels from pan0 on: their columns are contiguous
        g.Y = g.X;
        g.ldx = g.ldy = a.lda;
        g.M = g.N = a.n;
        g.Mread = g.Nread = npad;
        g.K = kpan * NB;
        if (!active) {  // the other half of the workgroup has a tile: keep its barriers company (same K: see the pairing)
            const int nbar = kpan * (NB / BK) + 1;
            for (int b = 0; b < nbar; ++b) __builtin_amdgcn_s_barrier();
        } else {
            // the accumulators start at -C_ij (loads in flight while the first stage is staged; rows up to the padded
            // order exist, what lies beyond n or above the diagonal is never stored) and come back negated
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        int64_t gj = j0 + wj * 64 + tj * 16 + hi + 4 * v;
                        gj = (gj < a.n) ? gj : a.n - 1;  // columns beyond n: any value will do, no branch
                        acc[ti][tj][v] = -a.A[i0 + wi * 64 + ti * 16 + lo + gj * a.lda];
                    }
            mainloop_dma(g, i0, j0, lds, wi, wj, lane, w4, acc);
        }
        MID_STAMP(1);
        if (active && !diag) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int64_t gi = i0 + wi * 64 + ti * 16 + lo;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int64_t gj = j0 + wj * 64 + tj * 16 + hi + 4 * v;
                        if (gi < a.n && gj < a.n && gi >= gj) a.A[gi + gj * a.lda] = -acc[ti][tj][v];
                    }
                }
            }
        }
        MID_STAMP(7);
    }
    if (!diag) return;

    // the diagonal tile: accumulators -> LDS image S[c*LDS_LD + r] (lower triangle, identity padding), factor, invert
    double* S = smem;
    const int nb = (int)((a.n - i0 < NB) ? (a.n - i0) : NB);
    if (k > 0 && !own_syrk) {
        // (the main loop ends with a barrier after its last LDS read: the staging buffers are free)
        if (half == 0) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int r = wi * 64 + ti * 16 + lo;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int c = wj * 64 + tj * 16 + hi + 4 * v;
                        double x = -acc[ti][tj][v];
                        if (r >= nb || c >= nb) x = (r == c) ? 1.0 : 0.0;
                        S[c * LDS_LD + r] = (r >= c) ? x : 0.0;  // the upper triangle is the inverse's work area
                    }
                }
            }
        }
        __syncthreads();
    }
    MID_STAMP(2);
    double* Wcm = a.winv + (int64_t)k * WBLK;
    if (a.lite)
        potf2_inv_body<MID_THREADS, 1>(a.A + i0 + i0 * a.lda, a.lda, nb, Wcm, Wcm + NB * NB, a.info, (int32_t)i0, S,
                                       smem + P2_S_DOUBLES, /*tile_in_lds=*/k > 0);
    else
        potf2_inv_body<MID_THREADS, 0>(a.A + i0 + i0 * a.lda, a.lda, nb, Wcm, Wcm + NB * NB, a.info, (int32_t)i0, S,
                                       smem + P2_S_DOUBLES, /*tile_in_lds=*/k > 0);
    MID_STAMP(3);
}

}  // namespace

extern "C" int32_t madqp_chol_create(madqp_ctx* ctx, int64_t n, madqp_chol** out) {
    ARG_TRY(ctx, ctx && out && n >= 0);
    *out = nullptr;
    madqp_chol* s = new (std::nothrow) madqp_chol();
    if (!s) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    s->ctx = ctx;
    s->n = n;
    s->npos = n;
    s->factored = false;
    s->A = nullptr;
    s->lda = 0;
    s->winv = nullptr;
    s->tmp = nullptr;
    s->d_jobs = nullptr;
    s->d_info = nullptr;
    const int64_t nblk = std::max<int64_t>(1, (n + NB - 1) / NB);
    hipError_t e = hipMalloc(&s->winv, nblk * WBLK * sizeof(double));
    if (e == hipSuccess) e = hipMemset(s->winv, 0, nblk * WBLK * sizeof(double));  // the kernel writes lower parts only
    // sweeps: block rows longer than `chunk` tiles are streamed by several workgroups (SweepPlan).  tmp holds the
    // intermediate vector (padded to whole blocks) followed by the partial-sum slots of both sweeps, so that one fill
    // per solve resets all of them.
    s->sweep_chunk = 64;
    if (const char* e_chunk = getenv("MADQP_SWEEP_CHUNK")) s->sweep_chunk = std::max(1, atoi(e_chunk));
    s->sweep_maxc = 0;
    s->sweep_njobs = 0;
    s->d_jobs = nullptr;
    std::vector<int32_t> jobs;
    if (nblk - 1 > s->sweep_chunk && nblk < (1 << 20)) {
        s->sweep_maxc = (int32_t)((nblk - 1 + s->sweep_chunk - 1) / s->sweep_chunk);
        for (int64_t r = 0; r < nblk; ++r) {
            const int64_t nc = std::max<int64_t>(1, (r + s->sweep_chunk - 1) / s->sweep_chunk);
            for (int64_t c = 0; c < nc; ++c) jobs.push_back((int32_t)(r | (c << 20)));
        }
        s->sweep_njobs = (int32_t)jobs.size();
    }
    s->tmp_len = nblk * NB + 2 * nblk * (int64_t)s->sweep_maxc * NB;
    if (e == hipSuccess) e = hipMalloc(&s->tmp, s->tmp_len * sizeof(double));
    if (e == hipSuccess && !jobs.empty()) {
        e = hipMalloc(&s->d_jobs, jobs.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemcpy(s->d_jobs, jobs.data(), jobs.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    s->upper = nullptr;
    s->upper_ld = 0;
    s->upper_ok = false;
    s->utmp = nullptr;
    s->utmp_len = 2 * nblk * NB + 2 * nblk * (int64_t)s->sweep_maxc * NB;
    if (e == hipSuccess && nblk > 1 && nblk <= 160) e = hipMalloc(&s->utmp, s->utmp_len * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&s->d_info, 8 * sizeof(int32_t));  // [0] info, [1..2] sweep tickets
    if (e != hipSuccess) {
        madqp_chol_destroy(s);
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "madqp_chol_create(%lld): %s", (long long)n,
                          hipGetErrorString(e));
    }
    *out = s;
    return MADQP_OK;
}

// Quasi-definite mode: the matrix handed to madqp_chol_factor is [P, .; B, Q] (lower triangle, P of order npos
// and Q positive definite) and stands for [P, B'; B, -Q]; factor computes L with [P, B'; B, -Q] = L diag(I_npos, -I) L',
// solve applies the inverse of that matrix.  npos a multiple of 128 (or n, which restores plain Cholesky).
extern "C" int32_t madqp_chol_set_signature(madqp_chol* s, int64_t npos) {
    if (!s) return MADQP_ERR_ARG;
    ARG_TRY(s->ctx, npos >= 0 && npos <= s->n && (npos % NB == 0 || npos == s->n));
    s->npos = npos;
    s->factored = false;
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_destroy(madqp_chol* s) {
    if (!s) return MADQP_OK;
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->winv) (void)hipFree(s->winv);
    if (s->tmp) (void)hipFree(s->tmp);
    if (s->d_jobs) (void)hipFree(s->d_jobs);
    if (s->d_info) (void)hipFree(s->d_info);
    if (s->d_mid_plan) (void)hipFree(s->d_mid_plan);
    if (s->upper) (void)hipFree(s->upper);
    if (s->utmp) (void)hipFree(s->utmp);
    delete[] s->mid_units;
    delete s;
    return MADQP_OK;
}

// Width (multiple of 128, 768..2560; up to 2048: +5 ms per iteration at C-main) of the next outer panel: the wide update GEMM runs
// (rows/128) x (W/128) tiles of equal cost on `slots` resident workgroups, so W is chosen to
// make the last round of tiles as full as possible (tail quantisation is the main loss of a
// left-looking factorisation; 8 fixed tile columns leave the last round 5-50 % full).
static int64_t outer_panel_width(int64_t rows, bool has_update, int64_t slots) {
    const int64_t mt = (rows + NB - 1) / NB;
    if (!has_update || mt <= 6) return NBO;
    int64_t best = NBO / NB;
    double best_eff = -1.0;
    static const int64_t wt_max = getenv("MADQP_CHOL_WMAX") ? atoll(getenv("MADQP_CHOL_WMAX")) : 20;
    static const int64_t wt_min = getenv("MADQP_CHOL_WMIN") ? atoll(getenv("MADQP_CHOL_WMIN")) : 6;
    for (int64_t wt = wt_min; wt <= wt_max && wt <= mt; ++wt) {
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
                            int64_t k0, int64_t width, int64_t kend = -1, double alpha = -1.0) {
    // C[row0:n, row0:row0+width] += alpha L[row0:n, k0:kend] * L[row0:row0+width, k0:kend]'   (kend = row0
    // unless given: the distributed factorisation applies one received panel at a time; alpha = +1: the
    // columns of the positive block applied to the negative block of a quasi-definite matrix)
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
    g.alpha = alpha;
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

// factor-only diagonal kernels + one inversion launch per factorisation (round 4); MADQP_CHOL_LITE=0: factor and invert in
// the diagonal kernel as before.  Needs the block-substitution panel solve (the product with the 128-inverse reads the image).
static bool panel_inv_mode();
static bool chol_lite() {
    static const bool on = !(getenv("MADQP_CHOL_LITE") && atoi(getenv("MADQP_CHOL_LITE")) == 0);
    return on && !panel_inv_mode();
}
// The diagonal step of the sweeps: 1 = unit block substitution with normalised images (default), 2 = plain block
// substitution over the 16 x 16 sub-blocks (MADQP_SWEEP_DIAG=sub16), 0 = the product with the stored 128 x 128 inverse of
// rounds 1-4 (MADQP_SWEEP_DIAG=inv; also whenever the diagonal kernels factor AND invert: MADQP_CHOL_LITE=0 /
// MADQP_CHOL_PANEL=inv, whose images are full inverses).
static int sweep_diag_mode() {
    static const int mode = [] {
        const char* e = getenv("MADQP_SWEEP_DIAG");
        if (e && strcmp(e, "inv") == 0) return 0;
        if (e && strcmp(e, "sub16") == 0) return 2;
        return 1;
    }();
    return chol_lite() ? mode : 0;
}
// what the factor-only diagonal kernels leave out of the serial spine, for all blocks j0/128 .. in ONE launch: the
// off-diagonal parts of the sweep images
static int32_t invert_blocks(madqp_chol* s, double* A, int64_t lda, int64_t j0, int64_t w) {
    if (w <= 0 || !chol_lite()) return MADQP_OK;
    madqp_ctx* ctx = s->ctx;
    ProfScope ps(ctx, MADQP_PROF_POTRF_DIAG);
    const unsigned nblk = (unsigned)((w + NB - 1) / NB);
    if (sweep_diag_mode() == 0)
        hipLaunchKernelGGL(potf2_invert_kernel, dim3(nblk), dim3(P2_KTHREADS), 0, ctx->stream, A, lda, s->n, j0, s->winv);
    else
        hipLaunchKernelGGL(sweep_image_kernel, dim3(nblk), dim3(256), 0, ctx->stream, A, lda, s->n, j0, s->winv,
                           sweep_diag_mode() != 2 ? 1 : 0);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
template <int DIAG>
static void launch_sweep(bool bwd, unsigned grid, hipStream_t st, const double* L, int64_t ld, const double* winv, const double* in,
                         double* out, int64_t n, int32_t* ctl, double* fault, int vec, const SweepPlan& plan) {
    if (bwd)
        hipLaunchKernelGGL(trsv_bwd_sweep_kernel<DIAG>, dim3(grid), dim3(1024), 0, st, L, ld, winv, in, out, n, ctl, fault, vec, plan);
    else
        hipLaunchKernelGGL(trsv_fwd_sweep_kernel<DIAG>, dim3(grid), dim3(1024), 0, st, L, ld, winv, in, out, n, ctl, fault, vec, plan,
                           (double*)nullptr);
}
static void launch_sweep(bool bwd, unsigned grid, hipStream_t st, const double* L, int64_t ld, const double* winv, const double* in,
                         double* out, int64_t n, int32_t* ctl, double* fault, int vec, const SweepPlan& plan) {
    switch (sweep_diag_mode()) {
        case 0: launch_sweep<0>(bwd, grid, st, L, ld, winv, in, out, n, ctl, fault, vec, plan); break;
        case 2: launch_sweep<2>(bwd, grid, st, L, ld, winv, in, out, n, ctl, fault, vec, plan); break;
        default: launch_sweep<1>(bwd, grid, st, L, ld, winv, in, out, n, ctl, fault, vec, plan); break;
    }
}
static bool panel_inv_mode() {
    static const bool inv = getenv("MADQP_CHOL_PANEL") && strcmp(getenv("MADQP_CHOL_PANEL"), "inv") == 0;
    return inv;
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
        if (chol_lite())
            hipLaunchKernelGGL(potf2_inv_kernel<1>, dim3(1), dim3(P2_KTHREADS), 0, ctx->stream, A + jb + jb * lda, lda,
                               (int)w, Wcm, Wrm, s->d_info, (int32_t)jb, Potf2Batch{0, 0, 0, nullptr});
        else
            hipLaunchKernelGGL(potf2_inv_kernel<0>, dim3(1), dim3(P2_KTHREADS), 0, ctx->stream, A + jb + jb * lda, lda,
                               (int)w, Wcm, Wrm, s->d_info, (int32_t)jb, Potf2Batch{0, 0, 0, nullptr});
        LAUNCH_CHECK(ctx);
    }
    static const bool pp_gemm = getenv("MADQP_CHOL_PP") && atoi(getenv("MADQP_CHOL_PP")) == 0;
    const int64_t npad_b = (n + NB - 1) / NB * NB;
    // the rows below the block: block substitution with the 16 x 16 diagonal inverses (panel_sub16_kernel); MADQP_CHOL_PANEL=inv
    // brings back the products with the 128 x 128 inverse of rounds 1-3 (for the A/B numbers in DESIGN.md)
    if (jb + w < n && w == NB && !panel_inv_mode())
        return panel_solve_sub16(ctx, A + (jb + NB) + jb * lda, lda, A + jb + jb * lda, lda, Wcm, n - jb - NB,
                                 lda >= npad_b ? npad_b - jb - NB : n - jb - NB);
    // whole block and few rows below it (one register-heavy workgroup per CU: beyond ~24 000 rows -- 3 rounds -- the GEMM
    // kernel's 128-row tiles, which read W once per 128 rows, are the faster form: n = 50 000 measured 1 316-1 319 against
    // 1 311-1 315 ms per iteration with this kernel on every block)
    if (jb + w < n && w == NB && lda >= npad_b && !pp_gemm && n - jb - NB <= 24576) {
        ProfScope ps(ctx, MADQP_PROF_POTRF_TRSM);
        hipLaunchKernelGGL(panel_inv_kernel, dim3((unsigned)((npad_b - jb - NB) / 32)), dim3(256), 0, ctx->stream,
                           A + (jb + NB) + jb * lda, lda, Wcm, n - jb - NB);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
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

// ---- mid-size schedule: which trailing columns a block step visits (round 4) ---------------------------------------
// The right-looking schedule rewrote the whole trailing matrix in every step; its first steps were bound by that (two
// rounds of tiles per step at n = 5 000) while the later ones wait for the chain diagonal tile -> panel -> diagonal tile
// with most CUs idle.  The update of column j with panel p is due only at step j, so the work can be moved: tile column
// j carries upto[j] (panels 0 .. upto[j]-1 applied); step k VISITS a column by applying up to MID_Q of its pending panels
// to all its tiles in one product (K = 128 q, one tile per workgroup, a column at most once per step).  Every step
// visits column k (completing it: its panel is solved next) and column k+1 (so that the next diagonal tile lacks panel
// k alone, the diagonal workgroup's own SYRK), then the columns with the least slack -- steps until the column is due
// minus the visits it still needs -- while the step's budget of `rounds` x `cap` tiles lasts; a column whose slack is
// used up is visited whatever the budget.  `rounds` is the smallest count for which no visit ever needs more than
// MID_Q panels: 1 up to n = 5 120 on 256 CUs (every step then fits the shadow of the chain: 40 steps of ~47 us at
// n = 5 000 instead of 8 of them at ~90 us), 2-4 up to 10 240, 5-6 up to 13 312.  The plan depends on (number of blocks, cap) only.
// builds and uploads the plan of s (once); false: no plan (the caller runs the two-panel schedule)
static bool mid_plan_build(madqp_chol* s, int nblk, int cap) {
    if (s->mid_plan_state) return s->mid_plan_state > 0;
    s->mid_plan_state = -1;
    if (nblk > 255 || cap < 1) return false;
    std::vector<std::vector<MidVisit>> steps;
    std::vector<int32_t> units;
    bool ok = false;
    for (int rounds = 1; rounds <= 64 && !ok; ++rounds) ok = mid_plan_steps(nblk, rounds * cap, steps, units, cap);
    if (!ok) return false;
    std::vector<uint32_t> w;
    s->mid_units = new (std::nothrow) int32_t[2 * nblk];
    if (!s->mid_units) return false;
    // every failure exit leaves the handle without a plan AND without its pieces (the fallback schedule runs next: no
    // sticky HIP error may reach its first launch check)
    auto give_up = [&]() {
        (void)hipGetLastError();
        if (s->d_mid_plan) (void)hipFree(s->d_mid_plan);
        s->d_mid_plan = nullptr;
        delete[] s->mid_units;
        s->mid_units = nullptr;
        return false;
    };
    for (int k = 0; k < nblk; ++k) {
        s->mid_units[nblk + k] = (int32_t)w.size();
        mid_plan_units(nblk, k, steps[k], w);
        s->mid_units[k] = (int32_t)w.size() - s->mid_units[nblk + k];
        if (s->mid_units[k] != units[k]) return give_up();
    }
    if (w.empty()) w.push_back(0);
    if (hipMalloc(&s->d_mid_plan, w.size() * sizeof(uint32_t)) != hipSuccess) {
        s->d_mid_plan = nullptr;
        return give_up();
    }
    // (synchronous, pageable source: the copy has left `w` when the call returns; once per handle)
    if (hipMemcpy(s->d_mid_plan, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return give_up();
    s->mid_plan_state = 1;
    return true;
}

// Everything of a factorisation except reading its info back: the launches are on the stream, info sits in s->d_info.
static int32_t chol_factor_enqueue(madqp_chol* s, double* A, int64_t lda) {
    madqp_ctx* ctx = s->ctx;
    const int64_t n = s->n;
    HIP_TRY(ctx, hipMemsetAsync(s->d_info, 0, sizeof(int32_t), ctx->stream));
    s->upper_ok = false;
    // mid-size matrices: right-looking, two launches per 128-column block (chol_mid_step_kernel)
    // (measured on bench.py, m = 0.4 n, ms per iteration against the left-looking schedule: 3.34 / 3.90 at n = 3 000,
    // 6.37 / 7.51 at 5 000, 13.7 / 15.1 at 8 000, 22.7 / 23.2 at 10 000, 34.4 / 33.0 at 12 000 -- every step rewrites
    // the whole trailing matrix, which the left-looking schedule does not)
    // (round 5, with the backward sweep on U = L' behind this path -- ms per iteration, this path / left-looking: 19.1 / 21.0 at
    // n = 10 000, 23.7 / 25.4 at 11 000, 29.2 / 30.5 at 12 000, 35.5 / 36.4 at 13 000, 42.8 / 42.6 at 14 000, 60.8 / 58.1 at 16 000)
    static const int64_t mid_max = getenv("MADQP_CHOL_MID_MAX") ? atoll(getenv("MADQP_CHOL_MID_MAX")) : 13312;
    const int64_t npad_m = (n + NB - 1) / NB * NB;
    if (n <= mid_max && n > NB && s->npos == n && lda >= npad_m && lda % 2 == 0 && (((uintptr_t)A) & 15) == 0) {
        const int32_t nblk = (int32_t)(npad_m / NB);
        // one timer scope for the whole factorisation (80 short launches: a scope each costs 0.7 ms per factorisation
        // at n = 5 000); the scopes of the launches inside are switched off meanwhile
        ProfScope ps_all(ctx, MADQP_PROF_POTRF_GEMM);
        struct ProfMute {
            madqp_ctx* c;
            decltype(c->prof) saved;
            explicit ProfMute(madqp_ctx* ctx_) : c(ctx_), saved(ctx_->prof) { c->prof = 0; }
            ~ProfMute() { c->prof = saved; }
        } mute(ctx);
        static const bool mid_dsyrk = !(getenv("MADQP_CHOL_MID_DSYRK") && atoi(getenv("MADQP_CHOL_MID_DSYRK")) == 0);
        // two panels per trailing pass (see the kernel): needs the diagonal workgroup's own SYRK
        static const bool mid_two = mid_dsyrk && !(getenv("MADQP_CHOL_MID_TWO") && atoi(getenv("MADQP_CHOL_MID_TWO")) == 0);
        // planned visits (mid_plan_build) unless switched off; the two-panel schedule otherwise
        static const bool mid_lazy = mid_dsyrk && !(getenv("MADQP_CHOL_MID_LAZY") && atoi(getenv("MADQP_CHOL_MID_LAZY")) == 0);
        static const int mid_cap = getenv("MADQP_CHOL_MID_CAP") ? atoi(getenv("MADQP_CHOL_MID_CAP")) : 0;  // (experiments)
        const bool planned = mid_lazy && mid_plan_build(s, nblk, mid_cap > 0 ? mid_cap : ctx->gemm_slots / 2 - 1);
        for (int32_t k = 0; k < nblk; ++k) {
            if (planned) {
                hipLaunchKernelGGL(chol_mid_step_kernel<1>, dim3((unsigned)(1 + s->mid_units[k])), dim3(MID_THREADS), 0,
                                   ctx->stream,
                                   MidArgs{A, lda, n, nblk, k, 1, 0, chol_lite() ? 1 : 0, 1, 0,
                                           s->d_mid_plan + s->mid_units[nblk + k], s->winv, s->d_info});
                LAUNCH_CHECK(ctx);
            } else {
                const int64_t rem = nblk - k;
                int64_t nt, extra = 0;  // tiles shared out by pack/npair; workgroups before them besides the diagonal one
                if (k == 0) {
                    nt = 0;
                } else if (!mid_two) {
                    nt = rem * (rem + 1) / 2 - 1;
                } else {
                    nt = rem - 1;  // column k below its diagonal tile, then columns k+2, k+4, ..
                    for (int64_t c = 1; 2 * c < rem; ++c) nt += rem - 2 * c;
                    extra = (k + 1 < nblk) ? 1 : 0;  // the next diagonal tile
                }
                // One workgroup per CU (LDS), one or two tiles each.  Up to a round of single tiles: singles; up to a round
                // of pairs: pairs; beyond that -- two rounds -- as few pairs as two rounds of workgroups need, FIRST in the
                // grid, singles behind them: a CU then works off three tiles (pair + single or three singles, 44 + 22 us)
                // instead of four (two pairs, 88 us) whenever three per CU are enough (up to 765 tiles).
                const int64_t cus = ctx->gemm_slots / 2 - 1 - extra;  // (the diagonal workgroup holds a CU)
                int32_t pack = (nt > cus) ? 2 : 1, npair = 0;
                unsigned grid = (unsigned)(1 + extra + (nt + pack - 1) / pack);
                if (pack == 2) {
                    npair = (int32_t)((nt + 1) / 2);
                    if (nt > 2 * cus && nt - 2 * cus <= cus) {
                        npair = (int32_t)(nt - 2 * cus);
                        grid = (unsigned)(1 + extra + npair + (nt - 2 * (int64_t)npair));
                    }
                }
                hipLaunchKernelGGL(chol_mid_step_kernel<0>, dim3(grid), dim3(MID_THREADS), 0, ctx->stream,
                                   MidArgs{A, lda, n, nblk, k, pack, npair, chol_lite() ? 1 : 0, mid_dsyrk ? 1 : 0,
                                           mid_two ? 1 : 0, nullptr, s->winv, s->d_info});
                LAUNCH_CHECK(ctx);
            }
            const int64_t jb = (int64_t)k * NB;
            static const bool pp_gemm = getenv("MADQP_CHOL_PP") && atoi(getenv("MADQP_CHOL_PP")) == 0;
            if (jb + NB < n && !panel_inv_mode()) {  // L[below, jb] = C[below, jb] L_kk^-T by block substitution
                const int32_t r = panel_solve_sub16(ctx, A + (jb + NB) + jb * lda, lda, A + jb + jb * lda, lda,
                                                    s->winv + (int64_t)k * WBLK, n - jb - NB, npad_m - jb - NB);
                if (r) return r;
            } else if (jb + NB < n && !pp_gemm) {  // L[below, jb] = C[below, jb] W_k'
                const int64_t below = n - jb - NB;
                hipLaunchKernelGGL(panel_inv_kernel, dim3((unsigned)((npad_m - jb - NB) / 32)), dim3(256), 0, ctx->stream,
                                   A + (jb + NB) + jb * lda, lda, s->winv + (int64_t)k * WBLK, below);
                LAUNCH_CHECK(ctx);
            } else if (jb + NB < n) {  // the same product by the GEMM kernel (see factor_block)
                GemmArgs g{};
                g.X = A + (jb + NB) + jb * lda;
                g.ldx = lda;
                g.Y = s->winv + (int64_t)k * WBLK;
                g.ldy = NB;
                g.C = A + (jb + NB) + jb * lda;
                g.ldc = lda;
                g.alpha = 1.0;
                g.beta = 0.0;
                g.M = n - jb - NB;
                g.N = NB;
                g.K = NB;
                g.Mread = npad_m - jb - NB;
                g.Nread = NB;
                const int32_t r = madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_TRSM);
                if (r) return r;
            }
        }
        const int32_t ri = invert_blocks(s, A, lda, 0, n);  // the inverse images of all blocks, one launch (factor-only diagonal kernels)
        if (ri) return ri;
        // U = L' for the backward sweep (trsv_fwd_sweep_kernel<1, true>): one pass over the factor, 37 us at n = 5 000
        static const bool upper_on = !(getenv("MADQP_SWEEP_UPPER") && atoi(getenv("MADQP_SWEEP_UPPER")) == 0);
        if (upper_on && s->utmp && sweep_diag_mode() == 1) {
            if (!s->upper) {
                if (hipMalloc(&s->upper, (size_t)npad_m * npad_m * sizeof(double)) != hipSuccess) {
                    (void)hipGetLastError();
                    s->upper = nullptr;
                    return MADQP_OK;  // (no room for the copy: the backward sweep on L serves)
                }
                s->upper_ld = npad_m;
            }
            hipLaunchKernelGGL(lower_to_upper_kernel, dim3(2 * nblk, 2 * nblk), dim3(256), 0, ctx->stream, A, lda, n, s->upper,
                               s->upper_ld);
            LAUNCH_CHECK(ctx);
            s->upper_ok = true;
        }
        return MADQP_OK;
    }
    // Quasi-definite mode (npos < n): A = [P, .; B, -Q] with P, Q positive definite and Q's block STORED AS +Q.
    // A = L diag(I, -I) L' with L = [L11, 0; W, L22], W = B L11^-T, L22 L22' = Q + W W': the same left-looking
    // sweep, except that an outer panel of the second block receives the columns of the first with a plus sign
    // (and never straddles the boundary).  No pivoting: stable for quasi-definite matrices.
    const int64_t npos = s->npos;
    int64_t W = 0;
    for (int64_t J0 = 0; J0 < n; J0 += W) {
        W = std::min<int64_t>(outer_panel_width(n - J0, J0 > 0, ctx->gemm_slots), n - J0);
        if (J0 < npos && J0 + W > npos) W = npos - J0;
        int32_t r = MADQP_OK;
        if (J0 > 0 && J0 < npos) {
            r = panel_update(ctx, A, lda, n, J0, 0, W);
        } else if (J0 > 0) {
            if (npos > 0) r = panel_update(ctx, A, lda, n, J0, 0, W, npos, 1.0);
            if (!r && J0 > npos) r = panel_update(ctx, A, lda, n, J0, npos, W);
        }
        if (r) return r;
        r = factor_range(s, A, lda, J0, W);
        if (r) return r;
    }
    return invert_blocks(s, A, lda, 0, n);
}

extern "C" int32_t madqp_chol_factor(madqp_chol* s, double* A, int64_t lda, int32_t* info_host) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, A && info_host && lda >= s->n);
    s->factored = false;
    s->A = A;
    s->lda = lda;
    *info_host = 0;
    if (s->n == 0) {
        s->factored = true;
        return MADQP_OK;
    }
    int32_t r = chol_factor_enqueue(s, A, lda);
    if (r) return r;
    int32_t info = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&info, s->d_info, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *info_host = info;
    s->factored = (info == 0);
    return MADQP_OK;
}

namespace {
__global__ void info_to_slot_kernel(const int32_t* __restrict__ info, double* __restrict__ slot) {
    *slot = (double)*info;
}
}  // namespace
// Queued form (mpc.hip): no read-back here; info goes to *d_slot (a word of the context's result block) and comes
// back with the caller's next madqp_read_results, who then reports it through madqp_chol_factor_result.  Until then
// the object counts as factored: solves enqueued behind a failed factorisation produce values the caller discards.
int32_t madqp_chol_factor_q(madqp_chol* s, double* A, int64_t lda, double* d_slot) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, A && d_slot && lda >= s->n);
    s->A = A;
    s->lda = lda;
    s->factored = true;
    if (s->n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_slot, 0, sizeof(double), ctx->stream));
        return MADQP_OK;
    }
    int32_t r = chol_factor_enqueue(s, A, lda);
    if (r) return r;
    hipLaunchKernelGGL(info_to_slot_kernel, dim3(1), dim3(1), 0, ctx->stream, s->d_info, d_slot);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
void madqp_chol_factor_result(madqp_chol* s, int32_t info) {
    if (s) s->factored = (info == 0);
}

// ---------------------------------------------------------------------------------------------
// The same recursive factorisation for a batch of B equally sized matrices at fixed strides (small
// QPs, batch.hip): every launch covers all problems (grid.y / grid.x = problem), problems with
// skip[b] != 0 are left alone.  winv: B x nblk x WBLK, info: B ints.
namespace {
struct CholBatch {
    madqp_ctx* ctx;
    double* A;
    int64_t lda, n, sA;
    double* winv;
    int64_t sW;
    int32_t* info;
    int64_t B;
    const int32_t* skip;
    const int32_t* list;   // compacted form (GemmBatch::list): B = the number of slots
    const int32_t* count;
};
int32_t bfactor_update(const CholBatch& c, int64_t row0, int64_t k0, int64_t width) {
    GemmArgs g{};
    g.X = c.A + row0 + k0 * c.lda;
    g.ldx = c.lda;
    g.Y = g.X;
    g.ldy = c.lda;
    g.C = c.A + row0 + row0 * c.lda;
    g.ldc = c.lda;
    g.Cin = g.C;
    g.ldcin = c.lda;
    g.alpha = -1.0;
    g.beta = 1.0;
    g.M = c.n - row0;
    g.N = width;
    g.K = row0 - k0;
    const int64_t npad = (c.n + NB - 1) / NB * NB;
    if (c.lda >= npad) {
        g.Mread = npad - row0;
        g.Nread = std::min<int64_t>(npad - row0, (width + NB - 1) / NB * NB);
    }
    g.lower_only = 1;
    GemmBatch bt{c.B, c.sA, c.sA, c.sA, c.sA, 0, c.skip, c.list, c.count};
    return madqp_gemm_tn(c.ctx, g, MADQP_PROF_POTRF_GEMM, nullptr, 0, &bt);
}
int32_t bfactor_block(const CholBatch& c, int64_t jb, int64_t w) {
    madqp_ctx* ctx = c.ctx;
    double* Wcm = c.winv + (jb / NB) * WBLK;
    {
        ProfScope ps(ctx, MADQP_PROF_POTRF_DIAG);
        hipLaunchKernelGGL(potf2_inv_kernel<0>, dim3((unsigned)c.B), dim3(P2_KTHREADS), 0, ctx->stream,
                           c.A + jb + jb * c.lda, c.lda, (int)w, Wcm, Wcm + NB * NB, c.info, (int32_t)jb,
                           Potf2Batch{c.sA, c.sW, 1, c.skip, c.list, c.count});
        LAUNCH_CHECK(ctx);
    }
    if (jb + w < c.n && w == NB && !panel_inv_mode()) {
        const int64_t npad = (c.n + NB - 1) / NB * NB;
        return panel_solve_sub16(ctx, c.A + (jb + NB) + jb * c.lda, c.lda, c.A + jb + jb * c.lda, c.lda, Wcm, c.n - jb - NB,
                                 c.lda >= npad ? npad - jb - NB : c.n - jb - NB, c.B, PanelBatch{c.sA, c.sA, c.sW, c.skip, c.list, c.count});
    }
    if (jb + w < c.n) {
        GemmArgs g{};
        g.X = c.A + (jb + w) + jb * c.lda;
        g.ldx = c.lda;
        g.Y = Wcm;
        g.ldy = NB;
        g.C = c.A + (jb + w) + jb * c.lda;
        g.ldc = c.lda;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.M = c.n - jb - w;
        g.N = w;
        g.K = w;
        const int64_t npad = (c.n + NB - 1) / NB * NB;
        if (c.lda >= npad) g.Mread = npad - jb - w;
        g.Nread = NB;
        GemmBatch bt{c.B, c.sA, c.sW, c.sA, 0, 0, c.skip, c.list, c.count};
        return madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_TRSM, nullptr, 0, &bt);
    }
    return MADQP_OK;
}
int32_t bfactor_range(const CholBatch& c, int64_t j0, int64_t w) {
    if (w <= NB) return bfactor_block(c, j0, w);
    const int64_t h = ((w + NB - 1) / NB + 1) / 2 * NB;
    int32_t r = bfactor_range(c, j0, h);
    if (r) return r;
    if ((r = bfactor_update(c, j0 + h, j0, w - h))) return r;
    return bfactor_range(c, j0 + h, w - h);
}
}  // namespace

// internal (batch.hip): asynchronous; info[b] = 0 or the first failing column of problem b (1-based)
int32_t madqp_chol_factor_batched(madqp_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t sA, double* winv,
                                  int64_t sW, int32_t* info, int64_t B, const int32_t* skip, int64_t slots,
                                  const int32_t* list, const int32_t* count) {
    ARG_TRY(ctx, A && winv && info && lda >= n && B >= 1 && (!list || (count && slots >= 1)));
    if (n == 0) return MADQP_OK;
    HIP_TRY(ctx, hipMemsetAsync(info, 0, (size_t)B * sizeof(int32_t), ctx->stream));
    CholBatch c{ctx, A, lda, n, sA, winv, sW, info, list ? slots : B, list ? nullptr : skip, list, count};
    return bfactor_range(c, 0, n);
}

// ---------------------------------------------------------------------------------------------
// Pieces of the factorisation for the multi-GPU path: the diagonal tile of a step of the 2-D block-cyclic
// Cholesky (dist.hip: dop_potrf_tile) is an order-nb matrix the distributed object owns -- factor_begin adopts it,
// factor_panel runs the blocked factorisation on it, panel_pack lays out [info | inverse diagonal blocks | L] for the
// broadcast.  All asynchronous on the context's stream.
namespace {
__global__ void info_store_kernel(const int32_t* __restrict__ info, double* __restrict__ hdr) {
    hdr[0] = (double)*info;
    hdr[1] = 0.0;
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
    ARG_TRY(ctx, s->npos == s->n);  // the panel pieces factor positive definite matrices only
    s->factored = false;
    s->A = A;
    s->lda = lda;
    HIP_TRY(ctx, hipMemsetAsync(s->d_info, 0, sizeof(int32_t), ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_factor_panel(madqp_chol* s, int64_t j0, int64_t w) {
    if (!s) return MADQP_ERR_ARG;
    ARG_TRY(s->ctx, panel_ok(s, j0, w));
    const int32_t r = factor_range(s, s->A, s->lda, j0, w);
    return r ? r : invert_blocks(s, s->A, s->lda, j0, w);  // (the caller packs / reads the images of this panel next)
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

// X (rows x 128, leading dimension ldx) <- X L^-T for ONE factored 128 x 128 block L (ldl) with the inverse image W of its
// 16 x 16 diagonal sub-blocks (dist.hip: the leaves of a tile's panel solve); false: this build multiplies with the
// 128 x 128 inverse instead (MADQP_CHOL_PANEL=inv) and the caller takes its GEMM form
bool madqp_chol_panel_sub16_on() { return !panel_inv_mode(); }
int32_t madqp_chol_panel_solve128(madqp_ctx* ctx, double* X, int64_t ldx, int64_t rows, int64_t rows_read, const double* L,
                                  int64_t ldl, const double* Wcm) {
    return panel_solve_sub16(ctx, X, ldx, L, ldl, Wcm, rows, rows_read);
}

// One sweep over a single order-w tile (multi-GPU solves, dist.hip): the same kernels as madqp_chol_solve.
int32_t madqp_trsv_tile(madqp_ctx* ctx, int32_t trans, const double* L, int64_t ld, const double* winv, double* v,
                        int64_t w, double* tmp, int32_t* ctl) {
    ARG_TRY(ctx, L && winv && v && tmp && ctl && w > 0 && ld >= w);
    ProfScope ps(ctx, MADQP_PROF_TRSV);
    const unsigned nblk = (unsigned)((w + NB - 1) / NB);
    const int vec = ((((uintptr_t)L) & 15) == 0) && (ld % 2 == 0);
    HIP_TRY(ctx, hipMemsetAsync(ctl, 0, 4 * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)tmp, 0x7FF8A5A5, 2 * (size_t)w, ctx->stream));
    const SweepPlan none{nullptr, nullptr, 0, 0};
    double* fault = ctx->d_res + MADQP_FAULT_SLOT;
    launch_sweep(trans != 0, nblk, ctx->stream, L, ld, winv, v, tmp, w, ctl, fault, vec, none);
    LAUNCH_CHECK(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(v, tmp, (size_t)w * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_chol_solve(madqp_chol* s, double* rhs) {
    if (!s) return MADQP_ERR_ARG;
    madqp_ctx* ctx = s->ctx;
    ARG_TRY(ctx, rhs != nullptr || s->n == 0);
    if (!s->A) return madqp_fail(ctx, MADQP_ERR_STATE, "madqp_chol_solve before madqp_chol_factor");
    const int64_t n = s->n, lda = s->lda;
    const double* A = s->A;
    if (n == 0) return MADQP_OK;
    ProfScope ps(ctx, MADQP_PROF_TRSV);
    if (s->upper_ok) {  // mid-size factor: y = L^-1 b, then x = U^-1 y with the same kernel on U = L'; three launches
        const int64_t nblk = (n + NB - 1) / NB;
        const unsigned grid = s->d_jobs ? (unsigned)s->sweep_njobs : (unsigned)nblk;
        double* yv = s->utmp;
        double* xv = yv + nblk * NB;
        double* part_f = xv + nblk * NB;
        double* part_b = part_f + nblk * s->sweep_maxc * NB;
        const SweepPlan pf{s->d_jobs, part_f, s->sweep_chunk, s->sweep_maxc}, pb{s->d_jobs, part_b, s->sweep_chunk, s->sweep_maxc};
        double* fault = ctx->d_res + MADQP_FAULT_SLOT;
        hipLaunchKernelGGL(sweep_prep_kernel, dim3((unsigned)std::min<int64_t>((s->utmp_len + 255) / 256, 256)), dim3(256), 0,
                           ctx->stream, reinterpret_cast<unsigned long long*>(s->utmp), s->utmp_len, s->d_info);
        hipLaunchKernelGGL((trsv_fwd_sweep_kernel<1, false>), dim3(grid), dim3(1024), 0, ctx->stream, A, lda, s->winv,
                           (const double*)rhs, yv, n, s->d_info, fault, 1, pf, (double*)nullptr);
        hipLaunchKernelGGL((trsv_fwd_sweep_kernel<1, true>), dim3(grid), dim3(1024), 0, ctx->stream, (const double*)s->upper,
                           s->upper_ld, s->winv, (const double*)yv, xv, n, s->d_info, fault, 1, pb, rhs);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    {
        // one launch per sweep (see trsv_*_sweep_kernel): y = L^-1 b into s->tmp, x = L^-T y into rhs
        const unsigned nblk = (unsigned)((n + NB - 1) / NB);
        const int vec = ((((uintptr_t)A) & 15) == 0) && (lda % 2 == 0);
        const int64_t nb64 = nblk;
        const unsigned grid = s->d_jobs ? (unsigned)s->sweep_njobs : nblk;
        double* part_f = s->tmp + nb64 * NB;
        double* part_b = part_f + nb64 * s->sweep_maxc * NB;
        const SweepPlan pf{s->d_jobs, part_f, s->sweep_chunk, s->sweep_maxc};
        const SweepPlan pb{s->d_jobs, part_b, s->sweep_chunk, s->sweep_maxc};
        HIP_TRY(ctx, hipMemsetAsync(s->d_info + 1, 0, 3 * sizeof(int32_t), ctx->stream));
        HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)s->tmp, 0x7FF8A5A5, 2 * (size_t)s->tmp_len, ctx->stream));
        launch_sweep(false, grid, ctx->stream, A, lda, s->winv, rhs, s->tmp, n, s->d_info, ctx->d_res + MADQP_FAULT_SLOT, vec, pf);
        LAUNCH_CHECK(ctx);
        if (s->npos < n) {  // y <- diag(I, -I) y
            const int64_t len = n - s->npos;
            hipLaunchKernelGGL(negate_kernel, dim3((unsigned)std::min<int64_t>((len + 255) / 256, 1024)), dim3(256), 0,
                               ctx->stream, s->tmp + s->npos, len);
            LAUNCH_CHECK(ctx);
        }
        HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)rhs, 0x7FF8A5A5, 2 * (size_t)n, ctx->stream));
        launch_sweep(true, grid, ctx->stream, A, lda, s->winv, s->tmp, rhs, n, s->d_info, ctx->d_res + MADQP_FAULT_SLOT, vec, pb);
        LAUNCH_CHECK(ctx);
    }
    return MADQP_OK;
}
