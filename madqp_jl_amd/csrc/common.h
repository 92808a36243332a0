// Internal helpers shared by the translation units of libmadqp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/madqp.h"

#define MADQP_RESULT_SLOTS 128
// last slot of the result block: device-side fault word (non-zero: a triangular sweep's hand-off timed out);
// it rides along with every synchronising scalar read-back (madqp_read_results) and turns into MADQP_ERR_HIP
#define MADQP_FAULT_SLOT (MADQP_RESULT_SLOTS - 1)

struct ProfEvent {
    hipEvent_t a, b;
    int cls;
};

struct madqp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t gemm_slots = 512;  // resident GEMM workgroups: 2 per CU
    // > 0: GEMM launches with more tiles than (gemm_slots - gemm_cap_slots) run as ONE persistent launch of that many
    // workgroups drawing tiles from per-XCD ticket counters, leaving gemm_cap_slots workgroup slots of the chip free
    // for kernels on other streams (dist.hip: the collectives beside a trailing update).  0: one workgroup per tile.
    int64_t gemm_cap_slots = 0;
    unsigned long long* d_tickets = nullptr;  // 8 counters, zeroed on the stream before every capped launch
    char err[512] = {0};
    // scalar results: device block + pinned host mirror
    double* d_res = nullptr;
    double* h_res = nullptr;
    // reduction partials (MADQP_MAX_BLOCKS x MADQP_RESULT_SLOTS doubles)
    double* d_part = nullptr;
    // generic workspace for deterministic two-pass gemv
    double* d_work = nullptr;
    size_t work_bytes = 0;
    // scaled copy of the assembly operand (Theta A), grow-only
    double* d_scaled = nullptr;
    size_t scaled_bytes = 0;
    // profiling
    uint32_t prof = 0;  // bit mask of enabled MADQP_PROF_* classes
    std::vector<ProfEvent> pending;
    std::vector<hipEvent_t> pool;
    double prof_ms[MADQP_PROF_COUNT] = {0};
    int64_t prof_n[MADQP_PROF_COUNT] = {0};
};

static inline int32_t madqp_fail(madqp_ctx* ctx, int32_t code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return madqp_fail((ctx), MADQP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,          \
                              hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

#define ARG_TRY(ctx, cond)                                                                   \
    do {                                                                                     \
        if (!(cond))                                                                         \
            return madqp_fail((ctx), MADQP_ERR_ARG, "bad argument: %s (%s:%d)", #cond,       \
                              __FILE__, __LINE__);                                           \
    } while (0)

#define LAUNCH_CHECK(ctx) HIP_TRY(ctx, hipGetLastError())

// ctx.hip
int32_t madqp_work_reserve(madqp_ctx* ctx, size_t bytes);
void madqp_prof_begin(madqp_ctx* ctx, int cls);
void madqp_prof_end(madqp_ctx* ctx);
int32_t madqp_read_results(madqp_ctx* ctx, int count, double* out_host);
// queued reductions (vec_kernels.hip, kkt.hip, chol.hip): results stay in ctx->d_res at the given slot until one
// madqp_read_results fetches the block (slots 0 .. MADQP_FAULT_SLOT-1)
int32_t madqp_q_compl(madqp_ctx* ctx, const madqp_state* st, int affine, double ap, double ad, const double* a8,
                      int slot0);                                                               // 2 slots: sums
int32_t madqp_q_alpha_max(madqp_ctx* ctx, const madqp_state* st, double tau, int slot0);        // 8 slots
// the same with a scalar that stays in device memory (mpc.hip, body_fused: no read-back between predictor and corrector)
int32_t madqp_q_alpha_max_dev(madqp_ctx* ctx, const madqp_state* st, double tau, const double* tau_dev, int slot0);
int32_t madqp_set_correction_rhs_dev(madqp_ctx* ctx, const madqp_state* st, const double* mu_dev);
// a8 != nullptr: the step lengths are min(a8[0], a8[2]), min(a8[4], a8[6]) read on the device (as madqp_q_compl's a8)
int32_t madqp_set_extra_correction_dev(madqp_ctx* ctx, const madqp_state* st, double alpha_p, double alpha_d, double beta_min,
                                       double beta_max, const double* mu_dev, const double* a8 = nullptr);
// Gondzio's trial step lengths min(alpha + delta, 1) from the 8-slot step-length block at `in`, written in the pattern
// madqp_q_compl / madqp_set_extra_correction_dev read (slots out+0, +2: primal; out+4, +6: dual)
int32_t madqp_q_mpc_trial_alpha(madqp_ctx* ctx, int in, int out, double delta);
int32_t madqp_q_mpc_mu(madqp_ctx* ctx, int in, int out, int64_t nb, double mu_min, int step_rule, double step_param);
int32_t madqp_q_mpc_muc(madqp_ctx* ctx, int in, int mu_curr_slot, int out, int64_t nb);
// body_fused's decisions behind the corrector, on the device (vec_kernels.hip, mpc_decide_kernel): slots of the block
struct MpcDecide {
    int info, nrm_pred, nrm_corr, alpha, alpha_gz, trial, out;
    int gondzio, max_ncorr, check_residual;
    double tol_linear_solve;
};
int32_t madqp_q_mpc_decide(madqp_ctx* ctx, const MpcDecide* a);
int32_t madqp_copy_if_dev(madqp_ctx* ctx, int64_t len, const double* src, double* dst, const double* flag_dev);
int32_t madqp_update_iterates_dev(madqp_ctx* ctx, const madqp_state* st, const double* dec_dev);
int32_t madqp_adjust_boundary_dev(madqp_ctx* ctx, const madqp_state* st, const double* dec_dev, const double* mu_dev);
// the result block on its way to a pinned host copy, marked by an event: the host waits for the event, not for the stream
int32_t madqp_results_post(madqp_ctx* ctx, double* h_dst, hipEvent_t ev);
int32_t madqp_results_wait(madqp_ctx* ctx, const double* h_src, hipEvent_t ev, int count, double* out_host);
int32_t madqp_q_inf(madqp_ctx* ctx, const madqp_state* st, int slot0);                          // 4 slots
void madqp_inf_from_block(const double* out4, double* out3);
int32_t madqp_q_norm_inf3(madqp_ctx* ctx, int64_t len, const double* a, const double* b, const double* c,
                          int slot0);                                                           // 3 slots
int32_t madqp_q_kkt_eval(madqp_kkt* k, const madqp_state* st, const double* q, const double* rhs, int slot0);  // 2
int32_t madqp_q_kkt_factorize(madqp_kkt* k, int slot0);                                         // 1 slot: info
// w = K^-1 p, optionally pcopy = p (kkt.hip: the condensed mode runs its per-variable passes fused)
int32_t madqp_kkt_solve_from(madqp_kkt* k, const madqp_state* st, const double* p, double* w, double* pcopy);
int32_t madqp_kkt_factor_result(madqp_kkt* k, int32_t info);
int32_t madqp_chol_factor_q(madqp_chol* s, double* A, int64_t lda, double* d_slot);  // info -> *d_slot, no read-back
void madqp_chol_factor_result(madqp_chol* s, int32_t info);                           // what the read-back said

struct ProfScope {
    madqp_ctx* c;
    bool on;
    ProfScope(madqp_ctx* ctx, int cls) : c(ctx), on((ctx->prof >> cls) & 1u) {
        if (on) madqp_prof_begin(c, cls);
    }
    ~ProfScope() {
        if (on) madqp_prof_end(c);
    }
};

// gemm_f64.hip
struct GemmArgs {
    const double* X;  // X[i + k*ldx], i in [0,M)
    int64_t ldx;
    const double* Y;  // Y[j + k*ldy], j in [0,N)
    int64_t ldy;
    double* C;        // out C[i + j*ldc]
    int64_t ldc;
    const double* Cin;  // optional addend, same indexing with ldcin (may alias C)
    int64_t ldcin;
    const double* dvec;  // optional: added where (i + diag_off == j), indexed by j
    double alpha, beta;  // out = alpha*acc + beta*Cin (+ dvec)
    int64_t M, N, K;
    int64_t Mread, Nread;  // rows of X / Y that may be READ (>= M, N; 0 = M, N): operands padded to a
                           // multiple of 128 let edge tiles take the fast path, stores stay masked to M, N
    int64_t diag_off;  // global_row(i) - global_col(j) = i - j + diag_off
    int lower_only;    // 1: write only elements with i + diag_off >= j; skip tiles above
    const int64_t* tile_row0;  // optional HOST array, one entry per 128-column tile of C: the first 128-row tile that is
                               // computed in that tile column (block-cyclic local matrices: "global tile row >= global
                               // tile column" is a per-column suffix, not an affine rule); whole tiles are written
};
// cols: optional host list of ncols column ranges [cols[2r], cols[2r+1]) (multiples of 128 relative to
// C) -- only output tiles whose columns fall in one of them are computed (multi-GPU path: the block
// columns a rank owns, in ONE launch)
// batch: optional -- the same product for B problems laid out at fixed strides (doubles) from the
// pointers of `a` (grid.y = problem); problems with skip[b] != 0 are left untouched
struct GemmBatch {
    int64_t B;
    int64_t sX, sY, sC, sCin, sD;
    const int32_t* skip;
    // compacted form (the masked x100-retry rounds of the batched engine): the launch has B "slots" in grid.y and slot y
    // works off the problems list[y], list[y + B], .. < *count -- with nothing to retry (the usual case) every workgroup
    // of the launch leaves after one load instead of B problems' worth of workgroups reading their skip word
    const int32_t* list = nullptr;
    const int32_t* count = nullptr;
};
int32_t madqp_gemm_tn(madqp_ctx* ctx, const GemmArgs& a, int prof_cls, const int64_t* cols = nullptr,
                      int64_t ncols = 0, const GemmBatch* batch = nullptr);

void madqp_gemm_release_tables(madqp_ctx* ctx);

// chol.hip: recursive blocked factorisation of B equally sized matrices (see there)
int32_t madqp_chol_factor_batched(madqp_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t sA, double* winv,
                                  int64_t sW, int32_t* info, int64_t B, const int32_t* skip, int64_t slots = 0,
                                  const int32_t* list = nullptr, const int32_t* count = nullptr);

// gemv.hip (internal entry with explicit class)
int32_t madqp_gemv_impl(madqp_ctx* ctx, int32_t trans, int64_t rows, int64_t cols, double alpha,
                        const double* A, int64_t lda, const double* x, double beta, double* y,
                        int prof_cls);

// y(n) = alpha H x + beta y for a symmetric H (row r at H + r*ldh) from its LOWER triangle only: half the bytes of the
// general product (gemv.hip)
bool madqp_symv_lower_reads_triangle(int64_t n, const double* H, int64_t ldh);
int32_t madqp_symv_lower(madqp_ctx* ctx, int64_t n, double alpha, const double* H, int64_t ldh, const double* x,
                         double beta, double* y, int prof_cls);
// only the entries H[r*ldh + c] with c >= r are read, at every size (H 16-byte aligned, ldh even)
int32_t madqp_symv_upper(madqp_ctx* ctx, int64_t n, double alpha, const double* H, int64_t ldh, const double* x,
                         double beta, double* y, int prof_cls);

// sparse.hip
int32_t madqp_spmv_csr(madqp_ctx* ctx, int64_t rows, const int64_t* rowptr, const int64_t* col, const double* val,
                       double alpha, const double* x, double beta, double* y, int prof_cls);
// rows of V as CSR (rowptr/col/val) and columns of V as CSR of V' (t_ptr/t_col/t_val)
int32_t madqp_sparse_gram(madqp_ctx* ctx, int64_t n, const int64_t* rowptr, const int64_t* col, const double* val,
                          const int64_t* t_ptr, const int64_t* t_col, const double* t_val, const double* w,
                          const double* base, int64_t ldbase, const double* dvec, double* C, int64_t ldc);

// chol.hip: one triangular sweep over an order-w tile whose factor and inverse diagonal blocks are given (dist.hip):
// trans = 0: v <- L^-1 v, trans = 1: v <- L^-T v.  tmp: w doubles, ctl: 4 ints of device scratch.
bool madqp_chol_panel_sub16_on();
int32_t madqp_chol_panel_solve128(madqp_ctx* ctx, double* X, int64_t ldx, int64_t rows, int64_t rows_read, const double* L,
                                  int64_t ldl, const double* Wcm);
int32_t madqp_trsv_tile(madqp_ctx* ctx, int32_t trans, const double* L, int64_t ld, const double* winv, double* v,
                        int64_t w, double* tmp, int32_t* ctl);

struct madqp_chol {
    madqp_ctx* ctx;
    int64_t n;
    double* winv;  // ceil(n/128) blocks of 128x128 (col-major, ld 128): inverse diagonal blocks
    double* tmp;   // tmp_len doubles: the intermediate vector of a solve, then the sweeps' partial-sum slots
    int64_t tmp_len;
    int32_t* d_jobs;  // sweep job list (chol.hip SweepPlan), nullptr when every block row is one job
    int32_t sweep_chunk, sweep_maxc, sweep_njobs;
    int32_t* d_info;
    double* A;  // last factored matrix (borrowed)
    int64_t lda;
    bool factored;
    int64_t npos;  // quasi-definite mode (madqp_chol_set_signature): A = L diag(I_npos, -I) L'; npos == n: Cholesky
    // mid-size schedule (chol.hip, mid_plan_build): which trailing tiles each block step updates, with which panels;
    // built at the first factorisation.  d_mid_plan: the packed units of all steps (mid_plan.inc); mid_units[k] /
    // mid_units[nblk + k]: number of units of step k / where they start
    uint32_t* d_mid_plan;
    int32_t* mid_units;
    int32_t mid_plan_state;  // state: 0 not built, 1 in use, -1 no plan (the two-panel schedule runs)
    // the backward sweep on U = L' (chol.hip, trsv_fwd_sweep_kernel<1, true>): U of the last mid-size factorisation (order
    // npad, allocated at the first one), and the scratch of such a solve -- [y | x | partial sums of both sweeps], filled
    // with the sentinel by one kernel per solve
    double* upper;
    int64_t upper_ld;
    double* utmp;
    int64_t utmp_len;
    bool upper_ok;  // the last factorisation left U and substitution images
};
