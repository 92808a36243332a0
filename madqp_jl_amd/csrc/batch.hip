// Batches of small, equally shaped QPs (BASELINE configs[3], SURVEY.md 8e "persistent kernel or per-QP
// convergence mask"): B problems advance in lock step, a handful of launches per iteration for the
// whole batch instead of ~60 launches (and ~12 host round trips) per problem.
//
//   * dense MFMA work is batched: ONE launch of the GEMM core assembles K_b = H_b + Sigma_b + A_b' Theta_b A_b
//     for all b (grid.y = problem), the recursive Cholesky runs as ~10 launches for all b
//     (madqp_chol_factor_batched: diagonal kernel with grid = B, TRSM / update GEMMs with grid.y = B);
//   * everything else of an iteration -- residuals, termination test, right-hand sides, the
//     condensed solves with their triangular sweeps, the residual check of solve_system!, step
//     lengths, centering, iterate update, model callbacks -- is ONE workgroup per problem running the
//     kernel bodies of vec_kernels.inc / kkt_kernels.inc back to back (compiled here as device
//     functions with a workgroup-stride loop): the per-problem scalars (mu, alpha, norms) never leave
//     the chip, and a problem that has converged (or failed) is masked out by its status word;
//   * the host only reads the status words (one small copy per call of madqp_batch_iterate).
//
// Same arithmetic as the single-problem path (src/solver.jl:127-182, 254-345 in the same order);
// sums are accumulated in a different order, so results agree to rounding, not bitwise.  All step
// rules, Gondzio's corrections and the x100 regularization retry of src/linear_solver.jl:6-17 (two extra, masked
// assembly + Cholesky rounds per iteration that only the problems whose factorisation failed take part in) are in,
// and so is the reference's own formulation: normal equations A Sigma^-1 A' of order m (opt.kkt_form = 1, LP only).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {
constexpr int NB = 128;
constexpr int64_t WBLK = 2 * NB * NB;


enum {
    S_MU = 0, S_ALPHA_P, S_ALPHA_D, S_OBJ, S_INF_PR, S_INF_DU, S_INF_COMPL, S_DNORM, S_NORM_B, S_NORM_C,
    S_DEL_W, S_DEL_C, S_RATIO, S_REG_P, S_REG_D, S_NFACT, S_COUNT
};
static_assert(S_COUNT == MADQP_BATCH_SCALARS, "scalar block layout is part of the ABI");

enum { ST_ACTIVE = 0, ST_SOLVED = 1, ST_MAXITER = 6, ST_STEP_ERROR = -3, ST_INTERNAL = -1 };

struct BQ {  // device view of the batch (by value in the kernel arguments); problem b at offset b * length
    int64_t B, nx, m, ns, n, nlb, nub, ntot, ldk, npad, kpad, nblk;
    int64_t dim;     // order of the factorised matrix: nx (condensed) or m (normal equations)
    int32_t normal;  // 1: the reference's NormalKKTSystem (src/KKT/normalkkt.jl), LP only
    const int64_t *ind_lb, *ind_ub, *ind_ineq, *slot;
    const double *H, *A, *q, *rhs, *c0;
    double *x, *xl, *xu, *zl, *zu, *y;
    double *f, *c, *jacl, *reg, *pr_diag, *du_diag, *d, *p, *w1, *w2;
    double *l_diag, *l_lower, *u_diag, *u_lower, *corr_lb, *corr_ub;
    double *theta, *t, *u, *K, *S, *winv, *tmp, *tn;
    double* sym;      // scratch of the symmetric H products (batch_wg.inc: wg_symv_lower), nullptr: full-matrix passes
    int64_t sym_len;  // doubles per problem
    // incremental model evaluation (round 5, MADQP_BATCH_INCR): H x of the current iterate and the two products the LAST
    // solve's residual check formed -- H dx and A' dy, unscaled -- per problem (nx each); nullptr: every iteration
    // evaluates H x, A x and A' y with passes of their own.  incr_ok[b]: the products belong to the step just taken.
    double *hx, *raw_h, *raw_at;
    double* at;  // A' dy of the LAST condensed solve, formed in the pass that formed A dx (batch_wg.inc: wg_gemv_n_then_t); nx each
    int32_t* incr_ok;
    double* scal;
    int32_t *status, *iters, *info, *retry_skip;
    int32_t *retry_list, *retry_count;  // the problems of the current x100-retry round, compacted (bq_retry_kernel appends)
    madqp_mpc_options opt;
    double mu_init, bound_fac;
};

__device__ __forceinline__ madqp_state state_of(const BQ& q, int64_t b) {
    madqp_state s;
    s.n = q.n;
    s.m = q.m;
    s.nlb = q.nlb;
    s.nub = q.nub;
    s.ind_lb = q.ind_lb;
    s.ind_ub = q.ind_ub;
    s.x = q.x + b * q.n;
    s.xl = q.xl + b * q.n;
    s.xu = q.xu + b * q.n;
    s.zl = q.zl + b * q.n;
    s.zu = q.zu + b * q.n;
    s.f = q.f + b * q.n;
    s.y = q.y + b * q.m;
    s.c = q.c + b * q.m;
    s.jacl = q.jacl + b * q.n;
    s.d = q.d + b * q.ntot;
    s.p = q.p + b * q.ntot;
    s.correction_lb = q.corr_lb + b * q.nlb;
    s.correction_ub = q.corr_ub + b * q.nub;
    s.reg = q.reg + b * q.n;
    s.pr_diag = q.pr_diag + b * q.n;
    s.du_diag = q.du_diag + b * q.m;
    s.l_diag = q.l_diag + b * q.nlb;
    s.l_lower = q.l_lower + b * q.nlb;
    s.u_diag = q.u_diag + b * q.nub;
    s.u_lower = q.u_lower + b * q.nub;
    return s;
}

// per-problem pointers that are not part of madqp_state
struct Prob {
    const double *H, *A, *qv, *rhs;
    double *theta, *t, *u, *K, *S, *winv, *tmp, *tn, *w1, *scal, *sym;
    double *hx, *raw_h, *raw_at, *at;
    double c0;
};
__device__ __forceinline__ Prob prob_of(const BQ& q, int64_t b) {
    Prob p;
    p.H = q.H ? q.H + b * q.nx * q.nx : nullptr;
    p.A = q.A + b * q.m * q.nx;
    p.qv = q.q + b * q.nx;
    p.rhs = q.rhs + b * q.m;
    p.theta = q.theta + b * q.m;
    p.t = q.t + b * q.m;
    p.u = q.u + b * q.m;
    p.K = q.K + b * q.ldk * q.ldk;
    p.S = q.S + b * q.kpad * q.npad;
    p.winv = q.winv + b * q.nblk * WBLK;
    p.tmp = q.tmp + b * q.npad;
    p.tn = q.tn ? q.tn + b * q.n : nullptr;
    p.sym = q.sym ? q.sym + b * q.sym_len : nullptr;
    p.hx = q.hx ? q.hx + b * q.nx : nullptr;
    p.raw_h = q.hx ? q.raw_h + b * q.nx : nullptr;
    p.raw_at = q.hx ? q.raw_at + b * q.nx : nullptr;
    p.at = q.at ? q.at + b * q.nx : nullptr;
    p.w1 = q.w1 + b * q.ntot;
    p.scal = q.scal + b * S_COUNT;
    p.c0 = q.c0[b];
    return p;
}

// The workgroup programs are compiled twice (batch_wg.inc): 256 threads per problem for large batches
// (many problems per CU, HBM bound) and 512 threads per problem for small ones (1024 would spill), where the time of a
// lock-step iteration is the serial time of ONE problem's vector work.
#define TPB 256
#define WGNS wg256
#ifdef MADQP_BATCH_STAMPS
__device__ unsigned long long madqp_batch_stamps[32];  // [31] = last stamp; slot 0 = everything not listed
#endif
#include "batch_wg.inc"
#undef TPB
#undef WGNS
#define TPB 512
#define WGNS wg512
#include "batch_wg.inc"
#undef TPB
#undef WGNS

__global__ void bq_count_active_kernel(const int32_t* __restrict__ status, int64_t B, int32_t* out) {
    int cnt = 0;
    for (int64_t i = threadIdx.x; i < B; i += blockDim.x) cnt += (status[i] == ST_ACTIVE);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (threadIdx.x == 0) *out = cnt;
}
}  // namespace

struct madqp_batch {
    madqp_ctx* ctx;
    BQ q;
    std::vector<void*> owned;
    int32_t* d_active;
    // one lock-step iteration (13 launches + 2 masked retry rounds) captured once as a hipGraph and replayed on an internal stream
    hipStream_t sG = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t evIn = nullptr, evOut = nullptr;
    int graph_state = 0;  // 0 not tried, 1 ready, -1 unavailable
    bool wide;  // 512 threads per problem (small batches)
};

namespace {
template <class T>
int32_t dalloc(madqp_batch* b, T** p, int64_t count, bool zero = false) {
    *p = nullptr;
    const size_t bytes = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess)
        return madqp_fail(b->ctx, MADQP_ERR_ALLOC, "madqp_batch_create: hipMalloc(%zu): %s", bytes,
                          hipGetErrorString(e));
    b->owned.push_back(*p);
    if (zero) {
        e = hipMemsetAsync(*p, 0, bytes, b->ctx->stream);
        if (e != hipSuccess) return madqp_fail(b->ctx, MADQP_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
    }
    return MADQP_OK;
}

// build_kkt! + factorize! for every active problem: one assembly launch, ~2 nx/128 launches of Cholesky
// retry == true: the masked x100-retry rounds -- the launches have RETRY_SLOTS problem slots in place of B problems and
// work off the compacted list bq_retry_kernel has just written (usually empty: every workgroup leaves after one load;
// round 3 launched B problems' worth of workgroups that each read a skip word: 1.8 % of the batch's time)
constexpr int64_t RETRY_SLOTS = 8;
int32_t factor_all(madqp_batch* b, const int32_t* skip, bool retry = false) {
    madqp_ctx* ctx = b->ctx;
    const BQ& q = b->q;
    if (q.dim == 0) return hipMemsetAsync(q.info, 0, q.B * sizeof(int32_t), ctx->stream) == hipSuccess
                               ? MADQP_OK
                               : MADQP_ERR_HIP;
    GemmArgs g{};
    g.X = q.S;
    g.ldx = q.npad;
    g.Y = q.S;
    g.ldy = q.npad;
    g.K = q.kpad;
    g.Mread = g.Nread = q.npad;
    g.C = q.K;
    g.ldc = q.ldk;
    g.Cin = q.normal ? nullptr : q.H;
    g.ldcin = q.nx;
    g.dvec = q.normal ? q.theta : q.pr_diag;  // normal equations: Sigma_s^-1 on the inequality rows
    g.alpha = 1.0;
    g.beta = 1.0;
    g.M = q.dim;
    g.N = q.dim;
    g.lower_only = 1;
    if ((q.normal ? q.nx : q.m) == 0) {  // empty product: K = base + diagonal through a K = 0 product
        g.K = 0;
        g.X = g.Y = q.K;
        g.Mread = g.Nread = 0;
    }
    static const bool compact = !(getenv("MADQP_BATCH_RETRY_COMPACT") && atoi(getenv("MADQP_BATCH_RETRY_COMPACT")) == 0);
    const bool lst = retry && compact;
    GemmBatch bt{lst ? RETRY_SLOTS : q.B, q.kpad * q.npad, q.kpad * q.npad, q.ldk * q.ldk, q.nx * q.nx, q.normal ? q.m : q.n,
                 lst ? nullptr : skip, lst ? q.retry_list : nullptr, lst ? q.retry_count : nullptr};
    int32_t r = madqp_gemm_tn(ctx, g, MADQP_PROF_SYRK, nullptr, 0, &bt);
    if (r) return r;
    return madqp_chol_factor_batched(ctx, q.K, q.ldk, q.dim, q.ldk * q.ldk, q.winv, q.nblk * WBLK, q.info, q.B, skip,
                                     RETRY_SLOTS, lst ? q.retry_list : nullptr, lst ? q.retry_count : nullptr);
}
}  // namespace

extern "C" int32_t madqp_batch_destroy(madqp_batch* b) {
    if (!b) return MADQP_OK;
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->sG) (void)hipStreamSynchronize(b->sG);
    if (b->exec) (void)hipGraphExecDestroy(b->exec);
    if (b->graph) (void)hipGraphDestroy(b->graph);
    if (b->evIn) (void)hipEventDestroy(b->evIn);
    if (b->evOut) (void)hipEventDestroy(b->evOut);
    if (b->sG) (void)hipStreamDestroy(b->sG);
    for (void* p : b->owned) (void)hipFree(p);
    delete b;
    return MADQP_OK;
}

extern "C" int32_t madqp_batch_create(madqp_ctx* ctx, int64_t B, int64_t nx, int64_t m, int64_t ns,
                                      const int64_t* ind_ineq_host, int64_t nlb, const int64_t* ind_lb,
                                      int64_t nub, const int64_t* ind_ub, const madqp_batch_data* data,
                                      const madqp_mpc_options* opt, madqp_batch** out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, out && data && opt && B >= 1 && B <= 65535 && nx >= 0 && m >= 0 && ns >= 0 && ns <= m);
    ARG_TRY(ctx, nlb >= 0 && nub >= 0 && (nlb == 0 || ind_lb) && (nub == 0 || ind_ub) && (ns == 0 || ind_ineq_host));
    ARG_TRY(ctx, (nx == 0 || (data->q && data->x && data->xl && data->xu && data->zl && data->zu)) &&
                     (m == 0 || (data->A && data->rhs && data->y)) && data->c0);
    ARG_TRY(ctx, opt->step_rule >= 0 && opt->step_rule <= 2 && opt->max_ncorr >= 0 && opt->regularization >= 0 &&
                     opt->regularization <= 2);
    const int32_t normal = opt->kkt_form;
    ARG_TRY(ctx, normal == 0 || normal == 1);
    ARG_TRY(ctx, !normal || !data->H);  // NormalKKTSystem supports only linear programs (src/KKT/normalkkt.jl:45-48)
    // the condensed form needs delta_d < 0 on equality rows (conventions)
    ARG_TRY(ctx, normal || ns == m || (opt->regularization != 0 && opt->delta_d < 0.0));
    *out = nullptr;
    madqp_batch* b = new (std::nothrow) madqp_batch();
    if (!b) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    b->ctx = ctx;
    // measured at (512, 256): 512 threads per problem are faster up to B = 512 and equal at 1024
    static const int wide_max = getenv("MADQP_BATCH_WIDE_MAX") ? atoi(getenv("MADQP_BATCH_WIDE_MAX")) : 1024;
    b->wide = B <= wide_max;
    BQ& q = b->q;
    memset(&q, 0, sizeof(q));
    q.B = B;
    q.nx = nx;
    q.m = m;
    q.ns = ns;
    q.n = nx + ns;
    q.nlb = nlb;
    q.nub = nub;
    q.ntot = q.n + m + nlb + nub;
    q.normal = normal;
    q.dim = normal ? m : nx;
    q.npad = std::max<int64_t>(128, (q.dim + 127) / 128 * 128);
    q.ldk = q.npad;
    q.kpad = std::max<int64_t>(16, ((normal ? nx : m) + 15) / 16 * 16);
    q.nblk = q.npad / 128;
    q.ind_lb = ind_lb;
    q.ind_ub = ind_ub;
    q.H = data->H;
    q.A = data->A;
    q.q = data->q;
    q.rhs = data->rhs;
    q.c0 = data->c0;
    q.x = data->x;
    q.xl = data->xl;
    q.xu = data->xu;
    q.zl = data->zl;
    q.zu = data->zu;
    q.y = data->y;
    q.opt = *opt;
    std::vector<int64_t> slot((size_t)std::max<int64_t>(m, 1), -1);
    for (int64_t k = 0; k < ns; ++k) {
        const int64_t r = ind_ineq_host[k];
        if (!(r >= 0 && r < m && slot[r] < 0 && (k == 0 || ind_ineq_host[k - 1] < r))) {
            delete b;
            return madqp_fail(ctx, MADQP_ERR_ARG, "ind_ineq must be strictly increasing row indices");
        }
        slot[r] = k;
    }
    int32_t r = MADQP_OK;
    int64_t *d_ineq = nullptr, *d_slot = nullptr;
#define BALLOC(ptr, count, ...)                                   \
    if (r == MADQP_OK) r = dalloc(b, &(ptr), (count), ##__VA_ARGS__)
    BALLOC(d_ineq, ns);
    BALLOC(d_slot, m);
    BALLOC(q.f, B * q.n);
    BALLOC(q.c, B * m);
    BALLOC(q.jacl, B * q.n);
    BALLOC(q.reg, B * q.n);
    BALLOC(q.pr_diag, B * q.n);
    BALLOC(q.du_diag, B * m);
    BALLOC(q.d, B * q.ntot);
    BALLOC(q.p, B * q.ntot);
    BALLOC(q.w1, B * q.ntot);
    BALLOC(q.w2, opt->max_ncorr > 0 ? B * q.ntot : 1);
    BALLOC(q.l_diag, B * nlb);
    BALLOC(q.l_lower, B * nlb);
    BALLOC(q.u_diag, B * nub);
    BALLOC(q.u_lower, B * nub);
    BALLOC(q.corr_lb, B * nlb);
    BALLOC(q.corr_ub, B * nub);
    BALLOC(q.theta, B * m);
    BALLOC(q.t, B * m);
    BALLOC(q.u, B * m);
    BALLOC(q.K, B * q.ldk * q.ldk + 128, true);
    BALLOC(q.S, B * q.kpad * q.npad);
    BALLOC(q.winv, B * q.nblk * WBLK, true);  // potf2_inv_kernel writes the lower parts only
    BALLOC(q.tmp, B * q.npad);
    if (normal) BALLOC(q.tn, B * q.n);
    {  // H products from the lower triangle (MADQP_BATCH_SYMV=0: full-matrix passes)
        static const bool symv = !(getenv("MADQP_BATCH_SYMV") && atoi(getenv("MADQP_BATCH_SYMV")) == 0);
        q.sym_len = (512 / 64 + 1) * 512;  // SYM_DOUBLES of the widest workgroup program
        if (symv && data->H && nx > 0 && nx <= 512) BALLOC(q.sym, B * q.sym_len);
    }
    {   // incremental model evaluation: condensed form without Gondzio corrections (a rejected trial would leave the
        // products of a direction that was not taken)
        static const bool incr = !(getenv("MADQP_BATCH_INCR") && atoi(getenv("MADQP_BATCH_INCR")) == 0);
        if (incr && !normal && opt->max_ncorr == 0 && nx > 0) {
            BALLOC(q.hx, B * nx, true);
            BALLOC(q.raw_h, B * nx, true);
            BALLOC(q.raw_at, B * nx, true);
        }
        BALLOC(q.incr_ok, B, true);
        // A dx and A' dy of a condensed solve in ONE pass over A (the workgroup programs are HBM bound; the residual check
        // of solve_system! then has no pass of its own over A): needs a row of A in a wave's registers, nx <= 512
        static const bool nt = !(getenv("MADQP_BATCH_NT") && atoi(getenv("MADQP_BATCH_NT")) == 0);
        if (nt && !normal && nx > 0 && nx <= 512 && m > 0) BALLOC(q.at, B * nx, true);
    }
    BALLOC(q.scal, B * S_COUNT, true);
    BALLOC(q.status, B, true);
    BALLOC(q.iters, B, true);
    BALLOC(q.info, B, true);
    BALLOC(q.retry_skip, B, true);
    BALLOC(q.retry_list, B, true);
    BALLOC(q.retry_count, 4, true);
    BALLOC(b->d_active, 1, true);
#undef BALLOC
    if (r == MADQP_OK && ns &&
        hipMemcpy(d_ineq, ind_ineq_host, ns * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
        r = madqp_fail(ctx, MADQP_ERR_HIP, "copy of ind_ineq failed");
    if (r == MADQP_OK && m && hipMemcpy(d_slot, slot.data(), m * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
        r = madqp_fail(ctx, MADQP_ERR_HIP, "copy of the slack map failed");
    if (r != MADQP_OK) {
        madqp_batch_destroy(b);
        return r;
    }
    q.ind_ineq = d_ineq;
    q.slot = d_slot;
    *out = b;
    return MADQP_OK;
}

// src/solver.jl:162-179 for every problem (the caller has done :127-159: bounds, interior push, scaling)
extern "C" int32_t madqp_batch_init(madqp_batch* b, double mu_init, double bound_fac) {
    if (!b) return MADQP_ERR_ARG;
    madqp_ctx* ctx = b->ctx;
    b->q.mu_init = mu_init;
    b->q.bound_fac = bound_fac;
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (b->wide)
            hipLaunchKernelGGL(wg512::bq_init_pre_kernel, dim3((unsigned)b->q.B), dim3(512), 0, ctx->stream, b->q);
        else
            hipLaunchKernelGGL(wg256::bq_init_pre_kernel, dim3((unsigned)b->q.B), dim3(256), 0, ctx->stream, b->q);
        LAUNCH_CHECK(ctx);
    }
    int32_t r = factor_all(b, b->q.status);
    if (r) return r;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (b->wide)
        hipLaunchKernelGGL(wg512::bq_init_post_kernel, dim3((unsigned)b->q.B), dim3(512), 0, ctx->stream, b->q);
    else
        hipLaunchKernelGGL(wg256::bq_init_post_kernel, dim3((unsigned)b->q.B), dim3(256), 0, ctx->stream, b->q);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

// one lock-step iteration on ctx->stream: loop head + operands, assembly + Cholesky, the rest
static int32_t launch_iteration(madqp_batch* b) {
    madqp_ctx* ctx = b->ctx;
    const BQ& q = b->q;
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        if (b->wide)
            hipLaunchKernelGGL(wg512::bq_iter_pre_kernel, dim3((unsigned)q.B), dim3(512), 0, ctx->stream, q);
        else
            hipLaunchKernelGGL(wg256::bq_iter_pre_kernel, dim3((unsigned)q.B), dim3(256), 0, ctx->stream, q);
        LAUNCH_CHECK(ctx);
    }
    int32_t r = factor_all(b, q.status);
    if (r) return r;
    for (int trial = 1; trial < 3; ++trial) {  // src/linear_solver.jl:7: three trials in all
        HIP_TRY(ctx, hipMemsetAsync(q.retry_count, 0, sizeof(int32_t), ctx->stream));
        {
            ProfScope ps(ctx, MADQP_PROF_VEC);
            if (b->wide)
                hipLaunchKernelGGL(wg512::bq_retry_kernel, dim3((unsigned)q.B), dim3(512), 0, ctx->stream, q);
            else
                hipLaunchKernelGGL(wg256::bq_retry_kernel, dim3((unsigned)q.B), dim3(256), 0, ctx->stream, q);
            LAUNCH_CHECK(ctx);
        }
        if ((r = factor_all(b, q.retry_skip, true))) return r;
    }
    ProfScope ps(ctx, MADQP_PROF_VEC);
    const bool gz = q.opt.max_ncorr > 0;
    if (b->wide && gz)
        hipLaunchKernelGGL(wg512::bq_iter_post_kernel<true>, dim3((unsigned)q.B), dim3(512), 0, ctx->stream, q);
    else if (b->wide)
        hipLaunchKernelGGL(wg512::bq_iter_post_kernel<false>, dim3((unsigned)q.B), dim3(512), 0, ctx->stream, q);
    else if (gz)
        hipLaunchKernelGGL(wg256::bq_iter_post_kernel<true>, dim3((unsigned)q.B), dim3(256), 0, ctx->stream, q);
    else
        hipLaunchKernelGGL(wg256::bq_iter_post_kernel<false>, dim3((unsigned)q.B), dim3(256), 0, ctx->stream, q);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

// Captures launch_iteration into a graph (MADQP_BATCH_GRAPH=0: off).  The launches are identical from one
// iteration to the next (problems drop out through their status words, not through the grid), the tile
// tables they need exist since madqp_batch_init, and nothing in between touches the host.
static bool graph_ready(madqp_batch* b) {
    static const int enabled = getenv("MADQP_BATCH_GRAPH") ? atoi(getenv("MADQP_BATCH_GRAPH")) : 1;
    madqp_ctx* ctx = b->ctx;
    if (!enabled || ctx->prof != 0 || b->graph_state < 0) return false;  // profiling events are not capturable
    if (b->graph_state == 1) return true;
    b->graph_state = -1;
    if (hipStreamCreateWithFlags(&b->sG, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&b->evIn, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&b->evOut, hipEventDisableTiming) != hipSuccess) return false;
    (void)hipStreamSynchronize(ctx->stream);
    hipStream_t saved = ctx->stream;
    ctx->stream = b->sG;
    bool ok = hipStreamBeginCapture(b->sG, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        const int32_t r = launch_iteration(b);
        const hipError_t e = hipStreamEndCapture(b->sG, &b->graph);
        ok = (r == MADQP_OK) && (e == hipSuccess) && b->graph;
    }
    ctx->stream = saved;
    if (ok) ok = hipGraphInstantiate(&b->exec, b->graph, nullptr, nullptr, 0) == hipSuccess;
    (void)hipGetLastError();  // a failed capture must not poison later launch checks
    if (getenv("MADQP_BATCH_GRAPH_VERBOSE")) fprintf(stderr, "madqp batch graph: %s\n", ok ? "captured" : "capture failed, direct launches");
    if (!ok) return false;
    b->graph_state = 1;
    return true;
}

// Up to max_steps lock-step iterations of mpc! (src/solver.jl:254-345); stops as soon as no problem
// is active (checked every `check_every` steps with one 4-byte read-back).  n_active_host: problems
// still active on return.
extern "C" int32_t madqp_batch_iterate(madqp_batch* b, int32_t max_steps, int32_t check_every,
                                       int32_t* n_active_host) {
    if (!b) return MADQP_ERR_ARG;
    madqp_ctx* ctx = b->ctx;
    ARG_TRY(ctx, max_steps >= 0 && check_every >= 1 && n_active_host);
    const BQ& q = b->q;
    const bool graph = max_steps > 0 && graph_ready(b);
    hipStream_t st = graph ? b->sG : ctx->stream;
    if (graph) {
        HIP_TRY(ctx, hipEventRecord(b->evIn, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(b->sG, b->evIn, 0));
    }
    auto count_active = [&](int32_t* active) -> int32_t {
        hipLaunchKernelGGL(bq_count_active_kernel, dim3(1), dim3(64), 0, st, q.status, q.B, b->d_active);
        LAUNCH_CHECK(ctx);
        HIP_TRY(ctx, hipMemcpyAsync(active, b->d_active, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        return MADQP_OK;
    };
    int32_t active = -1, r = MADQP_OK;
    for (int32_t it = 0; it < max_steps && !r; ++it) {
        if (graph) {
            if (hipGraphLaunch(b->exec, b->sG) != hipSuccess) r = madqp_fail(ctx, MADQP_ERR_HIP, "hipGraphLaunch failed");
        } else {
            r = launch_iteration(b);
        }
        if (!r && ((it + 1) % check_every == 0 || it + 1 == max_steps)) {
            r = count_active(&active);
            if (!r && active == 0) break;
        }
    }
    if (!r && active < 0) r = count_active(&active);
    if (graph) {  // later work on the context's stream is ordered after the replays
        (void)hipEventRecord(b->evOut, b->sG);
        (void)hipStreamWaitEvent(ctx->stream, b->evOut, 0);
    }
    if (r) return r;
    *n_active_host = active;
    return MADQP_OK;
}

// status (0 active, 1 SOLVE_SUCCEEDED, 6 MAXIMUM_ITERATIONS_EXCEEDED, -3 ERROR_IN_STEP_COMPUTATION,
// -1 INTERNAL_ERROR), iteration count and the MADQP_BATCH_SCALARS scalars of every problem
extern "C" int32_t madqp_batch_results(madqp_batch* b, int32_t* status_host, int32_t* iters_host,
                                       double* scal_host) {
    if (!b) return MADQP_ERR_ARG;
    madqp_ctx* ctx = b->ctx;
    const BQ& q = b->q;
    if (status_host)
        HIP_TRY(ctx, hipMemcpyAsync(status_host, q.status, q.B * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (iters_host)
        HIP_TRY(ctx, hipMemcpyAsync(iters_host, q.iters, q.B * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (scal_host)
        HIP_TRY(ctx, hipMemcpyAsync(scal_host, q.scal, q.B * S_COUNT * sizeof(double), hipMemcpyDeviceToHost,
                                    ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MADQP_OK;
}

#ifdef MADQP_BATCH_STAMPS
extern "C" int32_t madqp_batch_read_stamps(unsigned long long* out32, int32_t reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(madqp_batch_stamps), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(madqp_batch_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
