// Fused per-variable kernels of the predictor-corrector loop (src/kernels.jl, src/solver.jl) and
// the MadNLP vector helpers they call.  All are HBM-bound streaming passes; every reduction is a
// wave-shuffle + LDS block reduction into per-block partials followed by a fixed-order final
// pass (deterministic), the scalars land in the context's device result block and are copied to
// the host with one synchronisation.
//
// Arithmetic is written in the reference's evaluation order and the library is compiled with
// -ffp-contract=off, so elementwise results are bit-identical to the numpy oracle; max / min /
// arg-min reductions are order independent, sums are compared to a tolerance.
#include <algorithm>

#include "common.h"

#define MADQP_MAX_BLOCKS 1024
#define TPB 256

namespace {
#define MQ_KERNEL __global__ __launch_bounds__(TPB) void
#define MQ_BLOCK blockIdx.x
#define GRID_STRIDE(i, len) \
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < (len); i += (int64_t)gridDim.x * TPB)
#include "vec_kernels.inc"

inline int grid_for(int64_t len) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((len + TPB - 1) / TPB, MADQP_MAX_BLOCKS));
}

int32_t finalize(madqp_ctx* ctx, int nblocks, int nv, const int* ops, int slot0 = 0) {
    FinalOps f;
    f.nv = nv;
    for (int i = 0; i < nv; ++i) f.op[i] = (int8_t)ops[i];
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part, nblocks, f,
                       ctx->d_res + slot0);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

int32_t check_state(madqp_ctx* ctx, const madqp_state* st) {
    ARG_TRY(ctx, ctx != nullptr);
    ARG_TRY(ctx, st != nullptr);
    ARG_TRY(ctx, st->n >= 0 && st->m >= 0 && st->nlb >= 0 && st->nub >= 0);
    ARG_TRY(ctx, st->nlb == 0 || st->ind_lb);
    ARG_TRY(ctx, st->nub == 0 || st->ind_ub);
    return MADQP_OK;
}
}  // namespace

#define CHECK_STATE()                         \
    do {                                      \
        if (!ctx) return MADQP_ERR_ARG;       \
        int32_t r_ = check_state(ctx, st);    \
        if (r_) return r_;                    \
    } while (0)
#define LAUNCH(kern, len, ...)                                                                  \
    do {                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid_for(len)), dim3(TPB), 0, ctx->stream, __VA_ARGS__);  \
        LAUNCH_CHECK(ctx);                                                                      \
    } while (0)

static inline int64_t max4(int64_t a, int64_t b, int64_t c, int64_t d) {
    return std::max(std::max(a, b), std::max(c, d));
}

extern "C" int32_t madqp_set_aug_diagonal_reg(madqp_ctx* ctx, const madqp_state* st, double del_w,
                                              double del_c) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(aug_diag_fill_kernel, std::max(st->n, st->m), *st, del_w, del_c);
    if (st->nlb) LAUNCH(aug_diag_lb_kernel, st->nlb, *st);
    if (st->nub) LAUNCH(aug_diag_ub_kernel, st->nub, *st);
    return MADQP_OK;
}

static int32_t rhs_launch(madqp_ctx* ctx, const madqp_state* st, int mode, double mu, const double* mu_dev = nullptr) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(rhs_kernel, max4(st->n, st->m, st->nlb, st->nub), *st, mode, mu, mu_dev);
    return MADQP_OK;
}
// set_correction_rhs! with mu taken from device memory (mpc.hip, body_fused)
int32_t madqp_set_correction_rhs_dev(madqp_ctx* ctx, const madqp_state* st, const double* mu_dev) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, mu_dev != nullptr);
    return rhs_launch(ctx, st, 1, 0.0, mu_dev);
}
extern "C" int32_t madqp_set_initial_primal_rhs(madqp_ctx* ctx, const madqp_state* st) {
    return rhs_launch(ctx, st, 2, 0.0);
}
extern "C" int32_t madqp_set_initial_dual_rhs(madqp_ctx* ctx, const madqp_state* st) {
    return rhs_launch(ctx, st, 3, 0.0);
}
extern "C" int32_t madqp_set_predictive_rhs(madqp_ctx* ctx, const madqp_state* st) {
    return rhs_launch(ctx, st, 0, 0.0);
}
extern "C" int32_t madqp_set_correction_rhs(madqp_ctx* ctx, const madqp_state* st, double mu) {
    return rhs_launch(ctx, st, 1, mu);
}

extern "C" int32_t madqp_get_correction(madqp_ctx* ctx, const madqp_state* st) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(correction_kernel, std::max(st->nlb, st->nub), *st);
    return MADQP_OK;
}

extern "C" int32_t madqp_set_extra_correction(madqp_ctx* ctx, const madqp_state* st, double alpha_p,
                                              double alpha_d, double beta_min, double beta_max,
                                              double mu) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(extra_correction_kernel, std::max(st->nlb, st->nub), *st, alpha_p, alpha_d,
           beta_min * mu, beta_max * mu, nullptr, nullptr);
    return MADQP_OK;
}
// the same with mu in device memory: the kernel forms beta_min * mu, beta_max * mu
int32_t madqp_set_extra_correction_dev(madqp_ctx* ctx, const madqp_state* st, double alpha_p, double alpha_d,
                                       double beta_min, double beta_max, const double* mu_dev, const double* a8) {
    CHECK_STATE();
    ARG_TRY(ctx, mu_dev != nullptr);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(extra_correction_kernel, std::max(st->nlb, st->nub), *st, alpha_p, alpha_d, beta_min, beta_max, mu_dev, a8);
    return MADQP_OK;
}
// min(alpha + delta, 1) for the primal and the dual step length of the block at res[in] (std::min's operand order twice,
// as the host forms them), in the 8-slot pattern the step-length readers take
__global__ void mpc_trial_alpha_kernel(double* __restrict__ res, int in, int out, double delta) {
    if (threadIdx.x || blockIdx.x) return;
    const double a0 = res[in], a1 = res[in + 2], a2 = res[in + 4], a3 = res[in + 6];
    const double gp = ((a1 < a0) ? a1 : a0) + delta, gd = ((a3 < a2) ? a3 : a2) + delta;
    const double tp = (1.0 < gp) ? 1.0 : gp, td = (1.0 < gd) ? 1.0 : gd;
    res[out] = res[out + 2] = tp;
    res[out + 4] = res[out + 6] = td;
}
int32_t madqp_q_mpc_trial_alpha(madqp_ctx* ctx, int in, int out, double delta) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, in >= 0 && in + 8 <= MADQP_FAULT_SLOT && out >= 0 && out + 8 <= MADQP_FAULT_SLOT);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(mpc_trial_alpha_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_res, in, out, delta);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

// ---- scalars of the iteration that never visit the host (mpc.hip, body_fused) -------------------------------------------
// update_barrier!(Mehrotra) (src/kernels.jl:226-236) from the four complementarity sums of the result block:
// res[out] = mu, res[out+1] = tau of the step rule, res[out+2] = mu_curr -- the arithmetic of the host form, operation
// for operation ((mu_aff / mu_curr)^3 as t*t*t, which is what Julia's literal power is)
__global__ void mpc_mu_kernel(double* __restrict__ res, int in, int out, double nb, double mu_min, int step_rule,
                              double step_param) {
    if (threadIdx.x || blockIdx.x) return;
    const double mu_affine = nb > 0.0 ? (res[in] + res[in + 1]) / nb : 0.0;
    const double mu_curr = nb > 0.0 ? (res[in + 2] + res[in + 3]) / nb : 0.0;
    double sigma = 1.0;
    if (nb > 0.0) {
        const double t = mu_affine / mu_curr;
        sigma = fmin(fmax(t * t * t, 1e-6), 10.0);
    }
    const double sm = sigma * mu_curr;
    const double mu = (mu_min > sm) ? mu_min : sm;  // std::max(mu_min, sigma * mu_curr)
    res[out] = mu;
    const double om = 1.0 - mu;
    res[out + 1] = (step_rule == 0) ? step_param : ((om > step_param) ? om : step_param);  // std::max(1 - mu, step_param)
    res[out + 2] = mu_curr;
}
// gondzio_correction_direction! (src/solver.jl:200-251): mu_c = (ga / mu_curr)^2 ga from the two sums at res[in]
__global__ void mpc_muc_kernel(double* __restrict__ res, int in, int mu_curr_slot, int out, double nb) {
    if (threadIdx.x || blockIdx.x) return;
    const double ga = nb > 0.0 ? (res[in] + res[in + 1]) / nb : 0.0;
    const double mu_curr = res[mu_curr_slot];
    res[out] = (ga / mu_curr) * (ga / mu_curr) * ga;
}
int32_t madqp_q_mpc_mu(madqp_ctx* ctx, int in, int out, int64_t nb, double mu_min, int step_rule, double step_param) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, in >= 0 && in + 4 <= MADQP_FAULT_SLOT && out >= 0 && out + 3 <= MADQP_FAULT_SLOT);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(mpc_mu_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_res, in, out, (double)nb, mu_min, step_rule,
                       step_param);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
int32_t madqp_q_mpc_muc(madqp_ctx* ctx, int in, int mu_curr_slot, int out, int64_t nb) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, in >= 0 && in + 2 <= MADQP_FAULT_SLOT && out >= 0 && out < MADQP_FAULT_SLOT && mu_curr_slot >= 0 &&
                     mu_curr_slot < MADQP_FAULT_SLOT);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(mpc_muc_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_res, in, mu_curr_slot, out, (double)nb);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

// The decisions body_fused takes from its read-back behind the corrector, taken on the device as well so that the
// update of the iterates can be queued behind the corrector WITHOUT waiting for that read-back (mpc.hip): the residual
// verdicts of both solve_system! calls (src/linear_solver.jl:36-43), the step lengths min(lower, upper) and, with
// Gondzio corrections, whether the first trial is kept (src/solver.jl:231) -- the host's arithmetic, operation for
// operation, on the same words.  res[out] = 1: the iteration goes on as queued (0: the host takes over: failed
// factorisation, failed verdict, or a further trial to run); res[out+1], [out+2] = alpha_p, alpha_d;
// res[out+3] = 1: the trial's direction is dropped, the one saved before it comes back.
__global__ void mpc_decide_kernel(double* __restrict__ res, MpcDecide a) {
    if (threadIdx.x || blockIdx.x) return;
    auto bad = [&](const double* nrm) {
        const double den = (1.0 < nrm[1]) ? nrm[1] : 1.0;  // std::max(1.0, nrm[1])
        const double ratio = nrm[0] / den;
        return (ratio != ratio) || (a.check_residual && ratio > a.tol_linear_solve);
    };
    auto mn = [](double x, double y) { return (y < x) ? y : x; };  // std::min(x, y)
    bool go = res[a.info] == 0.0 && !bad(res + a.nrm_pred) && !bad(res + a.nrm_corr);
    double ap = mn(res[a.alpha + 0], res[a.alpha + 2]), ad = mn(res[a.alpha + 4], res[a.alpha + 6]);
    bool restore = false;
    if (a.gondzio) {
        const double g_ap = mn(res[a.alpha_gz + 0], res[a.alpha_gz + 2]), g_ad = mn(res[a.alpha_gz + 4], res[a.alpha_gz + 6]);
        const double* tb = res + a.trial;  // norms [0..2], steps at Gondzio's tau [3..10], steps at the rule's tau [11..18]
        go = go && !bad(tb);
        const double ha_p = mn(tb[3], tb[5]), ha_d = mn(tb[7], tb[9]);
        if (ha_p < 1.005 * g_ap || ha_d < 1.005 * g_ad) {
            restore = go;
        } else {
            ap = mn(tb[11], tb[13]);
            ad = mn(tb[15], tb[17]);
            if (a.max_ncorr > 1) go = false;  // the next trial is the host's to queue
        }
    }
    res[a.out] = go ? 1.0 : 0.0;
    res[a.out + 1] = ap;
    res[a.out + 2] = ad;
    res[a.out + 3] = restore ? 1.0 : 0.0;
}
int32_t madqp_q_mpc_decide(madqp_ctx* ctx, const MpcDecide* a) {
    if (!ctx || !a) return MADQP_ERR_ARG;
    ARG_TRY(ctx, a->out >= 0 && a->out + 4 <= MADQP_FAULT_SLOT);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(mpc_decide_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_res, *a);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
// dst = src where *flag != 0 (the dropped Gondzio trial's direction goes back, src/solver.jl:232)
__global__ __launch_bounds__(TPB) void copy_if_kernel(int64_t len, const double* __restrict__ src, double* __restrict__ dst,
                                                      const double* __restrict__ flag) {
    if (*flag == 0.0) return;
    GRID_STRIDE(i, len) dst[i] = src[i];
}
int32_t madqp_copy_if_dev(madqp_ctx* ctx, int64_t len, const double* src, double* dst, const double* flag_dev) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0 && (len == 0 || (src && dst)) && flag_dev);
    if (len == 0) return MADQP_OK;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(copy_if_kernel, len, len, src, dst, flag_dev);
    return MADQP_OK;
}
// update_iterates / adjust_boundary with their scalars in the result block, and nothing done when dec[0] == 0
__global__ __launch_bounds__(TPB) void update_iterates_dev_kernel(madqp_state s, const double* __restrict__ dec) {
    if (dec[0] == 0.0) return;
    const double ap = dec[1], ad = dec[2];
    const double* dx = s.d;
    const double* dy = s.d + s.n;
    const double* dzl = dy + s.m;
    const double* dzu = dzl + s.nlb;
    int64_t L = s.n;
    if (s.m > L) L = s.m;
    if (s.nlb > L) L = s.nlb;
    if (s.nub > L) L = s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.n) s.x[i] += ap * dx[i];
        if (i < s.m) s.y[i] += ad * dy[i];
        if (i < s.nlb) s.zl[s.ind_lb[i]] += ad * dzl[i];
        if (i < s.nub) s.zu[s.ind_ub[i]] += ad * dzu[i];
    }
}
__global__ __launch_bounds__(TPB) void adjust_boundary_dev_kernel(madqp_state s, const double* __restrict__ dec,
                                                                  const double* __restrict__ mu_dev, double eps, double c2) {
    if (dec[0] == 0.0) return;
    const double c1 = eps * *mu_dev;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            const double x = s.x[j], l = s.xl[j];
            if (x - l < c1) s.xl[j] = l - c2 * fmax(1.0, fabs(x));
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            const double x = s.x[j], u = s.xu[j];
            if (u - x < c1) s.xu[j] = u + c2 * fmax(1.0, fabs(x));
        }
    }
}
int32_t madqp_update_iterates_dev(madqp_ctx* ctx, const madqp_state* st, const double* dec_dev) {
    CHECK_STATE();
    ARG_TRY(ctx, dec_dev != nullptr);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(update_iterates_dev_kernel, max4(st->n, st->m, st->nlb, st->nub), *st, dec_dev);
    return MADQP_OK;
}
int32_t madqp_adjust_boundary_dev(madqp_ctx* ctx, const madqp_state* st, const double* dec_dev, const double* mu_dev) {
    CHECK_STATE();
    ARG_TRY(ctx, dec_dev && mu_dev);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (std::max(st->nlb, st->nub) > 0)
        LAUNCH(adjust_boundary_dev_kernel, std::max(st->nlb, st->nub), *st, dec_dev, mu_dev, 2.220446049250313e-16,
               1.8189894035458565e-12);
    return MADQP_OK;
}

// Queued forms (madqp_q_*): the reduction lands in the device result block at slot0 and stays there; the caller
// reads several of them back with ONE madqp_read_results (mpc.hip).  The extern "C" entry points below are the
// queued form at slot 0 followed by the read-back.
int32_t madqp_q_compl(madqp_ctx* ctx, const madqp_state* st, int affine, double ap, double ad, const double* a8,
                      int slot0) {
    CHECK_STATE();
    if (st->nlb + st->nub == 0) {  // src/kernels.jl:173-174
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_res + slot0, 0, 2 * sizeof(double), ctx->stream));
        return MADQP_OK;
    }
    const int64_t L = std::max(st->nlb, st->nub);
    const int nb = grid_for(L);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(compl_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, affine, ap, ad, a8, ctx->d_part);
    LAUNCH_CHECK(ctx);
    const int ops[2] = {OP_SUM, OP_SUM};
    return finalize(ctx, nb, 2, ops, slot0);
}

static int32_t compl_launch(madqp_ctx* ctx, const madqp_state* st, int affine, double ap, double ad,
                            double* mu_host) {
    CHECK_STATE();
    ARG_TRY(ctx, mu_host != nullptr);
    if (st->nlb + st->nub == 0) {  // src/kernels.jl:173-174
        *mu_host = 0.0;
        return MADQP_OK;
    }
    int32_t r = madqp_q_compl(ctx, st, affine, ap, ad, nullptr, 0);
    if (r) return r;
    double out[2];
    r = madqp_read_results(ctx, 2, out);
    if (r) return r;
    *mu_host = (out[0] + out[1]) / (double)(st->nlb + st->nub);
    return MADQP_OK;
}
extern "C" int32_t madqp_get_complementarity_measure(madqp_ctx* ctx, const madqp_state* st,
                                                     double* mu_host) {
    return compl_launch(ctx, st, 0, 0.0, 0.0, mu_host);
}
extern "C" int32_t madqp_get_affine_complementarity_measure(madqp_ctx* ctx, const madqp_state* st,
                                                            double alpha_p, double alpha_d,
                                                            double* mu_host) {
    return compl_launch(ctx, st, 1, alpha_p, alpha_d, mu_host);
}

namespace {
__global__ void alpha_none_kernel(double* __restrict__ res) {  // no bounds: alpha = 1, nothing blocks
    if (threadIdx.x < 4) {
        res[2 * threadIdx.x] = 1.0;
        res[2 * threadIdx.x + 1] = -1.0;
    }
}
}  // namespace
int32_t madqp_q_alpha_max_dev(madqp_ctx* ctx, const madqp_state* st, double tau, const double* tau_dev, int slot0);
int32_t madqp_q_alpha_max(madqp_ctx* ctx, const madqp_state* st, double tau, int slot0) {
    return madqp_q_alpha_max_dev(ctx, st, tau, nullptr, slot0);
}
// tau_dev != nullptr: tau is read from device memory
int32_t madqp_q_alpha_max_dev(madqp_ctx* ctx, const madqp_state* st, double tau, const double* tau_dev, int slot0) {
    CHECK_STATE();
    const int64_t L = std::max(st->nlb, st->nub);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (L == 0) {
        hipLaunchKernelGGL(alpha_none_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_res + slot0);
        LAUNCH_CHECK(ctx);
        return MADQP_OK;
    }
    const int nb = grid_for(L);
    hipLaunchKernelGGL(alpha_max_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, tau, ctx->d_part, tau_dev);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(alpha_max_final_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part, nb,
                       ctx->d_res + slot0);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

extern "C" int32_t madqp_get_alpha_max(madqp_ctx* ctx, const madqp_state* st, double tau,
                                       double* alpha_host, int64_t* iblock_host) {
    CHECK_STATE();
    ARG_TRY(ctx, alpha_host && iblock_host);
    const int64_t L = std::max(st->nlb, st->nub);
    if (L == 0) {
        for (int q = 0; q < 4; ++q) {
            alpha_host[q] = 1.0;
            iblock_host[q] = -1;
        }
        return MADQP_OK;
    }
    int32_t r = madqp_q_alpha_max(ctx, st, tau, 0);
    if (r) return r;
    double out[8];
    r = madqp_read_results(ctx, 8, out);
    if (r) return r;
    for (int q = 0; q < 4; ++q) {
        alpha_host[q] = out[2 * q];
        iblock_host[q] = (int64_t)out[2 * q + 1];
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_update_iterates(madqp_ctx* ctx, const madqp_state* st, double alpha_p,
                                         double alpha_d) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(update_iterates_kernel, max4(st->n, st->m, st->nlb, st->nub), *st, alpha_p, alpha_d);
    return MADQP_OK;
}
// 4 slots: |c|, dual, complementarity lower / upper (madqp_inf_from_block combines the last two)
int32_t madqp_q_inf(madqp_ctx* ctx, const madqp_state* st, int slot0) {
    CHECK_STATE();
    const int64_t L = max4(st->n, st->m, st->nlb, st->nub);
    if (L == 0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_res + slot0, 0, 4 * sizeof(double), ctx->stream));
        return MADQP_OK;
    }
    const int nb = grid_for(L);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(inf_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, ctx->d_part);
    LAUNCH_CHECK(ctx);
    const int ops[4] = {OP_MAX, OP_MAX, OP_MAX, OP_MAX};
    return finalize(ctx, nb, 4, ops, slot0);
}
void madqp_inf_from_block(const double* out, double* out3) {
    out3[0] = out[0];
    out3[1] = out[1];
    out3[2] = (out[2] != out[2]) ? out[2] : ((out[3] != out[3]) ? out[3] : std::max(out[2], out[3]));
}

extern "C" int32_t madqp_get_inf(madqp_ctx* ctx, const madqp_state* st, double* out_host) {
    CHECK_STATE();
    ARG_TRY(ctx, out_host != nullptr);
    const int64_t L = max4(st->n, st->m, st->nlb, st->nub);
    if (L == 0) {
        out_host[0] = out_host[1] = out_host[2] = 0.0;
        return MADQP_OK;
    }
    int32_t r = madqp_q_inf(ctx, st, 0);
    if (r) return r;
    double out[4];
    r = madqp_read_results(ctx, 4, out);
    if (r) return r;
    madqp_inf_from_block(out, out_host);
    return MADQP_OK;
}

extern "C" int32_t madqp_adjust_boundary(madqp_ctx* ctx, const madqp_state* st, double mu) {
    CHECK_STATE();
    const double eps = 2.220446049250313e-16;
    const double c1 = eps * mu;
    const double c2 = 1.8189894035458565e-12;  // eps^(3/4)
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (std::max(st->nlb, st->nub) > 0)
        LAUNCH(adjust_boundary_kernel, std::max(st->nlb, st->nub), *st, c1, c2);
    return MADQP_OK;
}

extern "C" int32_t madqp_reduce_rhs(madqp_ctx* ctx, const madqp_state* st, double* w) {
    CHECK_STATE();
    ARG_TRY(ctx, w != nullptr);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (st->nlb)
        LAUNCH(reduce_rhs_kernel, st->nlb, st->nlb, st->ind_lb, w, w + st->n + st->m, st->l_diag);
    if (st->nub)
        LAUNCH(reduce_rhs_kernel, st->nub, st->nub, st->ind_ub, w, w + st->n + st->m + st->nlb,
               st->u_diag);
    return MADQP_OK;
}

extern "C" int32_t madqp_finish_aug_solve(madqp_ctx* ctx, const madqp_state* st, double* w) {
    CHECK_STATE();
    ARG_TRY(ctx, w != nullptr);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (std::max(st->nlb, st->nub) > 0)
        LAUNCH(finish_aug_solve_kernel, std::max(st->nlb, st->nub), *st, w);
    return MADQP_OK;
}

extern "C" int32_t madqp_kktmul(madqp_ctx* ctx, const madqp_state* st, double* w, const double* v,
                                double alpha, double beta) {
    CHECK_STATE();
    ARG_TRY(ctx, w && v);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (std::max(st->n, st->m) > 0)
        LAUNCH(kktmul_diag_kernel, std::max(st->n, st->m), *st, w, v, alpha);
    if (st->nlb) LAUNCH(kktmul_lb_kernel, st->nlb, *st, w, v, alpha, beta);
    if (st->nub) LAUNCH(kktmul_ub_kernel, st->nub, *st, w, v, alpha, beta);
    return MADQP_OK;
}

int32_t madqp_q_norm_inf3(madqp_ctx* ctx, int64_t len, const double* a, const double* b, const double* c,
                          int slot0) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0);
    if (len == 0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_res + slot0, 0, 3 * sizeof(double), ctx->stream));
        return MADQP_OK;
    }
    const int nb = grid_for(len);
    ProfScope ps(ctx, MADQP_PROF_VEC);
    hipLaunchKernelGGL(norm_inf3_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, len, a, b, c, ctx->d_part);
    LAUNCH_CHECK(ctx);
    const int ops[3] = {OP_MAX, OP_MAX, OP_MAX};
    return finalize(ctx, nb, 3, ops, slot0);
}

extern "C" int32_t madqp_norm_inf3(madqp_ctx* ctx, int64_t len, const double* a, const double* b,
                                   const double* c, double* out_host) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0 && out_host);
    if (len == 0) {
        out_host[0] = out_host[1] = out_host[2] = 0.0;
        return MADQP_OK;
    }
    int32_t r = madqp_q_norm_inf3(ctx, len, a, b, c, 0);
    if (r) return r;
    return madqp_read_results(ctx, 3, out_host);
}

extern "C" int32_t madqp_norm_inf(madqp_ctx* ctx, int64_t len, const double* a, double* out_host) {
    double o[3];
    int32_t r = madqp_norm_inf3(ctx, len, a, nullptr, nullptr, o);
    if (r) return r;
    *out_host = o[0];
    return MADQP_OK;
}

extern "C" int32_t madqp_axpy(madqp_ctx* ctx, int64_t len, double alpha, const double* x, double* y) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0 && (len == 0 || (x && y)));
    if (len == 0) return MADQP_OK;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(axpy_kernel, len, len, alpha, x, y);
    return MADQP_OK;
}

extern "C" int32_t madqp_copy(madqp_ctx* ctx, int64_t len, const double* src, double* dst) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0 && (len == 0 || (src && dst)));
    if (len == 0) return MADQP_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, len * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_fill(madqp_ctx* ctx, int64_t len, double value, double* dst) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0 && (len == 0 || dst));
    if (len == 0) return MADQP_OK;
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(fill_kernel, len, len, value, dst);
    return MADQP_OK;
}

// ------------------------------------------------------------ starting point
extern "C" int32_t madqp_sp_init_duals(madqp_ctx* ctx, const madqp_state* st) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (st->n) LAUNCH(sp_init_duals_kernel, st->n, *st);
    return MADQP_OK;
}

extern "C" int32_t madqp_sp_mins(madqp_ctx* ctx, const madqp_state* st, double* out_host) {
    CHECK_STATE();
    ARG_TRY(ctx, out_host != nullptr);
    const int64_t L = std::max(st->nlb, st->nub);
    if (L == 0) {
        for (int q = 0; q < 4; ++q) out_host[q] = 0.0;
        return MADQP_OK;
    }
    const int nb = grid_for(L);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(sp_mins_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, ctx->d_part);
        LAUNCH_CHECK(ctx);
        const int ops[4] = {OP_MIN, OP_MIN, OP_MIN, OP_MIN};
        int32_t r = finalize(ctx, nb, 4, ops);
        if (r) return r;
    }
    return madqp_read_results(ctx, 4, out_host);
}

extern "C" int32_t madqp_sp_shift(madqp_ctx* ctx, const madqp_state* st, double dx, double dz) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (st->nlb) LAUNCH(sp_shift_kernel, st->nlb, st->nlb, st->ind_lb, st->x, dx);
    if (st->nub) LAUNCH(sp_shift_kernel, st->nub, st->nub, st->ind_ub, st->x, -dx);
    if (st->nlb) LAUNCH(sp_shift_kernel, st->nlb, st->nlb, st->ind_lb, st->zl, dz);
    if (st->nub) LAUNCH(sp_shift_kernel, st->nub, st->nub, st->ind_ub, st->zu, dz);
    return MADQP_OK;
}

extern "C" int32_t madqp_sp_sums(madqp_ctx* ctx, const madqp_state* st, double* out_host) {
    CHECK_STATE();
    ARG_TRY(ctx, out_host != nullptr);
    const int64_t L = std::max(st->nlb, st->nub);
    if (L == 0) {
        for (int q = 0; q < 8; ++q) out_host[q] = 0.0;
        return MADQP_OK;
    }
    const int nb = grid_for(L);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(sp_sums_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, ctx->d_part);
        LAUNCH_CHECK(ctx);
        const int ops[8] = {OP_SUM, OP_SUM, OP_SUM, OP_SUM, OP_SUM, OP_SUM, OP_SUM, OP_SUM};
        int32_t r = finalize(ctx, nb, 8, ops);
        if (r) return r;
    }
    return madqp_read_results(ctx, 8, out_host);
}

extern "C" int32_t madqp_sp_project(madqp_ctx* ctx, const madqp_state* st, double kappa) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    if (st->n) LAUNCH(sp_project_kernel, st->n, *st, kappa);
    return MADQP_OK;
}

extern "C" int32_t madqp_sp_check(madqp_ctx* ctx, const madqp_state* st, int32_t* ok_host) {
    CHECK_STATE();
    ARG_TRY(ctx, ok_host != nullptr);
    const int64_t L = std::max(st->nlb, st->nub);
    if (L == 0) {
        *ok_host = 1;
        return MADQP_OK;
    }
    const int nb = grid_for(L);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(sp_check_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, ctx->d_part);
        LAUNCH_CHECK(ctx);
        const int ops[1] = {OP_MAX};
        int32_t r = finalize(ctx, nb, 1, ops);
        if (r) return r;
    }
    double bad = 0.0;
    int32_t r = madqp_read_results(ctx, 1, &bad);
    if (r) return r;
    *ok_host = (bad == 0.0) ? 1 : 0;
    return MADQP_OK;
}
