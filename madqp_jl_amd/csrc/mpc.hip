// Native host driver of one Mehrotra predictor-corrector iteration (src/solver.jl:259-343).
//
// The same control flow as madqp_jl_amd/solver.py (MPCSolver.iteration_head / iteration_body), in
// C++ above the kernels of this library: one foreign call per iteration instead of ~60, so the host
// overhead of an interpreted driver disappears (it matters for small problems and for batches,
// where each host thread drives its own context).
//
// Read-backs.  The sequential form (body_sequential: every reduction returns its scalar at once) makes about
// ten per iteration, thirteen to sixteen with Gondzio corrections.  The fused form (body_fused; AdaptiveStep /
// ConservativeStep) queues the reductions of a phase in the context's result block and fetches the block TWICE per
// iteration, plus once per tried Gondzio correction beyond the first:
//   (b) behind the corrector: factorisation info; residual norms of both solve_system! calls, the predictor's step-length
//       minima and the complementarity sums, from which a one-thread kernel (mpc_mu_kernel) has formed sigma, mu and
//       the step rule's tau ON THE DEVICE -- the corrector's right-hand side and step-length kernels read them there --
//       then the corrector's step-length minima and |dx|  -> SolveException test BEFORE the iterates move; likewise
//       a Gondzio trial's mu_c (mpc_muc_kernel) and, for the FIRST trial, its step lengths min(alpha + delta, 1)
//       (mpc_trial_alpha_kernel): that trial is queued behind the corrector and decided from the same block; a
//       further trial costs the one read-back that decides whether it is kept;
//   (c) behind the update: objective sums and the residual norms of the NEXT termination test (madqp_mpc_head then
//       finds them cached).
// A blocking read-back is a 28 us round trip, and the launches behind it start from an empty queue -- 3.7 us each from
// the host against 1.6 us when they are already queued (tools/launch_probe.cpp) -- ~0.1 ms at n_x = 5 000 each, plus
// the driver's way around the loop behind (c).  Round 5: neither fetch leaves the queue empty any more.  The decisions
// of (b) are ALSO taken by a one-thread kernel (mpc_decide_kernel: the host's arithmetic on the same words), phase
// (c) is queued behind it as kernels that do nothing unless that kernel said "go on" (update_iterates / adjust_boundary
// are the two with side effects; the evaluations behind them recompute what they would have anyway), the block is
// copied to the host behind the decision AND behind (c), and the host waits for the two events: for (b) while (c)
// runs, and for (c) after it has queued the NEXT iteration's set_aug_diagonal_reg! and build_kkt! (their inputs are
// final once (c) has run, the regularisation is host arithmetic; K is rebuilt from scratch every pass, so a loop that
// ends here has lost one assembly).  The host replays every decision from the block and fails loudly if the device's
// differ; where the device said "the host takes over" (failed factorisation or verdict, a second Gondzio trial) the
// iterates have not moved and the round-4 flow continues from the same block.  MADQP_MPC_AHEAD=0: the round-4 form.
// Same kernels, the same arithmetic operation for operation (on the device where the scalar stays there), same order on
// the stream: the iterates are bitwise those of the sequential form (tests/test_gpu_solver.py).  A failed first factorisation (the x100 retries of src/linear_solver.jl:6-17)
// sends the rest of that iteration down the sequential form.
#include <cmath>
#include <cstdlib>

#include "common.h"

struct madqp_mpc {
    madqp_kkt* kkt;
    madqp_ctx* ctx;
    madqp_state st;
    double *w1, *w2;
    const double *q, *rhs;
    double c0, norm_b, norm_c;
    madqp_mpc_options opt;
    // mutable solver scalars (src/structure.jl:60-75)
    double mu, del_w, del_c, alpha_p, alpha_d, obj, inf_pr, inf_du, inf_compl, dnorm, residual_ratio;
    double reg_delta_p, reg_delta_d;  // AdaptiveRegularization state (src/kernels.jl:410-417)
    int64_t k, n_factorizations;
    int32_t last_info;
    bool fused;        // body_fused eligible (options) and not switched off (MADQP_MPC_FUSED=0)
    bool head_cached;  // nrm_cached = the residual norms of the current iterate, read with the last body's block (c)
    double nrm_cached[3];
    int64_t n_readbacks;  // blocking scalar read-backs issued by head/body (reported for DESIGN's count)
    // body_fused, queued ahead of its read-backs (see the head of this file)
    bool ahead;            // MADQP_MPC_AHEAD != 0
    double *h_early, *h_final;  // pinned copies of the result block
    hipEvent_t ev_early, ev_final;
    bool moved;            // a pass has completed since the scalars were set: f, c are those of the current x
    bool prebuilt;         // the NEXT iteration's diagonal and K are already queued, with these regularisations:
    double pre_del_w, pre_del_c;
    int64_t n_prebuilt, n_prebuilt_used;
};

madqp_ctx* madqp_kkt_ctx(madqp_kkt* kkt);  // kkt.hip

namespace {
#define TRY(expr)               \
    do {                        \
        int32_t r_ = (expr);    \
        if (r_) return r_;      \
    } while (0)

int64_t ntot(const madqp_state& s) { return s.n + s.m + s.nlb + s.nub; }

// src/kernels.jl:386-417
void update_regularization(madqp_mpc* s) {
    switch (s->opt.regularization) {
        case 0:
            s->del_w = 0.0;
            s->del_c = 0.0;
            break;
        case 1:
            s->del_w = s->opt.delta_p;
            s->del_c = s->opt.delta_d;
            break;
        default:
            s->reg_delta_p = std::max(s->reg_delta_p / 10.0, s->opt.delta_min);
            s->reg_delta_d = std::min(s->reg_delta_d / 10.0, -s->opt.delta_min);
            s->del_w = s->reg_delta_p;
            s->del_c = s->reg_delta_d;
    }
}

// src/linear_solver.jl:6-17
int32_t factorize_regularized_system(madqp_mpc* s, int first_trial = 0) {
    for (int trial = first_trial; trial < 3; ++trial) {
        TRY(madqp_kkt_set_aug_diagonal_reg(s->kkt, &s->st, s->del_w, s->del_c));  // dispatched on the KKT type
        TRY(madqp_kkt_build(s->kkt, &s->st));
        TRY(madqp_kkt_factorize(s->kkt, &s->last_info));
        s->n_factorizations += 1;
        if (s->last_info == 0) break;
        s->del_w *= 100.0;
        s->del_c *= 100.0;
    }
    return MADQP_OK;
}

// src/linear_solver.jl:19-45
int32_t solve_system(madqp_mpc* s) {
    const int64_t len = ntot(s->st);
    TRY(madqp_kkt_solve_from(s->kkt, &s->st, s->st.p, s->st.d, s->w1));   // d = K^-1 p, w1 = p
    TRY(madqp_kkt_mul_solved(s->kkt, &s->st, s->w1, s->st.d, -1.0, 1.0));  // d: the solve's result, untouched
    for (int32_t it = 0; it < s->opt.refine_steps; ++it) {  // extension (off by default): d += K^-1 (p - K d)
        TRY(madqp_kkt_solve(s->kkt, &s->st, s->w1));
        TRY(madqp_axpy(s->ctx, len, 1.0, s->w1, s->st.d));
        TRY(madqp_copy(s->ctx, len, s->st.p, s->w1));
        TRY(madqp_kkt_mul(s->kkt, &s->st, s->w1, s->st.d, -1.0, 1.0));
    }
    double nrm[3];
    TRY(madqp_norm_inf3(s->ctx, len, s->w1, s->st.p, s->st.d, nrm));
    const double ratio = nrm[0] / std::max(1.0, nrm[1]);
    s->residual_ratio = ratio;
    if (std::isnan(ratio) || (s->opt.check_residual && ratio > s->opt.tol_linear_solve))
        return MADQP_NUM_NAN;  // MadNLP.SolveException
    return MADQP_OK;
}

// src/kernels.jl:290-305
int32_t fraction_to_boundary(madqp_mpc* s, double tau, double* ap, double* ad, double* a4 = nullptr,
                             int64_t* i4 = nullptr) {
    double a[4];
    int64_t ib[4];
    TRY(madqp_get_alpha_max(s->ctx, &s->st, tau, a, ib));
    *ap = std::min(a[0], a[1]);
    *ad = std::min(a[2], a[3]);
    if (a4)
        for (int q = 0; q < 4; ++q) {
            a4[q] = a[q];
            i4[q] = ib[q];
        }
    return MADQP_OK;
}

int32_t read1(madqp_mpc* s, const double* dptr, double* out) {
    return madqp_memcpy_d2h(s->ctx, out, dptr, sizeof(double));
}
int32_t read_idx(madqp_mpc* s, const int64_t* dptr, int64_t* out) {
    return madqp_memcpy_d2h(s->ctx, out, dptr, sizeof(int64_t));
}

// update_step!(::MehrotraAdaptiveStep), src/kernels.jl:325-374 (scalar reads at the blocking indices)
int32_t mehrotra_adaptive_step(madqp_mpc* s) {
    const madqp_state& st = s->st;
    const double gamma_f = s->opt.step_param, gamma_a = 1.0 / (1.0 - gamma_f);
    double a[4], max_ap, max_ad;
    int64_t ib[4];
    TRY(fraction_to_boundary(s, 1.0, &max_ap, &max_ad, a, ib));
    double mu_full;
    TRY(madqp_get_affine_complementarity_measure(s->ctx, &s->st, max_ap, max_ad, &mu_full));
    mu_full /= gamma_a;
    const double* dx = st.d;
    const double* dzl = st.d + st.n + st.m;
    const double* dzu = dzl + st.nlb;
    double alpha_p = 1.0, alpha_d = 1.0;
    auto at_lb = [&](const double* vec, int64_t i, double* out) {
        int64_t j;
        int32_t r = read_idx(s, st.ind_lb + i, &j);
        return r ? r : read1(s, vec + j, out);
    };
    auto at_ub = [&](const double* vec, int64_t i, double* out) {
        int64_t j;
        int32_t r = read_idx(s, st.ind_ub + i, &j);
        return r ? r : read1(s, vec + j, out);
    };
    if (max_ap < 1.0) {
        double z, dz, x, b, d;
        if (a[0] <= a[1]) {
            const int64_t i = ib[0];
            TRY(at_lb(st.zl, i, &z));
            TRY(read1(s, dzl + i, &dz));
            TRY(at_lb(st.x, i, &x));
            TRY(at_lb(st.xl, i, &b));
            TRY(at_lb(dx, i, &d));
            const double tmp = mu_full / (z + max_ad * dz);
            alpha_p = (x - b - tmp) / (-d);
        } else {
            const int64_t i = ib[1];
            TRY(at_ub(st.zu, i, &z));
            TRY(read1(s, dzu + i, &dz));
            TRY(at_ub(st.x, i, &x));
            TRY(at_ub(st.xu, i, &b));
            TRY(at_ub(dx, i, &d));
            const double tmp = mu_full / (z + max_ad * dz);
            alpha_p = (b - x - tmp) / d;
        }
    }
    if (max_ad < 1.0) {
        double z, dz, x, b, d;
        if (a[2] <= a[3]) {
            const int64_t i = ib[2];
            TRY(at_lb(st.x, i, &x));
            TRY(at_lb(dx, i, &d));
            TRY(at_lb(st.xl, i, &b));
            TRY(at_lb(st.zl, i, &z));
            TRY(read1(s, dzl + i, &dz));
            const double tmp = mu_full / (x + max_ap * d - b);
            alpha_d = -(z - tmp) / dz;
        } else {
            const int64_t i = ib[3];
            TRY(at_ub(st.x, i, &x));
            TRY(at_ub(dx, i, &d));
            TRY(at_ub(st.xu, i, &b));
            TRY(at_ub(st.zu, i, &z));
            TRY(read1(s, dzu + i, &dz));
            const double tmp = mu_full / (b - x - max_ap * d);
            alpha_d = -(z - tmp) / dz;
        }
    }
    s->alpha_p = std::max(alpha_p, gamma_f * max_ap);
    s->alpha_d = std::max(alpha_d, gamma_f * max_ad);
    return MADQP_OK;
}

// src/solver.jl:200-251
int32_t gondzio(madqp_mpc* s, double mu_curr) {
    const double delta = 0.1, bmin = 0.1, bmax = 10.0, tau = 0.995;
    const int64_t len = ntot(s->st);
    double alpha_p, alpha_d;
    TRY(fraction_to_boundary(s, tau, &alpha_p, &alpha_d));
    for (int c = 0; c < s->opt.max_ncorr; ++c) {
        const double ta_p = std::min(alpha_p + delta, 1.0), ta_d = std::min(alpha_d + delta, 1.0);
        double ga;
        TRY(madqp_get_affine_complementarity_measure(s->ctx, &s->st, ta_p, ta_d, &ga));
        const double mu = (ga / mu_curr) * (ga / mu_curr) * ga;
        TRY(madqp_set_extra_correction(s->ctx, &s->st, ta_p, ta_d, bmin, bmax, mu));
        TRY(madqp_set_correction_rhs(s->ctx, &s->st, mu));
        TRY(madqp_copy(s->ctx, len, s->st.d, s->w2));
        TRY(solve_system(s));
        double ha_p, ha_d;
        TRY(fraction_to_boundary(s, tau, &ha_p, &ha_d));
        if (ha_p < 1.005 * alpha_p || ha_d < 1.005 * alpha_d) {
            TRY(madqp_copy(s->ctx, len, s->w2, s->st.d));
            break;
        }
        alpha_p = ha_p;
        alpha_d = ha_d;
    }
    return MADQP_OK;
}

void fill_info(const madqp_mpc* s, madqp_mpc_info* i) {
    if (!i) return;
    i->k = s->k;
    i->obj = s->obj;
    i->inf_pr = s->inf_pr;
    i->inf_du = s->inf_du;
    i->inf_compl = s->inf_compl;
    i->mu = s->mu;
    i->dnorm = s->dnorm;
    i->del_w = s->del_w;
    i->del_c = s->del_c;
    i->alpha_p = s->alpha_p;
    i->alpha_d = s->alpha_d;
    i->residual_ratio = s->residual_ratio;
    i->n_factorizations = s->n_factorizations;
    i->factor_info = s->last_info;
}
}  // namespace

extern "C" int32_t madqp_mpc_create(madqp_kkt* kkt, const madqp_state* st, double* w1, double* w2,
                                    const double* q, const double* rhs, double c0, double norm_b,
                                    double norm_c, const madqp_mpc_options* opt, madqp_mpc** out) {
    if (!kkt || !out) return MADQP_ERR_ARG;
    madqp_ctx* ctx = madqp_kkt_ctx(kkt);
    ARG_TRY(ctx, st && opt && (ntot(*st) == 0 || (w1 && w2)));
    ARG_TRY(ctx, opt->step_rule >= 0 && opt->step_rule <= 2 && opt->regularization >= 0 &&
                     opt->regularization <= 2 && opt->max_ncorr >= 0);
    madqp_mpc* s = new (std::nothrow) madqp_mpc();
    if (!s) return madqp_fail(ctx, MADQP_ERR_ALLOC, "host allocation failed");
    memset(s, 0, sizeof(*s));
    s->kkt = kkt;
    s->ctx = ctx;
    s->st = *st;
    s->w1 = w1;
    s->w2 = w2;
    s->q = q;
    s->rhs = rhs;
    s->c0 = c0;
    s->norm_b = norm_b;
    s->norm_c = norm_c;
    s->opt = *opt;
    s->reg_delta_p = opt->delta_p;
    s->reg_delta_d = opt->delta_d;
    const char* env = getenv("MADQP_MPC_FUSED");  // 0: the sequential form (A/B tests)
    s->fused = !(env && env[0] == '0') && opt->step_rule != 2;
    env = getenv("MADQP_MPC_AHEAD");  // 0: body_fused waits for its read-backs with nothing queued behind them (A/B tests)
    s->ahead = s->fused && !(env && env[0] == '0');
    if (s->ahead) {
        hipError_t e = hipHostMalloc((void**)&s->h_early, 2 * MADQP_RESULT_SLOTS * sizeof(double), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_early, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_final, hipEventDisableTiming);
        if (e != hipSuccess) {
            madqp_mpc_destroy(s);
            return madqp_fail(ctx, MADQP_ERR_HIP, "madqp_mpc_create: %s", hipGetErrorString(e));
        }
        s->h_final = s->h_early + MADQP_RESULT_SLOTS;
    }
    *out = s;
    return MADQP_OK;
}

extern "C" int32_t madqp_mpc_destroy(madqp_mpc* s) {
    if (s) {
        if (s->ev_early) (void)hipEventDestroy(s->ev_early);
        if (s->ev_final) (void)hipEventDestroy(s->ev_final);
        if (s->h_early) (void)hipHostFree(s->h_early);
    }
    delete s;
    return MADQP_OK;
}

extern "C" int32_t madqp_mpc_set_scalars(madqp_mpc* s, double mu, double del_w, double del_c, double obj,
                                         int64_t k) {
    if (!s) return MADQP_ERR_ARG;
    s->mu = mu;
    s->del_w = del_w;
    s->del_c = del_c;
    s->obj = obj;
    s->k = k;
    s->head_cached = false;
    s->prebuilt = false;
    s->moved = false;
    return MADQP_OK;
}

// src/solver.jl:259-283: residuals and the termination test.
// status_host: 0 = continue, 1 = SOLVE_SUCCEEDED, 6 = MAXIMUM_ITERATIONS_EXCEEDED
extern "C" int32_t madqp_mpc_head(madqp_mpc* s, madqp_mpc_info* info_host, int32_t* status_host) {
    if (!s || !status_host) return MADQP_ERR_ARG;
    double nrm[3];
    if (s->head_cached) {  // read with the last body's block (c): same iterate, same kernels
        for (int q = 0; q < 3; ++q) nrm[q] = s->nrm_cached[q];
    } else {
        TRY(madqp_kkt_jtprod(s->kkt, s->st.jacl, s->st.y));
        TRY(madqp_get_inf(s->ctx, &s->st, nrm));
        s->n_readbacks += 1;
    }
    s->inf_pr = nrm[0] / std::max(1.0, s->norm_b);
    s->inf_du = nrm[1] / std::max(1.0, s->norm_c);
    s->inf_compl = nrm[2] / std::max(1.0, s->norm_c);
    fill_info(s, info_host);
    if (std::max(s->inf_pr, std::max(s->inf_du, s->inf_compl)) <= s->opt.tol)
        *status_host = 1;
    else if (s->k >= s->opt.max_iter)
        *status_host = 6;
    else
        *status_host = 0;
    return MADQP_OK;
}

namespace {
// src/solver.jl:294-343 behind a successful factorisation, every reduction read back at once
int32_t body_after_factorization(madqp_mpc* s, madqp_mpc_info* info_host) {
    madqp_ctx* ctx = s->ctx;
    TRY(madqp_set_predictive_rhs(ctx, &s->st)); // :294
    TRY(solve_system(s));
    double a_aff_p, a_aff_d, mu_affine, mu_curr;
    TRY(fraction_to_boundary(s, 1.0, &a_aff_p, &a_aff_d));  // :295
    TRY(madqp_get_affine_complementarity_measure(ctx, &s->st, a_aff_p, a_aff_d, &mu_affine));  // :296
    TRY(madqp_get_correction(ctx, &s->st));     // :297
    // update_barrier!(Mehrotra), src/kernels.jl:226-236 (field-order quirk: "any bound")
    TRY(madqp_get_complementarity_measure(ctx, &s->st, &mu_curr));
    double sigma = 1.0;
    if (s->st.nlb + s->st.nub > 0) {
        const double t = mu_affine / mu_curr;
        sigma = std::min(std::max(t * t * t, 1e-6), 10.0);  // t^3 as Julia's literal power forms it; the Python driver and mpc_mu_kernel too
    }
    s->mu = std::max(s->opt.mu_min, sigma * mu_curr);
    TRY(madqp_set_correction_rhs(ctx, &s->st, s->mu));  // :307
    TRY(solve_system(s));
    if (s->opt.max_ncorr > 0) TRY(gondzio(s, mu_curr));  // :316-324
    // update_step!, src/kernels.jl:307-374
    if (s->opt.step_rule == 0) {
        TRY(fraction_to_boundary(s, s->opt.step_param, &s->alpha_p, &s->alpha_d));
    } else if (s->opt.step_rule == 1) {
        TRY(fraction_to_boundary(s, std::max(1.0 - s->mu, s->opt.step_param), &s->alpha_p, &s->alpha_d));
    } else {
        TRY(mehrotra_adaptive_step(s));
    }
    TRY(madqp_norm_inf(ctx, s->st.n, s->st.d, &s->dnorm));             // print_iter, src/structure.jl:190
    TRY(madqp_update_iterates(ctx, &s->st, s->alpha_p, s->alpha_d));   // :332-335
    TRY(madqp_kkt_eval(s->kkt, &s->st, s->q, s->rhs, s->c0, &s->obj)); // :338-340
    TRY(madqp_adjust_boundary(ctx, &s->st, s->mu));                    // :342
    s->k += 1;
    fill_info(s, info_host);
    return MADQP_OK;
}

int32_t body_sequential(madqp_mpc* s, madqp_mpc_info* info_host) {
    update_regularization(s);                   // :288
    TRY(factorize_regularized_system(s));       // :289
    return body_after_factorization(s, info_host);
}

// solve_system! up to its reduction: d = K^-1 p, w1 = p - K d, the three norms into the result block at slot0
int32_t solve_system_queue(madqp_mpc* s, int slot0) {
    const int64_t len = ntot(s->st);
    TRY(madqp_kkt_solve_from(s->kkt, &s->st, s->st.p, s->st.d, s->w1));  // d = K^-1 p, w1 = p
    TRY(madqp_kkt_mul_solved(s->kkt, &s->st, s->w1, s->st.d, -1.0, 1.0));
    for (int32_t it = 0; it < s->opt.refine_steps; ++it) {  // the refinement steps of solve_system, queued like the rest
        TRY(madqp_kkt_solve(s->kkt, &s->st, s->w1));
        TRY(madqp_axpy(s->ctx, len, 1.0, s->w1, s->st.d));
        TRY(madqp_copy(s->ctx, len, s->st.p, s->w1));
        TRY(madqp_kkt_mul(s->kkt, &s->st, s->w1, s->st.d, -1.0, 1.0));
    }
    return madqp_q_norm_inf3(s->ctx, len, s->w1, s->st.p, s->st.d, slot0);
}
int32_t residual_verdict(madqp_mpc* s, const double* nrm) {  // src/linear_solver.jl:36-43
    const double ratio = nrm[0] / std::max(1.0, nrm[1]);
    s->residual_ratio = ratio;
    if (std::isnan(ratio) || (s->opt.check_residual && ratio > s->opt.tol_linear_solve)) return MADQP_NUM_NAN;
    return MADQP_OK;
}
inline double min_like_host(double a, double b) { return std::min(a, b); }  // fraction_to_boundary's combination

// The same iteration with two blocking read-backs (see the head of this file).
int32_t body_fused(madqp_mpc* s, madqp_mpc_info* info_host) {
    madqp_ctx* ctx = s->ctx;
    const int64_t nb = s->st.nlb + s->st.nub;
    update_regularization(s);  // :288
    // (a) factorisation and predictor
    const bool have_k = s->prebuilt && s->pre_del_w == s->del_w && s->pre_del_c == s->del_c;  // queued by the last pass
    s->prebuilt = false;
    if (have_k) {
        s->n_prebuilt_used += 1;
    } else {
        TRY(madqp_kkt_set_aug_diagonal_reg(s->kkt, &s->st, s->del_w, s->del_c));
        TRY(madqp_kkt_build(s->kkt, &s->st));
    }
    TRY(madqp_q_kkt_factorize(s->kkt, 15));
    s->n_factorizations += 1;
    TRY(madqp_set_predictive_rhs(ctx, &s->st));                                  // :294
    TRY(solve_system_queue(s, 0));
    TRY(madqp_q_alpha_max(ctx, &s->st, 1.0, 3));                                 // :295
    TRY(madqp_q_compl(ctx, &s->st, 1, 0.0, 0.0, ctx->d_res + 3, 11));            // :296, step lengths from the block
    TRY(madqp_get_correction(ctx, &s->st));                                      // :297
    TRY(madqp_q_compl(ctx, &s->st, 0, 0.0, 0.0, nullptr, 13));
    // update_barrier! on the device: mu, the step rule's tau and mu_curr into the block (SL_MU ..) -- the corrector is
    // queued behind the predictor without a read-back in between (round 4: a read-back costs 28 us and leaves the
    // queue empty: the ~35 launches behind it are then issued at the host's 3.7 us each instead of the 1.6 us the
    // GPU takes them at from a filled queue, tools/launch_probe.cpp)
    const int SL_MU = 32, SL_TAU = 33, SL_MUCURR = 34, SL_NRM_B = 35, SL_ALPHA_B = 38, SL_DNORM_B = 46, SL_ALPHA_GZ = 49,
              SL_MUC = 58;
    TRY(madqp_q_mpc_mu(ctx, 11, SL_MU, nb, s->opt.mu_min, s->opt.step_rule, s->opt.step_param));
    // (b) corrector.  The step rule's minima and |dx| are queued behind every direction that may turn out to be the
    // final one, so that the Gondzio loop below ends without a read-back of its own.
    TRY(madqp_set_correction_rhs_dev(ctx, &s->st, ctx->d_res + SL_MU));  // :307
    TRY(solve_system_queue(s, SL_NRM_B));
    const bool gz = s->opt.max_ncorr > 0;
    const double gz_tau = 0.995, gz_delta = 0.1, gz_bmin = 0.1, gz_bmax = 10.0;  // src/solver.jl:200-251
    TRY(madqp_q_alpha_max_dev(ctx, &s->st, 0.0, ctx->d_res + SL_TAU, SL_ALPHA_B));
    TRY(madqp_q_norm_inf3(ctx, s->st.n, s->st.d, nullptr, nullptr, SL_DNORM_B));  // print_iter, src/structure.jl:190
    const int64_t len = ntot(s->st);
    // gondzio(), first trial: queued behind the corrector, its step lengths min(alpha + delta, 1) and its mu_c formed on
    // the device from the block -- the read-back that follows serves the corrector AND the trial
    const int SL_TA = 64, SL_T_COMPL = 72, SL_T_NRM = 74, SL_T_AGZ = 77, SL_T_ATAU = 85, SL_T_DN = 93;
    if (gz) {
        TRY(madqp_q_alpha_max(ctx, &s->st, gz_tau, SL_ALPHA_GZ));
        TRY(madqp_q_mpc_trial_alpha(ctx, SL_ALPHA_GZ, SL_TA, gz_delta));
        TRY(madqp_q_compl(ctx, &s->st, 1, 0.0, 0.0, ctx->d_res + SL_TA, SL_T_COMPL));
        TRY(madqp_q_mpc_muc(ctx, SL_T_COMPL, SL_MUCURR, SL_MUC, nb));
        TRY(madqp_set_extra_correction_dev(ctx, &s->st, 0.0, 0.0, gz_bmin, gz_bmax, ctx->d_res + SL_MUC, ctx->d_res + SL_TA));
        TRY(madqp_set_correction_rhs_dev(ctx, &s->st, ctx->d_res + SL_MUC));
        TRY(madqp_copy(ctx, len, s->st.d, s->w2));
        TRY(solve_system_queue(s, SL_T_NRM));
        TRY(madqp_q_alpha_max(ctx, &s->st, gz_tau, SL_T_AGZ));
        TRY(madqp_q_alpha_max_dev(ctx, &s->st, 0.0, ctx->d_res + SL_TAU, SL_T_ATAU));  // in case this direction is kept
        TRY(madqp_q_norm_inf3(ctx, s->st.n, s->st.d, nullptr, nullptr, SL_T_DN));      // and is the last one
    }
    // The decisions of the read-back below, on the device too (SL_DEC ..), and behind them -- ahead of the read-back -- the
    // update of the iterates and everything up to the next termination test's norms, as kernels that do nothing when
    // the device's verdict is "the host takes over".  The block travels to the host twice: behind the decision (the host
    // waits for THAT copy while the update runs) and behind the norms.
    const int SL_DEC = 100, SL_C = 106;
    double rb[MADQP_RESULT_SLOTS];
    const int nrb = gz ? SL_T_DN + 3 : SL_DNORM_B + 3;
    // (Not in the first pass behind the start point: f and c are then the values initialize! left, from BEFORE the start
    // point moved x -- src/solver.jl:100-146 -- and an evaluation queued ahead would refresh them even where the host
    // takes over and the reference's loop goes on with the stale ones.)
    const bool spec = s->ahead && s->moved;
    if (spec) {
        MpcDecide dec{15, 0, SL_NRM_B, SL_ALPHA_B, SL_ALPHA_GZ, SL_T_NRM, SL_DEC, gz ? 1 : 0, s->opt.max_ncorr,
                      s->opt.check_residual ? 1 : 0, s->opt.tol_linear_solve};
        TRY(madqp_q_mpc_decide(ctx, &dec));
        TRY(madqp_results_post(ctx, s->h_early, s->ev_early));
        if (gz) TRY(madqp_copy_if_dev(ctx, len, s->w2, s->st.d, ctx->d_res + SL_DEC + 3));
        TRY(madqp_update_iterates_dev(ctx, &s->st, ctx->d_res + SL_DEC));                   // :332-335
        TRY(madqp_q_kkt_eval(s->kkt, &s->st, s->q, s->rhs, SL_C));                          // :338-340
        TRY(madqp_adjust_boundary_dev(ctx, &s->st, ctx->d_res + SL_DEC, ctx->d_res + SL_MU)); // :342
        TRY(madqp_kkt_jtprod(s->kkt, s->st.jacl, s->st.y));                                 // :259-283 of the next pass
        TRY(madqp_q_inf(ctx, &s->st, SL_C + 2));
        TRY(madqp_results_post(ctx, s->h_final, s->ev_final));
        TRY(madqp_results_wait(ctx, s->h_early, s->ev_early, SL_DEC + 4, rb));
    } else {
        TRY(madqp_read_results(ctx, nrb, rb));
    }
    s->n_readbacks += 1;
    const bool went = spec && rb[SL_DEC] != 0.0;  // the device has gone on: the host's decisions must be the same
    auto differ = [&]() {
        return madqp_fail(ctx, MADQP_ERR_HIP, "body_fused: the device's decisions behind the corrector are not the host's");
    };
    s->last_info = (int32_t)rb[15];
    TRY(madqp_kkt_factor_result(s->kkt, s->last_info));
    if (s->last_info != 0) {  // src/linear_solver.jl:6-17: what was queued behind the failed factorisation is void
        if (went) return differ();
        s->del_w *= 100.0;
        s->del_c *= 100.0;
        TRY(factorize_regularized_system(s, 1));
        return body_after_factorization(s, info_host);
    }
    {
        int32_t v = residual_verdict(s, rb);  // the predictor's solve
        if (!v) {
            s->mu = rb[SL_MU];
            v = residual_verdict(s, rb + SL_NRM_B);  // the corrector's
        }
        if (v) return went ? differ() : v;
    }
    s->alpha_p = min_like_host(rb[SL_ALPHA_B + 0], rb[SL_ALPHA_B + 2]);
    s->alpha_d = min_like_host(rb[SL_ALPHA_B + 4], rb[SL_ALPHA_B + 6]);
    s->dnorm = rb[SL_DNORM_B];
    if (gz) {  // the first trial is in the block; a further one (while the last was kept) costs a read-back of its own
        double g_ap = min_like_host(rb[SL_ALPHA_GZ + 0], rb[SL_ALPHA_GZ + 2]);
        double g_ad = min_like_host(rb[SL_ALPHA_GZ + 4], rb[SL_ALPHA_GZ + 6]);
        const double* tb = rb + SL_T_NRM;  // norms [0..2], steps at gz_tau [3..10], steps at tau [11..18], |dx| [19]
        double tr[32];
        for (int c = 0; c < s->opt.max_ncorr; ++c) {
            if (c > 0) {
                if (went) return differ();
                const double ta_p = std::min(g_ap + gz_delta, 1.0), ta_d = std::min(g_ad + gz_delta, 1.0);
                TRY(madqp_q_compl(ctx, &s->st, 1, ta_p, ta_d, nullptr, 0));
                TRY(madqp_q_mpc_muc(ctx, 0, SL_MUCURR, SL_MUC, nb));
                TRY(madqp_set_extra_correction_dev(ctx, &s->st, ta_p, ta_d, gz_bmin, gz_bmax, ctx->d_res + SL_MUC));
                TRY(madqp_set_correction_rhs_dev(ctx, &s->st, ctx->d_res + SL_MUC));
                TRY(madqp_copy(ctx, len, s->st.d, s->w2));
                TRY(solve_system_queue(s, 0));
                TRY(madqp_q_alpha_max(ctx, &s->st, gz_tau, 3));
                TRY(madqp_q_alpha_max_dev(ctx, &s->st, 0.0, ctx->d_res + SL_TAU, 11));
                TRY(madqp_q_norm_inf3(ctx, s->st.n, s->st.d, nullptr, nullptr, 19));
                TRY(madqp_read_results(ctx, 22, tr));
                s->n_readbacks += 1;
                tb = tr;
            }
            {
                const int32_t v = residual_verdict(s, tb);
                if (v) return went ? differ() : v;
            }
            const double ha_p = min_like_host(tb[3], tb[5]), ha_d = min_like_host(tb[7], tb[9]);
            if (ha_p < 1.005 * g_ap || ha_d < 1.005 * g_ad) {
                if (went) {
                    if (rb[SL_DEC + 3] == 0.0) return differ();  // (the device has put the direction back itself)
                } else {
                    TRY(madqp_copy(ctx, len, s->w2, s->st.d));  // the direction before it, whose step is already known
                }
                break;
            }
            if (went && c == 0 && rb[SL_DEC + 3] != 0.0) return differ();
            g_ap = ha_p;
            g_ad = ha_d;
            s->alpha_p = min_like_host(tb[11], tb[13]);
            s->alpha_d = min_like_host(tb[15], tb[17]);
            s->dnorm = tb[19];
        }
    }
    if (went && (rb[SL_DEC + 1] != s->alpha_p || rb[SL_DEC + 2] != s->alpha_d)) return differ();
    // (c) update, objective, and the residuals the next termination test needs
    if (!went) {
        TRY(madqp_update_iterates(ctx, &s->st, s->alpha_p, s->alpha_d));  // :332-335
        TRY(madqp_q_kkt_eval(s->kkt, &s->st, s->q, s->rhs, SL_C));        // :338-340
        TRY(madqp_adjust_boundary(ctx, &s->st, s->mu));                   // :342
        TRY(madqp_kkt_jtprod(s->kkt, s->st.jacl, s->st.y));               // :259-283 of the next pass
        TRY(madqp_q_inf(ctx, &s->st, SL_C + 2));
        if (s->ahead) TRY(madqp_results_post(ctx, s->h_final, s->ev_final));
    }
    if (s->ahead) {
        // Queue the NEXT iteration's diagonal and assembly before waiting for the norms: the device does not run dry
        // across the read-back and the driver's way around the loop (K is rebuilt from scratch every iteration; should
        // the loop end here, a K nobody factors is all that happened)
        // (not when this pass is the last one allowed, nor when the iterate it started from was already within 100 x the
        // tolerance: the loop then usually ends behind this pass or the next, and the assembly would be for nobody)
        const double worst = std::max(s->inf_pr, std::max(s->inf_du, s->inf_compl));
        if (s->k + 1 < s->opt.max_iter && !(worst <= 100.0 * s->opt.tol)) {
            madqp_mpc nx = *s;  // the regularisation of the next pass, from a copy of the scalars
            update_regularization(&nx);
            TRY(madqp_kkt_set_aug_diagonal_reg(s->kkt, &s->st, nx.del_w, nx.del_c));
            TRY(madqp_kkt_build(s->kkt, &s->st));
            s->prebuilt = true;
            s->pre_del_w = nx.del_w;
            s->pre_del_c = nx.del_c;
            s->n_prebuilt += 1;
        }
        TRY(madqp_results_wait(ctx, s->h_final, s->ev_final, SL_C + 6, rb));
    } else {
        TRY(madqp_read_results(ctx, SL_C + 6, rb));
    }
    s->moved = true;
    s->k += 1;
    s->n_readbacks += 1;
    const double* r = rb + SL_C;
    s->obj = s->c0 + r[0] + 0.5 * r[1];
    madqp_inf_from_block(r + 2, s->nrm_cached);
    s->head_cached = true;
    fill_info(s, info_host);
    return MADQP_OK;
}
}  // namespace

// src/solver.jl:288-343: one predictor-corrector step.  Returns MADQP_NUM_NAN for the
// SolveException of src/linear_solver.jl:41-43.
extern "C" int32_t madqp_mpc_body(madqp_mpc* s, madqp_mpc_info* info_host) {
    if (!s) return MADQP_ERR_ARG;
    s->head_cached = false;
    if (!s->fused) return body_sequential(s, info_host);
    const int32_t r = body_fused(s, info_host);
    if (r) s->prebuilt = false;
    return r;
}

extern "C" int32_t madqp_mpc_readbacks(const madqp_mpc* s, int64_t* count) {
    if (!s || !count) return MADQP_ERR_ARG;
    *count = s->n_readbacks;
    return MADQP_OK;
}

extern "C" int32_t madqp_mpc_ahead_stats(const madqp_mpc* s, int64_t* queued, int64_t* used) {
    if (!s || !queued || !used) return MADQP_ERR_ARG;
    *queued = s->n_prebuilt;
    *used = s->n_prebuilt_used;
    return MADQP_OK;
}
