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
enum { OP_SUM = 0, OP_MAX = 1, OP_MIN = 2 };

__device__ __forceinline__ double nanmax(double a, double b) {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
template <int OP>
__device__ __forceinline__ double comb(double a, double b) {
    if (OP == OP_SUM) return a + b;
    if (OP == OP_MAX) return nanmax(a, b);
    return (b < a) ? b : a;
}
template <int OP>
__device__ __forceinline__ double block_reduce(double v, double* sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = comb<OP>(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = sm[0];
    for (int i = 1; i < TPB / 64; ++i) r = comb<OP>(r, sm[i]);
    return r;
}

struct FinalOps {
    int nv;
    int8_t op[16];
};
// partial layout: part[b*nv + i]
__global__ __launch_bounds__(TPB) void finalize_kernel(const double* __restrict__ part, int nblocks,
                                                       FinalOps ops, double* __restrict__ res) {
    __shared__ double sm[TPB / 64];
    for (int i = 0; i < ops.nv; ++i) {
        const int op = ops.op[i];
        double v = (op == OP_SUM) ? 0.0 : (op == OP_MAX ? -INFINITY : INFINITY);
        for (int b = threadIdx.x; b < nblocks; b += TPB) {
            const double p = part[b * ops.nv + i];
            v = (op == OP_SUM) ? v + p : (op == OP_MAX ? nanmax(v, p) : ((p < v) ? p : v));
        }
        double r;
        if (op == OP_SUM)
            r = block_reduce<OP_SUM>(v, sm);
        else if (op == OP_MAX)
            r = block_reduce<OP_MAX>(v, sm);
        else
            r = block_reduce<OP_MIN>(v, sm);
        if (threadIdx.x == 0) res[i] = r;
    }
}

inline int grid_for(int64_t len) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((len + TPB - 1) / TPB, MADQP_MAX_BLOCKS));
}
#define GRID_STRIDE(i, len) \
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < (len); i += (int64_t)gridDim.x * TPB)

// ------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(TPB) void aug_diag_fill_kernel(madqp_state s, double del_w, double del_c) {
    const int64_t L = s.n > s.m ? s.n : s.m;
    GRID_STRIDE(i, L) {
        if (i < s.n) {
            s.reg[i] = del_w;
            s.pr_diag[i] = del_w;
        }
        if (i < s.m) s.du_diag[i] = del_c;
    }
}
__global__ __launch_bounds__(TPB) void aug_diag_lb_kernel(madqp_state s) {
    GRID_STRIDE(i, s.nlb) {
        const int64_t j = s.ind_lb[i];
        const double ld = s.xl[j] - s.x[j];
        const double ll = s.zl[j];
        s.l_diag[i] = ld;
        s.l_lower[i] = ll;
        s.pr_diag[j] -= ll / ld;
    }
}
__global__ __launch_bounds__(TPB) void aug_diag_ub_kernel(madqp_state s) {
    GRID_STRIDE(i, s.nub) {
        const int64_t j = s.ind_ub[i];
        const double ud = s.x[j] - s.xu[j];
        const double ul = s.zu[j];
        s.u_diag[i] = ud;
        s.u_lower[i] = ul;
        s.pr_diag[j] -= ul / ud;
    }
}

// mode 0: predictive rhs; 1: correction rhs (mu, corrections); 2: initial primal; 3: initial dual
__global__ __launch_bounds__(TPB) void rhs_kernel(madqp_state s, int mode, double mu) {
    double* px = s.p;
    double* py = s.p + s.n;
    double* pzl = s.p + s.n + s.m;
    double* pzu = pzl + s.nlb;
    int64_t L = s.n;
    if (s.m > L) L = s.m;
    if (s.nlb > L) L = s.nlb;
    if (s.nub > L) L = s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.n) {
            if (mode <= 1)
                px[i] = -s.f[i] + s.zl[i] - s.zu[i] - s.jacl[i];
            else
                px[i] = (mode == 3) ? -s.f[i] : 0.0;
        }
        if (i < s.m) py[i] = (mode == 3) ? 0.0 : -s.c[i];
        if (i < s.nlb) {
            if (mode <= 1) {
                const int64_t j = s.ind_lb[i];
                double v = (s.xl[j] - s.x[j]) * s.zl[j];
                if (mode == 1) v = v + mu - s.correction_lb[i];
                pzl[i] = v;
            } else
                pzl[i] = 0.0;
        }
        if (i < s.nub) {
            if (mode <= 1) {
                const int64_t j = s.ind_ub[i];
                double v = (s.xu[j] - s.x[j]) * s.zu[j];
                if (mode == 1) v = v - mu - s.correction_ub[i];
                pzu[i] = v;
            } else
                pzu[i] = 0.0;
        }
    }
}

__global__ __launch_bounds__(TPB) void correction_kernel(madqp_state s) {
    const double* dx = s.d;
    const double* dzl = s.d + s.n + s.m;
    const double* dzu = dzl + s.nlb;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) s.correction_lb[i] = dx[s.ind_lb[i]] * dzl[i];
        if (i < s.nub) s.correction_ub[i] = dx[s.ind_ub[i]] * dzu[i];
    }
}

__global__ __launch_bounds__(TPB) void extra_correction_kernel(madqp_state s, double ap, double ad,
                                                               double tmin, double tmax) {
    const double* dx = s.d;
    const double* dzl = s.d + s.n + s.m;
    const double* dzu = dzl + s.nlb;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            const double x_ = s.x[j] + ap * dx[j] - s.xl[j];
            const double z_ = s.zl[j] + ad * dzl[i];
            const double v = x_ * z_;
            const double dl = (v < tmin) ? (tmin - v) : ((v > tmax) ? (tmax - v) : 0.0);
            s.correction_lb[i] = s.correction_lb[i] - dl;
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            const double x_ = s.xu[j] - ap * dx[j] - s.x[j];
            const double z_ = s.zu[j] + ad * dzu[i];
            const double v = x_ * z_;
            const double dl = (v < tmin) ? (tmin - v) : ((v > tmax) ? (tmax - v) : 0.0);
            s.correction_ub[i] = s.correction_ub[i] + dl;
        }
    }
}

__global__ __launch_bounds__(TPB) void update_iterates_kernel(madqp_state s, double ap, double ad) {
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

__global__ __launch_bounds__(TPB) void adjust_boundary_kernel(madqp_state s, double c1, double c2) {
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

// reduce_rhs!: two ordered scatter passes (an index may sit in both lists)
__global__ __launch_bounds__(TPB) void reduce_rhs_kernel(int64_t cnt, const int64_t* __restrict__ ind,
                                                         double* __restrict__ wx,
                                                         const double* __restrict__ wz,
                                                         const double* __restrict__ diag) {
    GRID_STRIDE(i, cnt) wx[ind[i]] -= wz[i] / diag[i];
}

__global__ __launch_bounds__(TPB) void finish_aug_solve_kernel(madqp_state s, double* w) {
    const double* wx = w;
    double* wzl = w + s.n + s.m;
    double* wzu = wzl + s.nlb;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) wzl[i] = (-wzl[i] + s.l_lower[i] * wx[s.ind_lb[i]]) / s.l_diag[i];
        if (i < s.nub) wzu[i] = (wzu[i] - s.u_lower[i] * wx[s.ind_ub[i]]) / s.u_diag[i];
    }
}

// _kktmul!: pass 1 over n and m, pass 2 over the lower-bound list, pass 3 over the upper-bound list
__global__ __launch_bounds__(TPB) void kktmul_diag_kernel(madqp_state s, double* w, const double* v,
                                                          double alpha) {
    const int64_t L = s.n > s.m ? s.n : s.m;
    GRID_STRIDE(i, L) {
        if (i < s.n) w[i] += alpha * s.reg[i] * v[i];
        if (i < s.m) w[s.n + i] += alpha * s.du_diag[i] * v[s.n + i];
    }
}
__global__ __launch_bounds__(TPB) void kktmul_lb_kernel(madqp_state s, double* w, const double* v,
                                                        double alpha, double beta) {
    double* wzl = w + s.n + s.m;
    const double* vzl = v + s.n + s.m;
    GRID_STRIDE(i, s.nlb) {
        const int64_t j = s.ind_lb[i];
        w[j] -= alpha * vzl[i];
        wzl[i] = beta * wzl[i] + alpha * (v[j] * s.l_lower[i] - vzl[i] * s.l_diag[i]);
    }
}
__global__ __launch_bounds__(TPB) void kktmul_ub_kernel(madqp_state s, double* w, const double* v,
                                                        double alpha, double beta) {
    double* wzu = w + s.n + s.m + s.nlb;
    const double* vzu = v + s.n + s.m + s.nlb;
    GRID_STRIDE(i, s.nub) {
        const int64_t j = s.ind_ub[i];
        w[j] += alpha * vzu[i];
        wzu[i] = beta * wzu[i] + alpha * (v[j] * s.u_lower[i] + vzu[i] * s.u_diag[i]);
    }
}

__global__ __launch_bounds__(TPB) void axpy_kernel(int64_t n, double a, const double* __restrict__ x,
                                                   double* __restrict__ y) {
    GRID_STRIDE(i, n) y[i] += a * x[i];
}
__global__ __launch_bounds__(TPB) void fill_kernel(int64_t n, double v, double* __restrict__ y) {
    GRID_STRIDE(i, n) y[i] = v;
}

// ------------------------------------------------------------------ reductions
__global__ __launch_bounds__(TPB) void compl_kernel(madqp_state s, int affine, double ap, double ad,
                                                    double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    const double* dx = s.d;
    const double* dzl = s.d + s.n + s.m;
    const double* dzu = dzl + s.nlb;
    double sl = 0.0, su = 0.0;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            if (affine)
                sl += ((s.x[j] + ap * dx[j]) - s.xl[j]) * (s.zl[j] + ad * dzl[i]);
            else
                sl += (s.x[j] - s.xl[j]) * s.zl[j];
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            if (affine)
                su += (s.xu[j] - (s.x[j] + ap * dx[j])) * (s.zu[j] + ad * dzu[i]);
            else
                su += (s.xu[j] - s.x[j]) * s.zu[j];
        }
    }
    sl = block_reduce<OP_SUM>(sl, sm);
    su = block_reduce<OP_SUM>(su, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = sl;
        part[blockIdx.x * 2 + 1] = su;
    }
}

struct ArgMin {
    double v;
    double i;
};
__device__ __forceinline__ ArgMin am_comb(ArgMin a, ArgMin b) {
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMin am_block(ArgMin a, double* sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ArgMin o;
        o.v = __shfl_down(a.v, off, 64);
        o.i = __shfl_down(a.i, off, 64);
        a = am_comb(a, o);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sm[2 * w] = a.v;
        sm[2 * w + 1] = a.i;
    }
    __syncthreads();
    ArgMin r{sm[0], sm[1]};
    for (int k = 1; k < TPB / 64; ++k) r = am_comb(r, ArgMin{sm[2 * k], sm[2 * k + 1]});
    return r;
}
// partials: part[b*8 + 2*q + {0,1}] = (value, index) of quantity q in {xl, xu, zl, zu}
__global__ __launch_bounds__(TPB) void alpha_max_kernel(madqp_state s, double tau,
                                                        double* __restrict__ part) {
    __shared__ double sm[2 * TPB / 64];
    const double* dx = s.d;
    const double* dzl = s.d + s.n + s.m;
    const double* dzu = dzl + s.nlb;
    const double BIG = 1e300;  // index of "nothing blocks"
    ArgMin a[4];
    for (int q = 0; q < 4; ++q) a[q] = ArgMin{INFINITY, BIG};
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            const double dxl = dx[j];
            const double v0 = (dxl < 0.0) ? (-s.x[j] + s.xl[j]) * tau / dxl : INFINITY;
            a[0] = am_comb(a[0], ArgMin{v0, (double)i});
            const double dz = dzl[i];
            const double v2 = (dz < 0.0) ? (-s.zl[j]) * tau / dz : INFINITY;
            a[2] = am_comb(a[2], ArgMin{v2, (double)i});
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            const double dxu = dx[j];
            const double v1 = (dxu > 0.0) ? (-s.x[j] + s.xu[j]) * tau / dxu : INFINITY;
            a[1] = am_comb(a[1], ArgMin{v1, (double)i});
            const double dz = dzu[i];
            const double zu = s.zu[j];
            const double v3 = ((dz < 0.0) && (zu + dz < 0.0)) ? (-zu) * tau / dz : INFINITY;
            a[3] = am_comb(a[3], ArgMin{v3, (double)i});
        }
    }
    for (int q = 0; q < 4; ++q) {
        ArgMin r = am_block(a[q], sm);
        if (threadIdx.x == 0) {
            part[blockIdx.x * 8 + 2 * q] = r.v;
            part[blockIdx.x * 8 + 2 * q + 1] = r.i;
        }
    }
}
__global__ __launch_bounds__(TPB) void alpha_max_final_kernel(const double* __restrict__ part,
                                                              int nblocks, double* __restrict__ res) {
    __shared__ double sm[2 * TPB / 64];
    for (int q = 0; q < 4; ++q) {
        ArgMin a{INFINITY, 1e300};
        for (int b = threadIdx.x; b < nblocks; b += TPB)
            a = am_comb(a, ArgMin{part[b * 8 + 2 * q], part[b * 8 + 2 * q + 1]});
        ArgMin r = am_block(a, sm);
        if (threadIdx.x == 0) {
            // the reference's init = (1.0, 0): alpha <= 1 and "nothing blocks" unless val < 1
            if (r.v < 1.0) {
                res[2 * q] = r.v;
                res[2 * q + 1] = r.i;
            } else {
                res[2 * q] = 1.0;
                res[2 * q + 1] = -1.0;
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void inf_kernel(madqp_state s, double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    double mc = 0.0, md = 0.0, ml = 0.0, mu = 0.0;
    int64_t L = s.n;
    if (s.m > L) L = s.m;
    if (s.nlb > L) L = s.nlb;
    if (s.nub > L) L = s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.m) mc = nanmax(mc, fabs(s.c[i]));
        if (i < s.n) md = nanmax(md, fabs(s.f[i] - s.zl[i] + s.zu[i] + s.jacl[i]));
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            ml = nanmax(ml, fabs((s.x[j] - s.xl[j]) * s.zl[j]));
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            mu = nanmax(mu, fabs((s.xu[j] - s.x[j]) * s.zu[j]));
        }
    }
    mc = block_reduce<OP_MAX>(mc, sm);
    md = block_reduce<OP_MAX>(md, sm);
    ml = block_reduce<OP_MAX>(ml, sm);
    mu = block_reduce<OP_MAX>(mu, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 4 + 0] = mc;
        part[blockIdx.x * 4 + 1] = md;
        part[blockIdx.x * 4 + 2] = ml;
        part[blockIdx.x * 4 + 3] = mu;
    }
}

__global__ __launch_bounds__(TPB) void norm_inf3_kernel(int64_t len, const double* __restrict__ a,
                                                        const double* __restrict__ b,
                                                        const double* __restrict__ c,
                                                        double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    double ma = 0.0, mb = 0.0, mc = 0.0;
    GRID_STRIDE(i, len) {
        if (a) ma = nanmax(ma, fabs(a[i]));
        if (b) mb = nanmax(mb, fabs(b[i]));
        if (c) mc = nanmax(mc, fabs(c[i]));
    }
    ma = block_reduce<OP_MAX>(ma, sm);
    mb = block_reduce<OP_MAX>(mb, sm);
    mc = block_reduce<OP_MAX>(mc, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = ma;
        part[blockIdx.x * 3 + 1] = mb;
        part[blockIdx.x * 3 + 2] = mc;
    }
}

// ------------------------------------------------------------ starting point
__global__ __launch_bounds__(TPB) void sp_init_duals_kernel(madqp_state s) {
    GRID_STRIDE(i, s.n) {
        const double r = s.jacl[i], l = s.xl[i], u = s.xu[i];
        const bool fl = isfinite(l), fu = isfinite(u);
        if (fl && fu) {
            s.zl[i] = 0.5 * r;
            s.zu[i] = -0.5 * r;
        } else {
            if (fl) s.zl[i] = r;
            if (fu) s.zu[i] = -r;
        }
    }
}
__global__ __launch_bounds__(TPB) void sp_mins_kernel(madqp_state s, double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            m0 = fmin(m0, s.x[j] - s.xl[j]);
            m2 = fmin(m2, s.zl[j]);
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            m1 = fmin(m1, s.xu[j] - s.x[j]);
            m3 = fmin(m3, s.zu[j]);
        }
    }
    m0 = block_reduce<OP_MIN>(m0, sm);
    m1 = block_reduce<OP_MIN>(m1, sm);
    m2 = block_reduce<OP_MIN>(m2, sm);
    m3 = block_reduce<OP_MIN>(m3, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 4 + 0] = m0;
        part[blockIdx.x * 4 + 1] = m1;
        part[blockIdx.x * 4 + 2] = m2;
        part[blockIdx.x * 4 + 3] = m3;
    }
}
__global__ __launch_bounds__(TPB) void sp_shift_kernel(int64_t cnt, const int64_t* __restrict__ ind,
                                                       double* __restrict__ v, double delta) {
    GRID_STRIDE(i, cnt) v[ind[i]] += delta;
}
__global__ __launch_bounds__(TPB) void sp_sums_kernel(madqp_state s, double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            const double x = s.x[j], l = s.xl[j], z = s.zl[j];
            a[0] += x * z;
            a[1] += l * z;
            a[4] += z;
            a[6] += x - l;
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            const double x = s.x[j], u = s.xu[j], z = s.zu[j];
            a[2] += u * z;
            a[3] += x * z;
            a[5] += z;
            a[7] += u - x;
        }
    }
    for (int q = 0; q < 8; ++q) {
        const double r = block_reduce<OP_SUM>(a[q], sm);
        if (threadIdx.x == 0) part[blockIdx.x * 8 + q] = r;
    }
}
__global__ __launch_bounds__(TPB) void sp_project_kernel(madqp_state s, double kappa) {
    GRID_STRIDE(i, s.n) {
        const double l = s.xl[i], u = s.xu[i], x = s.x[i];
        if (x < l) {
            const double pl = fmin(kappa * fmax(1.0, l), kappa * (u - l));
            s.x[i] = l + pl;
        } else if (u < x) {
            const double pu = fmin(kappa * fmax(1.0, u), kappa * (u - l));
            s.x[i] = u - pu;
        }
    }
}
// 1.0 when an assert fails
__global__ __launch_bounds__(TPB) void sp_check_kernel(madqp_state s, double* __restrict__ part) {
    __shared__ double sm[TPB / 64];
    double bad = 0.0;
    const int64_t L = s.nlb > s.nub ? s.nlb : s.nub;
    GRID_STRIDE(i, L) {
        if (i < s.nlb) {
            const int64_t j = s.ind_lb[i];
            if (!(s.zl[j] > 0.0) || !(s.x[j] > s.xl[j])) bad = 1.0;
        }
        if (i < s.nub) {
            const int64_t j = s.ind_ub[i];
            if (!(s.zu[j] > 0.0) || !(s.x[j] < s.xu[j])) bad = 1.0;
        }
    }
    bad = block_reduce<OP_MAX>(bad, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = bad;
}

int32_t finalize(madqp_ctx* ctx, int nblocks, int nv, const int* ops) {
    FinalOps f;
    f.nv = nv;
    for (int i = 0; i < nv; ++i) f.op[i] = (int8_t)ops[i];
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part, nblocks, f,
                       ctx->d_res);
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

static int32_t rhs_launch(madqp_ctx* ctx, const madqp_state* st, int mode, double mu) {
    CHECK_STATE();
    ProfScope ps(ctx, MADQP_PROF_VEC);
    LAUNCH(rhs_kernel, max4(st->n, st->m, st->nlb, st->nub), *st, mode, mu);
    return MADQP_OK;
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
           beta_min * mu, beta_max * mu);
    return MADQP_OK;
}

static int32_t compl_launch(madqp_ctx* ctx, const madqp_state* st, int affine, double ap, double ad,
                            double* mu_host) {
    CHECK_STATE();
    ARG_TRY(ctx, mu_host != nullptr);
    if (st->nlb + st->nub == 0) {  // src/kernels.jl:173-174
        *mu_host = 0.0;
        return MADQP_OK;
    }
    const int64_t L = std::max(st->nlb, st->nub);
    const int nb = grid_for(L);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(compl_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, affine, ap, ad,
                           ctx->d_part);
        LAUNCH_CHECK(ctx);
        const int ops[2] = {OP_SUM, OP_SUM};
        int32_t r = finalize(ctx, nb, 2, ops);
        if (r) return r;
    }
    double out[2];
    int32_t r = madqp_read_results(ctx, 2, out);
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
    const int nb = grid_for(L);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(alpha_max_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, tau,
                           ctx->d_part);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(alpha_max_final_kernel, dim3(1), dim3(TPB), 0, ctx->stream, ctx->d_part,
                           nb, ctx->d_res);
        LAUNCH_CHECK(ctx);
    }
    double out[8];
    int32_t r = madqp_read_results(ctx, 8, out);
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

extern "C" int32_t madqp_get_inf(madqp_ctx* ctx, const madqp_state* st, double* out_host) {
    CHECK_STATE();
    ARG_TRY(ctx, out_host != nullptr);
    const int64_t L = max4(st->n, st->m, st->nlb, st->nub);
    if (L == 0) {
        out_host[0] = out_host[1] = out_host[2] = 0.0;
        return MADQP_OK;
    }
    const int nb = grid_for(L);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(inf_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, *st, ctx->d_part);
        LAUNCH_CHECK(ctx);
        const int ops[4] = {OP_MAX, OP_MAX, OP_MAX, OP_MAX};
        int32_t r = finalize(ctx, nb, 4, ops);
        if (r) return r;
    }
    double out[4];
    int32_t r = madqp_read_results(ctx, 4, out);
    if (r) return r;
    out_host[0] = out[0];
    out_host[1] = out[1];
    out_host[2] = (out[2] != out[2]) ? out[2] : ((out[3] != out[3]) ? out[3] : std::max(out[2], out[3]));
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

extern "C" int32_t madqp_norm_inf3(madqp_ctx* ctx, int64_t len, const double* a, const double* b,
                                   const double* c, double* out_host) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, len >= 0 && out_host);
    if (len == 0) {
        out_host[0] = out_host[1] = out_host[2] = 0.0;
        return MADQP_OK;
    }
    const int nb = grid_for(len);
    {
        ProfScope ps(ctx, MADQP_PROF_VEC);
        hipLaunchKernelGGL(norm_inf3_kernel, dim3(nb), dim3(TPB), 0, ctx->stream, len, a, b, c,
                           ctx->d_part);
        LAUNCH_CHECK(ctx);
        const int ops[3] = {OP_MAX, OP_MAX, OP_MAX};
        int32_t r = finalize(ctx, nb, 3, ops);
        if (r) return r;
    }
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
