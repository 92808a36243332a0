// Position-addressable synthetic data generator, bit-identical to oracle/qp.py (integer hash +
// Irwin-Hall-4, no transcendental functions): any tile can be produced on any GPU or on the CPU.
#include "common.h"

namespace {
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double gen_normal(uint64_t key, uint64_t idx) {
    const uint64_t h = mix64(key + idx);
    const int64_t g = (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) +
                      (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48) - 131070;
    return (double)g * 0x1.bb67ae86627e7p-16;
}
__global__ __launch_bounds__(256) void gen_normal_kernel(uint64_t key, uint64_t idx0, int64_t count,
                                                         double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        out[i] = gen_normal(key, idx0 + (uint64_t)i);
}
__global__ __launch_bounds__(256) void gen_wigner_kernel(uint64_t key, int64_t n, double inv_sqrt_n,
                                                         double* __restrict__ H, int64_t ld) {
    const int64_t total = n * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / n, i = e % n;
        const uint64_t lo = (uint64_t)(i < j ? i : j), hi = (uint64_t)(i < j ? j : i);
        double v = gen_normal(key, lo * (uint64_t)n + hi) * inv_sqrt_n;
        if (i == j) v = 3.0 + v;
        H[i + j * ld] = v;
    }
}
// block-cyclic pieces (multi-GPU path, dist.hip): local column c of a direction with modulus R and residue r is the
// global column ((c / nb) * R + r) * nb + c % nb
__device__ __forceinline__ int64_t cyc_global(int64_t c, int64_t nb, int64_t R, int64_t r) {
    return ((c / nb) * R + r) * nb + c % nb;
}
__global__ __launch_bounds__(256) void gen_normal_cyclic_kernel(uint64_t key, int64_t m, int64_t nx, int64_t nb, int64_t R,
                                                                int64_t r, int64_t ncols, double* __restrict__ out,
                                                                int64_t ldo) {
    const int64_t total = m * ncols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / ncols, c = e % ncols;
        const int64_t g = cyc_global(c, nb, R, r);
        out[k * ldo + c] = (g < nx) ? gen_normal(key, (uint64_t)k * (uint64_t)nx + (uint64_t)g) : 0.0;
    }
}
__global__ __launch_bounds__(256) void gen_wigner_cyclic_kernel(uint64_t key, int64_t n, double inv_sqrt_n, int64_t nb,
                                                                int64_t P, int64_t p, int64_t Q, int64_t q, int64_t mloc,
                                                                int64_t nloc, double* __restrict__ H, int64_t ld) {
    const int64_t total = mloc * nloc;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t jl = e / mloc, il = e % mloc;
        const int64_t i = cyc_global(il, nb, P, p), j = cyc_global(jl, nb, Q, q);
        double v = 0.0;
        if (i < n && j < n) {
            const uint64_t lo = (uint64_t)(i < j ? i : j), hi = (uint64_t)(i < j ? j : i);
            v = gen_normal(key, lo * (uint64_t)n + hi) * inv_sqrt_n;
            if (i == j) v = 3.0 + v;
        }
        H[il + jl * ld] = v;
    }
}
}  // namespace

extern "C" int32_t madqp_gen_normal_cyclic(madqp_ctx* ctx, uint64_t key, int64_t m, int64_t nx, int64_t nb, int32_t R,
                                           int32_t r, int64_t ncols_local, double* out, int64_t ldo) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, m >= 0 && nx >= 0 && nb > 0 && R >= 1 && r >= 0 && r < R && ncols_local >= 0 && ldo >= ncols_local);
    if (m * ncols_local == 0) return MADQP_OK;
    ARG_TRY(ctx, out != nullptr);
    const int64_t blocks = std::min<int64_t>((m * ncols_local + 255) / 256, 65536);
    hipLaunchKernelGGL(gen_normal_cyclic_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, key, m, nx, nb,
                       (int64_t)R, (int64_t)r, ncols_local, out, ldo);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

extern "C" int32_t madqp_gen_wigner_cyclic(madqp_ctx* ctx, uint64_t key, int64_t n, double inv_sqrt_n, int64_t nb,
                                           int32_t P, int32_t p, int32_t Q, int32_t q, int64_t mloc, int64_t nloc,
                                           double* H, int64_t ld) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, n >= 0 && nb > 0 && P >= 1 && Q >= 1 && p >= 0 && p < P && q >= 0 && q < Q && mloc >= 0 && nloc >= 0 &&
                     ld >= mloc);
    if (mloc * nloc == 0) return MADQP_OK;
    ARG_TRY(ctx, H != nullptr);
    const int64_t blocks = std::min<int64_t>((mloc * nloc + 255) / 256, 65536);
    hipLaunchKernelGGL(gen_wigner_cyclic_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, key, n, inv_sqrt_n, nb,
                       (int64_t)P, (int64_t)p, (int64_t)Q, (int64_t)q, mloc, nloc, H, ld);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

extern "C" int32_t madqp_gen_normal(madqp_ctx* ctx, uint64_t key, uint64_t idx0, int64_t count,
                                    double* out) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, count >= 0 && (count == 0 || out));
    if (count == 0) return MADQP_OK;
    const int64_t blocks = std::min<int64_t>((count + 255) / 256, 65536);
    hipLaunchKernelGGL(gen_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, key, idx0,
                       count, out);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}

extern "C" int32_t madqp_gen_wigner(madqp_ctx* ctx, uint64_t key, int64_t n, double inv_sqrt_n,
                                    double* H, int64_t ld) {
    if (!ctx) return MADQP_ERR_ARG;
    ARG_TRY(ctx, n >= 0 && (n == 0 || (H && ld >= n)));
    if (n == 0) return MADQP_OK;
    const int64_t blocks = std::min<int64_t>((n * n + 255) / 256, 65536);
    hipLaunchKernelGGL(gen_wigner_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, key, n,
                       inv_sqrt_n, H, ld);
    LAUNCH_CHECK(ctx);
    return MADQP_OK;
}
