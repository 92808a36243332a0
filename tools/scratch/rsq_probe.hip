#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* a, double* y0, double* y1, double* y2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i];
    const double y = __builtin_amdgcn_rsq(x);
    y0[i] = y;
    {   // one third-order step: y (1 + e + 1.5 e^2), e = (1 - x y^2) / 2
        const double g = x * y, h = 0.5 * y;
        const double e = __builtin_fma(-h, g, 0.5);
        const double t = y * e, p = __builtin_fma(1.5, e, 1.0);
        y1[i] = __builtin_fma(t, p, y);
    }
    {   // two coupled steps (chol.hip fast_rsqrt)
        double g = x * y, h = 0.5 * y;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        r = __builtin_fma(-h, g, 0.5);
        h = __builtin_fma(h, r, h);
        y2[i] = h + h;
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<double> a(n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> um(1.0, 4.0), ue(-300, 300);
    for (int i = 0; i < n; ++i) a[i] = um(rng) * std::pow(2.0, std::floor(ue(rng)));
    double *da, *d0, *d1, *d2;
    hipMalloc(&da, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, d0, d1, d2, n);
    std::vector<double> y0(n), y1(n), y2(n);
    hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(y2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)a[i]);
        e0 = fmax(e0, (double)fabsl((y0[i] - t) / t));
        e1 = fmax(e1, (double)fabsl((y1[i] - t) / t));
        e2 = fmax(e2, (double)fabsl((y2[i] - t) / t));
    }
    printf("max rel err: estimate %.3e (2^%.1f)  one step %.3e (%.2f ulp)  two steps %.3e (%.2f ulp)\n", e0, log2(e0), e1, e1 / 1.11e-16, e2, e2 / 1.11e-16);
    return 0;
}
