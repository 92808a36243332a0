// Diagnostic (never part of the product): what a small dependent launch costs in a pre-filled in-order stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/launch_probe.cpp -o tools/launch_probe
// Prints microseconds per launch for an elementwise pass over n doubles at several n and grid sizes, and for a
// two-launch reduction (partials + final), back to back with no host synchronisation in between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void axpy_kernel(double* __restrict__ y, const double* __restrict__ x, double a, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = a * x[i] + y[i];
}
__global__ void partial_max_kernel(const double* __restrict__ x, double* __restrict__ part, int n) {
    __shared__ double s[256];
    double m = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmax(m, fabs(x[i]));
    s[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}
__global__ void final_max_kernel(const double* __restrict__ part, double* __restrict__ out, int nb) {
    double m = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) m = fmax(m, part[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o));
    if (threadIdx.x == 0) *out = m;
}

int main() {
    double *x, *y, *part, *out;
    const int nmax = 1 << 20;
    (void)hipMalloc(&x, nmax * 8);
    (void)hipMalloc(&y, nmax * 8);
    (void)hipMalloc(&part, 4096 * 8);
    (void)hipMalloc(&out, 8);
    (void)hipMemset(x, 0, nmax * 8);
    (void)hipMemset(y, 0, nmax * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int reps = 2000;
    for (int n : {5000, 20000, 100000}) {
        for (int grid : {1, 20, 80, 256}) {
            for (int w = 0; w < 2; ++w) {
                (void)hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, 0, y, x, 1.0, n);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("axpy n=%6d grid=%3d: %.2f us per launch\n", n, grid, ms * 1e3 / reps);
        }
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(partial_max_kernel, dim3(80), dim3(256), 0, 0, x, part, n);
            hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(64), 0, 0, part, out, 80);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("max  n=%6d partials(80) + final: %.2f us per pair\n", n, ms * 1e3 / reps);
    }
    // the runtime's own small copies and fills in the same position (what hipMemcpyAsync D2D / hipMemsetAsync cost in a chain)
    for (int n : {5000, 20000}) {
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(axpy_kernel, dim3(80), dim3(256), 0, 0, y, x, 1.0, n);
            (void)hipMemcpyAsync(y, x, (size_t)n * 8, hipMemcpyDeviceToDevice, 0);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("axpy + hipMemcpyAsync D2D n=%6d: %.2f us per pair\n", n, ms * 1e3 / reps);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(axpy_kernel, dim3(80), dim3(256), 0, 0, y, x, 1.0, n);
            (void)hipMemsetD32Async((hipDeviceptr_t)y, 0x7FF8A5A5, 2 * (size_t)n, 0);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("axpy + hipMemsetD32Async     n=%6d: %.2f us per pair\n", n, ms * 1e3 / reps);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(axpy_kernel, dim3(80), dim3(256), 0, 0, y, x, 1.0, n);
            hipLaunchKernelGGL(axpy_kernel, dim3(80), dim3(256), 0, 0, y, x, 1.0, n);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("axpy + axpy                  n=%6d: %.2f us per pair\n", n, ms * 1e3 / reps);
    }
    // the same chain replayed from a graph: no host work per launch
    {
        hipStream_t st;
        (void)hipStreamCreate(&st);
        hipGraph_t g;
        hipGraphExec_t ex;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(axpy_kernel, dim3(80), dim3(256), 0, st, y, x, 1.0, 20000);
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        for (int w = 0; w < 2; ++w) {
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 10; ++r) (void)hipGraphLaunch(ex, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
        }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("graph of 200 axpy launches (n=20000, grid=80), replayed: %.2f us per node\n", ms * 1e3 / 2000);
    }
    // host round trip: launch, read 8 bytes back, launch
    {
        double h;
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 200; ++r) {
            hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(64), 0, 0, part, out, 80);
            (void)hipMemcpyAsync(&h, out, 8, hipMemcpyDeviceToHost, 0);
            (void)hipStreamSynchronize(0);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("launch + 8-byte read-back + synchronise: %.2f us per round trip\n", ms * 1e3 / 200);
    }
    return 0;
}
