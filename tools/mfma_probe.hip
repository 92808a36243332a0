// Standalone hardware probe: issue rate and sustained clock of v_mfma_f64_16x16x4_f64 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
// Accumulators are pinned to VGPRs with inline asm (hipcc otherwise shuttles them through AGPRs
// every iteration, which measures v_accvgpr traffic instead of the matrix pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// KIND: 0 = v_fma_f32, 1 = ds_read_b64 (imm offset), 2 = s_add_u32, 3 = ds_read2_b64, 4 = v_add_u32
template <int NACC, int WPS, int FILL, int KIND = 0>
__global__ __launch_bounds__(256, WPS) void probe(int iters, unsigned long long* out, double seed) {
    __shared__ double lds[4096];
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned laddr = (threadIdx.x & 63) * 8;
    double dl0 = 0, dl1 = 0;
    double2_t dl2 = {0, 0};
    unsigned sc = 0, va = threadIdx.x;
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = seed + threadIdx.x * 1.0e-3, b = 1.0 - threadIdx.x * 3.0e-3;
    float f = (float)seed;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < FILL; ++q) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
                if (KIND == 1) asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(dl0) : "v"(laddr));
                if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                if (KIND == 3) asm volatile("ds_read2_b64 %0, %1 offset0:4 offset1:20" : "=v"(dl2) : "v"(laddr));
                if (KIND == 4) asm volatile("v_add_u32 %0, 4, %0" : "+v"(va));
            }
            if (KIND == 1 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double s = f + dl0 + dl1 + dl2.x + dl2.y + (double)sc + (double)va;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
    if (s == 1234.5) out[0] = 0;
}

template <int NACC, int WPS, int FILL, int KIND = 0>
void run(int iters) {
    const int blocks = 256 * WPS;
    unsigned long long* d;
    const size_t nw = (size_t)blocks * 4;
    (void)hipMalloc(&d, nw * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, WPS, FILL, KIND>), dim3(blocks), dim3(256), 0, 0, 100, d, 0.37);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, WPS, FILL, KIND>), dim3(blocks), dim3(256), 0, 0, iters, d, 0.37);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw * 2);
    (void)hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), ghz(nw);
    for (size_t w = 0; w < nw; ++w) {
        cyc[w] = (double)h[2 * w] / ((double)iters * NACC);
        ghz[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 0.1;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double flops = (double)nw * iters * NACC * 2048.0;
    printf("NACC=%2d waves/SIMD=%d fill=%d kind=%d: %8.3f ms %6.2f TFLOP/s  cycles/MFMA/wave %6.1f (per SIMD %5.1f)  clock %.3f GHz\n",
           NACC, WPS, FILL, KIND, ms, flops / (ms * 1e-3) * 1e-12, cyc[nw / 2], cyc[nw / 2] / WPS, ghz[nw / 2]);
    (void)hipFree(d);
}

int main() {
    run<4, 1, 0>(100000);
    run<8, 1, 0>(50000);
    run<16, 1, 0>(25000);
    run<16, 2, 0>(25000);
    run<8, 2, 0>(50000);
    run<4, 2, 0>(50000);
    run<4, 4, 0>(50000);
    run<8, 4, 0>(25000);
    run<4, 8, 0>(25000);
    run<2, 8, 0>(50000);
    run<16, 1, 4>(25000);
    run<16, 1, 8>(25000);
    run<16, 2, 4>(25000);
    run<16, 1, 1, 0>(25000);
    run<16, 2, 1, 0>(25000);
    run<16, 1, 1, 1>(25000);
    run<16, 1, 2, 1>(25000);
    run<16, 2, 1, 1>(25000);
    run<16, 2, 2, 1>(25000);
    run<16, 1, 1, 3>(25000);
    run<16, 2, 1, 3>(25000);
    run<16, 1, 2, 2>(25000);
    run<16, 1, 8, 2>(25000);
    run<16, 2, 4, 2>(25000);
    run<16, 1, 1, 4>(25000);
    run<16, 2, 1, 4>(25000);
    return 0;
}
