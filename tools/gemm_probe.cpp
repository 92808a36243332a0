// Diagnostic build of the GEMM core with per-workgroup clock stamps (never part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMADQP_STAMPS -x hip tools/gemm_probe.cpp \
//         madqp_jl_amd/csrc/{gemm_f64,ctx,gen}.hip -o tools/gemm_probe
// Reports TFLOP/s, the in-kernel shader clock (s_memtime / s_memrealtime) and cycles per K-stage.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/madqp.h"
#include "../madqp_jl_amd/csrc/common.h"
unsigned long long* madqp_stamp_buffer = nullptr;

static void run(madqp_ctx* ctx, int64_t n, int64_t k, bool scale, bool base) {
    double *B, *w, *H, *C;
    (void)hipMalloc(&B, sizeof(double) * n * k);
    (void)hipMalloc(&w, sizeof(double) * k);
    (void)hipMalloc(&H, sizeof(double) * n * n);
    (void)hipMalloc(&C, sizeof(double) * n * n);
    madqp_gen_normal(ctx, 123, 0, n * k, B);
    madqp_gen_normal(ctx, 77, 0, k, w);
    madqp_gen_normal(ctx, 99, 0, n * n, H);
    const int64_t tiles = (n + 127) / 128;
    const int64_t nt = tiles * (tiles + 1) / 2;
    (void)hipMalloc(&madqp_stamp_buffer, nt * 16);
    (void)hipMemset(madqp_stamp_buffer, 0, nt * 16);
    madqp_syrk_assemble(ctx, n, k, B, n, scale ? w : nullptr, base ? H : nullptr, n, nullptr, C, n);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    const int reps = 3;
    for (int r = 0; r < reps; ++r)
        madqp_syrk_assemble(ctx, n, k, B, n, scale ? w : nullptr, base ? H : nullptr, n, nullptr, C, n);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h(nt * 2);
    (void)hipMemcpy(h.data(), madqp_stamp_buffer, nt * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int64_t i = 0; i < nt; ++i)
        if (h[2 * i + 1]) {
            ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
            cyc.push_back((double)h[2 * i] / ((double)k / 16.0));
        }
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    const double flops = (double)k * n * n;  // lower triangle: 2 * k * n^2 / 2
    printf("n=%lld k=%lld scale=%d base=%d: %.2f ms  %.2f TFLOP/s  clock median %.3f GHz (p5 %.3f p95 %.3f)  cycles/stage median %.0f\n",
           (long long)n, (long long)k, scale, base, ms, flops / (ms * 1e-3) * 1e-12, ghz[ghz.size() / 2],
           ghz[ghz.size() / 20], ghz[ghz.size() * 19 / 20], cyc[cyc.size() / 2]);
    (void)hipFree(B); (void)hipFree(w); (void)hipFree(H); (void)hipFree(C); (void)hipFree(madqp_stamp_buffer);
    madqp_stamp_buffer = nullptr;
}

// left-looking panel update of chol.hip on an n x n column-major matrix: C[J0:n, J0:J0+W] -= L[J0:n,0:J0] L[J0:J0+W,0:J0]'
static void run_panel(madqp_ctx* ctx, int64_t n, int64_t J0, int64_t W, int variant) {
    double* A;
    (void)hipMalloc(&A, sizeof(double) * n * n);
    madqp_gen_normal(ctx, 5, 0, n * n, A);
    GemmArgs g{};
    g.X = A + J0; g.ldx = n; g.Y = A + J0; g.ldy = n;
    g.C = A + J0 + J0 * n; g.ldc = n; g.Cin = (variant & 1) ? nullptr : g.C; g.ldcin = n;
    g.alpha = -1.0; g.beta = 1.0; g.M = n - J0; g.N = W; g.K = J0; g.diag_off = 0; g.lower_only = (variant & 2) ? 0 : 1;
    const int64_t tm = (g.M + 127) / 128, tn = (g.N + 127) / 128;
    int64_t nt = tm * tn;
    (void)hipMalloc(&madqp_stamp_buffer, nt * 16);
    (void)hipMemset(madqp_stamp_buffer, 0, nt * 16);
    madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_GEMM);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    const int reps = 3;
    for (int r = 0; r < reps; ++r) madqp_gemm_tn(ctx, g, MADQP_PROF_POTRF_GEMM);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    int64_t active = g.lower_only ? tm * tn - tn * (tn - 1) / 2 : tm * tn;
    std::vector<unsigned long long> h(nt * 2);
    (void)hipMemcpy(h.data(), madqp_stamp_buffer, nt * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (int64_t i = 0; i < nt; ++i) if (h[2 * i + 1]) cyc.push_back((double)h[2 * i] / ((double)g.K / 16.0));
    std::sort(cyc.begin(), cyc.end());
    printf("panel n=%lld J0=%lld W=%lld variant=%d: tiles %lld (%.2f rounds) %.2f ms %.2f TFLOP/s cycles/stage median %.0f p90 %.0f\n",
           (long long)n, (long long)J0, (long long)W, variant, (long long)active, active / 512.0, ms,
           2.0 * active * 128 * 128 * g.K / (ms * 1e-3) * 1e-12, cyc.size() ? cyc[cyc.size() / 2] : 0.0,
           cyc.size() ? cyc[cyc.size() * 9 / 10] : 0.0);
    std::vector<double> dur;  // per-workgroup wall time of its tile, ms (s_memrealtime ticks of 10 ns)
    for (int64_t i = 0; i < nt; ++i) if (h[2 * i + 1]) dur.push_back((double)h[2 * i + 1] * 1e-5);
    std::sort(dur.begin(), dur.end());
    if (!dur.empty())
        printf("    per-workgroup tile time ms: min %.3f p10 %.3f p50 %.3f p90 %.3f max %.3f (n=%zu)\n", dur.front(),
               dur[dur.size() / 10], dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back(), dur.size());
    (void)hipFree(A); (void)hipFree(madqp_stamp_buffer); madqp_stamp_buffer = nullptr;
}

int main() {
    madqp_ctx* ctx;
    if (madqp_ctx_create(0, nullptr, &ctx)) return 1;
    if (getenv("PROBE_TAIL")) {  // single-round and few-round launches of the outer-panel update at C-main
        run_panel(ctx, 50000, 43008, 1280, 0);
        run_panel(ctx, 50000, 40960, 2048, 0);
        run_panel(ctx, 50000, 32000, 1920, 0);
        run_panel(ctx, 50000, 20864, 2048, 0);
    } else if (getenv("PROBE_PANEL")) {
        run_panel(ctx, 49920, 20480, 2048, 0);
        run_panel(ctx, 50000, 20480, 2048, 0);
        run_panel(ctx, 49920, 30720, 1920, 0);
        run_panel(ctx, 50000, 30720, 1920, 0);
        run_panel(ctx, 49920, 40960, 2048, 0);
        run_panel(ctx, 50000, 40960, 2048, 0);
    } else if (getenv("PROBE_MID")) {  // the assembly of the mid sizes (configs[1]: n = 5000, K = 2000) beside a long-K one
        run(ctx, 5000, 2000, true, true);
        run(ctx, 5120, 2048, true, true);
        run(ctx, 5000, 20000, true, true);
        run(ctx, 3000, 1200, true, true);
        run(ctx, 8000, 3200, true, true);
    } else if (getenv("PROBE_ONE")) {
        const char* e = getenv("PROBE_N");
        run(ctx, e ? atol(e) : 24576, 20480, true, getenv("PROBE_NOBASE") ? false : true);
    } else {
        run(ctx, 16384, 4096, false, false);
        run(ctx, 16384, 4096, true, true);
        run(ctx, 32768, 8192, true, true);
        run(ctx, 4096, 4096, false, false);
    }
    madqp_ctx_destroy(ctx);
    return 0;
}
