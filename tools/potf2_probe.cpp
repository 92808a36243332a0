// Diagnostic build of the diagonal-block kernel with phase stamps (never part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMADQP_POTF2_STAMPS -x hip tools/potf2_probe.cpp \
//         madqp_jl_amd/csrc/{gemm_f64,ctx,gen}.hip -o tools/potf2_probe
// Prints the time of each phase of potf2_inv_kernel (s_memrealtime, 100 MHz) for one 128 x 128 block.
#include "../madqp_jl_amd/csrc/chol.hip"

#include <cstdio>
#include <vector>

int main() {
    madqp_ctx* ctx = nullptr;
    if (madqp_ctx_create(0, nullptr, &ctx)) return 1;
    const int64_t n = 128, lda = 128;
    std::vector<double> h(n * lda, 0.0);
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < n; ++i) h[i + j * lda] = (i == j) ? 200.0 : 1.0 / (1.0 + (double)((i * 7 + j * 13) % 17));
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < j; ++i) h[i + j * lda] = h[j + i * lda];
    double* A;
    (void)hipMalloc(&A, sizeof(double) * n * lda);
    madqp_chol* ch = nullptr;
    if (madqp_chol_create(ctx, n, &ch)) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemcpy(A, h.data(), sizeof(double) * n * lda, hipMemcpyHostToDevice);
        int32_t info = -1;
        if (madqp_chol_factor(ch, A, lda, &info)) return 1;
        unsigned long long st[64];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(madqp_potf2_stamps), sizeof(st));
        auto us = [&](int a, int b) { return (double)(st[b] - st[a]) * 0.01; };
        printf("rep %d info %d: total %.1f us | load %.1f | J-steps %.1f | store L %.1f | inverse %.1f | store W %.1f\n", rep,
               info, us(0, 28), us(0, 1), us(1, 25), us(25, 26), us(26, 27), us(27, 28));
        printf("   first diag16 %.2f\n", us(1, 2));
        double p = 0, t = 0;
        for (int J = 0; J < 8; ++J) {
            const int prev = (J == 0) ? 2 : 4 + 3 * (J - 1);
            printf("   J=%d panel %.2f  trailing || next diag16 %.2f\n", J, us(prev, 3 + 3 * J), us(3 + 3 * J, 4 + 3 * J));
            p += us(prev, 3 + 3 * J);
            t += us(3 + 3 * J, 4 + 3 * J);
        }
        printf("   sum: panel %.1f  trailing || diag16 %.1f\n", p, t);
    }
    return 0;
}
