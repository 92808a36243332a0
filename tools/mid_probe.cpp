// Diagnostic build of the fused mid-size factorisation step with phase stamps (never part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMADQP_MID_STAMPS -x hip tools/mid_probe.cpp \
//         madqp_jl_amd/csrc/{gemm_f64,ctx,gen}.hip -ldl -o tools/mid_probe
// Prints, per block step k, the phases of the workgroup holding the diagonal tile and of workgroups 1 and 2;
// s_memrealtime, 100 MHz.  With -DMADQP_POTF2_STAMPS as well: the phases of the diagonal kernel body inside the last step.
// (A stamped build is scheduled differently from the product build: differences of a few percent between two variants
// must be confirmed on the product build -- tools/mid_steps.py on a kernel trace of bench.py.)
#include "../madqp_jl_amd/csrc/chol.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    setenv("MADQP_CHOL_MID_MAX", "100000", 1);
    madqp_ctx* ctx = nullptr;
    if (madqp_ctx_create(0, nullptr, &ctx)) return 1;
    const int64_t n = argc > 1 ? atoll(argv[1]) : 2048, lda = (n + 127) / 128 * 128;
    std::vector<double> h(lda * lda, 0.0);
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = j; i < n; ++i) h[i + j * lda] = (i == j) ? 2.0 * n : 1.0 / (1.0 + (double)((i * 7 + j * 13) % 17));
    double* A;
    (void)hipMalloc(&A, sizeof(double) * lda * lda);
    madqp_chol* ch = nullptr;
    if (madqp_chol_create(ctx, n, &ch)) return 1;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemcpy(A, h.data(), sizeof(double) * lda * lda, hipMemcpyHostToDevice);
        int32_t info = -1;
        if (madqp_chol_factor(ch, A, lda, &info)) return 1;
        static unsigned long long st[64][24];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(madqp_mid_stamps), sizeof(st));
        if (rep < 2) continue;
        const int nblk = (int)(lda / 128);
        printf("info %d; per step: workgroup 0 [update, to LDS, factor + invert]; workgroup 1 [update, store]\n", info);
        for (int k = 0; k < nblk && k < 64; ++k) {
            auto us = [&](int a, int b) { return (double)(st[k][b] - st[k][a]) * 0.01; };
            printf("k=%2d  diag %5.1f %5.1f %5.1f | wg1 %5.1f +%4.1f wg2 %5.1f +%4.1f (starts +%.1f)", k, us(0, 1), us(1, 2), us(2, 3), us(8, 9), us(9, 15), us(16, 17), us(17, 23), us(0, 16));
            if (st[k][4]) printf(" | syrk: operand in LDS +%.1f, products +%.1f, all waves done +%.1f, S written +%.1f", us(0, 4), us(4, 5), us(5, 6), us(6, 1));
            if (k + 1 < nblk) printf(" | next step starts +%.1f after the diagonal workgroup ends", (double)(st[k + 1][0] - st[k][3]) * 0.01);
            printf("\n");
        }
    }
#ifdef MADQP_POTF2_STAMPS
    {   // the phases of the diagonal kernel body inside the LAST block step (the stamps of every step land in the same slots)
        unsigned long long st[64];
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(madqp_potf2_stamps), sizeof(st));
        auto us = [&](int a, int b) { return (double)(st[b] - st[a]) * 0.01; };
        printf("last step, diagonal body: total %.1f us | to LDS %.1f | first 16x16 %.2f | J-steps %.1f | store L %.1f | store W %.1f\n",
               us(0, 28), us(0, 1), us(1, 2), us(2, 25), us(25, 26), us(27, 28));
        for (int J = 0; J < 8; ++J) {
            const int prev = (J == 0) ? 2 : 4 + 3 * (J - 1);
            printf("   J=%d panel %.2f  trailing || next 16x16 %.2f\n", J, us(prev, 3 + 3 * J), us(3 + 3 * J, 4 + 3 * J));
        }
    }
#endif
    return 0;
}
