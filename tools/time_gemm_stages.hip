// Dev aid: per-stage stamps of the K loop (thread 0 of 64 workgroups of the third round of a K = 512 SYRK-shaped update):
// stage start (after the barrier), MFMA phase issued, operands of the next stage stored, in 100 MHz ticks.
#ifndef GPAR_GEMM_STAGE_STAMPS
#define GPAR_GEMM_STAGE_STAMPS 1   // 2: stamps in shader-clock cycles (s_memtime) instead of 100 MHz ticks
#endif
#include "../gpar_amd/csrc/gemm_f64.h"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace gpar;
int main(int argc, char** argv) {
    const int n = 16384, K = argc > 1 ? atoi(argv[1]) : 512;
    double *C, *P; long long* st;
    (void)hipMalloc(&C, sizeof(double) * (size_t)n * n); (void)hipMalloc(&P, sizeof(double) * (size_t)n * 1024);
    (void)hipMemset(C, 0, sizeof(double) * (size_t)n * n); (void)hipMemset(P, 0, sizeof(double) * (size_t)n * 1024);
    GemmArgs p{}; p.A = P; p.B = P; p.C = C; p.m = n; p.n = n; p.k = K; p.lda = 1024; p.ldb = 1024; p.ldc = n;
    p.alpha = -1; p.beta = 1; p.flags = GPAR_GEMM_C_LOWER; p.tiles_m = n / 128; p.tiles_n = n / 128; p.fastA = p.fastB = p.fastC = 1;
    const int ntiles = gemm_num_tiles(p.tiles_m, p.tiles_n, p.flags);
    (void)hipMalloc(&st, sizeof(long long) * 64 * 40 * 3); (void)hipMemset(st, 0, sizeof(long long) * 64 * 40 * 3);
    p.stage_stamps = st;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((gemm_f64_kernel<false, true, 1>), dim3(ntiles), dim3(256), GEMM_LDS_BYTES, 0, p);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> h(64 * 40 * 3);
    (void)hipMemcpy(h.data(), st, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    const int nst = K / 16;
    std::vector<double> mf, ss, bar, tot;
    for (int b = 0; b < 64; ++b)
        for (int s = 0; s + 1 < nst && s + 1 < 40; ++s) {
            const long long* a = &h[((size_t)b * 40 + s) * 3];
            const long long* nx = &h[((size_t)b * 40 + s + 1) * 3];
            if (!a[0] || !nx[0]) continue;
            mf.push_back(0.01 * (a[1] - a[0])); ss.push_back(0.01 * (a[2] - a[1])); bar.push_back(0.01 * (nx[0] - a[2])); tot.push_back(0.01 * (nx[0] - a[0]));
        }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    auto mean = [](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return s / v.size(); };
    if (GPAR_GEMM_STAGE_STAMPS == 2) printf("(stamps are shader-clock cycles / 100: read \"us\" as hundreds of cycles; ideal paired stage 81.92)\n");
    printf("K=%d, %zu stages sampled (64 workgroups of the third round); us per stage (ideal paired: 8192 cycles = 3.41 us at 2.4 GHz)\n", K, tot.size());
    printf("stage total      mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f\n", mean(tot), q(tot, .1), q(tot, .5), q(tot, .9), q(tot, 1));
    printf("MFMA phase       mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f\n", mean(mf), q(mf, .1), q(mf, .5), q(mf, .9), q(mf, 1));
    printf("store next stage mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f\n", mean(ss), q(ss, .1), q(ss, .5), q(ss, .9), q(ss, 1));
    printf("barrier          mean %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f\n", mean(bar), q(bar, .1), q(bar, .5), q(bar, .9), q(bar, 1));
    // one workgroup's stages in full
    printf("workgroup 0 (us: total / mfma / store / barrier):");
    for (int s = 0; s + 1 < nst && s + 1 < 40; ++s) {
        const long long* a = &h[(size_t)s * 3]; const long long* nx = &h[(size_t)(s + 1) * 3];
        printf(" %.2f/%.2f/%.2f/%.2f", 0.01 * (nx[0] - a[0]), 0.01 * (a[1] - a[0]), 0.01 * (a[2] - a[1]), 0.01 * (nx[0] - a[2]));
    }
    printf("\n");
    return 0;
}
