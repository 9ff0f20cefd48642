// Dev aid: wall-clock (100 MHz) stamps along the hand-off chain of the second-generation panel kernel (panel2.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/time_panel2.hip -o tools/time_panel2 && tools/time_panel2 [N]
#include "../gpar_amd/csrc/panel2.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace gpar;

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, lda = N, S = 8;
    double* A;
    long long* st;
    const size_t nst = 16 * 17 * 8;
    hipMalloc(&A, sizeof(double) * (size_t)N * lda);
    hipMalloc(&st, 8 * nst);
    std::vector<double> h((size_t)N * 512, 0.0);
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < 512 && c <= r; ++c) h[(size_t)r * 512 + c] = (r == c) ? 600.0 : 0.5 / (1 + (r - c) % 7);
    PanelArgs p{A, N, lda, 0, S, nullptr, nullptr, st};
    p.progressive = argc > 2 ? atoi(argv[2]) : 1;
    p.pairs = 1;
    p.split = p.progressive && (argc > 3 ? atoi(argv[3]) : 1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_panel2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int R = (N + 63) / 64;
    for (int rep = 0; rep < 12; ++rep) {
        for (int r = 0; r < N; ++r) hipMemcpyAsync(A + (size_t)r * lda, h.data() + (size_t)r * 512, 512 * 8, hipMemcpyHostToDevice, 0);
        hipMemsetAsync(A + 8, 0, 56 * 8, 0);
        hipMemsetAsync(A + 2 * (size_t)lda + 8, 0, 56 * 8, 0);
        hipMemsetAsync(A + 3 * (size_t)lda + 8, 0, 56 * 8, 0);
        hipMemsetAsync(st, 0, 8 * nst, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(potrf_panel2_kernel, dim3(R - S + (p.split ? S + (S - 1) * (S - 2) / 2 : S)), dim3(256), P2_LDS_BYTES, 0, p);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 9) printf("panel2 kernel, N = %d (R = %d row blocks): %.1f us\n", N, R, ms * 1e3);
    }
    {   // checksum of the factored panel (compare the two modes: same bits)
        std::vector<double> out((size_t)N * 512);
        for (int r = 0; r < N; ++r) hipMemcpy(out.data() + (size_t)r * 512, A + (size_t)r * lda, 512 * 8, hipMemcpyDeviceToHost);
        unsigned long long h = 1469598103934665603ull;
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < 512 && c <= r; ++c) {
                unsigned long long v;
                memcpy(&v, &out[(size_t)r * 512 + c], 8);
                h = (h ^ v) * 1099511628211ull;
            }
        printf("progressive = %d split = %d: lower-trapezoid hash %016llx, L[N-1][511] = %.17g\n", p.progressive, p.split, h, out[(size_t)(N - 1) * 512 + 511]);
    }
    std::vector<long long> s(nst);
    hipMemcpy(s.data(), st, 8 * nst, hipMemcpyDeviceToHost);
    auto at = [&](int rb, int c, int k) { return s[((size_t)rb * 17 + c) * 8 + k]; };
    const long long t0 = at(0, 0, 0);
    auto us = [&](long long v) { return v ? (v - t0) * 0.01 : -1.0; };
    printf("team row t: per column c < t: start, chunks done, triangle available, in LDS, strip done, D accumulated, published [us since row 0 started]\n");
    for (int t = 0; t < S; ++t) {
        for (int c = 0; c < t; ++c)
            printf("  t=%d c=%d: %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f\n", t, c, us(at(t, c, 0)), us(at(t, c, 1)), us(at(t, c, 2)),
                   us(at(t, c, 3)), us(at(t, c, 4)), us(at(t, c, 5)), us(at(t, c, 6)));
        printf("  t=%d diag: start %7.2f assembled %7.2f factored %7.2f inverses+stores %7.2f published %7.2f\n", t, us(at(t, t, 0)),
               us(at(t, t, 1)), us(at(t, t, 2)), us(at(t, t, 3)), us(at(t, t, 4)));
    }
    for (int t = 0; t < S; ++t) {
        printf("  t=%d rounds end:", t);
        for (int k = 0; k < 8; ++k) printf(" %7.2f", us(at(t, 10, k)));
        printf("   helper (start r3 r4 r5 r6 r7 | end r3 r4 r5):");
        for (int k = 0; k < 8; ++k) printf(" %7.2f", us(at(t, 9, k)));
        printf("\n");
        for (int ww = 1; ww < 4; ++ww) {
            printf("      wave %d at the barrier:", ww);
            for (int k = 0; k < 8; ++k) printf(" %7.2f", us(at(t, 10 + ww, k)));
            printf("\n");
        }
    }
    for (int t = 1; t < S; ++t)
        for (int cc = 13; cc < 15; ++cc) {
            printf("  t=%d last strip, sub-step %d: flag/issue %7.2f landed %7.2f strip %7.2f xout+dself %7.2f barrier %7.2f xstore+dcross %7.2f\n", t, cc - 12,
                   us(at(t, cc, 0)), us(at(t, cc, 1)), us(at(t, cc, 2)), us(at(t, cc, 3)), us(at(t, cc, 4)), us(at(t, cc, 5)));
        }
    printf("bulk row blocks 8..15: per column c: start, chunks done, triangle available, in LDS, strip done\n");
    for (int rb = 8; rb < 16 && rb < R; rb += 7)
        for (int c = 0; c < S; ++c)
            printf("  rb=%d c=%d: %7.2f %7.2f %7.2f %7.2f %7.2f\n", rb, c, us(at(rb, c, 0)), us(at(rb, c, 1)), us(at(rb, c, 2)), us(at(rb, c, 3)),
                   us(at(rb, c, 4)));
    return 0;
}
