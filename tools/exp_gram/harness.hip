// Experiment harness: times variants of the generated C3 Gram kernel (gram_c3.inc = GPAR_JIT_DUMP output with three hooks:
// LB = occupancy bound, NOMATH = arithmetic replaced by one add, STORE_COND = store predicate).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef LB
#define LB 2
#endif
#ifndef NOMATH
#define NOMATH 0
#endif
#ifndef STORE_COND
#define STORE_COND true
#endif
#include SRC_INC

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const int strip = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<double> z((size_t)n * DZ);
    srand(1);
    for (auto& v : z) v = (rand() / (double)RAND_MAX - 0.5) * 3.0;
    double *dz, *dK;
    hipMalloc(&dz, z.size() * 8);
    const int ldk_alloc = n + (argc > 4 ? atoi(argv[4]) : 0);
    hipMalloc(&dK, (size_t)n * ldk_alloc * 8);
    hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice);
    gj_kspec ks{};
    ks.nterms = 3;
    ks.coef[0] = 1.0; ks.coef[1] = 0.3; ks.coef[2] = 0.7;
    const int nt = (n + 63) / 64;
    const int full = nt / strip, rest = nt - full * strip;
    const unsigned grid = (unsigned)((long long)strip * full * (full + 1) / 2 + (long long)rest * (full + 1));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const int ldk = n + (argc > 4 ? atoi(argv[4]) : 0);
    for (int it = 0; it < reps + 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gram_jit, dim3(grid), dim3(256), 0, 0, ks, dz, n, DZ, dz, n, DZ, dK, ldk, 1, nullptr, 1e-3, nullptr, 1, 0LL, 0LL, strip);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 3) { best = ms < best ? ms : best; sum += ms; }
        if (getenv("VERBOSE")) { long long d[2]; hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), 16); printf("%.3f(%.0fMHz,%.0fus) ", ms, d[0] / (d[1] / 100.0), d[1] / 100.0); }
    }
    double chk = 0;
    std::vector<double> row(n);
    hipMemcpy(row.data(), dK + (size_t)(n - 1) * ldk, n * 8, hipMemcpyDeviceToHost);
    for (double v : row) chk += v;
    const double bytes = 8.0 * ((double)n * (n + 1) / 2);
    printf("%-40s n=%d strip=%d ldpad=%d  best %.4f ms  mean %.4f ms  %.2f TB/s  checksum %.12g\n", VARIANT_NAME, n, strip, ldk - n, best, sum / reps, bytes / best * 1e-9, chk);
    return 0;
}
