// Dev aid: per-workgroup phase stamps (start / end of K loop / end) of the SYRK-shaped GEMM.
#include "../gpar_amd/csrc/gemm_f64.h"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace gpar;
int main(int argc, char** argv) {
    const int n = 16384, K = argc > 1 ? atoi(argv[1]) : 256;
    double *C, *P; long long* st;
    hipMalloc(&C, sizeof(double) * (size_t)n * n); hipMalloc(&P, sizeof(double) * (size_t)n * 512);
    hipMemset(C, 0, sizeof(double) * (size_t)n * n); hipMemset(P, 0, sizeof(double) * (size_t)n * 512);
    GemmArgs p; p.A = P; p.B = P; p.C = C; p.m = n; p.n = n; p.k = K; p.lda = 512; p.ldb = 512; p.ldc = n;
    p.alpha = -1; p.beta = 1; p.flags = GPAR_GEMM_C_LOWER; p.tiles_m = n / 128; p.tiles_n = n / 128; p.fastA = p.fastB = p.fastC = 1;
    const int ntiles = gemm_num_tiles(p.tiles_m, p.tiles_n, p.flags);
    hipMalloc(&st, sizeof(long long) * 4 * ntiles); p.stamps = st;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((gemm_f64_kernel<false, true, 1>), dim3(ntiles), dim3(256), GEMM_LDS_BYTES, 0, p);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(4 * (size_t)ntiles);
    hipMemcpy(h.data(), st, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    long long t0 = h[0]; for (int b = 0; b < ntiles; ++b) t0 = std::min(t0, h[4 * b]);
    std::vector<double> main_c, epi_c; long long tend = 0;
    for (int b = 0; b < ntiles; ++b) { main_c.push_back(h[4*b+1]-h[4*b]); epi_c.push_back(h[4*b+2]-h[4*b+1]); tend = std::max(tend, h[4*b+2]); }
    std::sort(main_c.begin(), main_c.end()); std::sort(epi_c.begin(), epi_c.end());
    printf("K=%d tiles=%d total %.1f us (counter @2.4GHz?)\n", K, ntiles, (tend - t0) / 2400.0);
    printf("main loop cycles: min %.0f median %.0f p90 %.0f max %.0f\n", main_c[0], main_c[ntiles/2], main_c[ntiles*9/10], main_c.back());
    printf("epilogue  cycles: min %.0f median %.0f p90 %.0f max %.0f\n", epi_c[0], epi_c[ntiles/2], epi_c[ntiles*9/10], epi_c.back());
    // first-generation blocks: start offsets
    printf("block start offsets (us) of blocks 0,1,255,256,511,512,1024: ");
    for (int b : {0, 1, 255, 256, 511, 512, 1024}) printf("%.1f ", (h[4*b] - t0) / 2400.0);
    printf("\n");
    return 0;
}
