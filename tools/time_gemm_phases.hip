// Dev aid: per-workgroup phase stamps (start / end of K loop / end, 100 MHz chip-wide clock) and placement of the SYRK-shaped
// GEMM: K-loop and epilogue durations, and the gap between a workgroup retiring and its successor starting on the same CU.
#include "../gpar_amd/csrc/gemm_f64.h"
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
using namespace gpar;
int main(int argc, char** argv) {
    const int n = 16384, K = argc > 1 ? atoi(argv[1]) : 512;
    double *C, *P; long long* st;
    hipMalloc(&C, sizeof(double) * (size_t)n * n); hipMalloc(&P, sizeof(double) * (size_t)n * 1024);
    hipMemset(C, 0, sizeof(double) * (size_t)n * n); hipMemset(P, 0, sizeof(double) * (size_t)n * 1024);
    GemmArgs p{}; p.A = P; p.B = P; p.C = C; p.m = n; p.n = n; p.k = K; p.lda = 1024; p.ldb = 1024; p.ldc = n;
    p.alpha = -1; p.beta = 1; p.flags = GPAR_GEMM_C_LOWER; p.tiles_m = n / 128; p.tiles_n = n / 128; p.fastA = p.fastB = p.fastC = 1;
    const int ntiles = gemm_num_tiles(p.tiles_m, p.tiles_n, p.flags);
    hipMalloc(&st, sizeof(long long) * 4 * ntiles); p.stamps = st;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gemm_f64_kernel<false, true, 1>), dim3(ntiles), dim3(256), GEMM_LDS_BYTES, 0, p);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %.1f us (events)\n", 1000.0 * ms);
    }
    std::vector<long long> h(4 * (size_t)ntiles);
    hipMemcpy(h.data(), st, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    long long t0 = h[0], tend = 0;
    for (int b = 0; b < ntiles; ++b) { t0 = std::min(t0, h[4 * b]); tend = std::max(tend, h[4 * b + 2]); }
    std::vector<double> main_c, epi_c;
    std::map<long long, std::vector<std::pair<long long, long long>>> per_cu;   // (xcc, se, sh, cu) -> (start, end)
    for (int b = 0; b < ntiles; ++b) {
        main_c.push_back(0.01 * (h[4*b+1] - h[4*b])); epi_c.push_back(0.01 * (h[4*b+2] - h[4*b+1]));
        const long long id = h[4*b+3];
        const long long key = ((id >> 16) & 15) << 16 | (id & 0xff00);   // xcc, se/sh/cu
        per_cu[key].push_back({h[4*b], h[4*b+2]});
    }
    std::sort(main_c.begin(), main_c.end()); std::sort(epi_c.begin(), epi_c.end());
    printf("K=%d tiles=%d kernel %.1f us, CUs seen %zu\n", K, ntiles, 0.01 * (tend - t0), per_cu.size());
    printf("K loop   us: min %.1f median %.1f p90 %.1f max %.1f\n", main_c[0], main_c[ntiles/2], main_c[ntiles*9/10], main_c.back());
    printf("epilogue us: min %.1f median %.1f p90 %.1f max %.1f\n", epi_c[0], epi_c[ntiles/2], epi_c[ntiles*9/10], epi_c.back());
    // per CU: two slots; a retiring workgroup's slot is taken by the next workgroup to start on that CU
    std::vector<double> gaps; double occ = 0, cnt = 0;
    for (auto& kv : per_cu) {
        auto v = kv.second; std::sort(v.begin(), v.end());
        std::vector<long long> ends; for (auto& e : v) ends.push_back(e.second); std::sort(ends.begin(), ends.end());
        // starts after the first two pair up with ends in order
        for (size_t i = 2; i < v.size(); ++i) gaps.push_back(0.01 * (v[i].first - ends[i - 2]));
        long long busy = 0; for (auto& e : v) busy += e.second - e.first;
        occ += (double)busy / (double)(ends.back() - v[0].first); cnt += 1;
    }
    std::sort(gaps.begin(), gaps.end());
    printf("retire -> successor start on the same CU, us: min %.2f median %.2f p90 %.2f max %.2f\n", gaps[0], gaps[gaps.size()/2], gaps[gaps.size()*9/10], gaps.back());
    printf("average resident workgroups per CU while it had work: %.2f\n", occ / cnt);
    return 0;
}
