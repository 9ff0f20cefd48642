// Dev aid: cycle stamps along the critical chain of the fused panel kernel (same-XCD stamps only are comparable;
// durations within one workgroup are exact).
#include "../gpar_amd/csrc/panel.h"
#include <cstdio>
#include <vector>
using namespace gpar;
int main() {
    const int N = 16384, lda = 16384;
    double* A; long long* st;
    hipMalloc(&A, sizeof(double) * (size_t)N * lda); hipMalloc(&st, 8 * (64 + 256 * 9));
    std::vector<double> h((size_t)N * 520, 0.0);
    // SPD-ish first 512 columns: diag dominant
    for (int r = 0; r < N; ++r) for (int c = 0; c < 512 && c <= r; ++c) h[(size_t)r * 520 + c] = (r == c) ? 600.0 : 0.5 / (1 + (r - c) % 7);
    for (int r = 0; r < N; ++r) hipMemcpy(A + (size_t)r * lda, h.data() + (size_t)r * 520, 512 * 8, hipMemcpyHostToDevice);
    PanelArgs p{A, N, lda, 0, 8, nullptr, nullptr, st};
    hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PNL_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        for (int r = 0; r < N; ++r) hipMemcpyAsync(A + (size_t)r * lda, h.data() + (size_t)r * 520, 512 * 8, hipMemcpyHostToDevice, 0);
        hipMemsetAsync(A + 8, 0, 56 * 8, 0); hipMemsetAsync(A + lda + 8, 0, 56 * 8, 0);
        hipMemsetAsync(st, 0, 8 * (64 + 256 * 9), 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(256), dim3(256), PNL_LDS_BYTES, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("panel kernel: %.1f us\n", ms * 1e3);
    }
    long long s[64]; hipMemcpy(s, st, sizeof(s), hipMemcpyDeviceToHost);
    printf("step: diag_load diag_compute diag_store+publish | (next owner) wait strip(incl loads) update   [cycles]\n");
    for (int k = 0; k < 8; ++k)
        printf("%d: %6lld %6lld %6lld | %6lld %6lld %6lld\n", k, s[k*8+1]-s[k*8+0], s[k*8+2]-s[k*8+1], s[k*8+3]-s[k*8+2],
               s[k*8+5]-s[k*8+4], s[k*8+6]-s[k*8+5], s[k*8+7]-s[k*8+6]);
    // ---- the same panel kernel launched ~200 us into a trailing-update-shaped SYRK running on a low-priority side stream
    {
        int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t side; hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo);
        double* C; hipMalloc(&C, sizeof(double) * (size_t)N * lda); hipMemset(C, 0, sizeof(double) * (size_t)N * lda);
        for (int r = 0; r < N; ++r) hipMemcpyAsync(A + (size_t)r * lda, h.data() + (size_t)r * 520, 512 * 8, hipMemcpyHostToDevice, 0);
        hipMemsetAsync(A + 8, 0, 56 * 8, 0); hipMemsetAsync(A + lda + 8, 0, 56 * 8, 0);
        hipMemsetAsync(st, 0, 8 * (64 + 256 * 9), 0);
        hipDeviceSynchronize();
        hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
        hipEventRecord(t0, side);
        gemm_launch(0, 1, N - 1024, N - 1024, 512, -1.0, A + (size_t)1024 * lda, lda, A + (size_t)1024 * lda, lda, 1.0, C, lda, GPAR_GEMM_C_LOWER, side, 1);
        hipEventRecord(t1, side);
        gemm_launch(0, 1, N - 512, 512, 512, -1.0, A + (size_t)512 * lda, lda, A + (size_t)512 * lda, lda, 1.0, C, lda, GPAR_GEMM_C_LOWER, 0, 0);  // the LA slice
        hipEventRecord(e0);
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(256), dim3(256), PNL_LDS_BYTES, 0, p);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms, ms2, ms3; hipEventElapsedTime(&ms, e0, e1); hipEventElapsedTime(&ms2, t0, t1); hipEventElapsedTime(&ms3, t0, e1);
        printf("co-run: panel kernel %.1f us (ends %.1f us after the SYRK started), SYRK %.1f us\n", ms * 1e3, ms3 * 1e3, ms2 * 1e3);
        std::vector<long long> w(64 + 256 * 9); hipMemcpy(w.data(), st, 8 * w.size(), hipMemcpyDeviceToHost);
        long long first = w[64]; for (int g = 0; g < 256; ++g) first = std::min(first, w[64 + g]);
        printf("workgroup arrival (us after the first): ");
        for (int g = 0; g < 256; g += 16) printf("%d:%.0f ", g, (w[64 + g] - first) * 0.01);
        std::vector<long long> arr(w.begin() + 64, w.begin() + 64 + 256); std::sort(arr.begin(), arr.end());
        printf("\n  arrival percentiles: p50 %.0f p90 %.0f max %.0f us\n", (arr[128] - first) * 0.01, (arr[230] - first) * 0.01, (arr[255] - first) * 0.01);
        for (int s2 = 0; s2 < 8; ++s2) {
            long long mx = 0, mn = 1LL << 62; for (int g = 0; g < 256; ++g) { mx = std::max(mx, w[64 + 256 * (1 + s2) + g]); mn = std::min(mn, w[64 + 256 * (1 + s2) + g]); }
            printf("  step %d finished: first workgroup %.0f us, last workgroup %.0f us (after the first arrival)\n", s2, (mn - first) * 0.01, (mx - first) * 0.01);
        }
        printf("chain under co-run: diag_load diag_compute diag_store+publish | wait strip update [cycles]\n");
        for (int k = 0; k < 8; ++k)
            printf("%d: %6lld %6lld %6lld | %6lld %6lld %6lld\n", k, w[k*8+1]-w[k*8+0], w[k*8+2]-w[k*8+1], w[k*8+3]-w[k*8+2],
                   w[k*8+5]-w[k*8+4], w[k*8+6]-w[k*8+5], w[k*8+7]-w[k*8+6]);
    }
    return 0;
}
