// Dev aid: cycle stamps along the critical chain of the fused panel kernel (same-XCD stamps only are comparable;
// durations within one workgroup are exact).
#include "../gpar_amd/csrc/panel.h"
#include <cstdio>
#include <vector>
using namespace gpar;
int main() {
    const int N = 16384, lda = 16384;
    double* A; long long* st;
    hipMalloc(&A, sizeof(double) * (size_t)N * lda); hipMalloc(&st, 8 * 64);
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
        hipMemsetAsync(st, 0, 8 * 64, 0);
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
    return 0;
}
