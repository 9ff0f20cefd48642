// Dev aid: shader clock during fp64 MFMA work = delta(s_memtime) / delta(s_memrealtime, 100 MHz), per workgroup of a launch that
// keeps every SIMD busy with independent v_mfma_f64_16x16x4_f64 (waves per SIMD = argv[1], default 2) for ~argv[2] ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burn(long long* out, int iters) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = threadIdx.x * 1e-3, b = 1.0 - a;
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = c1 - c0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (long long)s; }
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int nb = 256 * wps;
    long long* d; (void)hipMalloc(&d, sizeof(long long) * 3 * nb);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(burn, dim3(nb), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(3 * nb);
        (void)hipMemcpy(h.data(), d, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<double> f, us;
        for (int b = 0; b < nb; ++b) { f.push_back(1e-1 * (double)h[3 * b] / (double)h[3 * b + 1]); us.push_back(0.01 * h[3 * b + 1]); }
        std::sort(f.begin(), f.end()); std::sort(us.begin(), us.end());
        const double flops = (double)nb * 4 * iters * 8 * 2048.0;
        printf("waves/SIMD %d: shader clock GHz min %.3f median %.3f max %.3f; %.0f us per workgroup; %.1f TFLOP/s; cycles per MFMA and SIMD %.1f\n", wps, f[0],
               f[nb / 2], f[nb - 1], us[nb / 2], flops / (us[nb / 2] * 1e-6) * 1e-12, f[nb / 2] * 1e3 * us[nb / 2] / (iters * 8.0 * wps));
    }
    return 0;
}
