// Dev aid: cycle counts of the 64 x 64 diagonal-tile factorisation alone (one workgroup, tile in LDS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/time_diag.hip -o tools/time_diag && tools/time_diag
#include "../gpar_amd/csrc/panel2.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace gpar;

template <int V>
__global__ __launch_bounds__(256, 2) void diag_kernel(const double* __restrict__ in, double* __restrict__ out, long long* st, int reps) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    const int t = threadIdx.x;
    PanelArgs p{nullptr, 64, 64, 0, 1, nullptr, nullptr, nullptr};
    long long best = 1ll << 60;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = t; e < 64 * 64; e += 256) psm[(e >> 6) * PNL_LD + (e & 63)] = in[e];
        __syncthreads();
        const long long c0 = (long long)__builtin_readcyclecounter();
        const long long w0 = (long long)__builtin_amdgcn_s_memrealtime();
        if (V == 0) pnl_diag(psm, 0, p, t);
        else p3_diag<true>(psm, 0, p, t, st + 8);
        __syncthreads();
        const long long c1 = (long long)__builtin_readcyclecounter();
        const long long w1 = (long long)__builtin_amdgcn_s_memrealtime();
        if (t == 0 && c1 - c0 < best) { best = c1 - c0; st[0] = c1 - c0; st[1] = w1 - w0; }
        __syncthreads();
    }
    for (int e = t; e < 64 * 64; e += 256) out[e] = psm[(e >> 6) * PNL_LD + (e & 63)];
}

int main() {
    std::vector<double> h(64 * 64), L(64 * 64, 0.0), o(64 * 64);
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 64; ++c) h[r * 64 + c] = (r == c) ? 70.0 + r : 1.0 / (1 + std::abs(r - c)) + 0.3 * std::cos(r * c);
    for (int j = 0; j < 64; ++j) {   // reference
        double d = h[j * 64 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 64 + k] * L[j * 64 + k];
        L[j * 64 + j] = std::sqrt(d);
        for (int i2 = j + 1; i2 < 64; ++i2) {
            double v = h[i2 * 64 + j];
            for (int k = 0; k < j; ++k) v -= L[i2 * 64 + k] * L[j * 64 + k];
            L[i2 * 64 + j] = v / L[j * 64 + j];
        }
    }
    double *din, *dout;
    long long* st;
    hipMalloc(&din, 8 * 4096);
    hipMalloc(&dout, 8 * 4096);
    hipMalloc(&st, 8 * 64);
    hipMemcpy(din, h.data(), 8 * 4096, hipMemcpyHostToDevice);
    for (int v = 0; v < 3; v += 2) {
        hipMemset(st, 0, 8 * 64);
        if (v == 0) hipLaunchKernelGGL(diag_kernel<0>, dim3(1), dim3(256), P2_LDS_BYTES, 0, din, dout, st, 20);
        else if (v == 2) hipLaunchKernelGGL(diag_kernel<2>, dim3(1), dim3(256), P2_LDS_BYTES, 0, din, dout, st, 20);
        hipDeviceSynchronize();
        long long s[64];
        hipMemcpy(s, st, 8 * 64, hipMemcpyDeviceToHost);
        hipMemcpy(o.data(), dout, 8 * 4096, hipMemcpyDeviceToHost);
        double err = 0.0;
        for (int r = 0; r < 64; ++r)
            for (int c = 0; c <= r; ++c) err = std::fmax(err, std::fabs(o[r * 64 + c] - L[r * 64 + c]) / std::fabs(L[r * 64 + r]));
        printf("%s: %lld cycles (s_memtime), %.2f us wall, max rel err %.2e\n", v == 0 ? "pnl_diag" : "p3_diag", s[0], s[1] * 0.01, err);
        if (v == 2) {
            printf("   max |err| per 8 x 8 block (rows down, columns across):\n");
            for (int br = 0; br < 8; ++br) {
                printf("     ");
                for (int bc = 0; bc <= br; ++bc) {
                    double e = 0.0;
                    for (int r = 8 * br; r < 8 * br + 8; ++r)
                        for (int c = 8 * bc; c < 8 * bc + 8 && c <= r; ++c) {
                            const double dd = std::fabs(o[r * 64 + c] - L[r * 64 + c]);
                            e = (dd == dd) ? std::fmax(e, dd) : 1e300;
                        }
                    printf(" %8.1e", e);
                }
                printf("\n");
            }
            printf("   first rows of column 0: got %g %g %g %g  want %g %g %g %g\n", o[0], o[64], o[128], o[64 * 9], L[0], L[64], L[128], L[64 * 9]);
        }
        if (v >= 1)
            for (int jb = 0; jb < 8; ++jb)
                printf("   round %d: block %lld, to next round %lld\n", jb,
                       s[8 + 3 * jb + 2] - s[8 + 3 * jb], jb < 7 ? s[8 + 3 * (jb + 1)] - s[8 + 3 * jb + 2] : 0ll);
    }
    return 0;
}
