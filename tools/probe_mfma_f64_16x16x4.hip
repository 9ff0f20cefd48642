// Probe: lane / register layout of v_mfma_f64_16x16x4_f64 on gfx950 (A, B and the four D registers per lane), and a
// clean issue-rate measurement with accumulators that stay in registers (the first micro-benchmark of this
// instruction, tools/ubench_mfma_f64.hip, let the compiler shuttle the accumulators between AGPRs and VGPRs every
// iteration and so measured the shuttle, not the instruction: rocBLAS reaches 76.8 TFLOP/s with it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(int* table) {   // table[(la*64+lb)*4 + v] = ballot of lanes whose D register v is non-zero
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
            d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            for (int v = 0; v < 4; ++v) {
                unsigned long long m = __ballot(d[v] != 0.0);
                if (lane == 0) { table[((la * 64 + lb) * 4 + v) * 2] = (int)(m & 0xffffffffu); table[((la * 64 + lb) * 4 + v) * 2 + 1] = (int)(m >> 32); }
            }
        }
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    int* d; hipMalloc(&d, 64 * 64 * 4 * 2 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64 * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    auto mask = [&](int la, int lb, int v) { return (unsigned long long)(unsigned)h[((la * 64 + lb) * 4 + v) * 2] | ((unsigned long long)(unsigned)h[((la * 64 + lb) * 4 + v) * 2 + 1] << 32); };
    // summarise: for each (A lane, B lane) pair that produces output, which (lane, v) receives it
    printf("(A lane la, B lane lb) -> D (lane, reg): only pairs with la/16 == lb/16 should contribute (same k)\n");
    for (int la : {0, 1, 5, 15, 16, 17, 33, 63})
        for (int lb : {0, 1, 7, 15, 16, 18, 35, 63}) {
            printf("  A%2d B%2d ->", la, lb);
            for (int v = 0; v < 4; ++v) { unsigned long long m = mask(la, lb, v); for (int o = 0; o < 64; ++o) if ((m >> o) & 1ull) printf(" (lane %d, reg %d)", o, v); }
            printf("\n");
        }
    // full map check against the hypothesis: A lane = (i = l%16, k = l/16), B lane = (j = l%16, k = l/16), D[i][j] at lane (j + 16*(i/4))?, reg i%4
    int bad1 = 0, bad2 = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int v = 0; v < 4; ++v) {
        unsigned long long m = mask(la, lb, v);
        const int i = la % 16, j = lb % 16; const bool same_k = la / 16 == lb / 16;
        unsigned long long h1 = (same_k && (i % 4) == v) ? (1ull << (j + 16 * (i / 4))) : 0ull;   // rows split as i = 4*(lane/16) + reg
        unsigned long long h2 = (same_k && (i / 4) == v) ? (1ull << (j + 16 * (i % 4))) : 0ull;   // rows split as i = (lane/16) + 4*reg
        bad1 += m != h1; bad2 += m != h2;
    }
    printf("hypothesis 1 (D[i][j]: lane = j + 16*(i/4), reg = i%%4): %d mismatches\n", bad1);
    printf("hypothesis 2 (D[i][j]: lane = j + 16*(i%%4), reg = i/4): %d mismatches\n", bad2);
    double* out; hipMalloc(&out, 1024 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int nacc, int waves) {
        const int iters = 2000; dim3 grid(256 * waves), block(256);
        hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.0, 1e-30); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.0, 1e-30); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2048.0 * 8 * nacc * iters * 4.0 * 256 * waves;
        printf("v_mfma_f64_16x16x4 (inline asm, accumulators pinned) nacc=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", nacc, waves, ms, flops / ms * 1e-9);
    };
    run(rate<4>, 4, 1); run(rate<8>, 8, 1); run(rate<4>, 4, 2); run(rate<8>, 8, 2);
    return 0;
}
