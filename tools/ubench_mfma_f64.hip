// µbench: issue rate / dependent latency of v_mfma_f64_16x16x4_f64 on gfx950, and
// fp64 VALU FMA rate, to establish the fp64 matrix roofline denominator.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o ubench_mfma_f64 ubench_mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x4 ; MODE 1: 4x4x4_4b ; MODE 2: mixed mfma 16x16x4 + VALU fma (NV fmas per mfma)
template <int NACC, int MODE, int NV>
__global__ __launch_bounds__(256) void k_mfma(double* out, long long* cyc, int iters, double a0, double b0) {
    d4 acc[NACC];
    double vacc[NV > 0 ? NV : 1];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) vacc[i] = threadIdx.x * 1e-9 + i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 1) {
                double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
                acc[i][0] = r;
            } else {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            if (MODE == 2) {
#pragma unroll
                for (int v = 0; v < NV; ++v) vacc[v] = __builtin_fma(vacc[v], a0, b0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += vacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, long long* cyc, int iters, double a0, double b0) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    double a = a0, b = b0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

long long* dcyc;

template <int NACC, int MODE, int NV>
void run_mfma(double* d, int blocks, int threads, int iters) {
    float ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<NACC, MODE, NV>), dim3(blocks), dim3(threads), 0, 0, d, dcyc, iters, 1.0, 1e-3); });
    long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    double waves = double(blocks) * threads / 64.0;
    double nmfma = waves * iters * NACC;
    double fl_per = (MODE == 1) ? 512.0 : 2048.0;
    double flops = nmfma * fl_per + (MODE == 2 ? waves * 64.0 * iters * NACC * NV * 2.0 : 0.0);
    double wps = waves / 1024.0; if (wps < 1) wps = 1;
    printf("%s nacc=%d nv=%d waves/SIMD=%.0f: %.3f ms  %.2f TFLOP/s  counter %.1f ticks/MFMA/wave -> %.1f ticks/MFMA/SIMD  (counter rate %.1f MHz)\n",
           MODE == 1 ? "mfma_f64_4x4x4_4b" : (MODE == 2 ? "mfma16+valu" : "mfma_f64_16x16x4"), NACC, NV, wps, ms,
           flops / ms * 1e-9, double(c) / (iters * NACC), double(c) / (iters * NACC) / wps, double(c) / ms * 1e-3);
}

template <int NACC>
void run_fma(double* d, int blocks, int threads, int iters) {
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma<NACC>, dim3(blocks), dim3(threads), 0, 0, d, dcyc, iters, 1.0000001, 1e-9); });
    long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    double flops = double(blocks) * threads * iters * NACC * 2.0;
    double wps = double(blocks) * threads / 64.0 / 1024.0;
    printf("v_fma_f64 nacc=%d waves/SIMD=%.0f: %.3f ms  %.2f TFLOP/s  %.2f ticks/FMA/SIMD (counter rate %.1f MHz)\n", NACC, wps, ms,
           flops / ms * 1e-9, double(c) / (iters * NACC) / wps, double(c) / ms * 1e-3);
}

int main() {
    double* d; hipMalloc(&d, sizeof(double) * 4096 * 1024);
    hipMalloc(&dcyc, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    int iters = 20000;
    run_mfma<1, 0, 0>(d, 256, 256, iters);
    run_mfma<2, 0, 0>(d, 256, 256, iters);
    run_mfma<4, 0, 0>(d, 256, 256, iters);
    run_mfma<8, 0, 0>(d, 256, 256, iters);
    run_mfma<4, 0, 0>(d, 512, 256, iters);
    run_mfma<4, 0, 0>(d, 768, 256, iters);
    run_mfma<4, 0, 0>(d, 1024, 256, iters);
    run_mfma<4, 0, 0>(d, 2048, 256, iters);
    run_mfma<2, 0, 0>(d, 1024, 256, iters);
    run_mfma<1, 0, 0>(d, 2048, 256, iters);
    run_mfma<4, 0, 0>(d, 1, 64, iters);
    run_mfma<4, 1, 0>(d, 256, 256, iters);
    run_mfma<4, 1, 0>(d, 1024, 256, iters);
    run_mfma<4, 2, 4>(d, 256, 256, iters);
    run_mfma<4, 2, 8>(d, 256, 256, iters);
    run_mfma<4, 2, 16>(d, 256, 256, iters);
    run_mfma<4, 2, 8>(d, 512, 256, iters);
    run_mfma<4, 2, 16>(d, 512, 256, iters);
    run_mfma<4, 2, 16>(d, 1024, 256, iters);
    run_fma<8>(d, 256, 256, iters);
    run_fma<8>(d, 256 * 2, 256, iters);
    run_fma<8>(d, 256 * 4, 256, iters);
    run_fma<16>(d, 256 * 8, 256, iters);
    hipFree(d);
    return 0;
}
