// Issue cost of the fp64 vector instructions the Gram kernels are made of (gfx950): cycles per wave-instruction with four
// waves per SIMD issuing independent chains.  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_f64.hip -o tools/ubench_valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 2048
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int WHICH>
__global__ __launch_bounds__(256) void k(double* out, double seed, long long* cycles) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + 0.001 * (threadIdx.x + i);
    int e[8];
    for (int i = 0; i < 8; ++i) e[i] = (int)threadIdx.x & 3;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; ++it) {
#define OP_FMA(i) x[i] = __builtin_fma(x[i], 1.0000001, 1e-9);
#define OP_ADD(i) x[i] = x[i] + 1e-9;
#define OP_MUL(i) x[i] = x[i] * 1.0000001;
#define OP_RNDNE(i) x[i] = __builtin_rint(x[i]) + 0.25;
#define OP_CVT(i) e[i] = (int)x[i]; x[i] += 1e-9;
#define OP_LDEXP(i) x[i] = __builtin_amdgcn_ldexp(x[i], e[i]);
#define OP_FREXPM(i) x[i] = __builtin_amdgcn_frexp_mant(x[i]) + 1.0;
#define OP_FREXPE(i) e[i] = __builtin_amdgcn_frexp_exp(x[i]); x[i] += 1e-9;
#define OP_RCP(i) x[i] = __builtin_amdgcn_rcp(x[i]) + 1.5;
#define OP_CVTF(i) x[i] = (double)e[i] + x[i];
#define OP_IADD(i) e[i] = e[i] * 3 + 1;
        if (WHICH == 0) { BODY8(OP_FMA) }
        if (WHICH == 1) { BODY8(OP_ADD) }
        if (WHICH == 2) { BODY8(OP_MUL) }
        if (WHICH == 3) { BODY8(OP_RNDNE) }
        if (WHICH == 4) { BODY8(OP_CVT) }
        if (WHICH == 5) { BODY8(OP_LDEXP) }
        if (WHICH == 6) { BODY8(OP_FREXPM) }
        if (WHICH == 7) { BODY8(OP_FREXPE) }
        if (WHICH == 8) { BODY8(OP_RCP) }
        if (WHICH == 9) { BODY8(OP_CVTF) }
        if (WHICH == 10) { BODY8(OP_IADD) }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
template <int W> void run(const char* name, int extra, double* out, long long* cyc) {
    // 256 CUs x 4 workgroups of 256 threads: 4 waves per SIMD
    hipLaunchKernelGGL(k<W>, dim3(1024), dim3(256), 0, 0, out, 1.0, cyc);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k<W>, dim3(1024), dim3(256), 0, 0, out, 1.0, cyc); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per SIMD: 4 waves x 8 x N_ITER instruction groups
    const double groups = 4.0 * 8 * N_ITER;
    printf("%-34s %8.3f ms  wave-0 cycles %lld  -> %.2f cycles per wave-instruction group (%d extra plain op(s) in the group)\n", name, ms, c,
           ms * 1e-3 * 2.4e9 / groups, extra);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 8);
    run<0>("v_fma_f64", 0, out, cyc); run<1>("v_add_f64", 0, out, cyc); run<2>("v_mul_f64", 0, out, cyc);
    run<3>("v_rndne_f64 + v_add_f64", 1, out, cyc); run<4>("v_cvt_i32_f64 + v_add_f64", 1, out, cyc); run<5>("v_ldexp_f64", 0, out, cyc);
    run<6>("v_frexp_mant_f64 + v_add_f64", 1, out, cyc); run<7>("v_frexp_exp_i32_f64 + v_add_f64", 1, out, cyc);
    run<8>("v_rcp_f64 + v_add_f64", 1, out, cyc); run<9>("v_cvt_f64_i32 + v_add_f64", 1, out, cyc); run<10>("v_mad_u32 (32-bit int)", 0, out, cyc);
    return 0;
}
