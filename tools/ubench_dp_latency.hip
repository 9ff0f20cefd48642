// Dev aid: dependent-issue latencies (cycles, s_memtime) of the fp64 vector instructions on the pivot chain, one wave alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_dp_latency.hip -o tools/ubench_dp_latency && tools/ubench_dp_latency
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double rl(double v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <int MODE>
__global__ void chain(double* out, long long* st, double a, double b) {
    double x = out[threadIdx.x], y = x + 1.0, z = x + 2.0, u = x + 3.0;
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(u));
    long long c0, c1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
#pragma unroll
    for (int i = 0; i < 128; ++i) {
        if (MODE == 0) x = fma(x, a, b);                                   // dependent fma
        if (MODE == 1) x = __builtin_amdgcn_rsq(x);                        // dependent rsq
        if (MODE == 2) x = rl(x, i & 63) + a;                              // readlane -> add -> readlane
        if (MODE == 3) { x = fma(x, a, b); y = fma(y, a, b); }             // two independent chains
        if (MODE == 4) { x = fma(x, a, b); y = fma(y, a, b); z = fma(z, a, b); u = fma(u, a, b); }
        if (MODE == 5) x = x * a;                                          // dependent mul
        if (MODE == 6) x = fma(__builtin_amdgcn_rsq(x), a, b);             // rsq -> fma -> rsq
        if (MODE == 7) x = __builtin_amdgcn_rcp(x);
    }
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(u));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    out[threadIdx.x] = x + y + z + u;
    if (threadIdx.x == 0) st[MODE] = c1 - c0;
}

int main() {
    double* o;
    long long* st;
    hipMalloc(&o, 8 * 64);
    hipMalloc(&st, 8 * 16);
    hipMemset(o, 0, 8 * 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<3>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<4>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<5>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<6>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipLaunchKernelGGL(chain<7>, dim3(1), dim3(64), 0, 0, o, st, 0.999, 0.5);
        hipDeviceSynchronize();
    }
    long long s[16];
    hipMemcpy(s, st, 8 * 16, hipMemcpyDeviceToHost);
    const char* names[] = {"dependent v_fma_f64", "dependent v_rsq_f64", "readlane x2 -> v_add_f64 (per round)", "2 independent fma chains (per pair)",
                           "4 independent fma chains (per 4)", "dependent v_mul_f64", "v_rsq_f64 -> v_fma_f64 (per pair)", "dependent v_rcp_f64"};
    for (int m = 0; m < 8; ++m) printf("%-40s %6.1f cycles\n", names[m], s[m] / 128.0);
    return 0;
}
