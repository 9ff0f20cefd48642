// Dev aid: can a kernel launched on stream B get workgroups onto the chip while a long multi-round kernel from stream A
// is still dispatching?  Prints, per stream configuration, when B finished relative to A's start and A's duration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void busy(long long cycles, int* sink) {
    extern __shared__ double lds[];
    const long long t0 = __builtin_readcyclecounter();
    double x = threadIdx.x;
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) { x = x * 1.0000001 + 1e-9; __builtin_amdgcn_s_sleep(8); }
    lds[threadIdx.x] = x;
    if (x == 12345.678) sink[0] = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)busy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    struct Cfg { const char* name; int pa, pb; bool a_null, b_null; };
    std::vector<Cfg> cfgs = {{"A=low  B=normal", lo, 0, false, false}, {"A=low  B=high", lo, hi, false, false}, {"A=normal B=high", 0, hi, false, false},
                             {"A=low  B=null", lo, 0, false, true}, {"A=null B=high", 0, hi, true, false}, {"A=low B=low", lo, lo, false, false}};
    for (auto& c : cfgs) {
        for (int extra = 0; extra < 3; ++extra) {   // a few extra dummy streams in between to land on other HW queues
            hipStream_t sa = nullptr, sb = nullptr; std::vector<hipStream_t> dummies;
            if (!c.a_null) hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, c.pa);
            for (int i = 0; i < extra; ++i) { hipStream_t d; hipStreamCreateWithPriority(&d, hipStreamNonBlocking, 0); dummies.push_back(d); }
            if (!c.b_null) hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, c.pb);
            hipEvent_t a0, a1, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b1);
            hipDeviceSynchronize();
            hipEventRecord(a0, sa);
            hipLaunchKernelGGL(busy, dim3(512 * 16), dim3(256), 73728, sa, 240000LL, sink);   // ~100 us per workgroup, 16 rounds
            hipEventRecord(a1, sa);
            hipLaunchKernelGGL(busy, dim3(256), dim3(256), 76 * 1024, sb, 48000LL, sink);      // ~20 us per workgroup
            hipEventRecord(b1, sb);
            hipDeviceSynchronize();
            float ta, tb; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, a0, b1);
            printf("%-16s extra=%d: A took %.3f ms, B finished %.3f ms after A started\n", c.name, extra, ta, tb);
            if (sa) hipStreamDestroy(sa); if (sb) hipStreamDestroy(sb); for (auto d : dummies) hipStreamDestroy(d);
        }
    }
    return 0;
}
