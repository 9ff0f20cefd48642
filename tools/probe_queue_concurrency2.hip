// Dev aid: which of several created streams runs concurrently with the NULL stream?
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
    for (int flags = 1; flags < 2; ++flags) {
        std::vector<hipStream_t> ss(6);
        for (auto& s : ss) hipStreamCreateWithPriority(&s, flags ? hipStreamNonBlocking : hipStreamDefault, (&s - &ss[0]) < 3 ? lo : 0);
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < 6; ++i) {
                hipEvent_t a0, a1, b1, go; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b1); hipEventCreate(&go);
                hipDeviceSynchronize();
                // mimic potrf: null stream records an event, side waits on it, side launches long kernel, null launches small kernel
                hipEventRecord(go, 0);
                hipStreamWaitEvent(ss[i], go, 0);
                hipEventRecord(a0, ss[i]);
                hipLaunchKernelGGL(busy, dim3(512 * 16), dim3(256), 73728, ss[i], 240000LL, sink);
                hipEventRecord(a1, ss[i]);
                hipLaunchKernelGGL(busy, dim3(1), dim3(64), 1024, 0, 720000LL, sink);   // ~300 us delay: A has filled the chip by now
                hipLaunchKernelGGL(busy, dim3(256), dim3(256), 76 * 1024, 0, 48000LL, sink);
                hipEventRecord(b1, 0);
                hipDeviceSynchronize();
                float ta, tb; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, a0, b1);
                printf("flags=%s side=stream#%d rep %d: A took %.3f ms, B (null stream) finished %.3f ms after A started\n",
                       flags ? "nonblocking" : "default", i, rep, ta, tb);
            }
        for (auto& s : ss) hipStreamDestroy(s);
    }
    return 0;
}
