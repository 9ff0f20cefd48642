// Probe: lane layout of v_mfma_f64_4x4x4_4b_f64 (A/B/D maps) and CBSZ/ABID broadcast on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void probe(int* table) {
    int lane = threadIdx.x;
    for (int l0 = 0; l0 < 64; ++l0)
        for (int l1 = 0; l1 < 64; ++l1) {
            double a = (lane == l0) ? 1.0 : 0.0;
            double b = (lane == l1) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
            // each (l0,l1) pair can light several output lanes (with broadcast); record a bitmask
            unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) { table[(l0 * 64 + l1) * 2] = (int)(m & 0xffffffffu); table[(l0 * 64 + l1) * 2 + 1] = (int)(m >> 32); }
        }
}

template <int CBSZ, int ABID>
void run(const char* name) {
    int* d; hipMalloc(&d, 64 * 64 * 2 * sizeof(int));
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    printf("== %s: for each output lane, contributing (A-lane,B-lane) pairs\n", name);
    for (int out = 0; out < 64; ++out) {
        printf("D[%2d] <-", out);
        for (int l0 = 0; l0 < 64; ++l0)
            for (int l1 = 0; l1 < 64; ++l1) {
                unsigned long long m = (unsigned)h[(l0 * 64 + l1) * 2] | ((unsigned long long)(unsigned)h[(l0 * 64 + l1) * 2 + 1] << 32);
                if ((m >> out) & 1ull) printf(" (%d,%d)", l0, l1);
            }
        printf("\n");
    }
    hipFree(d);
}

int main() {
    run<0, 0>("cbsz=0 abid=0");
    run<2, 0>("cbsz=2 abid=0");
    run<2, 1>("cbsz=2 abid=1");
    run<2, 3>("cbsz=2 abid=3");
    run<1, 0>("cbsz=1 abid=0");
    run<1, 1>("cbsz=1 abid=1");
    return 0;
}
