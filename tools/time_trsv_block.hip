// Dev aid: the single-workgroup TRSV block kernel alone on one 512 x 512 block, with 100 MHz stamps per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPAR_TRSV_STAMPS tools/time_trsv_block.hip -o tools/time_trsv_block
#include "../gpar_amd/csrc/potrf.h"
#include <cstdio>
#include <vector>
using namespace gpar;
int main() {
    const int n = 2048, ld = 2048;
    std::vector<double> L((size_t)n * ld, 0.0), b(n);
    srand(1);
    for (int i = 0; i < n; ++i) { for (int j = 0; j < i; ++j) L[(size_t)i * ld + j] = 0.01 * (rand() / (double)RAND_MAX); L[(size_t)i * ld + i] = 1.0; b[i] = rand() / (double)RAND_MAX; }
    double *dL, *db;
    hipMalloc(&dL, L.size() * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        hipLaunchKernelGGL(trsv_block_kernel, dim3(1), dim3(256), 0, 0, dL, ld, db, 1536, 2048);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("block kernel: %.1f us\n", ms * 1e3);
    }
#ifdef GPAR_TRSV_STAMPS
    long long st[4][64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_trsv_stamps), sizeof(st));
    for (int w = 0; w < 4; ++w) {
        printf("wave %d:", w);
        for (int k = 0; k < 64; ++k) if (st[w][k]) printf(" %d:%.2f", k, (st[w][k] - st[0][0]) / 100.0);
        printf("\n");
    }
#endif
    return 0;
}
