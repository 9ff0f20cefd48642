// Dev aid: throughput of the in-launch update tile (grp_la_tile, panel2.h) with every column block already published:
// T tiles of panel q's columns, K = kq - k0 deep, one workgroup each.  Prints time per launch, us per column block and workgroup,
// and the aggregate rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/time_la_tile.hip -o tools/time_la_tile && tools/time_la_tile [K] [rows]
#include "../gpar_amd/csrc/panel2.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace gpar;

__global__ __launch_bounds__(256, 2) void la_tile_kernel(PanelArgs p, int kq, int rows, int map) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    const int x = blockIdx.x;
    int ti, tj;   // tiles below the diagonal block: 8 column blocks x rows
    if (map == 0) { ti = x % rows; tj = x / rows; }            // a column block's tiles consecutive
    else if (map == 1) { ti = x / 8; tj = x % 8; }             // a row's tiles consecutive: one per XCD (the group kernel's order)
    else { const int g = x / 64, r = x % 64; tj = r / 8; ti = 8 * g + r % 8; }   // a row's tiles on ONE XCD
    grp_la_tile(p, kq, 8 + ti, tj, psm);
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2048, rows = argc > 2 ? atoi(argv[2]) : 128, map = argc > 3 ? atoi(argv[3]) : 0;
    const int kq = K, N = kq + 64 * (8 + rows), lda = N;
    double* A;
    hipMalloc(&A, sizeof(double) * (size_t)N * lda);
    std::vector<double> h((size_t)N * lda);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    // progress words: everything published
    for (int arb = 1; arb <= N / 64; ++arb) {
        unsigned long long big = 1ull << 40, zero = 0;
        memcpy(&h[(size_t)(64 * (arb - 1) + 2) * lda + 64 * (arb - 1) + 8], &big, 8);
        memcpy(&h[(size_t)(64 * (arb - 1) + 3) * lda + 64 * (arb - 1) + 8], &zero, 8);
    }
    hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    PanelArgs p{A, N, lda, 0, 8, nullptr, nullptr, nullptr};
    hipFuncSetAttribute(reinterpret_cast<const void*>(&la_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int tiles = 8 * rows;
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(la_tile_kernel, dim3(tiles), dim3(256), P2_LDS_BYTES, 0, p, kq, rows, map);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double flops = 2.0 * 64 * 64 * (double)K * tiles;
    const double rounds = (tiles + 511) / 512;
    printf("map %d  K = %d, %d tiles: %.1f us per launch, %.2f us per column block and workgroup (%.0f rounds of 512), %.1f TFLOP/s\n", map, K, tiles, best * 1e3,
           best * 1e3 / rounds / (K / 64), rounds, flops / best * 1e-9);
    return 0;
}
