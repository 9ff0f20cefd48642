// Dev aid: what does the chip sustain on WRITE-ONLY streams of fp64, and how much does the shape of the stores matter?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o tools/ubench_store && tools/ubench_store
// (the Gram build writes 8 bytes per entry and reads almost nothing: this is its roofline)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

// linear: every lane writes 16 bytes, consecutive lanes consecutive addresses, `per` stores per thread strided by the grid
__global__ __launch_bounds__(256) void fill_linear(double* __restrict__ out, size_t n2, int per) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (int k = 0; k < per; ++k, i += stride)
        if (i < n2) reinterpret_cast<d2*>(out)[i] = d2{1.0 + k, 2.0};
}

// tiles of TR rows x TC columns of a row-major n x n matrix (lower-triangle tiles only when `lower`): a workgroup writes its
// tile row by row, each wave instruction covering 1024 bytes of `1024 / (8 TC)`... see body
template <int TR, int TC>
__global__ __launch_bounds__(256) void fill_tiles(double* __restrict__ out, int n, int ld, int lower) {
    const int tiles_c = n / TC;
    int bm, bn;
    if (lower) {
        // enumerate tiles (bm, bn) with bn * TC <= bm * TR + TR - 1
        const long long tile = blockIdx.x;
        // rows of tiles hold ((bm * TR + TR - 1) / TC + 1) tiles each: solve by search (cheap: dev aid)
        long long acc = 0;
        bm = 0;
        while (true) {
            const long long cnt = (long long)(bm * TR + TR - 1) / TC + 1;
            if (tile < acc + cnt) break;
            acc += cnt;
            ++bm;
        }
        bn = (int)(tile - acc);
    } else {
        bm = blockIdx.x / tiles_c;
        bn = blockIdx.x % tiles_c;
    }
    const int t = threadIdx.x;
    constexpr int PAIRS_PER_ROW = TC / 2;              // 16-byte pairs per tile row
    constexpr int ROWS_PER_PASS = 256 / PAIRS_PER_ROW;  // rows covered by one store instruction of the workgroup
    const int pc = t % PAIRS_PER_ROW, pr = t / PAIRS_PER_ROW;
#pragma unroll
    for (int r = pr; r < TR; r += ROWS_PER_PASS) {
        double* dst = out + (size_t)(bm * TR + r) * ld + bn * TC + 2 * pc;
        *reinterpret_cast<d2*>(dst) = d2{1.0 + r, 2.0};
    }
}

int main() {
    const int n = 16384, ld = 16400;
    double* A;
    hipMalloc(&A, sizeof(double) * (size_t)n * ld);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](auto launch, double bytes, const char* name) {
        launch();
        hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-58s %8.3f ms  %6.2f TB/s\n", name, best, bytes / best * 1e-9);
    };
    const double full = 8.0 * n * (double)n, low = 8.0 * n * (n + 1.0) / 2;
    time([&] { hipMemsetAsync(A, 0, (size_t)full, 0); }, full, "hipMemsetAsync, n x n doubles");
    for (int per : {1, 4, 16}) {
        const size_t n2 = (size_t)n * n / 2;
        const int grid = (int)((n2 + 256ull * per - 1) / (256ull * per));
        char name[96];
        snprintf(name, sizeof name, "linear 16 B / lane, %d stores per thread", per);
        time([&] { hipLaunchKernelGGL(fill_linear, dim3(grid), dim3(256), 0, 0, A, n2, per); }, full, name);
    }
    time([&] { hipLaunchKernelGGL((fill_tiles<64, 64>), dim3((n / 64) * (n / 64)), dim3(256), 0, 0, A, n, ld, 0); }, full, "tiles 64 x 64 (512 B per row), full matrix");
    time([&] { hipLaunchKernelGGL((fill_tiles<32, 128>), dim3((n / 32) * (n / 128)), dim3(256), 0, 0, A, n, ld, 0); }, full, "tiles 32 x 128 (1 KB per row), full matrix");
    time([&] { hipLaunchKernelGGL((fill_tiles<16, 256>), dim3((n / 16) * (n / 256)), dim3(256), 0, 0, A, n, ld, 0); }, full, "tiles 16 x 256 (2 KB per row), full matrix");
    time([&] { hipLaunchKernelGGL((fill_tiles<8, 512>), dim3((n / 8) * (n / 512)), dim3(256), 0, 0, A, n, ld, 0); }, full, "tiles 8 x 512 (4 KB per row), full matrix");
    auto count_lower = [&](int TR, int TC) {
        long long c = 0;
        for (int bm = 0; bm < n / TR; ++bm) c += (long long)(bm * TR + TR - 1) / TC + 1;
        return c;
    };
    time([&] { hipLaunchKernelGGL((fill_tiles<64, 64>), dim3((unsigned)count_lower(64, 64)), dim3(256), 0, 0, A, n, ld, 1); }, low, "tiles 64 x 64, lower triangle (algorithmic bytes)");
    time([&] { hipLaunchKernelGGL((fill_tiles<32, 128>), dim3((unsigned)count_lower(32, 128)), dim3(256), 0, 0, A, n, ld, 1); }, low, "tiles 32 x 128, lower triangle (algorithmic bytes)");
    time([&] { hipLaunchKernelGGL((fill_tiles<16, 256>), dim3((unsigned)count_lower(16, 256)), dim3(256), 0, 0, A, n, ld, 1); }, low, "tiles 16 x 256, lower triangle (algorithmic bytes)");
    return 0;
}
