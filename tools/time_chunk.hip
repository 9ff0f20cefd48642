// Dev aid: matrix-core issue rate of the panel kernel's K = 64 chunk (p2_chunk) for a lone workgroup, and variants of its
// operand-fragment schedule.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/time_chunk.hip -o tools/time_chunk
#include "../gpar_amd/csrc/panel2.h"
#include <cstdio>
using namespace gpar;

// variant 1: the compiler's own order
__device__ __forceinline__ void chunk_v1(const double* __restrict__ Ls, const double* __restrict__ Xs, pan_d4 (&acc)[4], int w, int l15, int lk) {
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        double a[4], b;
        p2_frag(Ls, Xs, w, l15, 4 * k4 + lk, a, b);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b, acc[mi], 0, 0, 0);
    }
}

// variant 2: fragments two steps ahead (three register sets)
__device__ __forceinline__ void chunk_v2(const double* __restrict__ Ls, const double* __restrict__ Xs, pan_d4 (&acc)[4], int w, int l15, int lk) {
    double a[3][4], b[3];
    p2_frag(Ls, Xs, w, l15, lk, a[0], b[0]);
    p2_frag(Ls, Xs, w, l15, 4 + lk, a[1], b[1]);
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        if (k4 + 2 < 16) p2_frag(Ls, Xs, w, l15, 4 * (k4 + 2) + lk, a[(k4 + 2) % 3], b[(k4 + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4 % 3][mi], b[k4 % 3], acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// variant 3: all 80 fragments of the chunk in registers first (160 VGPRs), then 64 products back to back
__device__ __forceinline__ void chunk_v3(const double* __restrict__ Ls, const double* __restrict__ Xs, pan_d4 (&acc)[4], int w, int l15, int lk) {
    double a[16][4], b[16];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) p2_frag(Ls, Xs, w, l15, 4 * k4 + lk, a[k4], b[k4]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4][mi], b[k4], acc[mi], 0, 0, 0);
}

// variant 4: operands in registers, no LDS at all (the ceiling for one wave per SIMD)
__device__ __forceinline__ void chunk_v4(pan_d4 (&acc)[4], double a0, double b0) {
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 + mi, b0, acc[mi], 0, 0, 0);
}

// the chunk as the kernel runs it: tiles arrive from global memory through registers, two barriers per chunk
template <int STEP>
__global__ __launch_bounds__(256, 2) void loopk(const double* __restrict__ G, double* out, long long* st, int reps, int span) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    double* Cs = psm; double* Xs = psm + PNL_TILE;
    for (int e = t; e < 2 * PNL_TILE; e += 256) psm[e] = 1e-3 * ((e * 7) % 13);
    __syncthreads();
    pan_d4 acc[4];
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0, 0, 0, 0};
    pan_d2 la[8], xa[8];
    const double* base = G + (size_t)blockIdx.x * 64 * 4096;
    p2_gload(base, 4096, 64, 0, 0, t, la);
    p2_gload(base, 4096, 64, 0, 64, t, xa);
    const long long c0 = (long long)__builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (STEP >= 1) __syncthreads();
        if (STEP >= 2) { p2_sstore(Cs, t, la); p2_sstore(Xs, t, xa); }
        if (STEP >= 1) __syncthreads();
        if (STEP >= 3) { p2_gload(base, 4096, 64, 0, 64 * ((2 * r + 2) & span), t, la); p2_gload(base, 4096, 64, 0, 64 * ((2 * r + 3) & span), t, xa); }
        __builtin_amdgcn_sched_barrier(0);
        p2_chunk(Cs, Xs, acc, w, l15, lk);
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    double s = 0;
    for (int mi = 0; mi < 4; ++mi) s += acc[mi][0] + acc[mi][1] + acc[mi][2] + acc[mi][3];
    if (STEP < 3) s += la[0][0] + xa[0][0];
    out[blockIdx.x * 256 + t] = s;
    if (t == 0 && blockIdx.x == 0) st[8 + STEP] = c1 - c0;
}

// LDS-direct variant: K = 32 half-chunks, operands global -> LDS without passing through registers (global_load_lds_dwordx4),
// double-buffered in four 16 KB half-tiles (unpadded rows of 256 bytes, 16-byte chunks XOR-swizzled by row), one barrier each.
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void half_load(const double* __restrict__ tile, int ld, int kbase, double* __restrict__ buf, int t) {
    const int lane = t & 63, w = t >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * (4 * q + w) + (lane >> 4), p = lane & 15, j = p ^ ((r & 7) << 1);
        __builtin_amdgcn_global_load_lds(tile + (size_t)r * ld + kbase + 2 * j, (lds_ptr)(buf + 128 * (4 * q + w)), 16, 0, 0);
    }
}
__device__ __forceinline__ double half_at(const double* __restrict__ buf, int r, int k) {
    return buf[r * 32 + 2 * (((k >> 1) ^ ((r & 7) << 1))) + (k & 1)];
}
__device__ __forceinline__ void half_chunk(const double* __restrict__ Lh, const double* __restrict__ Xh, pan_d4 (&acc)[4], int w, int l15, int lk) {
    double a0[4], a1[4], b0, b1;
    b0 = half_at(Xh, 16 * w + l15, lk);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a0[mi] = half_at(Lh, 16 * mi + l15, lk);
#pragma unroll
    for (int k4 = 0; k4 < 8; k4 += 2) {
        b1 = half_at(Xh, 16 * w + l15, 4 * (k4 + 1) + lk);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a1[mi] = half_at(Lh, 16 * mi + l15, 4 * (k4 + 1) + lk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[mi], b0, acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 2 < 8) {
            b0 = half_at(Xh, 16 * w + l15, 4 * (k4 + 2) + lk);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a0[mi] = half_at(Lh, 16 * mi + l15, 4 * (k4 + 2) + lk);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[mi], b1, acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__global__ __launch_bounds__(256, 2) void ldsk(const double* __restrict__ G, double* out, long long* st, int reps, int span) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    pan_d4 acc[4];
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0, 0, 0, 0};
    const double* base = G + (size_t)blockIdx.x * 64 * 4096;
    // buffers: [L half 0 | X half 0 | L half 1 | X half 1], 2048 doubles each
    half_load(base, 4096, 0, psm, t);
    half_load(base, 4096, 64, psm + 2048, t);
    const long long c0 = (long long)__builtin_readcyclecounter();
    for (int h = 0; h < 2 * reps; ++h) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double* nb = psm + 4096 * ((h + 1) & 1);
        half_load(base, 4096, (32 * (h + 1)) & (64 * span + 63), nb, t);
        half_load(base, 4096, (32 * (h + 1) + 2048) & (64 * span + 63) , nb + 2048, t);
        __builtin_amdgcn_sched_barrier(0);
        const double* cb = psm + 4096 * (h & 1);
        half_chunk(cb, cb + 2048, acc, w, l15, lk);
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    double s = 0;
    for (int mi = 0; mi < 4; ++mi) s += acc[mi][0] + acc[mi][1] + acc[mi][2] + acc[mi][3];
    out[blockIdx.x * 256 + t] = s;
    if (t == 0 && blockIdx.x == 0) st[12] = c1 - c0;
}

template <int V>
__global__ __launch_bounds__(256, 2) void k(double* out, long long* st, int reps) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    for (int e = t; e < 2 * PNL_TILE; e += 256) psm[e] = 1e-3 * ((e * 7) % 13);
    __syncthreads();
    pan_d4 acc[4];
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0, 0, 0, 0};
    const long long c0 = (long long)__builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (V == 0) p2_chunk(psm, psm + PNL_TILE, acc, w, l15, lk);
        if (V == 1) chunk_v1(psm, psm + PNL_TILE, acc, w, l15, lk);
        if (V == 2) chunk_v2(psm, psm + PNL_TILE, acc, w, l15, lk);
        if (V == 3) chunk_v3(psm, psm + PNL_TILE, acc, w, l15, lk);
        if (V == 4) chunk_v4(acc, psm[lane], psm[lane + 64]);
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    double s = 0;
    for (int mi = 0; mi < 4; ++mi) s += acc[mi][0] + acc[mi][1] + acc[mi][2] + acc[mi][3];
    out[blockIdx.x * 256 + t] = s;
    if (t == 0 && blockIdx.x == 0) st[V] = c1 - c0;
}

int main() {
    double* o; long long* st;
    hipMalloc(&o, 8 * 256 * 512); hipMalloc(&st, 8 * 16); hipMemset(st, 0, 8 * 16);
    const int reps = 200;
    const char* names[] = {"p2_chunk (fragments one step ahead, scheduling barriers)", "compiler's order", "fragments two steps ahead", "all fragments first, then 64 products", "operands in registers (no LDS)"};
    for (int grid : {1, 256, 512}) {
        printf("grid = %d workgroups (256 CUs)\n", grid);
#define RUN(V) hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES); \
        hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), P2_LDS_BYTES, 0, o, st, reps); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), P2_LDS_BYTES, 0, o, st, reps);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
        hipDeviceSynchronize();
        long long s[16]; hipMemcpy(s, st, 8 * 16, hipMemcpyDeviceToHost);
        for (int v = 0; v < 5; ++v) printf("   %-58s %6.1f cycles per v_mfma_f64_16x16x4 (64 per chunk per wave)\n", names[v], s[v] / (double)(reps * 64));
    }
    double* G;
    hipMalloc(&G, 8ull * 512 * 64 * 4096);
    hipMemset(G, 0, 8ull * 512 * 64 * 4096);
    const char* ln[] = {"chunk alone", "+ two barriers", "+ 16 ds_write_b128 of the next tiles", "+ the next tiles' global loads in flight"};
    for (int span : {63, 3})
    for (int grid : {1, 256}) {
        printf("the chunk loop as the kernel runs it, grid = %d, operand tiles from %s\n", grid, span == 63 ? "HBM (every tile read once)" : "L2 (four tiles re-read)");
#define RUNL(V) hipFuncSetAttribute(reinterpret_cast<const void*>(&loopk<V>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES); \
        hipLaunchKernelGGL(loopk<V>, dim3(grid), dim3(256), P2_LDS_BYTES, 0, G, o, st, reps, span); hipLaunchKernelGGL(loopk<V>, dim3(grid), dim3(256), P2_LDS_BYTES, 0, G, o, st, reps, span);
        RUNL(0) RUNL(1) RUNL(2) RUNL(3)
        hipDeviceSynchronize();
        long long s2[16]; hipMemcpy(s2, st, 8 * 16, hipMemcpyDeviceToHost);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ldsk), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
        hipLaunchKernelGGL(ldsk, dim3(grid), dim3(256), P2_LDS_BYTES, 0, G, o, st, reps, span); hipLaunchKernelGGL(ldsk, dim3(grid), dim3(256), P2_LDS_BYTES, 0, G, o, st, reps, span);
        hipDeviceSynchronize();
        hipMemcpy(s2, st, 8 * 16, hipMemcpyDeviceToHost);
        for (int v = 0; v < 4; ++v) printf("   %-45s %7.0f cycles per chunk (64 products = 4096 cycles of matrix-core time)\n", ln[v], s2[8 + v] / (double)reps);
        printf("   %-45s %7.0f cycles per chunk\n", "LDS-direct loads, K = 32 halves, double-buffered", s2[12] / (double)reps);
    }
    return 0;
}
