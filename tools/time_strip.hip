// Dev aid: phase timing of the strip kernel with s_memtime stamps.
#include "../gpar_amd/csrc/potrf.h"
#include <cstdio>
#include <vector>
using namespace gpar;

template <bool FWD>
__global__ __launch_bounds__(64) void strip_timed(const double* __restrict__ Ld, int ldl, int cb,
                                                  double* __restrict__ B, int ldb, int nrows, long long* stamps) {
    __shared__ __attribute__((aligned(16))) double Cs[64 * PAN_LD];
    __shared__ __attribute__((aligned(16))) double Xs[64 * PAN_LD];
    __shared__ double rinvs[64];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    long long t0 = __builtin_readcyclecounter();
    {
        double v[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int rc = r < cb ? r : cb - 1;
            v[r] = Ld[(size_t)rc * ldl + (lane < rc ? lane : rc)];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = (r < cb && lane <= r) ? v[r] : ((r == lane) ? 1.0 : 0.0);
#pragma unroll
        for (int r = 0; r < 64; ++r) Cs[r * PAN_LD + lane] = v[r];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int rr = (row0 + r < nrows) ? row0 + r : nrows - 1;
            v[r] = B[(size_t)rr * ldb + (lane < cb ? lane : cb - 1)];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) Xs[r * PAN_LD + lane] = (row0 + r < nrows && lane < cb) ? v[r] : 0.0;
    }
    __syncthreads();
    rinvs[lane] = 1.0 / Cs[lane * PAN_LD + lane];
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    for (int jb = 0; jb < 8; ++jb) {
        double acc[8];
        {
            const pan_d2* src = reinterpret_cast<const pan_d2*>(&Xs[lane * PAN_LD + 8 * jb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const pan_d2 t = src[q]; acc[2 * q] = t[0]; acc[2 * q + 1] = t[1]; }
        }
        for (int kb = 0; kb < jb; ++kb) {
            double xk[8];
            const pan_d2* xs = reinterpret_cast<const pan_d2*>(&Xs[lane * PAN_LD + 8 * kb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const pan_d2 t = xs[q]; xk[2 * q] = t[0]; xk[2 * q + 1] = t[1]; }
            pan_d2 c[8][4];   // issue all 32 broadcast reads, then k-outer FMAs (see potrf_diag64_kernel)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const pan_d2* cs = reinterpret_cast<const pan_d2*>(&Cs[(8 * jb + j) * PAN_LD + 8 * kb]);
#pragma unroll
                for (int q = 0; q < 4; ++q) c[j][q] = cs[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fma(-xk[2 * q], c[j][q][0], acc[j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fma(-xk[2 * q + 1], c[j][q][1], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double* crow = &Cs[(8 * jb + j) * PAN_LD + 8 * jb];
            double sacc = acc[j];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc = fma(-acc[k], crow[k], sacc);
            acc[j] = sacc * rinvs[8 * jb + j];
        }
        {
            pan_d2* dst = reinterpret_cast<pan_d2*>(&Xs[lane * PAN_LD + 8 * jb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = pan_d2{acc[2 * q], acc[2 * q + 1]};
        }
    }
    __syncthreads();
    long long t2 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (row0 + r < nrows && lane < cb) B[(size_t)(row0 + r) * ldb + lane] = Xs[r * PAN_LD + lane];
    long long t3 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; }
}

__global__ void empty_kernel(long long* s) { if (threadIdx.x == 0 && s) s[3] = 1; }

int main() {
    const int n = 16384, ld = 16384;
    double *L, *B; long long* st;
    hipMalloc(&L, sizeof(double) * 64 * ld); hipMalloc(&B, sizeof(double) * (size_t)n * ld / 16); hipMalloc(&st, 64);
    std::vector<double> h(64 * (size_t)ld, 0.01);
    for (int i = 0; i < 64; ++i) h[i * (size_t)ld + i] = 2.0;
    hipMemcpy(L, h.data(), sizeof(double) * 64 * ld, hipMemcpyHostToDevice);
    hipMemset(B, 0, sizeof(double) * (size_t)n * ld / 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rows : {64, 1024, 16384 - 64}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((strip_timed<true>), dim3((rows + 63) / 64), dim3(64), 0, 0, L, ld, 64, B, 1024, rows, st);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long s[4]; hipMemcpy(s, st, 32, hipMemcpyDeviceToHost);
            printf("rows=%5d: %.1f us  | cycles load %lld main %lld store %lld\n", rows, ms * 1e3, s[0], s[1], s[2]);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel: %.1f us\n", ms * 1e3);
    }
    return 0;
}
