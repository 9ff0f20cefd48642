// HBM-bound vector / matrix-vector helpers of the GP layer algebra (all deterministic: fixed reduction orders).
//
//   gemv_t      out[j] = sum_i A[i][j] v[i]        (tall A, e.g. c = B D^-1 y of the inducing-point bound: n x M)
//   rownorm2    out[i] = sum_j A[i][j]^2           (posterior marginal variances k** - |V_i|^2, VFE diagonal terms)
//   dot         out    = sum_i x[i] y[i]
// Algorithmic traffic: the matrix once (8 rows * cols bytes).  Every lane keeps 8-16 loads in flight (a rolled loop around a
// single load waits out one memory round trip per iteration).
#pragma once
#include "common.h"

namespace gpar {

typedef double b1_d2 __attribute__((ext_vector_type(2)));

constexpr int GEMVT_COLS = 512;    // columns per workgroup (256 threads x 2 adjacent columns: 16-byte loads, 4 KB per row)
constexpr int GEMVT_ROWS = 512;    // rows per workgroup: n = 65536, M = 1024 -> 2 x 128 workgroups, 128 partial rows to sum

// partial[chunk][j] = sum over the chunk's rows of A[i][j] v[i]; rows are consumed in order, 8 at a time.
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const double* __restrict__ A, int rows, int cols, int lda,
                                                             const double* __restrict__ v, double* __restrict__ partial) {
    const int j = blockIdx.x * GEMVT_COLS + 2 * threadIdx.x;
    const int r0 = blockIdx.y * GEMVT_ROWS;
    const int r1 = min(rows, r0 + GEMVT_ROWS);
    const bool vec = ((lda & 1) == 0) && gpar_aligned16(A) && j + 1 < cols;
    double a0 = 0.0, a1 = 0.0;
    if (j < cols) {
        const int jc = min(j + 1, cols - 1);
        for (int r = r0; r < r1; r += 8) {
            b1_d2 x[8];
            double w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rr = min(r + q, r1 - 1);   // clamped, never behind a branch; the weight is zeroed instead
                const double* row = A + (size_t)rr * lda;
                x[q] = vec ? *reinterpret_cast<const b1_d2*>(row + j) : b1_d2{row[j], row[jc]};
                w[q] = (r + q < r1) ? v[rr] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { a0 = fma(x[q][0], w[q], a0); a1 = fma(x[q][1], w[q], a1); }
        }
        partial[(size_t)blockIdx.y * cols + j] = a0;
        if (j + 1 < cols) partial[(size_t)blockIdx.y * cols + j + 1] = a1;
    }
}

// out[j] = sum over chunks, in a FIXED order: 64 columns per workgroup (lane = column, coalesced), the chunk range cut into
// four contiguous quarters (one per wave, eight loads in flight), the four quarter sums added in order.
__global__ __launch_bounds__(256) void gemv_t_reduce_kernel(const double* __restrict__ partial, int nchunks, int cols,
                                                            double* __restrict__ out) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int jc = min(j, cols - 1);
    const int per = (nchunks + 3) / 4, c0 = min(w * per, nchunks), c1 = min(c0 + per, nchunks);
    double s = 0.0;
    for (int c = c0; c < c1; c += 8) {
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = partial[(size_t)min(c + q, c1 - 1) * cols + jc];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += (c + q < c1) ? x[q] : 0.0;
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && j < cols) out[j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

static inline int gemv_t_chunks(int rows) { return gpar_ceil_div(rows, GEMVT_ROWS); }

static int gemv_t_run(const double* A, int rows, int cols, int lda, const double* v, double* out, double* workspace,
                      hipStream_t stream) {
    if (cols <= 0) return 0;
    if (rows <= 0) {
        GPAR_HIP_TRY(hipMemsetAsync(out, 0, sizeof(double) * cols, stream));
        return 0;
    }
    const int nchunks = gemv_t_chunks(rows);
    hipLaunchKernelGGL(gemv_t_partial_kernel, dim3(gpar_ceil_div(cols, GEMVT_COLS), nchunks), dim3(256), 0, stream, A, rows,
                       cols, lda, v, workspace);
    hipLaunchKernelGGL(gemv_t_reduce_kernel, dim3(gpar_ceil_div(cols, 64)), dim3(256), 0, stream, (const double*)workspace,
                       nchunks, cols, out);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// One wave per row, lanes stride the row in 16-byte steps, 4 loads in flight per lane, fixed-order wave reduction.
__global__ __launch_bounds__(256) void rownorm2_kernel(const double* __restrict__ A, int rows, int cols, int lda,
                                                       double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double* r = A + (size_t)row * lda;
    const bool vec = ((lda & 1) == 0) && gpar_aligned16(A);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (vec) {
        const int npair = cols >> 1;
        for (int p = lane; p < npair; p += 256) {
            b1_d2 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pp = p + 64 * q;
                x[q] = *reinterpret_cast<const b1_d2*>(r + 2 * (pp < npair ? pp : p));
                if (pp >= npair) x[q] = b1_d2{0.0, 0.0};
            }
            a0 = fma(x[0][0], x[0][0], a0); a0 = fma(x[0][1], x[0][1], a0);
            a1 = fma(x[1][0], x[1][0], a1); a1 = fma(x[1][1], x[1][1], a1);
            a2 = fma(x[2][0], x[2][0], a2); a2 = fma(x[2][1], x[2][1], a2);
            a3 = fma(x[3][0], x[3][0], a3); a3 = fma(x[3][1], x[3][1], a3);
        }
        if ((cols & 1) && lane == 0) a0 = fma(r[cols - 1], r[cols - 1], a0);
    } else {
        for (int j = lane; j < cols; j += 64) a0 = fma(r[j], r[j], a0);
    }
    double s = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[row] = s;
}

// Lower triangle (diagonal included) of a row-major n x n matrix <-> packed storage, row r at offset r (r + 1) / 2: what one
// rank sends another when a Cholesky factor travels (half the bytes of the padded square buffer).  One workgroup per row.
__global__ __launch_bounds__(256) void pack_lower_kernel(const double* __restrict__ A, int n, int lda, double* __restrict__ out) {
    const int r = blockIdx.x;
    const double* src = A + (size_t)r * lda;
    double* dst = out + (size_t)r * (r + 1) / 2;
    for (int c = threadIdx.x; c <= r; c += 256) dst[c] = src[c];
}

__global__ __launch_bounds__(256) void unpack_lower_kernel(const double* __restrict__ in, int n, double* __restrict__ A, int lda) {
    const int r = blockIdx.x;
    const double* src = in + (size_t)r * (r + 1) / 2;
    double* dst = A + (size_t)r * lda;
    for (int c = threadIdx.x; c <= r; c += 256) dst[c] = src[c];
}

// single-workgroup, fixed-order reduction: deterministic
__global__ __launch_bounds__(1024) void dot_kernel(const double* __restrict__ x, long incx, const double* __restrict__ y,
                                                   long incy, int n, double* __restrict__ out, int accumulate) {
    __shared__ double part[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s = fma(x[(size_t)i * incx], y[(size_t)i * incy], s);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + part[0];
}

}  // namespace gpar
