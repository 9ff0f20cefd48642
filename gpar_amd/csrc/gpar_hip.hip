// libgpar_hip.so — C ABI (include/gpar_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC gpar_hip.hip -o ../libgpar_hip.so
#include <mutex>
#include <string.h>
#include "common.h"
#include "gemm_f64.h"
#include "potrf.h"
#include "panel.h"
#include "panel2.h"
#include "gram.h"
#include "gram_jit.h"
#include "grad_jit.h"
#include "blas1.h"

namespace gpar {

// Philox-4x32-10 (Salmon et al. 2011).  counter = (pair index lo, hi, offset lo, hi), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&r)[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__global__ __launch_bounds__(256) void randn_kernel(uint64_t seed, uint64_t offset, double* __restrict__ out, int rows,
                                                    int cols, int ldo) {
    const size_t total = (size_t)rows * cols;
    const size_t pair = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair * 2 >= total) return;
    uint32_t r[4];
    philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const uint64_t a = ((uint64_t)r[1] << 32) | r[0], b = ((uint64_t)r[3] << 32) | r[2];
    const double u1 = ((double)(a >> 11) + 0.5) * 1.1102230246251565404e-16;  // (0, 1)
    const double u2 = ((double)(b >> 11) + 0.5) * 1.1102230246251565404e-16;
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    const double zv[2] = {rad * cos(ang), rad * sin(ang)};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const size_t idx = pair * 2 + e;
        if (idx < total) {
            const size_t rr = idx / cols, cc = idx - rr * cols;
            out[rr * ldo + cc] = zv[e];
        }
    }
}

}  // namespace gpar

using namespace gpar;

// ---- y = L x for a lower-triangular L and one vector: one wave per row, lanes stride the row (coalesced), wave reduce
__global__ __launch_bounds__(256) void trmv_lower_kernel(const double* __restrict__ L, int n, int ldl, const double* __restrict__ x,
                                                         int incx, double* __restrict__ y, int incy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* Lr = L + (size_t)row * ldl;
    double a0 = 0.0, a1 = 0.0;
    int j = lane;
    for (; j + 64 <= row; j += 128) {
        a0 = fma(Lr[j], x[(size_t)j * incx], a0);
        a1 = fma(Lr[j + 64], x[(size_t)(j + 64) * incx], a1);
    }
    if (j <= row) a0 = fma(Lr[j], x[(size_t)j * incx], a0);
    double s = a0 + a1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) y[(size_t)row * incy] = s;
}

// ---- y = U x for an UPPER-triangular U (row i: columns i .. n - 1; the strict lower triangle is never read) and one vector:
// one wave per row, lanes stride the row, wave reduce.  alpha = (L L^T)^-1 y = L^-T (L^-1 y) = X z with X = L^-T, which the
// inverse of the gradient pass has just formed: n^2 / 2 multiply-adds at memory speed instead of a backward substitution's chain of
// n / 512 block kernels with an update each (n = 4096: 8 x 31 + 7 x 8 us -> ~15 us).
__global__ __launch_bounds__(256) void trmv_upper_kernel(const double* __restrict__ U, int n, int ldu, const double* __restrict__ x,
                                                         int incx, double* __restrict__ y, int incy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* Ur = U + (size_t)row * ldu;
    double a0 = 0.0, a1 = 0.0;
    int j = row + lane;
    for (; j + 64 < n; j += 128) {
        a0 = fma(Ur[j], x[(size_t)j * incx], a0);
        a1 = fma(Ur[j + 64], x[(size_t)(j + 64) * incx], a1);
    }
    if (j < n) a0 = fma(Ur[j], x[(size_t)j * incx], a0);
    double s = a0 + a1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) y[(size_t)row * incy] = s;
}

// ---- y = alpha A x for a general row-major A and one vector: one wave per row (coalesced along the row), wave reduce.  A 128-wide
// GEMM tile for ONE column is a K-long chain of stages for nothing (M = 1024: 99 us; this: ~6 us)
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* __restrict__ A, int rows, int cols, int lda, const double* __restrict__ x, int incx,
                                                     double alpha, double* __restrict__ y, int incy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double* Ar = A + (size_t)row * lda;
    double a0 = 0.0, a1 = 0.0;
    int j = lane;
    for (; j + 64 < cols; j += 128) {
        a0 = fma(Ar[j], x[(size_t)j * incx], a0);
        a1 = fma(Ar[j + 64], x[(size_t)(j + 64) * incx], a1);
    }
    if (j < cols) a0 = fma(Ar[j], x[(size_t)j * incx], a0);
    double s = a0 + a1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) y[(size_t)row * incy] = alpha * s;
}

// ---- the same for `batch` matrices and one vector each (gridDim.y = batch): y_b = L_b x_b (+ add_b)
__global__ __launch_bounds__(256) void trmv_lower_batch_kernel(const double* __restrict__ L, long long stride_l, int n, int ldl,
                                                               const double* __restrict__ x, int incx, long long stride_x,
                                                               const double* __restrict__ add, int inca, long long stride_add,
                                                               double* __restrict__ y, int incy, long long stride_y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const size_t b = blockIdx.y;
    const double* Lr = L + b * stride_l + (size_t)row * ldl;
    const double* xb = x + b * stride_x;
    double a0 = 0.0, a1 = 0.0;
    int j = lane;
    for (; j + 64 <= row; j += 128) {
        a0 = fma(Lr[j], xb[(size_t)j * incx], a0);
        a1 = fma(Lr[j + 64], xb[(size_t)(j + 64) * incx], a1);
    }
    if (j <= row) a0 = fma(Lr[j], xb[(size_t)j * incx], a0);
    double s = a0 + a1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) y[b * stride_y + (size_t)row * incy] = add ? s + add[b * stride_add + (size_t)row * inca] : s;
}

// ---- Monte-Carlo reduction over posterior samples (reference regression.py:589-595)
__device__ __forceinline__ double np_lerp(double a, double b, double t) {
    // numpy's _lerp: a + (b - a) t, replaced by b - (b - a)(1 - t) when t >= 0.5; numpy rounds the product and the
    // sum separately, so no fused multiply-add here
#pragma clang fp contract(off)
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

__global__ __launch_bounds__(256) void sample_stats_kernel(const double* __restrict__ x, int S, long long count, long long stride,
                                                           int k_lo, double g_lo, int k_hi, double g_hi, double* __restrict__ mean,
                                                           double* __restrict__ lo, double* __restrict__ hi) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    const double* col = x + e;
    double sum = 0.0;
    bool has_nan = false;
    for (int s = 0; s < S; ++s) {
        const double v = col[(long long)s * stride];
        has_nan |= (v != v);
        sum += v;
    }
    mean[e] = sum / (double)S;
    if (!lo && !hi) return;
    if (has_nan) {
        // np.percentile of a column that holds a NaN is NaN (NaN sorts last and poisons the interpolation); rank counting
        // with < and == would silently mis-rank instead
        if (lo) lo[e] = __builtin_nan("");
        if (hi) hi[e] = __builtin_nan("");
        return;
    }
    const int k_lo1 = min(k_lo + 1, S - 1), k_hi1 = min(k_hi + 1, S - 1);
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
    for (int s = 0; s < S; ++s) {
        const double v = col[(long long)s * stride];
        int rank = 0;   // position of sample s in the stable ascending order
        for (int t = 0; t < S; ++t) {
            const double u = col[(long long)t * stride];
            rank += (u < v || (u == v && t < s)) ? 1 : 0;
        }
        if (rank == k_lo) a0 = v;
        if (rank == k_lo1) b0 = v;
        if (rank == k_hi) a1 = v;
        if (rank == k_hi1) b1 = v;
    }
    if (lo) lo[e] = np_lerp(a0, b0, g_lo);
    if (hi) hi[e] = np_lerp(a1, b1, g_hi);
}

// The library keeps process-global state (look-ahead side streams and events, the profile hook, one-time kernel
// attributes).  Entry points only ENQUEUE work, so serialising them costs nothing measurable and makes the library safe
// to call from several host threads (GPARRegressor.fit trains independent layers from two threads, each on its stream).
static std::mutex g_api_mutex;
// Every entry point also makes the device of the caller's stream current for its duration (and restores the previous one):
// the auxiliary streams and events the library creates must live on that device, whatever the calling thread's current
// device happens to be (a fresh host thread starts on device 0).
struct GparDeviceGuard {
    int prev = -1, dev = -1;
    explicit GparDeviceGuard(void* stream) {
        hipDevice_t d;
        // (a null stream is the current device's default stream: nothing to switch, and per-device state is looked up by
        // the current device)
        if (stream && hipGetDevice(&prev) == hipSuccess && hipStreamGetDevice(static_cast<hipStream_t>(stream), &d) == hipSuccess) {
            dev = (int)d;
            if (dev != prev) GPAR_HIP_IGNORE(hipSetDevice(dev));   // a failure shows up in the launch that follows
        }
    }
    ~GparDeviceGuard() {
        if (dev >= 0 && dev != prev) GPAR_HIP_IGNORE(hipSetDevice(prev));
    }
};
#define GPAR_API_GUARD                                         \
    std::lock_guard<std::mutex> gpar_api_guard(g_api_mutex);   \
    GparDeviceGuard gpar_device_guard(stream)
#define GPAR_API_GUARD_NOSTREAM std::lock_guard<std::mutex> gpar_api_guard(g_api_mutex)

static int featurize_launch(const gpar_fspec_t* fs, const double* x, int n, int ldx, double* z, int ldz, hipStream_t stream) {
    if (!fs || fs->dz < 0 || fs->dz > GPAR_MAX_DIMS) return GPAR_ARG_ERROR(2);
    if (n <= 0 || fs->dz == 0) return 0;
    const long total = (long)n * fs->dz;
    hipLaunchKernelGGL(featurize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, *fs, x, n, ldx, z, ldz);
    GPAR_LAUNCH_CHECK();
    return 0;
}

static int gram_launch(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2, int dz, double* K,
                       int ldk, int flags, const double* diag_add, double diag_const, const double* row_scale, hipStream_t stream,
                       int batch = 1, long long batch_z = 0, long long batch_k = 0) {
    if (!ks || ks->nterms < 0 || ks->nterms > GPAR_MAX_TERMS || ks->nfactors < 0 || ks->nfactors > GPAR_MAX_FACTORS)
        return GPAR_ARG_ERROR(3);
    if (dz < 0 || dz > GPAR_MAX_DIMS) return GPAR_ARG_ERROR(4);
    if (n1 <= 0 || n2 <= 0) return 0;
    const int sym = (z1 == z2 && n1 == n2) ? 1 : 0;
    if ((flags & GPAR_GRAM_LOWER) && !sym) return GPAR_ARG_ERROR(5);
    const size_t lds = ((size_t)2 * (dz > 0 ? dz : 1) * GRAM_LD + GRAM_TAB_DOUBLES) * sizeof(double);
    const int nt1 = gpar_ceil_div(n1, GRAM_T), nt2 = gpar_ceil_div(n2, GRAM_T);
    dim3 grid(nt2, nt1, batch);
    if (flags & GPAR_GRAM_LOWER) grid = dim3((unsigned)((long long)nt1 * (nt1 + 1) / 2), 1, batch);
    // per-specification kernel (gram_jit.h) for problems large enough to repay its compilation; the interpreter otherwise
    if (gram_jit_launch(ks, z1, n1, ldz1, z2, n2, ldz2, dz, K, ldk, flags, diag_add, diag_const, row_scale, sym, grid, stream, batch_z, batch_k))
        return 0;
    hipLaunchKernelGGL(gram_kernel, grid, dim3(256), lds, stream, *ks, z1, n1, ldz1, z2, n2, ldz2, dz, K, ldk, flags, diag_add,
                       diag_const, row_scale, sym, batch_z, batch_k);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// row n of the augmented matrix <- observations, corner / log-determinant / info word <- 0
__global__ __launch_bounds__(256) void logpdf_prepare_kernel(const double* __restrict__ y, long incy, int n, double* __restrict__ A, int lda,
                                                             double* __restrict__ logdet, int* __restrict__ info) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) A[(size_t)n * lda + i] = y[(size_t)i * incy];
    if (i == n) {
        A[(size_t)n * lda + n] = 0.0;
        logdet[0] = 0.0;
        info[0] = 0;
    }
}

__global__ void logpdf_value_kernel(const double* __restrict__ A, int lda, int n, double n_log_2pi, const double* __restrict__ logdet,
                                    double* __restrict__ value, long long batch_a = 0) {
    A += (size_t)blockIdx.x * batch_a;   // batched: one block per matrix, its words at logdet[b] / value[b]
    logdet += blockIdx.x;
    value += blockIdx.x;
    // the corner holds -|L^-1 y|^2 (potrf.h: the augmented row's Schur complement).  Two additions and an exact scaling, the
    // product n log 2 pi formed on the host: the same roundings as the host-side expression this replaces (a multiply next to
    // an add would be contracted into a fused multiply-add here), so that the fused and the separate paths return the same bits.
    value[0] = -0.5 * ((logdet[0] + n_log_2pi) - A[(size_t)n * lda + n]);
}


// ---- one call for a whole lock-step evaluation (gpar_logpdf_lockstep) ---------------------------------------------------------
// observation-noise variance and observed column of up to LOCKSTEP_CHUNK layers, by value
constexpr int LOCKSTEP_CHUNK = 64;
struct LockstepCols {
    double noise[LOCKSTEP_CHUNK];
    int ycol[LOCKSTEP_CHUNK];
};

// For layer b = blockIdx.y of the chunk: row n of its augmented matrix <- its observations, its noise diagonal noise_b / w (an IEEE
// division, as the host-side tensor expression it replaces) when weights are given, corner / log-determinant / info word <- 0.
__global__ __launch_bounds__(256) void lockstep_prepare_kernel(LockstepCols lc, const double* __restrict__ y, int ldy, const double* __restrict__ w,
                                                               int ldw, int n, double* __restrict__ nd, double* __restrict__ A, int lda,
                                                               long long stride_a, double* __restrict__ logdet, int* __restrict__ info) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double* Ab = A + (size_t)b * stride_a;
    const int col = lc.ycol[b];
    if (i < n) {
        Ab[(size_t)n * lda + i] = y[(size_t)i * ldy + col];
        if (nd) nd[(size_t)b * n + i] = lc.noise[b] / w[(size_t)i * ldw + col];
    }
    if (i == n) {
        Ab[(size_t)n * lda + n] = 0.0;
        logdet[b] = 0.0;
        info[b] = 0;
    }
}

// value[b] as logpdf_value_kernel computes it, then total = ((0 + value[0]) + value[1]) + ... in layer order (the host-side sum it replaces)
__global__ __launch_bounds__(64) void lockstep_finish_kernel(const double* __restrict__ A, int lda, long long stride_a, int n, double n_log_2pi,
                                                             const double* __restrict__ logdet, double* __restrict__ value, int batch,
                                                             double* __restrict__ total) {
    for (int b = threadIdx.x; b < batch; b += 64) {
        const double* Ab = A + (size_t)b * stride_a;
        value[b] = -0.5 * ((logdet[b] + n_log_2pi) - Ab[(size_t)n * lda + n]);
    }
    __syncthreads();
    if (threadIdx.x == 0 && total) {
        double t = 0.0;
        for (int b = 0; b < batch; ++b) t = t + value[b];
        total[0] = t;
    }
}

// row n of the augmented factor -> a contiguous vector (alpha before its backward solve); 1/2 diag(W) -> a vector
__global__ __launch_bounds__(256) void copy_row_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void half_diag_kernel(const double* __restrict__ W, int ldw, int n, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 0.5 * W[(size_t)i * ldw + i];
}

// ---- the scalar side of the inducing-point bound (gpar_vfe_assemble / gpar_vfe_value) ------------------------------------------
// A <- [[G + diag_add I (lower triangle), .], [c^T, 0]], log-determinant and info words zeroed: one launch instead of six tensor operations
__global__ __launch_bounds__(256) void vfe_assemble_kernel(const double* __restrict__ G, int M, int ldg, const double* __restrict__ c, double diag_add,
                                                           double* __restrict__ A, int lda, double* __restrict__ logdet, int* __restrict__ info) {
    const int r = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (r < M) {
        if (j <= r) A[(size_t)r * lda + j] = G[(size_t)r * ldg + j] + (j == r ? diag_add : 0.0);
    } else {
        if (j < M) A[(size_t)M * lda + j] = c[j];
        if (j == M) {
            A[(size_t)M * lda + M] = 0.0;
            logdet[0] = 0.0;
            info[0] = 0;
        }
    }
}
// Partial sums of  ys^2, kdiag / d, log d  (n terms each) and of the diagonal of G (M terms): VFE_PARTS workgroups, block b writes
// scal[4 b .. 4 b + 3]; vfe_value_kernel adds the parts in order (a fixed summation order; one workgroup over 65536 terms with a
// logarithm and a division each took 67 us)
constexpr int VFE_PARTS = 64;
__global__ __launch_bounds__(256) void vfe_sums_kernel(const double* __restrict__ ys, const double* __restrict__ kdiag, const double* __restrict__ d, int n,
                                                       const double* __restrict__ G, int M, int ldg, double* __restrict__ scal) {
    __shared__ double sm[4][256];
    const int t = threadIdx.x, b = blockIdx.x;
    const int per_n = (n + VFE_PARTS - 1) / VFE_PARTS, per_m = (M + VFE_PARTS - 1) / VFE_PARTS;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int i = b * per_n + t; i < min(n, (b + 1) * per_n); i += 256) {
        const double y = ys[i], di = d[i];
        a0 = fma(y, y, a0);
        a1 += kdiag[i] / di;
        a2 += log(di);
    }
    for (int i = b * per_m + t; i < min(M, (b + 1) * per_m); i += 256) a3 += G[(size_t)i * ldg + i];
    sm[0][t] = a0; sm[1][t] = a1; sm[2][t] = a2; sm[3][t] = a3;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sm[q][t] += sm[q][t + off];
        }
        __syncthreads();
    }
    if (t < 4) scal[4 * b + t] = sm[t][0];
}
// the bound from its pieces: -1/2 (trace + sum log d + n log 2 pi + log|A| + y^T D^-1 y - |L_A^-1 c|^2)
__global__ void vfe_value_kernel(const double* __restrict__ scal, const double* __restrict__ logdet, const double* __restrict__ A, int lda, int M,
                                 double n_log_2pi, int with_trace, double* __restrict__ out) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < VFE_PARTS; ++b)
        for (int q = 0; q < 4; ++q) s[q] += scal[4 * b + q];
    const double quad = -A[(size_t)M * lda + M];
    const double trace = with_trace ? s[1] - s[3] : 0.0;
    out[0] = -0.5 * (trace + s[2] + n_log_2pi + logdet[0] + s[0] - quad);
}

extern "C" {

int gpar_abi_version(void) { return GPAR_ABI_VERSION; }
size_t gpar_sizeof_fspec(void) { return sizeof(gpar_fspec_t); }
size_t gpar_sizeof_kspec(void) { return sizeof(gpar_kspec_t); }
size_t gpar_sizeof_layer(void) { return sizeof(gpar_layer_t); }

// Everything the library creates lazily, created now: the look-ahead side stream paired with `stream`, the event ring and
// every kernel's dynamic-LDS attribute on the device that owns `stream`.  After it, entry points called
// on `stream` make no HIP object-creation call - the precondition of capturing them into a hipGraph (the run-time compiled
// kernels excepted: a structure is compiled at its first large launch, which must therefore happen once outside the capture).
int gpar_init(void* stream) {
    GPAR_API_GUARD;
    if (!la_init()) return -(int)hipErrorOutOfMemory;
    if (!la_side((hipStream_t)stream)) return -(int)hipErrorOutOfMemory;
    if (!spin_chain_init()) return -(int)hipErrorOutOfMemory;
    const void* big[] = {reinterpret_cast<const void*>(&gemm_f64_kernel<false, false, 0>), reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 0>),
                         reinterpret_cast<const void*>(&gemm_f64_kernel<true, false, 0>), reinterpret_cast<const void*>(&gemm_f64_kernel<true, true, 0>),
                         reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1>), reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 0, 64>),
                         reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1, 64>), reinterpret_cast<const void*>(&gram_grad_kernel),
                         reinterpret_cast<const void*>(&gram_input_grad_kernel)};
    for (const void* fn : big) GPAR_HIP_TRY(gpar_set_max_lds(fn, 160 * 1024));
    const void* p2[] = {reinterpret_cast<const void*>(&potrf_panel2_kernel), reinterpret_cast<const void*>(&trsm_block2_kernel),
                        reinterpret_cast<const void*>(&trsm_block2_back_kernel), reinterpret_cast<const void*>(&trinv_blocks2_kernel)};
    for (const void* fn : p2) GPAR_HIP_TRY(gpar_set_max_lds(fn, P2_LDS_BYTES));
    return 0;
}

// ---- run-time specialisation (jit.h) ----------------------------------------------------------------------------------
static bool jit_request(int kind, const gpar_kspec_t& ks, int dz, int& jkind, int& extra, std::string& entry, std::string& source, bool want_source);

int gpar_jit_compile_check(int kind, const gpar_kspec_t* ks, int dz, const char* arch, char* log, int log_len) {
    // (no library lock: nothing shared is touched - source generation and hiprtc only -, and several threads may check at once)
    if (!ks || !arch || ks->nterms < 0 || ks->nterms > GPAR_MAX_TERMS || ks->nfactors < 0 || ks->nfactors > GPAR_MAX_FACTORS || dz < 0 ||
        dz > GPAR_MAX_DIMS)
        return GPAR_ARG_ERROR(1);
    std::string source, entry;
    int jkind = 0, extra = 0;
    if (!jit_request(kind, *ks, dz, jkind, extra, entry, source, true)) return GPAR_ARG_ERROR(2);
    std::string code, text;
    const bool ok = jit_compile(source, entry.c_str(), arch, code, text);
    if (log && log_len > 0) {
        const size_t n = text.size() < (size_t)(log_len - 1) ? text.size() : (size_t)(log_len - 1);
        memcpy(log, text.data(), n);
        log[n] = '\0';
    }
    return ok ? (int)code.size() : -1;
}

// Compile the kernel of `kind` for this structure and hand back the code object together with its archive key: the build step
// (gpar_amd/aot.py) collects these into gpar_aot_<arch>.bin.  Needs no GPU.  Returns the code size (> capacity: nothing copied), or -1.
long long gpar_jit_compile(int kind, const gpar_kspec_t* ks, int dz, const char* arch, void* code_out, long long capacity, char* key_out,
                           int key_len, char* log, int log_len) {
    if (!ks || !arch || ks->nterms < 0 || ks->nterms > GPAR_MAX_TERMS || ks->nfactors < 0 || ks->nfactors > GPAR_MAX_FACTORS || dz < 0 ||
        dz > GPAR_MAX_DIMS)
        return GPAR_ARG_ERROR(1);
    std::string source, entry;
    int jkind = 0, extra = 0;
    if (!jit_request(kind, *ks, dz, jkind, extra, entry, source, true)) return GPAR_ARG_ERROR(2);
    std::string code, text;
    const bool ok = jit_compile(source, entry.c_str(), arch, code, text);
    if (log && log_len > 0) {
        const size_t n = text.size() < (size_t)(log_len - 1) ? text.size() : (size_t)(log_len - 1);
        memcpy(log, text.data(), n);
        log[n] = '\0';
    }
    if (!ok) return -1;
    const std::string key = aot_key(jkind, *ks, dz, extra);
    if (key_out && key_len > 0) {
        const size_t n = key.size() < (size_t)(key_len - 1) ? key.size() : (size_t)(key_len - 1);
        memcpy(key_out, key.data(), n);
        key_out[n] = '\0';
    }
    if (code_out && (long long)code.size() <= capacity) memcpy(code_out, code.data(), code.size());
    return (long long)code.size();
}

// Fingerprint of the kernel generators: FNV-1a over the sources of every kernel kind for one probe structure that uses every factor
// type (EQ x RQ product, a linear term, a constant term; six feature dims), + the ABI version.  Pure host code, needs no GPU.
unsigned long long gpar::aot_fingerprint() {
    static unsigned long long cached = 0ull;
    if (cached) return cached;
    gpar_kspec_t ks;
    memset(&ks, 0, sizeof ks);
    ks.nterms = 3;
    ks.nfactors = 3;
    ks.coef[0] = 1.0; ks.coef[1] = 1.0; ks.coef[2] = 1.0;
    ks.factor[0] = gpar_factor_t{GPAR_K_EQ, 0, 0, 2, 0.0};
    ks.factor[1] = gpar_factor_t{GPAR_K_RQ, 0, 2, 2, 0.5};
    ks.factor[2] = gpar_factor_t{GPAR_K_LINEAR, 1, 4, 2, 0.0};
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const std::string& text) {
        for (unsigned char ch : text) h = (h ^ ch) * 1099511628211ull;
        h = (h ^ 0xffu) * 1099511628211ull;
    };
    const int kinds[] = {JIT_GRAM, JIT_GRAD, JIT_GRAD + 10, JIT_GRAD + 20, JIT_GRAD + 30, JIT_INPUT_GRAD, JIT_INPUT_GRAD + 10};
    for (int kind : kinds) {
        std::string source, entry;
        int jkind = 0, extra = 0;
        if (jit_request(kind, ks, 6, jkind, extra, entry, source, true)) mix(entry + "\n" + source);
        else mix("-");
    }
    {   // ... and the wide form of the Gram generator (more than GRAM_JIT_MAX_DZ feature dims): the same factor types over 20 dims
        gpar_kspec_t wide = ks;
        wide.factor[0].nd = 8; wide.factor[1].off = 8; wide.factor[1].nd = 7; wide.factor[2].off = 15; wide.factor[2].nd = 5;
        std::string source, entry;
        int jkind = 0, extra = 0;
        if (jit_request(JIT_GRAM, wide, 20, jkind, extra, entry, source, true)) mix(entry + "\n" + source);
        else mix("-");
    }
    mix("abi " + std::to_string(GPAR_ABI_VERSION));
    cached = h ? h : 1ull;
    return cached;
}

unsigned long long gpar_aot_fingerprint(void) { return gpar::aot_fingerprint(); }

int gpar_aot_stats(int* entries, int* loaded) {
    GPAR_API_GUARD_NOSTREAM;
    if (entries) *entries = (int)g_aot.entries.size();
    if (loaded) *loaded = g_aot.loaded;
    return 0;
}

// (kind, structure) -> what the launch paths would ask jit_get for: cache key ingredients, entry point, source
static bool jit_request(int kind, const gpar_kspec_t& ks, int dz, int& jkind, int& extra, std::string& entry, std::string& source, bool want_source) {
    if (kind == JIT_GRAM) {
        const int strip = gram_jit_strip(1 << 20, dz);
        if (strip <= 0) return false;   // wide structure: there is no generated Gram kernel (gram_jit.h)
        jkind = JIT_GRAM; extra = 1; entry = "gram_jit";
        if (want_source) source = gram_jit_source(ks, dz, strip);
    } else if (kind == JIT_GRAD || kind == JIT_GRAD + 10 || kind == JIT_GRAD + 20 || kind == JIT_GRAD + 30) {
        const int mode = (kind / 10) & 1, has_zd = kind >= 20;
        jkind = JIT_GRAD; extra = 100 + mode * 2 + has_zd; entry = "gram_grad_jit";
        if (want_source) source = grad_jit_source(ks, dz, mode, has_zd);
    } else if (kind == JIT_INPUT_GRAD || kind == JIT_INPUT_GRAD + 10) {
        jkind = JIT_INPUT_GRAD; extra = 200 + kind / 10; entry = "gram_input_grad_jit";
        if (want_source) source = input_grad_jit_source(ks, dz, kind / 10);
    } else {
        return false;
    }
    return true;
}

// Compile the kernel of `kind` for this structure NOW if it is not cached yet, on the calling thread and OUTSIDE the library's
// mutex: a host that knows which structures a model will need (all layers of a GPARRegressor) calls this from several threads at
// once, so that p structures cost one compilation time instead of p, and no compilation happens inside a timed or pipelined
// evaluation (where it would also hold the mutex every other thread's launches need).  Returns 0 (ready), 1 (compilation failed:
// the interpreter will be used) or an argument error.
int gpar_jit_prepare(int kind, const gpar_kspec_t* ks, int dz, void* stream) {
    if (!ks || ks->nterms < 0 || ks->nterms > GPAR_MAX_TERMS || ks->nfactors < 0 || ks->nfactors > GPAR_MAX_FACTORS || dz < 0 || dz > GPAR_MAX_DIMS)
        return GPAR_ARG_ERROR(1);
    int jkind = 0, extra = 0;
    std::string entry, source, key, arch;
    {
        GPAR_API_GUARD;
        if (!jit_request(kind, *ks, dz, jkind, extra, entry, source, false)) return GPAR_ARG_ERROR(2);
        key = jit_key(jkind, *ks, dz, extra);
        if (key.empty()) return -(int)hipErrorInvalidDevice;
        auto it = g_jit.cache.find(key);
        if (it != g_jit.cache.end()) return it->second.failed ? 1 : 0;
        arch = jit_device_arch();
        if (aot_install(jkind, *ks, dz, extra, key, entry.c_str(), arch)) return 0;   // compiled when the library was built
    }
    jit_request(kind, *ks, dz, jkind, extra, entry, source, true);
    std::string code, log;
    const bool ok = !arch.empty() && jit_compile(source, entry.c_str(), arch, code, log);
    {
        GPAR_API_GUARD;
        auto it = g_jit.cache.find(key);   // (another thread may have been faster)
        if (it != g_jit.cache.end()) return it->second.failed ? 1 : 0;
        return jit_install(key, code, entry.c_str(), ok, log) ? 0 : 1;
    }
}

int gpar_jit_stats(int* compiled, int* failures, int* cached) {
    GPAR_API_GUARD_NOSTREAM;
    if (compiled) *compiled = g_jit.compiled;
    if (failures) *failures = g_jit.failures;
    if (cached) *cached = (int)g_jit.cache.size();
    return 0;
}

int gpar_featurize(const gpar_fspec_t* fs, const double* x, int n, int ldx, double* z, int ldz, void* stream) {
    GPAR_API_GUARD;
    return featurize_launch(fs, x, n, ldx, z, ldz, (hipStream_t)stream);
}

int gpar_gram(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2, int dz,
              double* K, int ldk, int flags, const double* diag_add, double diag_const, const double* row_scale, void* stream) {
    GPAR_API_GUARD;
    return gram_launch(ks, z1, n1, ldz1, z2, n2, ldz2, dz, K, ldk, flags, diag_add, diag_const, row_scale, (hipStream_t)stream);
}

// One dense layer's log marginal likelihood in one call (SURVEY 8(b)'s fused a2-a4; gpar/model.py:226 for a prior process):
// features -> Gram + noise + jitter into the top-left n x n of the (n + 1) x (n + 1) buffer A -> y into row n -> partial
// factorisation -> value = -1/2 (log|K| + n log 2 pi + |L^-1 y|^2).  The same launches the separate entry points make; what
// it saves is the caller's side: a Python host spends ~0.1 ms per layer on the five calls and the small tensor operations
// between them, which at n = 4096 decides when the last of four pipelined layers starts.
int gpar_logpdf_dense(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, const double* y, long incy,
                      const double* noise_diag, double jitter, double* z, int ldz, double* A, int lda, double* logdet, int* info,
                      double* value, int potrf_flags, void* stream) {
    GPAR_API_GUARD;
    if (!fs || !ks || !A || !logdet || !info || !value || (n > 0 && (!x || !y))) return GPAR_ARG_ERROR(1);
    hipStream_t st = (hipStream_t)stream;
    int rc = featurize_launch(fs, x, n, ldx, z, ldz, st);
    if (!rc && n > 0) rc = gram_launch(ks, z, n, ldz, z, n, ldz, fs->dz, A, lda, GPAR_GRAM_LOWER, noise_diag, jitter, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(logpdf_prepare_kernel, dim3(gpar_ceil_div(n + 1, 256)), dim3(256), 0, st, y, incy, n, A, lda, logdet, info);
    if (n > 0) rc = potrf_run(A, n + 1, n, lda, logdet, info, st, potrf_flags);
    if (rc) return rc;
    hipLaunchKernelGGL(logpdf_value_kernel, dim3(1), dim3(1), 0, st, (const double*)A, lda, n, (double)n * 1.8378770664093453,
                       (const double*)logdet, value);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_gram_batch(const gpar_kspec_t* ks, const double* z, int n, int ldz, long long stride_z, int dz, double* K, int ldk,
                    long long stride_k, int flags, const double* diag_add, double diag_const, int batch, void* stream) {
    GPAR_API_GUARD;
    if (batch <= 0) return 0;
    if (stride_z < 0 || stride_k < 0) return GPAR_ARG_ERROR(9);
    return gram_launch(ks, z, n, ldz, z, n, ldz, dz, K, ldk, flags, diag_add, diag_const, nullptr, (hipStream_t)stream, batch, stride_z, stride_k);
}

int gpar_logpdf_dense_build(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, const double* y, long incy,
                            const double* noise_diag, double jitter, double* z, int ldz, double* A, int lda, double* logdet, int* info,
                            void* stream) {
    GPAR_API_GUARD;
    if (!fs || !ks || !A || !logdet || !info || (n > 0 && (!x || !y))) return GPAR_ARG_ERROR(1);
    hipStream_t st = (hipStream_t)stream;
    int rc = featurize_launch(fs, x, n, ldx, z, ldz, st);
    if (!rc && n > 0) rc = gram_launch(ks, z, n, ldz, z, n, ldz, fs->dz, A, lda, GPAR_GRAM_LOWER, noise_diag, jitter, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(logpdf_prepare_kernel, dim3(gpar_ceil_div(n + 1, 256)), dim3(256), 0, st, y, incy, n, A, lda, logdet, info);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_potrf_batch(double* A, int batch, long long stride_a, int N, int nf, int lda, double* logdet, int* info, int flags, void* stream) {
    GPAR_API_GUARD;
    if (N <= 0 || nf <= 0 || batch <= 0) return 0;
    if (!A || !logdet || !info || stride_a < 0) return GPAR_ARG_ERROR(1);
    return potrf_run_batch(A, batch, stride_a, N, nf, lda, logdet, info, (hipStream_t)stream, flags);
}

int gpar_logpdf_dense_finish(const double* A, int batch, long long stride_a, int n, int lda, const double* logdet, double* value,
                             void* stream) {
    GPAR_API_GUARD;
    if (batch <= 0) return 0;
    if (!A || !logdet || !value || n < 0) return GPAR_ARG_ERROR(1);
    hipLaunchKernelGGL(logpdf_value_kernel, dim3(batch), dim3(1), 0, (hipStream_t)stream, A, lda, n, (double)n * 1.8378770664093453, logdet,
                       value, stride_a);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_logpdf_lockstep(const gpar_layer_t* layers, int batch, const double* x, int n, int ldx, const double* y, int ldy, const double* w,
                         int ldw, double jitter, double* z, int ldz, double* nd, double* A, int lda, long long stride_a, double* logdet,
                         int* info, double* value, double* total, int potrf_flags, void* stream) {
    GPAR_API_GUARD;
    if (batch <= 0) return 0;
    if (!layers || !A || !logdet || !info || !value || stride_a < 0 || (n > 0 && (!x || !y || !z)) || (w && !nd)) return GPAR_ARG_ERROR(1);
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < batch; ++b)
        if (!layers[b].fs || !layers[b].ks || layers[b].y_col < 0) return GPAR_ARG_ERROR(2);
    // Small evaluations: ONE launch per LS_CHUNK layers builds everything (gram.h: lockstep_build_kernel; same bits as the launches
    // below) - when every feature map fits its compact form and the build is short enough for launch latency to matter.
    bool fused_build = n > 0 && n <= env_int("GPAR_LOCKSTEP_FUSED_BUILD_ROWS", 4096) && w != nullptr;
    for (int b = 0; b < batch && fused_build; ++b) {
        const gpar_fspec_t& fs = *layers[b].fs;
        if (fs.dz < 0 || fs.dz > LS_MAXDZ) fused_build = false;
        for (int q = 0; q < fs.dz && fused_build; ++q)
            if (fs.col[q] < 0 || fs.col[q] > 255) fused_build = false;
    }
    if (fused_build) {
        GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&lockstep_build_kernel), 64 * 1024));
        const int nt = gpar_ceil_div(n, GRAM_T);
        for (int b0 = 0; b0 < batch; b0 += LS_CHUNK) {
            const int cnt = batch - b0 < LS_CHUNK ? batch - b0 : LS_CHUNK;
            LockstepSpecs sp;
            memset(&sp, 0, sizeof(sp));
            int dzmax = 1;
            for (int b = 0; b < cnt; ++b) {
                const gpar_layer_t& L = layers[b0 + b];
                sp.ks[b] = *L.ks;
                sp.fs[b].dz = L.fs->dz;
                for (int q = 0; q < L.fs->dz; ++q) {
                    sp.fs[b].col[q] = (unsigned char)L.fs->col[q];
                    sp.fs[b].embed[q] = (unsigned char)L.fs->embed[q];
                    sp.fs[b].inv_scale[q] = L.fs->inv_scale[q];
                    sp.fs[b].freq[q] = L.fs->freq[q];
                }
                sp.noise[b] = L.noise;
                sp.ycol[b] = L.y_col;
                if (L.fs->dz > dzmax) dzmax = L.fs->dz;
            }
            const size_t lds = ((size_t)2 * dzmax * GRAM_LD + GRAM_TAB_DOUBLES) * sizeof(double);
            hipLaunchKernelGGL(lockstep_build_kernel, dim3((unsigned)((long long)nt * (nt + 1) / 2), 1, cnt), dim3(256), lds, st, sp, x, n, ldx, y, ldy,
                               w, ldw, jitter, A + (size_t)b0 * stride_a, lda, stride_a, logdet + b0, info + b0);
        }
        GPAR_LAUNCH_CHECK();
    } else {
    for (int b0 = 0; b0 < batch; b0 += LOCKSTEP_CHUNK) {
        const int cnt = batch - b0 < LOCKSTEP_CHUNK ? batch - b0 : LOCKSTEP_CHUNK;
        LockstepCols lc;
        for (int b = 0; b < cnt; ++b) { lc.noise[b] = layers[b0 + b].noise; lc.ycol[b] = layers[b0 + b].y_col; }
        hipLaunchKernelGGL(lockstep_prepare_kernel, dim3(gpar_ceil_div(n + 1, 256), cnt), dim3(256), 0, st, lc, y, ldy, w, ldw, n,
                           w ? nd + (size_t)b0 * n : nullptr, A + (size_t)b0 * stride_a, lda, stride_a, logdet + b0, info + b0);
    }
    GPAR_LAUNCH_CHECK();
    for (int b = 0; b < batch && n > 0; ++b) {
        const gpar_layer_t& L = layers[b];
        double* zb = z + (size_t)b * n * ldz;
        int rc = featurize_launch(L.fs, x, n, ldx, zb, ldz, st);
        // unit weights: the diagonal is noise + jitter, added exactly as the kernel adds diag_add[row] + diag_const
        if (!rc) rc = gram_launch(L.ks, zb, n, ldz, zb, n, ldz, L.fs->dz, A + (size_t)b * stride_a, lda, GPAR_GRAM_LOWER, w ? nd + (size_t)b * n : nullptr,
                                  w ? jitter : L.noise + jitter, nullptr, st);
        if (rc) return rc;
    }
    }
    if (n > 0) {
        const int rc = potrf_run_batch(A, batch, stride_a, n + 1, n, lda, logdet, info, st, potrf_flags);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(lockstep_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)A, lda, stride_a, n, (double)n * 1.8378770664093453,
                       (const double*)logdet, value, batch, total);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_gram_diag(const gpar_kspec_t* ks, const double* z, int n, int ldz, int dz, double* out, void* stream) {
    GPAR_API_GUARD;
    (void)dz;
    if (!ks) return GPAR_ARG_ERROR(3);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gram_diag_kernel, dim3(gpar_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, *ks, z, n, ldz, out);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_featurize_dfreq(const gpar_fspec_t* fs, const double* x, int n, int ldx, double* zd, int ldz, void* stream) {
    GPAR_API_GUARD;
    if (!fs || fs->dz < 0 || fs->dz > GPAR_MAX_DIMS) return GPAR_ARG_ERROR(2);
    if (n <= 0 || fs->dz == 0) return 0;
    const long total = (long)n * fs->dz;
    hipLaunchKernelGGL(featurize_dfreq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *fs, x,
                       n, ldx, zd, ldz);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_grad_nacc(void) { return GRAD_NACC; }

static int gram_grad_launch(const gpar_kspec_t* ks, const double* z1, const double* zd1, int n1, int ldz1, const double* z2,
                            const double* zd2, int n2, int ldz2, int dz, const double* W, int ldw, int mode, double* workspace,
                            int nblocks, double* out, void* stream, bool reduce = true) {
    if (!ks || ks->nterms < 0 || ks->nterms > GPAR_MAX_TERMS || ks->nfactors < 0 || ks->nfactors > GPAR_MAX_FACTORS)
        return GPAR_ARG_ERROR(3);
    if (dz < 0 || dz > GPAR_MAX_DIMS || nblocks <= 0) return GPAR_ARG_ERROR(4);
    if (mode != GPAR_GRAD_SYM && mode != GPAR_GRAD_RECT && mode != GPAR_GRAD_DIAG) return GPAR_ARG_ERROR(13);
    if (mode != GPAR_GRAD_RECT && (n2 != n1 || z2 != z1)) return GPAR_ARG_ERROR(9);
    if ((zd1 == nullptr) != (zd2 == nullptr)) return GPAR_ARG_ERROR(8);
    {   // the pass handles at most GRAD_MAXF factors per product term
        int cnt[GPAR_MAX_TERMS] = {0};
        for (int f = 0; f < ks->nfactors; ++f) {
            const int t = ks->factor[f].term;
            if (t < 0 || t >= ks->nterms || ++cnt[t] > GRAD_MAXF) return GPAR_ARG_ERROR(6);
        }
    }
    const size_t lds = ((size_t)4 * (dz > 0 ? dz : 1) * GRAM_LD + 4 * GRAD_NACC) * sizeof(double);
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&gram_grad_kernel), 160 * 1024));
    if (lds > 160 * 1024) return GPAR_ARG_ERROR(7);
    // per-specification kernel (grad_jit.h) for problems large enough to repay its compilation; the interpreter otherwise
    if (!grad_jit_launch(ks, z1, zd1, n1, ldz1, z2, zd2, n2, ldz2, dz, W, ldw, mode, workspace, nblocks, (hipStream_t)stream))
        hipLaunchKernelGGL(gram_grad_kernel, dim3(nblocks), dim3(256), lds, (hipStream_t)stream, *ks, z1, zd1, n1, ldz1, z2, zd2, n2, ldz2,
                           dz, W, ldw, mode, workspace);
    // (`reduce` false: the caller sums the partials itself - dense_grad_epilogue_kernel, in the same order)
    if (reduce) hipLaunchKernelGGL(gram_grad_reduce_kernel, dim3(GRAD_NACC), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, nblocks, out);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_gram_grad(const gpar_kspec_t* ks, const double* z, const double* zd, int n, int ldz, int dz, const double* W,
                   int ldw, double* workspace, int nblocks, double* out, void* stream) {
    GPAR_API_GUARD;
    return gram_grad_launch(ks, z, zd, n, ldz, z, zd, n, ldz, dz, W, ldw, GPAR_GRAD_SYM, workspace, nblocks, out, stream);
}

int gpar_gram_grad_cross(const gpar_kspec_t* ks, const double* z1, const double* zd1, int n1, int ldz1, const double* z2,
                         const double* zd2, int n2, int ldz2, int dz, const double* W, int ldw, int mode, double* workspace,
                         int nblocks, double* out, void* stream) {
    GPAR_API_GUARD;
    return gram_grad_launch(ks, z1, zd1, n1, ldz1, z2, zd2, n2, ldz2, dz, W, ldw, mode, workspace, nblocks, out, stream);
}

// ---- launches folded together for the one-call training objective (round 6).  Below ~1000 rows an evaluation is a CHAIN of ~23 short
// kernels and every link costs 8-9 us of latency whatever it computes: the element-wise ones are merged - the same expressions per
// element, so the same bits as the separate kernels (featurize_kernel, featurize_dfreq_kernel, logpdf_prepare_kernel; logpdf_value_kernel,
// gram_grad_reduce_kernel, half_diag_kernel).
__global__ __launch_bounds__(256) void dense_grad_prep_kernel(gpar_fspec_t fs, const double* __restrict__ x, int n, int ldx, double* __restrict__ z,
                                                              double* __restrict__ zd, int ldz, const double* __restrict__ y, long incy,
                                                              double* __restrict__ A, int lda, double* __restrict__ logdet, int* __restrict__ info) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dz = fs.dz;
    if (idx < n * dz) {
        const int r = idx / dz, q = idx - r * dz;
        const double v = x[(size_t)r * ldx + fs.col[q]];
        double e, d = 0.0;
        if (fs.embed[q] == GPAR_EMBED_SIN) { e = sin(v * fs.freq[q]); d = v * cos(v * fs.freq[q]); }
        else if (fs.embed[q] == GPAR_EMBED_COS) { e = cos(v * fs.freq[q]); d = -v * sin(v * fs.freq[q]); }
        else e = v;
        z[(size_t)r * ldz + q] = e * fs.inv_scale[q];
        if (zd) zd[(size_t)r * ldz + q] = d * fs.inv_scale[q];
    }
    if (idx < n) A[(size_t)n * lda + idx] = y[(size_t)idx * incy];
    if (idx == n) {
        A[(size_t)n * lda + n] = 0.0;
        logdet[0] = 0.0;
        info[0] = 0;
    }
}

// blocks [0, GRAD_NACC): the gradient pass's partial sums in their fixed order (lanes 0-63, as gram_grad_reduce_kernel); blocks
// [GRAD_NACC, GRAD_NACC + ceil(n / 256)): 1/2 diag W; the last block: the value from the corner of the factor and its log-determinant.
__global__ __launch_bounds__(256) void dense_grad_epilogue_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out,
                                                                  const double* __restrict__ W, int ldw, int n, double* __restrict__ half_diag,
                                                                  const double* __restrict__ A, int lda, double n_log_2pi,
                                                                  const double* __restrict__ logdet) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < GRAD_NACC) {
        if (t >= 64) return;
        double s = 0.0;
        for (int q = t; q < nblocks; q += 64) s += partial[(size_t)q * GRAD_NACC + b];
        s = wave_sum(s);
        if (t == 0) out[2 + b] = s;
        return;
    }
    const int hb = b - GRAD_NACC, nh = (n + 255) / 256;
    if (hb < nh) {
        const int i = hb * 256 + t;
        if (i < n) half_diag[i] = 0.5 * W[(size_t)i * ldw + i];
        return;
    }
    if (t == 0) out[0] = -0.5 * ((logdet[0] + n_log_2pi) - A[(size_t)n * lda + n]);
}

// Everything of gpar_logpdf_dense_grad behind the factorisation: value, K^-1 from L, alpha^T = (L^-1 y)^T L^-1, W = alpha alpha^T - K^-1,
// the fused weighted-sum pass, 1/2 diag W.  `logdet`: the word the factorisation left (out + 1 itself in the one-call form).
static int logpdf_grad_finish_run(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, double* z, double* zd,
                                  int ldz, double* A, int lda, const double* logdet, double* X, int ldxw, double* W, int ldw, double* alpha,
                                  double* workspace, int nblocks, double* out, double* half_diag, void* stream, bool dfreq = true) {
    hipStream_t st = (hipStream_t)stream;
    if (dfreq && zd && fs->dz > 0) {
        const long total = (long)n * fs->dz;
        hipLaunchKernelGGL(featurize_dfreq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *fs, x, n, ldx, zd, ldz);
    }
    int rc = chol_inverse_run(A, n, lda, X, ldxw, W, ldw, st);
    if (rc) return rc;
    // alpha = (K + D)^-1 y = X (L^-1 y): X = L^-T is what the inverse has just left in its workspace, L^-1 y is row n of the factor
    hipLaunchKernelGGL(trmv_upper_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const double*)X, n, ldxw,
                       (const double*)(A + (size_t)n * lda), 1, alpha, 1);
    rc = gemm_launch(1, 0, n, n, 1, 1.0, alpha, n, alpha, n, -1.0, W, ldw, GPAR_GEMM_C_LOWER, st);
    if (!rc) rc = gram_grad_launch(ks, z, zd, n, ldz, z, zd, n, ldz, fs->dz, W, ldw, GPAR_GRAD_SYM, workspace, nblocks, out + 2, stream, false);
    if (rc) return rc;
    // the partial sums of the gradient pass, 1/2 diag W and the value (from the corner of the factor, untouched since the factorisation)
    hipLaunchKernelGGL(dense_grad_epilogue_kernel, dim3((unsigned)(GRAD_NACC + (n + 255) / 256 + 1)), dim3(256), 0, st, (const double*)workspace,
                       nblocks, out, (const double*)W, ldw, n, half_diag, (const double*)A, lda, (double)n * 1.8378770664093453, logdet);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_logpdf_dense_grad(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, const double* y, long incy,
                           const double* noise_diag, double jitter, double* z, double* zd, int ldz, double* A, int lda, double* X, int ldxw,
                           double* W, int ldw, double* alpha, double* workspace, int nblocks, double* out, double* half_diag, int* info,
                           int potrf_flags, void* stream) {
    GPAR_API_GUARD;
    if (!fs || !ks || !x || !y || !z || !A || !X || !W || !alpha || !workspace || !out || !half_diag || !info || n <= 0 || nblocks <= 0)
        return GPAR_ARG_ERROR(1);
    hipStream_t st = (hipStream_t)stream;
    // ---- what gpar_logpdf_dense does up to the factor: features (+ their frequency derivatives) and observations in ONE launch, Gram,
    // the augmented factorisation
    if (fs->dz < 0 || fs->dz > GPAR_MAX_DIMS) return GPAR_ARG_ERROR(2);
    {
        const long total = (long)n * fs->dz > (long)n + 1 ? (long)n * fs->dz : (long)n + 1;
        hipLaunchKernelGGL(dense_grad_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *fs, x, n, ldx, z, zd, ldz, y, incy, A, lda,
                           out + 1, info);
    }
    int rc = gram_launch(ks, z, n, ldz, z, n, ldz, fs->dz, A, lda, GPAR_GRAM_LOWER, noise_diag, jitter, nullptr, st);
    if (rc) return rc;
    rc = potrf_run(A, n + 1, n, lda, out + 1, info, st, potrf_flags);
    if (rc) return rc;
    // ---- value and gradient ingredients from the factor
    return logpdf_grad_finish_run(fs, ks, x, n, ldx, z, zd, ldz, A, lda, out + 1, X, ldxw, W, ldw, alpha, workspace, nblocks, out, half_diag, stream,
                                  false);
}

int gpar_logpdf_dense_grad_finish(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, double* z, double* zd,
                                  int ldz, double* A, int lda, const double* logdet, const int* info, double* X, int ldxw, double* W, int ldw,
                                  double* alpha, double* workspace, int nblocks, double* out, double* half_diag, int* info_out, void* stream) {
    GPAR_API_GUARD;
    if (!fs || !ks || !x || !z || !A || !logdet || !X || !W || !alpha || !workspace || !out || !half_diag || n <= 0 || nblocks <= 0)
        return GPAR_ARG_ERROR(1);
    hipStream_t st = (hipStream_t)stream;
    GPAR_HIP_TRY(hipMemcpyAsync(out + 1, logdet, sizeof(double), hipMemcpyDeviceToDevice, st));
    if (info && info_out) GPAR_HIP_TRY(hipMemcpyAsync(info_out, info, sizeof(int), hipMemcpyDeviceToDevice, st));
    return logpdf_grad_finish_run(fs, ks, x, n, ldx, z, zd, ldz, A, lda, logdet, X, ldxw, W, ldw, alpha, workspace, nblocks, out, half_diag, stream);
}

int gpar_gram_input_grad(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2, int dz,
                         const double* W, int ldw, int mode, int nsplit, double* workspace, double* out, int ldo, void* stream) {
    GPAR_API_GUARD;
    if (!ks || ks->nterms < 0 || ks->nterms > GPAR_MAX_TERMS || ks->nfactors < 0 || ks->nfactors > GPAR_MAX_FACTORS)
        return GPAR_ARG_ERROR(3);
    if (dz < 0 || dz > GPAR_MAX_DIMS || nsplit <= 0) return GPAR_ARG_ERROR(4);
    if (mode != GPAR_GRAD_SYM && mode != GPAR_GRAD_RECT) return GPAR_ARG_ERROR(13);
    if (mode == GPAR_GRAD_SYM && (n2 != n1 || z2 != z1)) return GPAR_ARG_ERROR(9);
    {
        int cnt[GPAR_MAX_TERMS] = {0};
        for (int f = 0; f < ks->nfactors; ++f) {
            const int t = ks->factor[f].term;
            if (t < 0 || t >= ks->nterms || ++cnt[t] > GRAD_MAXF) return GPAR_ARG_ERROR(6);
        }
    }
    if (n1 <= 0 || dz == 0) return 0;
    if (n2 <= 0) {
        for (int r = 0; r < n1; ++r) GPAR_HIP_TRY(hipMemsetAsync(out + (size_t)r * ldo, 0, sizeof(double) * dz, (hipStream_t)stream));
        return 0;
    }
    const size_t lds = ((size_t)2 * dz * GRAM_LD + (size_t)GRAM_T * dz) * sizeof(double);
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&gram_input_grad_kernel), 160 * 1024));
    if (lds > 160 * 1024) return GPAR_ARG_ERROR(7);
    if (!input_grad_jit_launch(ks, z1, n1, ldz1, z2, n2, ldz2, dz, W, ldw, mode, nsplit, workspace, (hipStream_t)stream))
        hipLaunchKernelGGL(gram_input_grad_kernel, dim3(gpar_ceil_div(n1, GRAM_T), nsplit), dim3(256), lds, (hipStream_t)stream, *ks, z1, n1,
                           ldz1, z2, n2, ldz2, dz, W, ldw, mode, nsplit, workspace);
    const long long total = (long long)n1 * dz;
    hipLaunchKernelGGL(gram_input_grad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const double*)workspace, nsplit, n1, dz, out, ldo);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_potrf(double* A, int N, int nf, int lda, double* logdet, int* info, void* stream) {
    GPAR_API_GUARD;
    if (N <= 0 || nf <= 0) return 0;
    return potrf_run(A, N, nf, lda, logdet, info, (hipStream_t)stream);
}

int gpar_potrf_ex(double* A, int N, int nf, int lda, double* logdet, int* info, int flags, void* stream) {
    GPAR_API_GUARD;
    if (N <= 0 || nf <= 0) return 0;
    return potrf_run(A, N, nf, lda, logdet, info, (hipStream_t)stream, flags);
}

int gpar_trsm_rlt(const double* L, int n, int ldl, double* B, int nrows, int ldb, void* stream) {
    GPAR_API_GUARD;
    return trsm_rlt_run(L, n, ldl, B, nrows, ldb, (hipStream_t)stream);
}

int gpar_trsm_rlt_if(const double* L, int n, int ldl, double* B, int nrows, int ldb, const int* flag, int run_if, void* stream) {
    GPAR_API_GUARD;
    g_pred.flag = flag;
    g_pred.sense = run_if;
    const int rc = trsm_rlt_run(L, n, ldl, B, nrows, ldb, (hipStream_t)stream);
    g_pred.flag = nullptr;
    g_pred.sense = 0;
    return rc;
}

int gpar_vfe_assemble(const double* G, int M, int ldg, const double* c, const double* ys, const double* kdiag, const double* d, int n,
                      double diag_add, double* A, int lda, double* scal, double* logdet, int* info, void* stream) {
    GPAR_API_GUARD;
    if (M <= 0 || n < 0) return 0;
    if (!G || !c || !A || !scal || !logdet || !info || (n > 0 && (!ys || !kdiag || !d))) return GPAR_ARG_ERROR(1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vfe_assemble_kernel, dim3(gpar_ceil_div(M + 1, 256), M + 1), dim3(256), 0, st, G, M, ldg, c, diag_add, A, lda, logdet, info);
    hipLaunchKernelGGL(vfe_sums_kernel, dim3(VFE_PARTS), dim3(256), 0, st, ys, kdiag, d, n, G, M, ldg, scal);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_vfe_value(const double* scal, const double* logdet, const double* A, int lda, int M, int n, int with_trace, double* out, void* stream) {
    GPAR_API_GUARD;
    if (!scal || !logdet || !A || !out) return GPAR_ARG_ERROR(1);
    hipLaunchKernelGGL(vfe_value_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scal, logdet, A, lda, M, (double)n * 1.8378770664093453, with_trace, out);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_chol_spread(const double* L, int n, int ldl, double limit, double* spread, int* flag, void* stream) {
    GPAR_API_GUARD;
    if (n <= 0 || (!spread && !flag)) return 0;
    if (!L) return GPAR_ARG_ERROR(1);
    hipLaunchKernelGGL(chol_spread_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, L, n, ldl, limit, spread, flag);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_trsm_rln(const double* L, int n, int ldl, double* B, int nrows, int ldb, void* stream) {
    GPAR_API_GUARD;
    return trsm_rln_run(L, n, ldl, B, nrows, ldb, (hipStream_t)stream);
}

int gpar_chol_inverse(const double* L, int n, int ldl, double* X, int ldx, double* Kinv, int ldk, void* stream) {
    GPAR_API_GUARD;
    return chol_inverse_run(L, n, ldl, X, ldx, Kinv, ldk, (hipStream_t)stream);
}

int gpar_gemm(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B, int ldb,
              double beta, double* C, int ldc, int flags, void* stream) {
    GPAR_API_GUARD;
    return gemm_launch(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, flags, (hipStream_t)stream);
}

int gpar_gemm_batch(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, long long stride_a, const double* B,
                    int ldb, long long stride_b, double beta, double* C, int ldc, long long stride_c, int flags, int batch, void* stream) {
    GPAR_API_GUARD;
    if (batch <= 0) return 0;
    if (stride_a < 0 || stride_b < 0 || stride_c < 0) return GPAR_ARG_ERROR(9);
    return gemm_launch(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, flags, (hipStream_t)stream, 0, batch, stride_a, stride_b, stride_c);
}

int gpar_gemm_splitk(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B, int ldb,
                     double beta, double* C, int ldc, int flags, int splits, double* workspace, void* stream) {
    GPAR_API_GUARD;
    return gemm_splitk_launch(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, flags, splits, workspace, (hipStream_t)stream);
}

int gpar_dot(const double* x, int incx, const double* y, int incy, int n, double* out, int accumulate, void* stream) {
    GPAR_API_GUARD;
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long)incx, y, (long)incy, n, out, accumulate);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_gemv_t(const double* A, int rows, int cols, int lda, const double* v, double* out, double* workspace, void* stream) {
    GPAR_API_GUARD;
    if (!out || (rows > 0 && cols > 0 && (!A || !v || !workspace))) return GPAR_ARG_ERROR(2);
    return gemv_t_run(A, rows, cols, lda, v, out, workspace, (hipStream_t)stream);
}

int gpar_rownorm2(const double* A, int rows, int cols, int lda, double* out, void* stream) {
    GPAR_API_GUARD;
    if (rows <= 0) return 0;
    if (!A || !out) return GPAR_ARG_ERROR(2);
    hipLaunchKernelGGL(rownorm2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, rows, cols, lda, out);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_pack_lower(const double* A, int n, int lda, double* out, void* stream) {
    GPAR_API_GUARD;
    if (n <= 0) return 0;
    if (!A || !out) return GPAR_ARG_ERROR(2);
    hipLaunchKernelGGL(pack_lower_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, A, n, lda, out);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_unpack_lower(const double* in, int n, double* A, int lda, void* stream) {
    GPAR_API_GUARD;
    if (n <= 0) return 0;
    if (!A || !in) return GPAR_ARG_ERROR(2);
    hipLaunchKernelGGL(unpack_lower_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, in, n, A, lda);
    GPAR_LAUNCH_CHECK();
    return 0;
}

long long gpar_workspace_doubles(int op, int a, int b, int c) {
    switch (op) {
        case GPAR_WS_GEMM_SPLITK: return (long long)a * b * (c > 1 ? c : 1);           /* m, n, splits */
        case GPAR_WS_GEMV_T: return (long long)gemv_t_chunks(a) * (b > 0 ? b : 0);      /* rows, cols */
        case GPAR_WS_GRAM_GRAD: return (long long)(a > 0 ? a : 0) * GRAD_NACC;          /* nblocks */
        case GPAR_WS_CHOL_INVERSE: return (long long)a * b;                             /* n, ldx: the X matrix */
        case GPAR_WS_INPUT_GRAD: return (long long)a * b * (c > 1 ? c : 1);             /* n1, dz, nsplit */
        default: return -1;
    }
}

int gpar_randn(uint64_t seed, uint64_t offset, double* out, int rows, int cols, int ldo, void* stream) {
    GPAR_API_GUARD;
    if (rows <= 0 || cols <= 0) return 0;
    const size_t pairs = ((size_t)rows * cols + 1) / 2;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, offset, out,
                       rows, cols, ldo);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_trmv_lower(const double* L, int n, int ldl, const double* x, int incx, double* y, int incy, void* stream) {
    GPAR_API_GUARD;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(trmv_lower_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, L, n, ldl, x, incx, y, incy);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_trmv_upper(const double* U, int n, int ldu, const double* x, int incx, double* y, int incy, void* stream) {
    GPAR_API_GUARD;
    if (n <= 0) return 0;
    if (!U || !x || !y) return GPAR_ARG_ERROR(2);
    hipLaunchKernelGGL(trmv_upper_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, U, n, ldu, x, incx, y, incy);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_gemv(const double* A, int rows, int cols, int lda, const double* x, int incx, double alpha, double* y, int incy, void* stream) {
    GPAR_API_GUARD;
    if (rows <= 0) return 0;
    if (!A || !y || (cols > 0 && !x)) return GPAR_ARG_ERROR(2);
    hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, rows, cols, lda, x, incx, alpha, y, incy);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_trmv_lower_batch(const double* L, int batch, long long stride_l, int n, int ldl, const double* x, int incx, long long stride_x,
                          const double* add, int inca, long long stride_add, double* y, int incy, long long stride_y, void* stream) {
    GPAR_API_GUARD;
    if (n <= 0 || batch <= 0) return 0;
    if (!L || !x || !y) return GPAR_ARG_ERROR(2);
    hipLaunchKernelGGL(trmv_lower_batch_kernel, dim3((unsigned)((n + 3) / 4), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, L, stride_l,
                       n, ldl, x, incx, stride_x, add, inca, stride_add, y, incy, stride_y);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_sample_stats(const double* samples, int S, long long count, long long stride, int k_lo, double g_lo, int k_hi,
                      double g_hi, double* mean, double* lo, double* hi, void* stream) {
    GPAR_API_GUARD;
    if (count <= 0) return 0;
    if (S <= 0 || S > 65536 || !samples || !mean) return GPAR_ARG_ERROR(2);
    if ((lo || hi) && (k_lo < 0 || k_lo >= S || k_hi < 0 || k_hi >= S)) return GPAR_ARG_ERROR(5);
    hipLaunchKernelGGL(sample_stats_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, samples, S,
                       count, stride, k_lo, g_lo, k_hi, g_hi, mean, lo, hi);
    GPAR_LAUNCH_CHECK();
    return 0;
}

int gpar_profile_enable(int on) {
    GPAR_API_GUARD_NOSTREAM;
    g_prof.on = on != 0;
    return 0;
}

int gpar_profile_read(int* launches, double* ms, double* busy_ms, double* flops, int reset) {
    GPAR_API_GUARD_NOSTREAM;
    profile_collect();
    if (launches) *launches = g_prof.launches;
    if (ms) *ms = g_prof.ms_done;
    if (busy_ms) *busy_ms = g_prof.ms_busy;
    if (flops) *flops = g_prof.flops;
    if (reset) { g_prof.launches = 0; g_prof.ms_done = 0.0; g_prof.ms_busy = 0.0; g_prof.flops = 0.0; }
    return 0;
}

}  // extern "C"
