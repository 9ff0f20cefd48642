// Blocked right-looking Cholesky (partial), triangular solves.
//
// Two-level blocking: outer panels of NBO columns whose trailing update is one SYRK-shaped MFMA GEMM with
// K = NBO (arithmetic intensity NBO/8 flop per HBM byte of C traffic: 32 at NBO = 256, above the fp64
// ridge of ~12.5), inner steps of 64 columns inside a panel:
//     potrf_diag64   one wave factors the 64 x 64 diagonal block: lane i owns row i in registers, the
//                    pivot is broadcast with a wave shuffle, the scaled column goes through LDS;
//     trsm_strip     rows below the block: one row per lane, substitution against L_cc^T held in LDS;
//     gemm (NT)      rank-64 update of the rest of the panel, then rank-NBO update of the trailing matrix.
#pragma once
#include "common.h"
#include "gemm_f64.h"

namespace gpar {

constexpr int POTRF_NBI = 64;   // inner block (diag / strip width)

// ---------------------------------------------------------------------------------------------------
// 64 x 64 (or smaller, cb <= 64) diagonal block, one wave.  A points at the block's (0,0).
// Lane i owns row i in registers; step j: the pivot is broadcast with a wave shuffle (v_readlane), the
// column is scaled by the reciprocal pivot (as LAPACK's dpotf2 does: DSCAL by 1/ajj), published through LDS
// and applied as a rank-1 update.  Only the lower triangle is read or written.
__global__ __launch_bounds__(64) void potrf_diag64_kernel(double* __restrict__ A, int lda, int cb, int col0,
                                                          double* __restrict__ logdet, int* __restrict__ info) {
    __shared__ double T[64 * 65];
    __shared__ double colbuf[2][64];
    const int i = threadIdx.x;
    {
        double v[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = (r < cb && i < cb && i <= r) ? A[(size_t)r * lda + i] : ((r == i) ? 1.0 : 0.0);
#pragma unroll
        for (int r = 0; r < 64; ++r) T[r * 65 + i] = v[r];
    }
    __syncthreads();
    double a[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) a[j] = T[i * 65 + j];

    double mydiag = 1.0;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const double d = gpar_readlane_f64(a[j], j);
        if (!(d > 0.0) && bad == 0 && j < cb) bad = col0 + j + 1;
        const double s = sqrt(d);
        const double rinv = 1.0 / s;
        const double lij = (i == j) ? s : a[j] * rinv;
        if (i == j) mydiag = s;
        a[j] = lij;
        colbuf[j & 1][i] = lij;
        __syncthreads();
#pragma unroll
        for (int k = j + 1; k < 64; ++k) a[k] = fma(-lij, colbuf[j & 1][k], a[k]);
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) T[i * 65 + j] = a[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (r < cb && i <= r) A[(size_t)r * lda + i] = T[r * 65 + i];
    double ld = (i < cb) ? 2.0 * log(mydiag) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
    if (i == 0) {
        if (logdet) atomicAdd(logdet, ld);
        if (bad && info) atomicCAS(info, 0, bad);
    }
}

// ---------------------------------------------------------------------------------------------------
// Strip solve against a full 64 x 64 lower-triangular block Ld, 64 rows of B per wave, one row per lane
// held in registers.  Only rows of the lower triangle of Ld are read.
//   FWD:  X Ld^T = B   left-looking:  x_j = (b_j - sum_{k<j} x_k L[j][k]) / L_jj   (4 partial sums)
//   !FWD: X Ld   = B   right-looking, j descending: x_j = b_j / L_jj, then b_q -= x_j L[j][q] for q < j
// Ld is staged once into LDS (one coalesced round trip); every coefficient is then a wave-uniform
// (broadcast) LDS read, two per ds_read_b128.  Reciprocal pivots are computed one per lane and broadcast with
// v_readlane, so there is no division on the dependency chain.  B goes through a padded LDS tile so global
// traffic stays coalesced; all global loads of a phase are issued before the first is consumed.
template <bool FWD>
__global__ __launch_bounds__(64) void trsm_strip64_kernel(const double* __restrict__ Ld, int ldl,
                                                          double* __restrict__ B, int ldb, int nrows) {
    __shared__ __attribute__((aligned(16))) double Ls[64 * 64];
    __shared__ double Bt[64 * 65];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    {
        double v[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = (lane <= r) ? Ld[(size_t)r * ldl + lane] : 0.0;
#pragma unroll
        for (int r = 0; r < 64; ++r) Ls[r * 64 + lane] = v[r];
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = (row0 + r < nrows) ? B[(size_t)(row0 + r) * ldb + lane] : 0.0;
#pragma unroll
        for (int r = 0; r < 64; ++r) Bt[r * 65 + lane] = v[r];
    }
    __syncthreads();
    const double my_rinv = 1.0 / Ls[lane * 64 + lane];
    double a[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) a[j] = Bt[lane * 65 + j];
    if (FWD) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            double s0 = a[j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int k = 0; k < j; ++k) {
                const double c = Ls[j * 64 + k];
                if ((k & 3) == 0) s0 = fma(-a[k], c, s0);
                else if ((k & 3) == 1) s1 = fma(-a[k], c, s1);
                else if ((k & 3) == 2) s2 = fma(-a[k], c, s2);
                else s3 = fma(-a[k], c, s3);
            }
            a[j] = ((s0 + s1) + (s2 + s3)) * gpar_readlane_f64(my_rinv, j);
        }
    } else {
#pragma unroll
        for (int j = 63; j >= 0; --j) {
            const double x = a[j] * gpar_readlane_f64(my_rinv, j);
            a[j] = x;
#pragma unroll
            for (int q = 0; q < j; ++q) a[q] = fma(-x, Ls[j * 64 + q], a[q]);
        }
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) Bt[lane * 65 + j] = a[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (row0 + r < nrows) B[(size_t)(row0 + r) * ldb + lane] = Bt[r * 65 + lane];
}

// Generic (cb < 64) strip solve: coefficients staged in LDS with identity padding.  Only the ragged last
// block of a matrix takes this path.
template <bool FWD>
__global__ __launch_bounds__(64) void trsm_strip_kernel(const double* __restrict__ Ld, int ldl, int cb,
                                                        double* __restrict__ B, int ldb, int nrows) {
    __shared__ double Ls[64 * 65];
    __shared__ double Bt[64 * 65];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    // coefficient matrix: Ls[j][q] = FWD ? L[q][j] : L[j][q], identity outside cb
    {
        double v[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            v[r] = (r == lane) ? 1.0 : 0.0;
            if (r < cb && lane < cb && lane <= r) v[r] = Ld[(size_t)r * ldl + lane];   // L[r][lane], coalesced
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (FWD) Ls[lane * 65 + r] = v[r]; else Ls[r * 65 + lane] = v[r];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = (row0 + r < nrows && lane < cb) ? B[(size_t)(row0 + r) * ldb + lane] : 0.0;
#pragma unroll
        for (int r = 0; r < 64; ++r) Bt[r * 65 + lane] = v[r];
    }
    __syncthreads();
    double a[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) a[j] = Bt[lane * 65 + j];
    if (FWD) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const double x = a[j] / Ls[j * 65 + j];
            a[j] = x;
#pragma unroll
            for (int q = j + 1; q < 64; ++q) a[q] = fma(-x, Ls[j * 65 + q], a[q]);
        }
    } else {
#pragma unroll
        for (int j = 63; j >= 0; --j) {
            const double x = a[j] / Ls[j * 65 + j];
            a[j] = x;
#pragma unroll
            for (int q = 0; q < j; ++q) a[q] = fma(-x, Ls[j * 65 + q], a[q]);
        }
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) Bt[lane * 65 + j] = a[j];
    __syncthreads();
    for (int r = 0; r < 64; ++r)
        if (row0 + r < nrows && lane < cb) B[(size_t)(row0 + r) * ldb + lane] = Bt[r * 65 + lane];
}

template <bool FWD>
static void launch_strip(const double* Ld, int ldl, int cb, double* B, int ldb, int nrows, hipStream_t stream) {
    if (nrows <= 0) return;
    const dim3 grid(gpar_ceil_div(nrows, 64)), block(64);
    if (cb == 64) hipLaunchKernelGGL((trsm_strip64_kernel<FWD>), grid, block, 0, stream, Ld, ldl, B, ldb, nrows);
    else hipLaunchKernelGGL((trsm_strip_kernel<FWD>), grid, block, 0, stream, Ld, ldl, cb, B, ldb, nrows);
}

// ---------------------------------------------------------------------------------------------------
struct ProfileState {
    bool on = false;
    int launches = 0;
    double flops = 0.0;
    double ms_done = 0.0;
    static constexpr int MAXEV = 4096;
    hipEvent_t ev[MAXEV][2];
    int nev = 0;
    bool created = false;
};
static ProfileState g_prof;

static void profile_collect() {
    for (int i = 0; i < g_prof.nev; ++i) {
        float ms = 0.f;
        hipEventSynchronize(g_prof.ev[i][1]);
        hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]);
        g_prof.ms_done += ms;
    }
    g_prof.nev = 0;
}

static int potrf_outer_block(int N) {
    // measured trade-off: wider panels raise SYRK intensity, narrower panels shorten the serial panel chain
    if (N >= 6144) return 256;
    if (N >= 1536) return 128;
    return 64;
}

static int potrf_run(double* A, int N, int nf, int lda, double* logdet, int* info, hipStream_t stream) {
    if (nf > N) return GPAR_ARG_ERROR(1);
    const int NBO = potrf_outer_block(N);
    for (int k0 = 0; k0 < nf; k0 += NBO) {
        const int kb = (nf - k0 < NBO) ? nf - k0 : NBO;
        const int kend = k0 + kb;
        for (int c = k0; c < kend; c += POTRF_NBI) {
            const int cb = (kend - c < POTRF_NBI) ? kend - c : POTRF_NBI;
            double* Acc = A + (size_t)c * lda + c;
            hipLaunchKernelGGL(potrf_diag64_kernel, dim3(1), dim3(64), 0, stream, Acc, lda, cb, c, logdet, info);
            const int r0 = c + cb;
            const int below = N - r0;
            if (below > 0) {
                launch_strip<true>(Acc, lda, cb, A + (size_t)r0 * lda + c, lda, below, stream);
                const int ncols = kend - r0;   // remaining columns of this panel
                if (ncols > 0) {
                    const double* P = A + (size_t)r0 * lda + c;
                    int rc = gemm_launch(0, 1, below, ncols, cb, -1.0, P, lda, P, lda, 1.0,
                                         A + (size_t)r0 * lda + r0, lda, GPAR_GEMM_C_LOWER, stream);
                    if (rc) return rc;
                }
            }
        }
        const int rem = N - kend;
        if (rem > 0) {
            const double* P = A + (size_t)kend * lda + k0;
            const bool prof = g_prof.on && g_prof.nev < ProfileState::MAXEV;
            if (prof) {
                if (!g_prof.created) {
                    for (int i = 0; i < ProfileState::MAXEV; ++i) { hipEventCreate(&g_prof.ev[i][0]); hipEventCreate(&g_prof.ev[i][1]); }
                    g_prof.created = true;
                }
                hipEventRecord(g_prof.ev[g_prof.nev][0], stream);
            }
            int rc = gemm_launch(0, 1, rem, rem, kb, -1.0, P, lda, P, lda, 1.0, A + (size_t)kend * lda + kend, lda,
                                 GPAR_GEMM_C_LOWER, stream);
            if (rc) return rc;
            if (prof) {
                hipEventRecord(g_prof.ev[g_prof.nev][1], stream);
                g_prof.nev++;
                g_prof.launches++;
                // algorithmic flops of the lower-triangular rank-kb update
                g_prof.flops += (double)rem * ((double)rem + 1.0) * (double)kb;
            }
        }
    }
    GPAR_LAUNCH_CHECK();
    return 0;
}

// B <- B L^-T : forward over column blocks.  B[:, c:c+cb] solved against L_cc, then
// B[:, c+cb:] -= X_c L[c+cb:, c:c+cb]^T  (NT GEMM, K = cb)
static int trsm_rlt_run(const double* L, int n, int ldl, double* B, int nrows, int ldb, hipStream_t stream) {
    if (nrows <= 0) return 0;
    for (int c = 0; c < n; c += POTRF_NBI) {
        const int cb = (n - c < POTRF_NBI) ? n - c : POTRF_NBI;
        launch_strip<true>(L + (size_t)c * ldl + c, ldl, cb, B + c, ldb, nrows, stream);
        const int rest = n - (c + cb);
        if (rest > 0) {
            int rc = gemm_launch(0, 1, nrows, rest, cb, -1.0, B + c, ldb, L + (size_t)(c + cb) * ldl + c, ldl, 1.0,
                                 B + c + cb, ldb, 0, stream);
            if (rc) return rc;
        }
    }
    GPAR_LAUNCH_CHECK();
    return 0;
}

// B <- B L^-1 : backward over column blocks.  B[:, c:c+cb] solved against L_cc, then
// B[:, :c] -= X_c L[c:c+cb, :c]  (NN GEMM, K = cb)
static int trsm_rln_run(const double* L, int n, int ldl, double* B, int nrows, int ldb, hipStream_t stream) {
    if (nrows <= 0) return 0;
    const int nblk = gpar_ceil_div(n, POTRF_NBI);
    for (int b = nblk - 1; b >= 0; --b) {
        const int c = b * POTRF_NBI;
        const int cb = (n - c < POTRF_NBI) ? n - c : POTRF_NBI;
        launch_strip<false>(L + (size_t)c * ldl + c, ldl, cb, B + c, ldb, nrows, stream);
        if (c > 0) {
            int rc = gemm_launch(0, 0, nrows, c, cb, -1.0, B + c, ldb, L + (size_t)c * ldl, ldl, 1.0, B, ldb, 0, stream);
            if (rc) return rc;
        }
    }
    GPAR_LAUNCH_CHECK();
    return 0;
}

}  // namespace gpar
