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
__global__ __launch_bounds__(64) void potrf_diag64_kernel(double* __restrict__ A, int lda, int cb, int col0,
                                                          double* __restrict__ logdet, int* __restrict__ info) {
    __shared__ double colbuf[2][64];
    const int i = threadIdx.x;
    double a[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) a[j] = (i < cb && j < cb && j <= i) ? A[(size_t)i * lda + j] : ((i == j) ? 1.0 : 0.0);

    double ld = 0.0;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        if (j < cb) {
            const double d = __shfl(a[j], j, 64);
            if (!(d > 0.0) && bad == 0) bad = col0 + j + 1;
            const double s = sqrt(d);
            ld += 2.0 * log(s);
            const double lij = (i == j) ? s : a[j] / s;
            a[j] = lij;
            colbuf[j & 1][i] = lij;
            __syncthreads();
#pragma unroll
            for (int k = j + 1; k < 64; ++k) a[k] = fma(-lij, colbuf[j & 1][k], a[k]);
        }
    }
    if (i < cb) {
#pragma unroll
        for (int j = 0; j < 64; ++j)
            if (j < cb && j <= i) A[(size_t)i * lda + j] = a[j];
    }
    if (i == 0) {
        if (logdet) atomicAdd(logdet, ld);
        if (bad && info) atomicCAS(info, 0, bad);
    }
}

// ---------------------------------------------------------------------------------------------------
// Strip solve against a (cb <= 64) triangular block Ld, 64 rows of B per wave (one row per lane).
//   FWD:  X Ld^T = B   (x_j = (b_j - sum_{k<j} x_k L[j][k]) / L[j][j], right-looking elimination)
//   !FWD: X Ld   = B   (x_j = (b_j - sum_{i>j} x_i L[i][j]) / L[j][j], j descending)
// LDS holds Ld^T (FWD) or Ld (!FWD) so that the coefficients needed after x_j is known are one contiguous,
// wave-uniform (broadcast) row; B is staged through a padded LDS tile so global traffic is coalesced.
template <bool FWD>
__global__ __launch_bounds__(64) void trsm_strip_kernel(const double* __restrict__ Ld, int ldl, int cb,
                                                        double* __restrict__ B, int ldb, int nrows) {
    __shared__ double Ls[64 * 65];
    __shared__ double Bt[64 * 65];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    // coefficient matrix: Ls[j][q] = FWD ? L[q][j] : L[j][q], identity outside cb
    for (int r = 0; r < 64; ++r) {
        double v = (r == lane) ? 1.0 : 0.0;
        if (r < cb && lane < cb && lane <= r) v = Ld[(size_t)r * ldl + lane];   // L[r][lane], coalesced
        if (FWD) Ls[lane * 65 + r] = v; else Ls[r * 65 + lane] = v;
    }
    for (int r = 0; r < 64; ++r) {
        double v = 0.0;
        if (row0 + r < nrows && lane < cb) v = B[(size_t)(row0 + r) * ldb + lane];
        Bt[r * 65 + lane] = v;
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
                hipLaunchKernelGGL((trsm_strip_kernel<true>), dim3(gpar_ceil_div(below, 64)), dim3(64), 0, stream,
                                   (const double*)Acc, lda, cb, A + (size_t)r0 * lda + c, lda, below);
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
        hipLaunchKernelGGL((trsm_strip_kernel<true>), dim3(gpar_ceil_div(nrows, 64)), dim3(64), 0, stream,
                           L + (size_t)c * ldl + c, ldl, cb, B + c, ldb, nrows);
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
        hipLaunchKernelGGL((trsm_strip_kernel<false>), dim3(gpar_ceil_div(nrows, 64)), dim3(64), 0, stream,
                           L + (size_t)c * ldl + c, ldl, cb, B + c, ldb, nrows);
        if (c > 0) {
            int rc = gemm_launch(0, 0, nrows, c, cb, -1.0, B + c, ldb, L + (size_t)c * ldl, ldl, 1.0, B, ldb, 0, stream);
            if (rc) return rc;
        }
    }
    GPAR_LAUNCH_CHECK();
    return 0;
}

}  // namespace gpar
