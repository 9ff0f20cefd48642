// Blocked right-looking Cholesky (partial), triangular solves.
//
// Two-level blocking: outer panels of NBO = 512 columns, each factored by ONE persistent fused kernel (panel.h), whose
// trailing update is a SYRK-shaped MFMA GEMM with K = NBO (K = 2 NBO while panels are paired: potrf_run); the next
// panel runs under that update on a side stream (look-ahead).  The leaf kernels below serve ragged tails, unaligned
// inputs and the GPAR_POTRF_FUSED=0 fallback - inner steps of 64 columns inside a panel:
//     potrf_diag64   one wave factors the 64 x 64 diagonal block: lane i owns row i in registers, the
//                    pivot is broadcast with a wave shuffle, the scaled column goes through LDS;
//     trsm_strip     rows below the block: one row per lane, substitution against L_cc^T held in LDS;
//     gemm (NT)      rank-64 update of the rest of the panel, then rank-NBO update of the trailing matrix.
#pragma once
#include "common.h"
#include "gemm_f64.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <utility>
#include <vector>

namespace gpar {

constexpr int POTRF_NBI = 64;   // inner block (diag / strip width)

// ---------------------------------------------------------------------------------------------------
// Panel kernels.  Both are written as SHORT LOOPS over 8-wide sub-blocks with the matrix resident in LDS and
// only an 8-element register window per lane: a fully unrolled 64-step formulation is ~30 KB of straight-line
// code and ran at instruction-fetch speed (37 us per launch, profiles/r01_potrf_v1_kernels.txt); the looped form
// re-executes ~1.5 KB bodies from the instruction cache.
constexpr int PAN_LD = 66;   // LDS row pitch (doubles): even -> 16-byte aligned rows for ds_read_b128

typedef double pan_d2 __attribute__((ext_vector_type(2)));
typedef double pan_d4 __attribute__((ext_vector_type(4)));

// 64 x 64 (or smaller, cb <= 64) diagonal block, one wave; lane i owns row i.  Blocked left-looking Cholesky:
// for each 8-column block, (1) subtract the contribution of all previous columns (own row entries: lane-private
// LDS reads; the other row: wave-uniform broadcast reads), (2) factor the 8 columns with pivots and column
// entries broadcast by v_readlane, scaling by the reciprocal pivot as LAPACK's dpotf2 does.
// (blockIdx.x selects matrix b of a lock-step batch: A + b * batch_a, logdet[b], info[b])
__global__ __launch_bounds__(64) void potrf_diag64_kernel(double* __restrict__ A, int lda, int cb, int col0,
                                                          double* __restrict__ logdet, int* __restrict__ info, long long batch_a = 0) {
    __shared__ __attribute__((aligned(16))) double T[64 * PAN_LD];
    const int i = threadIdx.x;
    A += (size_t)blockIdx.x * batch_a;
    if (logdet) logdet += blockIdx.x;
    if (info) info += blockIdx.x;
    {
        double v[64];
#pragma unroll
        // unconditional loads from clamped (always valid) addresses, selected afterwards: a guarded load would put
        // every one of the 64 loads behind its own branch + vmcnt(0) wait (64 serial memory round trips)
        for (int r = 0; r < 64; ++r) {
            const int rc = r < cb ? r : cb - 1;
            v[r] = A[(size_t)rc * lda + (i < rc ? i : rc)];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) T[r * PAN_LD + i] = (r < cb && i <= r) ? v[r] : ((r == i) ? 1.0 : 0.0);
    }
    __syncthreads();
    double mydiag = 1.0;
    int bad = 0;
    for (int jb = 0; jb < 8; ++jb) {
        double acc[8];
        {
            const pan_d2* src = reinterpret_cast<const pan_d2*>(&T[i * PAN_LD + 8 * jb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const pan_d2 t = src[q]; acc[2 * q] = t[0]; acc[2 * q + 1] = t[1]; }
        }
        for (int kb = 0; kb < jb; ++kb) {
            double mine[8];
            const pan_d2* ms = reinterpret_cast<const pan_d2*>(&T[i * PAN_LD + 8 * kb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const pan_d2 t = ms[q]; mine[2 * q] = t[0]; mine[2 * q + 1] = t[1]; }
            // all 32 broadcast reads are issued before the first FMA, and the FMAs run k-outer so the 8
            // accumulators form independent chains (a j-outer order serialises read -> wait -> 8 dependent FMAs)
            pan_d2 c[8][4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const pan_d2* cs = reinterpret_cast<const pan_d2*>(&T[(8 * jb + j) * PAN_LD + 8 * kb]);
#pragma unroll
                for (int q = 0; q < 4; ++q) c[j][q] = cs[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fma(-mine[2 * q], c[j][q][0], acc[j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fma(-mine[2 * q + 1], c[j][q][1], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = 8 * jb + j;
            const double d = gpar_readlane_f64(acc[j], col);
            if (!(d > 0.0) && bad == 0 && col < cb) bad = col0 + col + 1;
            const double sd = sqrt(d);
            const double rinv = 1.0 / sd;
            const double lij = (i == col) ? sd : acc[j] * rinv;
            if (i == col) mydiag = sd;
            acc[j] = lij;
#pragma unroll
            for (int j2 = j + 1; j2 < 8; ++j2) acc[j2] = fma(-lij, gpar_readlane_f64(lij, 8 * jb + j2), acc[j2]);
        }
        {
            pan_d2* dst = reinterpret_cast<pan_d2*>(&T[i * PAN_LD + 8 * jb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = pan_d2{acc[2 * q], acc[2 * q + 1]};
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (r < cb && i <= r) A[(size_t)r * lda + i] = T[r * PAN_LD + i];
    double ld = (i < cb) ? 2.0 * log(mydiag) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
    if (i == 0) {
        if (logdet) atomicAdd(logdet, ld);
        if (bad && info) atomicCAS(info, 0, bad);
    }
}

// ---------------------------------------------------------------------------------------------------
// Strip solve against a (cb <= 64) lower-triangular block Ld, 64 rows of B per wave, one row per lane.
//   FWD:  X Ld^T = B :  x_j = (b_j - sum_{k<j} x_k L[j][k]) / L_jj
//   !FWD: X Ld   = B :  x_j = (b_j - sum_{i>j} x_i L[i][j]) / L_jj   (j descending)
// The backward case is run as a forward solve on the index-reversed system (a -> 63 - a), i.e. its coefficient
// image in LDS is the reversed transpose of Ld, so one loop nest serves both.  Only the lower triangle of Ld is
// read.  Blocked left-looking substitution, 8 columns at a time: previously solved entries of the lane's row come
// back from LDS (lane-private), coefficients are wave-uniform broadcast reads (ds_read_b128), reciprocal pivots
// are precomputed (no division on the dependency chain).  B is staged through LDS so global traffic is coalesced,
// and all global loads of a phase are issued before the first is consumed.
template <bool FWD>
__global__ __launch_bounds__(64) void trsm_strip_kernel(const double* __restrict__ Ld, int ldl, int cb,
                                                        double* __restrict__ B, int ldb, int nrows, long long batch_stride = 0,
                                                        const int* pred = nullptr, int pred_sense = 0) {
    __shared__ __attribute__((aligned(16))) double Cs[64 * PAN_LD];   // Cs[j][k]: coefficient of x_k in equation j
    __shared__ __attribute__((aligned(16))) double Xs[64 * PAN_LD];   // Xs[lane][a]: row `lane` of B / X (reversed if !FWD)
    __shared__ double rinvs[64];
    if (gpar_pred_skip(pred, pred_sense)) return;   // (predicated solve, common.h)
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    Ld += (size_t)blockIdx.y * batch_stride;   // (lock-step batch: triangle and right-hand sides live in the same matrix b)
    B += (size_t)blockIdx.y * batch_stride;
    {
        // both 64 x 64 tiles are requested before either is consumed (one memory round trip instead of two);
        // loads are unconditional from clamped, always valid addresses (see potrf_diag64_kernel)
        double v[64], u[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int rc = r < cb ? r : cb - 1;
            v[r] = Ld[(size_t)rc * ldl + (lane < rc ? lane : rc)];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int rr = (row0 + r < nrows) ? row0 + r : nrows - 1;
            u[r] = B[(size_t)rr * ldb + (lane < cb ? lane : cb - 1)];
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r] = (r < cb && lane <= r) ? v[r] : ((r == lane) ? 1.0 : 0.0);
        if (FWD) {
#pragma unroll
            for (int r = 0; r < 64; ++r) Cs[r * PAN_LD + lane] = v[r];
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r) Cs[(63 - lane) * PAN_LD + (63 - r)] = v[r];   // Cs[a][b] = L[63-b][63-a]
        }
#pragma unroll
        for (int r = 0; r < 64; ++r) Xs[r * PAN_LD + (FWD ? lane : 63 - lane)] = (row0 + r < nrows && lane < cb) ? u[r] : 0.0;
    }
    __syncthreads();
    rinvs[lane] = 1.0 / Cs[lane * PAN_LD + lane];
    __syncthreads();

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
#pragma unroll
    for (int r = 0; r < 64; ++r)
        if (row0 + r < nrows && lane < cb) B[(size_t)(row0 + r) * ldb + lane] = Xs[r * PAN_LD + (FWD ? lane : 63 - lane)];
}

template <bool FWD>
static void launch_strip(const double* Ld, int ldl, int cb, double* B, int ldb, int nrows, hipStream_t stream, int batch = 1, long long batch_stride = 0) {
    if (nrows <= 0) return;
    hipLaunchKernelGGL((trsm_strip_kernel<FWD>), dim3(gpar_ceil_div(nrows, 64), batch), dim3(64), 0, stream, Ld, ldl, cb, B, ldb, nrows, batch_stride,
                       g_pred.flag, g_pred.sense);
}

// ---------------------------------------------------------------------------------------------------
struct ProfileState {
    bool on = false;
    int launches = 0;
    double flops = 0.0;
    double ms_done = 0.0;    // sum of the launches' own durations
    double ms_busy = 0.0;    // union of their intervals (launches issued from different streams may overlap)
    static constexpr int MAXEV = 8192;
    hipEvent_t ev[MAXEV][2];
    int shape[MAXEV][3];     // rows, cols, kb (x batch) of the update behind event pair i (GPAR_PROFILE_DUMP)
    int nev = 0;
    bool created = false;
};
static ProfileState g_prof;

static void profile_collect() {
    if (g_prof.nev == 0) return;
    std::vector<std::pair<float, float>> iv(g_prof.nev);
    GPAR_HIP_IGNORE(hipEventSynchronize(g_prof.ev[0][0]));
    for (int i = 0; i < g_prof.nev; ++i) {
        float t0 = 0.f, t1 = 0.f;   // a failed query leaves zeros: a measurement aid, never a result
        GPAR_HIP_IGNORE(hipEventSynchronize(g_prof.ev[i][1]));
        GPAR_HIP_IGNORE(hipEventElapsedTime(&t0, g_prof.ev[0][0], g_prof.ev[i][0]));
        GPAR_HIP_IGNORE(hipEventElapsedTime(&t1, g_prof.ev[0][0], g_prof.ev[i][1]));
        g_prof.ms_done += t1 - t0;
        iv[i] = {t0, t1};
    }
    if (const char* path = getenv("GPAR_PROFILE_DUMP")) {
        // development aid (tools/r06_launch_table.py): one line per counted update launch - start and end in ms since the first
        // one, rows, cols, K x batch - appended to the named file
        if (FILE* f = fopen(path, "a")) {
            for (int i = 0; i < g_prof.nev; ++i)
                fprintf(f, "%.6f %.6f %d %d %d\n", iv[i].first, iv[i].second, g_prof.shape[i][0], g_prof.shape[i][1], g_prof.shape[i][2]);
            fprintf(f, "#\n");
            fclose(f);
        }
    }
    std::sort(iv.begin(), iv.end());
    float cs = iv[0].first, ce = iv[0].second;
    for (size_t i = 1; i < iv.size(); ++i) {
        if (iv[i].first > ce) { g_prof.ms_busy += ce - cs; cs = iv[i].first; ce = iv[i].second; }
        else if (iv[i].second > ce) ce = iv[i].second;
    }
    g_prof.ms_busy += ce - cs;
    g_prof.nev = 0;
}

// Conditioning estimate of a Cholesky factor that comes for free: the spread of its pivots, (max L_jj / min L_jj)^2 - a lower
// bound of cond(L L^T).  spread[0] <- the estimate, flag[0] <- 1 if it exceeds `limit` or a pivot is not positive (a failed
// factorisation), 0 otherwise: the word a predicated solve (gpar_trsm_rlt_if) dispatches on.
__global__ __launch_bounds__(256) void chol_spread_kernel(const double* __restrict__ L, int n, int ldl, double limit, double* __restrict__ spread,
                                                          int* __restrict__ flag) {
    __shared__ double smax[256], smin[256];
    double mx = 0.0, mn = __builtin_inf();
    bool bad = false;
    for (int j = threadIdx.x; j < n; j += 256) {
        const double v = L[(size_t)j * ldl + j];
        if (!(v > 0.0)) bad = true;   // (also NaN)
        mx = fmax(mx, v);
        mn = fmin(mn, v);
    }
    smax[threadIdx.x] = mx;
    smin[threadIdx.x] = bad ? -1.0 : mn;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + off]);
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double lo = smin[0], hi = smax[0];
        const bool ok = lo > 0.0;
        const double r = ok ? hi / lo : __builtin_inf();
        const double s = r * r;
        if (spread) spread[0] = s;
        if (flag) flag[0] = (ok && s <= limit) ? 0 : 1;
    }
}

// ---- blocking policy (overridable for experiments: GPAR_POTRF_NBO / GPAR_POTRF_NBM / GPAR_POTRF_LOOKAHEAD) ----------
struct PotrfPolicy {
    int nbo;        // top-level panel width: K of the trailing SYRK
    int nbm;        // mid-level width inside a panel
    int lookahead;  // overlap panel k+1 with the trailing update of panel k on a second stream
    int split;      // factor the diagonal block first, then solve the rows below (see potrf_panel_split)
    int fused;      // factor each top-level panel with the persistent fused kernel (panel.h)
    int pair_rows;  // panels with at least this many rows left are factored in groups (one rank-group*nbo trailing update)
    int group;      // panels per group
};

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static PotrfPolicy potrf_policy(int N) {
    PotrfPolicy p;
    // wider top-level panels amortise the read-modify-write of the trailing matrix over more flops (measured SYRK
    // rate at n = 16384: K = 128 31, K = 256 42, K = 512 53 TFLOP/s); the panel itself is factored recursively
    // with the fused panel kernel (panel.h) the panel is cheap, so the widest panel it supports wins at every size
    // (measured n = 1024 .. 16384, profiles/r01_potrf_nbo_sweep.txt); the unfused fallback prefers narrower ones
    p.fused = env_int("GPAR_POTRF_FUSED", 1);
    if (p.fused) p.nbo = 512;
    else if (N >= 12288) p.nbo = 512;
    else if (N >= 6144) p.nbo = 256;
    else if (N >= 1536) p.nbo = 128;
    else p.nbo = 64;
    p.nbm = p.nbo >= 512 ? 128 : 64;
    // Look-ahead: the fused panel kernel of panel k+1 (70.8 KB LDS: fits on a CU beside one SYRK workgroup) runs on
    // the caller's stream under the trailing update of panel k on a low-priority side stream.  Pays once the
    // trailing updates are long enough to hide it (measured with half-tile workgroups for the small launches: n = 5120
    // 3.12 -> 2.97 ms, 6144 4.06 -> 3.81, 8192 6.75 -> 5.98; a wash at 4096, a loss at 3072).  The unfused fallback keeps
    // it off (its small kernels starve behind the SYRK).
    // (round 4, with the small look-ahead update kernel and two panels per launch: n = 3072 1.152 -> 1.097 ms, 4096 1.767 -> 1.631, 4600
    // 2.248 -> 1.96, a wash at 2048: profiles/r04_exp_potrf_fuse2.txt)
    p.lookahead = (p.fused && N >= 2560) ? 1 : 0;
    p.nbo = env_int("GPAR_POTRF_NBO", p.nbo);
    p.nbm = env_int("GPAR_POTRF_NBM", p.nbm);
    p.lookahead = env_int("GPAR_POTRF_LOOKAHEAD", p.lookahead);
    p.split = env_int("GPAR_POTRF_SPLIT", 0);
    // n = 8192 measured slower grouped (5.38 vs 5.19 ms alone).  From N = 12288 on panels stay grouped until 6144 rows are left:
    // with two factorisations in flight (the pipelined C3 evaluation) the longer serial stretch hides under the other stream's
    // updates, 197.9 -> 194.9 ms per evaluation; alone it costs 0.5 % at n = 16384 (tools/exp_pair_rows.sh).  The rule depends
    // on the size only, so that a factorisation returns the same bits whatever runs beside it.
    p.pair_rows = env_int("GPAR_POTRF_PAIR_ROWS", N >= 12288 ? 6144 : 9216);
    p.group = env_int("GPAR_POTRF_GROUP", 3);
    if (p.nbo < 64) p.nbo = 64;
    if (p.nbm < 64) p.nbm = 64;
    return p;
}

struct PotrfCtx {
    double* A;
    int N, lda;
    double* logdet;
    int* info;
    int nbm;
    int batch = 1;           // lock-step factorisation of `batch` matrices (matrix b at A + b * batch_a): the trailing updates
    long long batch_a = 0;   // are batched launches
};

// The trailing update when only a few rows are left - the augmented rows under the last panel: [y^T, 0] of the log marginal
// likelihood, whose corner becomes -|L^-1 y|^2 - one wave per stored element (i, j), j <= i: lanes stride K, butterfly sum.
// A 64 x 64 tile of the GEMM kernel for ONE element is all latency (42 us per evaluation at every size; this: ~4 us).
constexpr int POTRF_SMALL_ROWS = 16;
// (org: first row / column of the updated block; the K columns [k0, k0 + K) of its rows are the operand)
__global__ __launch_bounds__(64) void potrf_small_update_kernel(double* __restrict__ A, int lda, int org, int k0, int K, int rows, int cols, long long batch_a) {
    const int i = blockIdx.x / cols, j = blockIdx.x - i * cols;
    if (j > i) return;
    A += (size_t)blockIdx.y * batch_a;
    const double* Pi = A + (size_t)(org + i) * lda + k0;
    const double* Pj = A + (size_t)(org + j) * lda + k0;
    double a0 = 0.0, a1 = 0.0;
    int k = threadIdx.x;
    for (; k + 64 < K; k += 128) {
        a0 = fma(Pi[k], Pj[k], a0);
        a1 = fma(Pi[k + 64], Pj[k + 64], a1);
    }
    if (k < K) a0 = fma(Pi[k], Pj[k], a0);
    double s = a0 + a1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) A[(size_t)(org + i) * lda + org + j] -= s;
}

static int potrf_gemm_update(const PotrfCtx& c, int k0, int kend, int col_end, hipStream_t stream, int role = 0) {
    // A[kend:N, kend:col_end] -= A[kend:N, k0:kend] A[kend:col_end, k0:kend]^T   (lower part only)
    const int rows = c.N - kend, cols = col_end - kend;
    if (rows <= 0 || cols <= 0) return 0;
    if (rows <= POTRF_SMALL_ROWS && env_int("GPAR_POTRF_SMALL_UPDATE", 1)) {
        hipLaunchKernelGGL(potrf_small_update_kernel, dim3(rows * cols, c.batch), dim3(64), 0, stream, c.A, c.lda, kend, k0, kend - k0, rows, cols, c.batch_a);
        return 0;
    }
    const double* P = c.A + (size_t)kend * c.lda + k0;
    return gemm_launch(0, 1, rows, cols, kend - k0, -1.0, P, c.lda, P, c.lda, 1.0, c.A + (size_t)kend * c.lda + kend, c.lda,
                       GPAR_GEMM_C_LOWER, stream, role, c.batch, c.batch_a, c.batch_a, c.batch_a);
}

static int potrf_la_update_small(double* A, int N, int lda, int k0, int kend, int ncols, hipStream_t stream, int batch, long long batch_a);   // panel2.h

// Does the update of the NEXT panel's columns [kend, la_end) take the one-tile-per-workgroup kernel (panel2.h) instead of the GEMM?
// While it has at most GPAR_POTRF_LA_SMALL_TILES 64 x 64 tiles over the whole batch: then its duration is one tile's, and the
// small kernel's tile is several times shorter.  The rule looks at the geometry only - never at the look-ahead setting - so that a
// factorisation returns the same bits with and without look-ahead (potrf_run splits its single update accordingly).
static bool potrf_la_is_small(const PotrfCtx& c, int k0, int kend, int la_end) {
    const int cols = la_end - kend, K = kend - k0;
    if (cols <= 0 || cols % 64 != 0 || K <= 0 || K % 64 != 0 || (c.lda & 1) || !gpar_aligned16(c.A) || (c.batch_a & 1) || (k0 & 1) || c.N - kend <= POTRF_SMALL_ROWS)
        return false;
    const int nc = cols / 64, tr = (c.N - kend + 63) / 64;
    if (tr < nc) return false;
    const long long tiles = ((long long)nc * (nc + 1) / 2 + (long long)(tr - nc) * nc) * c.batch;
    // behind a fused pair of panels (K >= 1024, potrf_group_kernel) a tile is 16 chunks and the alternative a K = 1024 GEMM tile of
    // ~200 us: several rounds of small tiles still win
    if (K >= 1024) return tiles <= env_int("GPAR_POTRF_LA_SMALL_TILES2", 2048);
    return tiles <= env_int("GPAR_POTRF_LA_SMALL_TILES", 512);
}

// The update of the next panel's columns by whichever kernel the rule picks.
static int potrf_la_update(const PotrfCtx& c, int k0, int kend, int la_end, hipStream_t stream) {
    if (potrf_la_is_small(c, k0, kend, la_end)) return potrf_la_update_small(c.A, c.N, c.lda, k0, kend, la_end - kend, stream, c.batch, c.batch_a);
    return potrf_gemm_update(c, k0, kend, la_end, stream, 1);
}

// Everything to the right of the next step's columns: A[from:N, from:N] -= P P^T (lower), P = A[from:N, k0:kend).  A handful of rows
// (the augmented row once the last panel is next) by the one-wave kernel, else the batched GEMM.
static int potrf_rest_update(const PotrfCtx& c, int k0, int kend, int from, hipStream_t stream) {
    const int rows = c.N - from;
    if (rows <= 0) return 0;
    if (rows <= POTRF_SMALL_ROWS && env_int("GPAR_POTRF_SMALL_UPDATE", 1)) {
        hipLaunchKernelGGL(potrf_small_update_kernel, dim3(rows * rows, c.batch), dim3(64), 0, stream, c.A, c.lda, from, k0, kend - k0, rows, rows, c.batch_a);
        return 0;
    }
    const double* P = c.A + (size_t)from * c.lda + k0;
    return gemm_launch(0, 1, rows, rows, kend - k0, -1.0, P, c.lda, P, c.lda, 1.0, c.A + (size_t)from * c.lda + from, c.lda, GPAR_GEMM_C_LOWER, stream, 1,
                       c.batch, c.batch_a, c.batch_a, c.batch_a);
}

// Factor columns [c0, c1) (already up to date with respect to all columns < c0): on exit rows c0..N of those
// columns hold L.  Recursive right-looking restricted to the panel: widths nb(level) -> ... -> 64.
static int potrf_panel(const PotrfCtx& c, int c0, int c1, int nb, hipStream_t stream) {
    const int w = c1 - c0;
    if (w <= POTRF_NBI) {
        double* Acc = c.A + (size_t)c0 * c.lda + c0;
        // (a lock-step batch: one launch for all its matrices - a ragged last panel used to cost `batch` serial single-wave launches,
        // 11 % of a predict with 50 samples at n* = 1000)
        hipLaunchKernelGGL(potrf_diag64_kernel, dim3(c.batch), dim3(64), 0, stream, Acc, c.lda, w, c0, c.logdet, c.info, c.batch_a);
        launch_strip<true>(Acc, c.lda, w, c.A + (size_t)c1 * c.lda + c0, c.lda, c.N - c1, stream, c.batch, c.batch_a);
        return 0;
    }
    if (nb >= w) nb = (w > c.nbm && c.nbm >= POTRF_NBI) ? c.nbm : POTRF_NBI;
    const int next_nb = nb > c.nbm ? c.nbm : POTRF_NBI;
    for (int k0 = c0; k0 < c1; k0 += nb) {
        const int kend = (k0 + nb < c1) ? k0 + nb : c1;
        int rc = potrf_panel(c, k0, kend, next_nb, stream);
        if (rc) return rc;
        rc = potrf_gemm_update(c, k0, kend, c1, stream);
        if (rc) return rc;
    }
    return 0;
}

static int trsm_rlt_run(const double* L, int n, int ldl, double* B, int nrows, int ldb, hipStream_t stream);
static int trsm_rlt_run2(const double* L, int n, int ldl, double* B, int nrows, int ldb, int upper_tri, hipStream_t stream);

// Panel = (a) the w x w diagonal block, factored recursively with kernels that only span the block's own rows
// (1-4 workgroups each: they slip in beside a running trailing update instead of queueing for 256 CU slots), then
// (b) the rows below, X = A21 L11^-T, as a blocked forward substitution over all rows (bulk work).
static int potrf_panel_split(const PotrfCtx& c, int k0, int kend, int nb, hipStream_t stream) {
    PotrfCtx blk = c;
    blk.N = kend;
    int rc = potrf_panel(blk, k0, kend, nb, stream);
    if (rc) return rc;
    if (c.N > kend)
        rc = trsm_rlt_run(c.A + (size_t)k0 * c.lda + k0, kend - k0, c.lda, c.A + (size_t)kend * c.lda + k0, c.N - kend, c.lda, stream);
    return rc;
}

// defined in panel.h (one launch per 64-row block of B for up to 8 column steps)
// defined in panel.h (one persistent launch per panel)
static void potrf_zero_flags(double* A, int N, int lda, hipStream_t stream, int batch, long long batch_a);
// the fused panel kernels (panel2.h): left-looking row-block tasks, strips on the matrix cores
static int potrf_panel_fused2(double* A, int N, int lda, int k0, int W, double* logdet, int* info, hipStream_t stream, bool prezeroed,
                              int batch = 1, long long batch_a = 0);
static int trsm_block_fused2(const double* L, int n, int ldl, double* B, int nrows, int ldb, int c0, int S, int upper_tri,
                             hipStream_t stream);
static int potrf_la_update_small(double* A, int N, int lda, int k0, int kend, int ncols, hipStream_t stream, int batch, long long batch_a);
static int potrf_group_fused(double* A, int N, int lda, int k0, int W, int G, double* logdet, int* info, hipStream_t stream, int batch,
                             long long batch_a, unsigned long long la_base);
static int env_int(const char* name, int dflt);
static int trinv_blocks_fused2(const double* L, int n, int ldl, double* X, int ldx, int S, hipStream_t stream);
static int trsm_block_back_fused2(const double* L, int n, int ldl, double* B, int nrows, int ldb, int c0, int S, hipStream_t stream);

static inline int potrf_panel_any(double* A, int N, int lda, int k0, int W, double* logdet, int* info, hipStream_t stream, bool prezeroed,
                                  int batch = 1, long long batch_a = 0) {
    // (a chain-only launch followed by the solve block kernel for the rows below - GPAR_POTRF_PANEL_SPLIT_ROWS, round 2 - was slower at
    // every size and is retired: NOTES.md section 7, lesson 34)
    return potrf_panel_fused2(A, N, lda, k0, W, logdet, info, stream, prezeroed, batch, batch_a);
}
static inline int trsm_block_any(const double* L, int n, int ldl, double* B, int nrows, int ldb, int c0, int S, int upper_tri,
                                 hipStream_t stream) {
    return trsm_block_fused2(L, n, ldl, B, nrows, ldb, c0, S, upper_tri, stream);
}

struct LookaheadState {
    // one low-priority side stream per caller stream (callers that pipeline independent layers over two or three
    // streams must not have their trailing updates queued behind one another on a shared side stream)
    static constexpr int MAXS = 8;
    hipStream_t caller[MAXS];
    hipStream_t side[MAXS];
    int nside = 0;
    int next_victim = 0;
    static constexpr int MAXE = 4096;
    hipEvent_t ev[MAXE];
    int nev = 0;
    bool ok = false;
};
// One state per DEVICE: streams and events belong to the device that was current when they were created, and a process may
// drive several GPUs (every entry point makes the device of the caller's stream current - the current device for a null
// stream - before it gets here: GparDeviceGuard).
constexpr int GPAR_MAX_DEVICES = 16;
static LookaheadState g_la_devices[GPAR_MAX_DEVICES];
static LookaheadState& la_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return g_la_devices[dev % GPAR_MAX_DEVICES];
}
#define g_la la_state()

static hipEvent_t la_event() {
    LookaheadState& st = la_state();
    if (st.nev >= LookaheadState::MAXE) st.nev = 0;   // ring: far more than a few factorisations' worth in flight
    return st.ev[st.nev++];
}

static bool la_init() {
    LookaheadState& st = la_state();
    if (st.ok) return true;
    for (int i = 0; i < LookaheadState::MAXE; ++i)
        if (hipEventCreateWithFlags(&st.ev[i], hipEventDisableTiming) != hipSuccess) return false;
    st.ok = true;
    return true;
}

// Side stream paired with `caller`: lowest priority, non-blocking (a blocking stream would serialise with the null
// stream: tools/probe_queue_concurrency2.hip).
static hipStream_t la_side(hipStream_t caller) {
    for (int i = 0; i < g_la.nside; ++i)
        if (g_la.caller[i] == caller) return g_la.side[i];
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;   // lo = least urgent
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
    int slot = g_la.nside;
    if (slot >= LookaheadState::MAXS) {
        // table full (a process that keeps creating caller streams): recycle round-robin; destroying a stream lets its
        // pending work finish first, and every gpar_potrf joins its side stream before it returns anyway
        slot = g_la.next_victim;
        g_la.next_victim = (g_la.next_victim + 1) % LookaheadState::MAXS;
        GPAR_HIP_IGNORE(hipStreamDestroy(g_la.side[slot]));
    } else {
        ++g_la.nside;
    }
    g_la.caller[slot] = caller;
    g_la.side[slot] = s;
    return s;
}

// (`rows`: the rows of the update about to be launched.  An update of at most POTRF_SMALL_ROWS rows - the augmented row once the last
// panel is next - is the one-wave kernel, not the matrix-core update: it is not counted, so that `launches` equals the dispatches of
// gemm_f64_kernel<false, true, 1, *> a kernel trace of the same evaluation shows.)
static void prof_begin(hipStream_t s, bool& active, int rows) {
    active = g_prof.on && g_prof.nev < ProfileState::MAXEV && !(rows <= POTRF_SMALL_ROWS && env_int("GPAR_POTRF_SMALL_UPDATE", 1));
    if (!active) return;
    if (!g_prof.created) {
        for (int i = 0; i < ProfileState::MAXEV; ++i)
            if (hipEventCreate(&g_prof.ev[i][0]) != hipSuccess || hipEventCreate(&g_prof.ev[i][1]) != hipSuccess) { active = false; return; }
        g_prof.created = true;
    }
    GPAR_HIP_IGNORE(hipEventRecord(g_prof.ev[g_prof.nev][0], s));
}

static void prof_end(hipStream_t s, bool active, int rows, int cols, int kb) {
    if (!active) return;
    GPAR_HIP_IGNORE(hipEventRecord(g_prof.ev[g_prof.nev][1], s));
    g_prof.shape[g_prof.nev][0] = rows; g_prof.shape[g_prof.nev][1] = cols; g_prof.shape[g_prof.nev][2] = kb;
    g_prof.nev++;
    g_prof.launches++;
    // algorithmic flops of the lower-trapezoid rank-kb update: 2 * kb per stored element
    g_prof.flops += 2.0 * (double)kb * ((double)cols * ((double)cols + 1.0) * 0.5 + (double)(rows - cols) * (double)cols);
}

// Top level: panels of `nbo` columns; the trailing update of panel k is split into the next panel's columns
// (look-ahead part, stays on the caller's stream ahead of the next panel factorisation) and the rest (on a
// low-priority side stream), so the serial diag/strip chain of panel k+1 runs under the big SYRK of panel k.
static int potrf_run(double* A, int N, int nf, int lda, double* logdet, int* info, hipStream_t stream, int flags = 0, int batch = 1,
                     long long batch_a = 0) {
    if (nf > N) return GPAR_ARG_ERROR(1);
    // The schedule below - panel widths, which steps are grouped, which are fused into one launch, the tile form of every update -
    // is a function of the SHAPE (N, nf, lda alignment) AND OF `batch`, and of nothing else: the same bits with and without
    // look-ahead, alone or beside other work, on any stream.  A matrix factored inside a lock-step batch takes another summation
    // order than the same matrix alone (pair_rows, fuse2_rows and the half-tile rule of the updates all look at the batch): equal
    // to rounding, not to the bit (tests/test_full_size_gpu.py::test_batch_geometry_changes_the_summation_order_not_the_factor).
    PotrfPolicy pol = potrf_policy(N);
    // a LONE large factorisation stops grouping earlier (from 7680 rows on the steps are single panels with look-ahead, the last 7680 rows
    // then one fused launch of fifteen panels): n = 16384 25.10 -> 24.97 ms, n = 12288 11.84 -> 11.79; a lock-step batch keeps 6144 (C3:
    // 180.2 against 180.7 ms with 7680; profiles/r05_exp_fuse_rows.txt)
    if (batch == 1 && N >= 12288 && !getenv("GPAR_POTRF_PAIR_ROWS")) pol.pair_rows = 7680;
    // a wide lock-step batch of large matrices groups its panels down to 2560 rows: its rank-1536 updates are `batch` times a lone
    // matrix's and hide the longer serial stretch, and spare the launch-wide read and write of the trailing matrices two times in three
    // (C5, 16 x 8193: 55.5 -> 53.7 ms; 12 x 8192 40.8 -> 39.5, 12 x 10240 73.7 -> 72.5; batches of 2-8 and matrices below 8192 rows:
    // equal or slower, C3 - 8 x 16385 - equal: profiles/r05_exp_batch_pair.txt)
    if (batch >= 12 && N >= 8192 && !getenv("GPAR_POTRF_PAIR_ROWS")) pol.pair_rows = 2560;
    // the caller runs several factorisations at once (three or more layer streams): each one's look-ahead side stream would
    // add a queue to an already over-subscribed chip (C5, three streams at n = 8192: 78 -> 72 ms per evaluation without)
    if ((flags & GPAR_POTRF_NO_LOOKAHEAD) && !getenv("GPAR_POTRF_LOOKAHEAD")) pol.lookahead = 0;
    if (flags & GPAR_POTRF_UNFUSED) {   // the caller's retry after a hand-off timeout: separate leaf kernels, nothing spins
        pol.fused = 0;
        pol.lookahead = 0;
        pol.nbo = N >= 12288 ? 512 : (N >= 6144 ? 256 : (N >= 1536 ? 128 : 64));
        pol.nbm = pol.nbo >= 512 ? 128 : 64;
    }
    PotrfCtx c{A, N, lda, logdet, info, pol.nbm};
    c.batch = batch;
    c.batch_a = batch_a;
    // lock-step batch: its updates are `batch` times longer than one matrix's - long enough to hide a panel kernel behind at any N
    if (batch > 1 && pol.fused && !getenv("GPAR_POTRF_LOOKAHEAD")) pol.lookahead = env_int("GPAR_POTRF_BATCH_LOOKAHEAD", 1);
    const int nbo = pol.nbo;
    hipStream_t side = (pol.lookahead && nf > nbo && la_init()) ? la_side(stream) : nullptr;
    const bool la = side != nullptr;
    hipEvent_t trail_done = nullptr;   // completion of the side-stream update issued in the previous step
    // Early in the factorisation two panels are factored back to back (the second after a narrow update of its own
    // columns by the first) and the rest of the matrix then receives ONE rank-2*nbo update: the trailing update reads and
    // writes every remaining element once per 1024 columns instead of once per 512, and a K = 1024 SYRK runs ~8 % faster
    // than two K = 512 ones.  The price is a longer serial stretch per step (two panels + the narrow update), so the
    // pairing stops once the trailing update is too short to hide it (`pair_rows`).
    const int G = pol.group;
    // hand-off flags of all panels zeroed once, ahead of the first panel (panel.h)
    const bool prezero = pol.fused && env_int("GPAR_POTRF_PREZERO", 1) && nf >= env_int("GPAR_POTRF_LOCKSTEP_MIN", 1);
    if (prezero) potrf_zero_flags(A, N, lda, stream, batch, batch_a);
    auto groupable = [&](int k) {
        return G > 1 && k > 0 && pol.fused && nbo % 64 == 0 && k + G * nbo <= nf && (N - k) >= pol.pair_rows && (k % 2 == 0) && (lda % 2 == 0) &&
               gpar_aligned16(A);
    };
    // (The last columns through ONE panel kernel of up to 16 column blocks - GPAR_POTRF_TAIL, round 3 - measured slower: n = 1024 0.43
    // against 0.33 ms, C4 18.1 -> 19.7 ms; retired in round 6.)
    // Two or more panels in ONE launch (potrf_group_kernel, panel2.h; at most GPAR_POTRF_FUSE_MAX) once the rows that are left make a step latency-bound: at most
    // GPAR_POTRF_FUSE2_ROWS rows from the step's first column on (a lock-step batch: GPAR_POTRF_FUSE2_BATCH_ROWS over the batch - it
    // fills the chip sooner).  Measured (tools/exp_potrf_fuse2.py, profiles/r04_exp_potrf_fuse2.txt): lone n = 1024 / 2048 / 3072 /
    // 4096 0.333 -> 0.310 / 0.753 -> 0.651 / 1.246 -> 1.146 / 1.869 -> 1.765 ms with every step fused; where a trailing update runs
    // beside the panels on the side stream (look-ahead, N >= 2560) the waiting tile workgroups cost it compute-unit slots, and only the
    // last ~2500 rows gain (n = 8192 5.04 -> 4.92 ms, 5.30 fused from 5120 rows on; n = 4096 1.640 -> 1.579, 3072 1.130 -> 1.079, 2560
    // 0.924 -> 0.866; n = 4600 1.94 / 1.95 / 2.00 / 2.18 ms fused never / from 2560 / 4200 / 5120 rows; n = 16384 inside the noise); a lock-step batch of
    // four gains 1 % on its last pair of panels and loses when more are fused (4 x 4096: 2.74 -> 2.71 / 2.84 ms).  Geometry only, like
    // every other rule here: the same bits with and without look-ahead.
    // Round 5 (the next team's rows updated tile by tile, progressive hand-off): a factorisation of up to 5200 rows is ONE launch (n = 4096
    // 1.56 -> 1.26 ms); a larger one fuses its last eight panels (profiles/r05_exp_fuse_rows.txt: n = 8192 4.77 / 4.72 / 4.60 / 4.71 / 4.66 / 4.79 ms
    // fused from 5200 / 4700 / 4200 / 3600 / 3100 / 2560 rows; n = 5632 .. 16384 all flat within 2 % between 3600 and 4700).
    // With the launch's tiles taking published column blocks without polling and fetching the next one under the current product
    // (grp_la_tile: a tile's share of the earlier panels 6 -> ~3.5 us per column block) a lock-step batch gains from fusing too: four matrices
    // of 4096 rows in ONE launch 3.55 -> 2.81 ms against 3.05 with their last 1536 rows fused (rows x batch <= 16500: C2, 4 x 3072 1.82 ->
    // 1.62 ms, 8 x 2048 1.55 -> 1.36; C3 and C5 - the last 2048 / 1024 rows - unchanged; profiles/r05_exp_batch_fuse.txt).
    const int fuse2_rows = batch == 1 ? env_int("GPAR_POTRF_FUSE2_ROWS", N <= 5200 ? 5200 : (N >= 12288 ? 8300 : 4200)) : env_int("GPAR_POTRF_FUSE2_BATCH_ROWS", 16500) / batch;
    const bool fuse2_on = pol.fused && prezero && !(flags & GPAR_POTRF_UNFUSED) && nbo == 512 &&
                          env_int("GPAR_PANEL_PAIRS", 1) && (lda % 2 == 0) && (batch_a % 2 == 0) && gpar_aligned16(A) && fuse2_rows > 0;
    unsigned long long fuse_counted = 0;   // tiles every row block below the fused launches so far has counted (panel2.h)
    // (more than two panels per launch add little - between panels inside a launch the next team waits ~45 us for the last column
    // blocks of its own rows, which the bulk row blocks of the panel before finish behind the chain - : n = 1536 0.497 -> 0.452 ms with
    // three, n = 2048 0.647 -> 0.637 with four, nothing beyond; 4 measured equal or better than 2 / 3 / 8 at every size)
    // (from N = 12288 on the last sixteen panels: with the faster update tiles n = 12288 12.0-12.1 -> 11.9 ms, n = 16384 25.4-25.5 -> 25.1-25.2;
    // n = 8192 4.66-4.70 / 4.59 / 4.74-4.77 ms fused from 4200 / 6200 / 8300 rows: profiles/r05_exp_fuse_rows.txt)
    const int fuse_max = env_int("GPAR_POTRF_FUSE_MAX", N >= 12288 ? 16 : 10);
    // panels the step at column k takes in one launch (0: the step is not fused)
    auto fuse_panels = [&](int k) {
        if (!fuse2_on || groupable(k) || k % 64 != 0 || N - k > fuse2_rows) return 0;
        int G = (nf - k) / nbo;
        if (G > fuse_max) G = fuse_max;
        return G >= 2 ? G : 0;
    };
    auto fusable2 = [&](int k) { return fuse_panels(k) > 0; };
    auto panel_end = [&](int k) { return fusable2(k) ? k + fuse_panels(k) * nbo : (k + nbo >= nf ? nf : k + nbo); };
    for (int k0 = 0, knext = 0; k0 < nf; k0 = knext) {
        int kend = panel_end(k0);
        // a ragged tail (nf not a multiple of 64) becomes its own narrow panel so the wide part stays fusable
        if (pol.fused && (kend - k0) > 64 && (kend - k0) % 64 != 0) kend = k0 + (kend - k0) / 64 * 64;
        int rc = 0;
        if (groupable(k0)) {
            kend = k0 + G * nbo;
            for (int i = 0; i < G && !rc; ++i) {
                const int ks = k0 + i * nbo;
                if (i > 0) {   // this panel's columns: one update by the i panels of the group factored so far
                    bool pb;
                    prof_begin(stream, pb, N - ks);
                    rc = potrf_gemm_update(c, k0, ks, ks + nbo, stream, 1);
                    prof_end(stream, pb, N - ks, nbo, (ks - k0) * batch);
                }
                if (!rc) rc = potrf_panel_any(A, N, lda, ks, nbo, logdet, info, stream, prezero, batch, batch_a);
            }
        } else {
            const int w = kend - k0;
            const bool fused_ok = pol.fused && w % 64 == 0 && w <= 1024 && N - k0 >= 64 && (k0 % 2 == 0) && (lda % 2 == 0) && gpar_aligned16(A);
            if (fusable2(k0) && w == fuse_panels(k0) * nbo) {
                const int G = fuse_panels(k0);
                rc = potrf_group_fused(A, N, lda, k0, nbo, G, logdet, info, stream, batch, batch_a, fuse_counted);
                fuse_counted += 8ull * (unsigned long long)(G - 1);
            } else if (fused_ok) {
                rc = potrf_panel_any(A, N, lda, k0, w, logdet, info, stream, prezero, batch, batch_a);
            } else {
                if (batch > 1 && w <= POTRF_NBI && !pol.split) {   // a narrow (ragged last) panel of a lock-step batch: batched leaf kernels
                    rc = potrf_panel(c, k0, kend, nbo, stream);
                } else
                for (int b = 0; b < batch && !rc; ++b) {   // leaf kernels (the unfused path): matrix by matrix
                    PotrfCtx cb{A + (size_t)b * batch_a, N, lda, logdet ? logdet + b : nullptr, info ? info + b : nullptr, pol.nbm};
                    rc = pol.split ? potrf_panel_split(cb, k0, kend, nbo, stream) : potrf_panel(cb, k0, kend, nbo, stream);
                }
            }
        }
        knext = kend;
        if (rc) return rc;
        if (kend >= N) break;
        // columns the next step factors (one panel, or two if it pairs): [kend, next_end)
        const int next_end = groupable(kend) ? kend + G * nbo : panel_end(kend);
        bool pa;
        if (!la || kend >= nf) {
            // no further panel to overlap with (or look-ahead off): one update of everything that is left
            if (la && trail_done) { GPAR_HIP_TRY(hipStreamWaitEvent(stream, trail_done, 0)); trail_done = nullptr; }
            if (kend < nf && next_end <= N && potrf_la_is_small(c, k0, kend, next_end)) {
                // (the same split the look-ahead schedule makes, so that both produce the same bits: the next panel's columns by the
                // small kernel, everything to their right by the GEMM)
                rc = potrf_la_update(c, k0, kend, next_end, stream);
                if (!rc) {
                    prof_begin(stream, pa, N - next_end);
                    rc = potrf_rest_update(c, k0, kend, next_end, stream);
                    prof_end(stream, pa, N - next_end, N - next_end, (kend - k0) * batch);
                }
                if (rc) return rc;
                continue;
            }
            if (kend < nf && next_end < N) {
                // (the same two launches as the look-ahead schedule below - the next step's columns, then everything to their right:
                // the update kernel picks its tile shape by the size of the launch, and a tile that preloads C rounds differently
                // from one that adds it at the end, so ONE launch over everything would not return the look-ahead schedule's bits)
                prof_begin(stream, pa, N - kend);
                rc = potrf_gemm_update(c, k0, kend, next_end, stream, 1);
                prof_end(stream, pa, N - kend, next_end - kend, (kend - k0) * batch);
                if (!rc) {
                    prof_begin(stream, pa, N - next_end);
                    rc = potrf_rest_update(c, k0, kend, next_end, stream);
                    prof_end(stream, pa, N - next_end, N - next_end, (kend - k0) * batch);
                }
                if (rc) return rc;
                continue;
            }
            prof_begin(stream, pa, N - kend);
            rc = potrf_gemm_update(c, k0, kend, N, stream, 1);
            prof_end(stream, pa, N - kend, N - kend, (kend - k0) * batch);
            if (rc) return rc;
            continue;
        }
        // (1) next panel's columns, on the caller's stream; they were last written by the previous side update.
        // (Measured negative and retired - NOTES.md section 7 and R5.11: updating only the first panel's columns of a following GROUP
        // ahead of it, GPAR_POTRF_LA_SPLIT (n = 16384 25.94 -> 26.12 ms, C3 180.5 -> 182.0); releasing the big update only after the
        // look-ahead update, GPAR_POTRF_REST_AFTER_LA / GPAR_POTRF_BATCH_REST_AFTER_LA; factoring the very first panel inside a group,
        // GPAR_POTRF_PAIR_FIRST.)
        if (trail_done) GPAR_HIP_TRY(hipStreamWaitEvent(stream, trail_done, 0));
        const int la_end = next_end;
        hipEvent_t panel_done = la_event();
        GPAR_HIP_TRY(hipEventRecord(panel_done, stream));
        if (potrf_la_is_small(c, k0, kend, la_end)) {
            rc = potrf_la_update(c, k0, kend, la_end, stream);
        } else {
            prof_begin(stream, pa, N - kend);
            rc = potrf_gemm_update(c, k0, kend, la_end, stream, 1);   // same kernel symbol: it is part of the trailing update
            prof_end(stream, pa, N - kend, la_end - kend, (kend - k0) * batch);
        }
        if (rc) return rc;
        // (2) everything to the right of the next step's columns, on the side stream
        GPAR_HIP_TRY(hipStreamWaitEvent(side, panel_done, 0));
        {
            const int rows = N - next_end, cols = N - next_end;
            if (rows > 0) {
                prof_begin(side, pa, rows);
                rc = potrf_rest_update(c, k0, kend, next_end, side);
                prof_end(side, pa, rows, cols, (kend - k0) * batch);
                if (rc) return rc;
            }
        }
        trail_done = la_event();
        GPAR_HIP_TRY(hipEventRecord(trail_done, side));
    }
    if (trail_done) GPAR_HIP_TRY(hipStreamWaitEvent(stream, trail_done, 0));   // join
    GPAR_LAUNCH_CHECK();
    return 0;
}

// `batch` factorisations of the same shape in lock-step (matrix b at A + b * batch_a, its words logdet[b] / info[b]): potrf_run
// with ONE panel launch (gridDim.y = batch) and ONE batched trailing update where a single factorisation has a launch of its own.
// Independent factorisations on separate streams contend for compute-unit slots - a panel workgroup of one finds every unit
// holding two update workgroups of another, and its whole team waits; in lock-step every launch carries `batch` times the parallel
// work (no partial last round of update tiles to speak of), the chain of panel kernels is paid once and hides behind the batched
// update of the step before.  Four layers at n = 4096: 4.2 -> 3.3 ms; sixteen at n = 8192: 65 -> 60 ms.  Anything the
// fused path cannot take (alignment, first-generation panel kernel, unfused retry) is factored matrix by matrix.
static int potrf_run_batch(double* A, int batch, long long batch_a, int N, int nf, int lda, double* logdet, int* info, hipStream_t stream,
                           int flags = 0) {
    if (batch <= 0) return 0;
    if (nf > N) return GPAR_ARG_ERROR(1);
    const PotrfPolicy pol = potrf_policy(N);
    const bool lockstep = batch > 1 && pol.fused && !(flags & GPAR_POTRF_UNFUSED) && pol.nbo % 64 == 0 &&
                          (lda % 2 == 0) && (batch_a % 2 == 0) && gpar_aligned16(A) && nf >= env_int("GPAR_POTRF_LOCKSTEP_MIN", 1);
    if (lockstep) return potrf_run(A, N, nf, lda, logdet, info, stream, flags, batch, batch_a);
    for (int b = 0; b < batch; ++b) {
        const int rc = potrf_run(A + (size_t)b * batch_a, N, nf, lda, logdet + b, info + b, stream, flags);
        if (rc) return rc;
    }
    return 0;
}

// B <- B L^-T : forward over column blocks.  B[:, c:c+cb] solved against L_cc, then
// B[:, c+cb:] -= X_c L[c+cb:, c:c+cb]^T  (NT GEMM, K = cb)
static int trsm_rlt_run(const double* L, int n, int ldl, double* B, int nrows, int ldb, hipStream_t stream) {
    return trsm_rlt_run2(L, n, ldl, B, nrows, ldb, 0, stream);
}

// Two-level blocked forward substitution.  Columns are processed in blocks of NB: inside a block, 64-wide strip
// solves with rank-64 updates confined to the block; the rest of the row block is then updated with ONE GEMM of
// K = NB (the K = 64 updates over all remaining columns that a one-level scheme issues run at a third of the rate).
// `upper_tri`: B is upper triangular on entry (e.g. the identity when forming L^-T): row r has nothing left of
// column r, so block [c0, c1) only involves rows < c1 - the n^3 of a full solve becomes n^3 / 3.
static int trsm_rlt_run2(const double* L, int n, int ldl, double* B, int nrows, int ldb, int upper_tri, hipStream_t stream) {
    if (nrows <= 0) return 0;
    // Block width: 512-column fused blocks (panel2.h) + one GEMM update each from n = 1024 on.  (Measured on the inducing-point
    // solve of C4, K_xz L_z^-T with 65536 x 1024: 256-column blocks 20.4 ms per evaluation, 512 19.9, the whole factor in one
    // launch 20.6 - the flops are the same and the block kernel runs them at the same ~55 % of the matrix-core rate.)
    // Below n = 1024 the fused blocks serve as well (round 3: the 64-column strips + K = 64 updates they replace were 1.8-2.8x
    // slower at n = 256 .. 1000 - 20000 rows: 0.18 / 0.48 / 0.83 / 1.31 ms against 0.06 / 0.20 / 0.40 / 0.75).
    const int NB = env_int("GPAR_TRSM_NB", n >= 128 ? 512 : 64);
    const bool fusable = env_int("GPAR_TRSM_FUSED", 1) && NB > 64 && gpar_aligned16(L) && gpar_aligned16(B) && (ldl % 2 == 0) && (ldb % 2 == 0);
    // As in gpar_potrf, while many columns remain G blocks are solved back to back (each after a narrow update of its own
    // columns by the ones before it) and everything to the right then gets ONE rank-G*NB update instead of G rank-NB ones.
    const int pair_cols = env_int("GPAR_TRSM_PAIR_COLS", 4096);
    const int G = env_int("GPAR_TRSM_GROUP", 4);   // blocks per group (2 / 3 / 4 at n = 16384, 204800 rows: 69.6 / 70.4 / 70.9 TFLOP/s; n = 4096, 65536 rows: 63.7 / - / 65.3)
    for (int c0 = 0, c1 = 0; c0 < n; c0 = c1) {
        const bool group = fusable && G > 1 && NB == 512 && c0 + G * NB < n && (n - c0) >= pair_cols;
        if (group) {
            c1 = c0 + G * NB;
            int rc = 0;
            for (int i = 0; i < G && !rc; ++i) {
                const int ci = c0 + i * NB;
                // an upper-triangular right-hand side (upper_tri): row r is zero left of column r, so rows >= ci contribute
                // nothing to the narrow update of block i and rows >= ci + NB have nothing in it to solve
                const int rows_upd = upper_tri ? (nrows < ci ? nrows : ci) : nrows;
                const int rows_blk = upper_tri ? (nrows < ci + NB ? nrows : ci + NB) : nrows;
                if (i > 0)   // block i's columns: one update by the i blocks of the group solved so far
                    rc = gemm_launch(0, 1, rows_upd, NB, i * NB, -1.0, B + c0, ldb, L + (size_t)ci * ldl + c0, ldl, 1.0, B + ci, ldb, 0, stream);
                if (!rc) rc = trsm_block_any(L, n, ldl, B, rows_blk, ldb, ci, NB / 64, upper_tri, stream);
            }
            const int rows = upper_tri ? (nrows < c1 ? nrows : c1) : nrows;
            if (!rc) rc = gemm_launch(0, 1, rows, n - c1, G * NB, -1.0, B + c0, ldb, L + (size_t)c1 * ldl + c0, ldl, 1.0, B + c1, ldb, 0, stream);
            if (rc) return rc;
            continue;
        }
        c1 = (c0 + NB < n) ? c0 + NB : n;
        // a ragged tail (n not a multiple of 64) becomes its own narrow block so the wide part stays fusable
        if (fusable && (c1 - c0) > 64 && (c1 - c0) % 64 != 0) c1 = c0 + (c1 - c0) / 64 * 64;
        const int rows = upper_tri ? (nrows < c1 ? nrows : c1) : nrows;
        if (fusable && (c1 - c0) % 64 == 0 && (c1 - c0) > 64) {
            int rc = trsm_block_any(L, n, ldl, B, rows, ldb, c0, (c1 - c0) / 64, upper_tri, stream);
            if (rc) return rc;
        } else {
            for (int c = c0; c < c1; c += POTRF_NBI) {
                const int cb = (c1 - c < POTRF_NBI) ? c1 - c : POTRF_NBI;
                launch_strip<true>(L + (size_t)c * ldl + c, ldl, cb, B + c, ldb, rows, stream);
                const int rest = c1 - (c + cb);
                if (rest > 0) {
                    int rc = gemm_launch(0, 1, rows, rest, cb, -1.0, B + c, ldb, L + (size_t)(c + cb) * ldl + c, ldl, 1.0,
                                         B + c + cb, ldb, 0, stream);
                    if (rc) return rc;
                }
            }
        }
        if (c1 < n) {
            int rc = gemm_launch(0, 1, rows, n - c1, c1 - c0, -1.0, B + c0, ldb, L + (size_t)c1 * ldl + c0, ldl, 1.0, B + c1, ldb,
                                 0, stream);
            if (rc) return rc;
        }
    }
    GPAR_LAUNCH_CHECK();
    return 0;
}

// Lower triangle of (L L^T)^-1 into Kinv, with X (n x n) as workspace:  X = L^-T (upper triangular, by the
// triangular-aware solve of the identity), Kinv = X X^T where only k >= row contributes.  2 n^3 / 3 flops.
// [torch autograd through cholesky/solve_triangular forms the same quantity implicitly, gpar/regression.py:459]
__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ X, int n, int ldx) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c < n) X[(size_t)r * ldx + c] = (r == c) ? 1.0 : 0.0;
}

// X = L^-T by recursive blocked inversion (n a multiple of 512):
//     [A 0; B C]^-T = [A^-T, -A^-T B^T C^-T; 0, C^-T]
// leaves: all 512 x 512 diagonal blocks at once (trinv_blocks2_kernel); level s = 512, 1024, ...: for every pair of
// neighbouring s-blocks  T = X11 B^T  (NT, k from the tile's first row: X11 is upper triangular) into the scratch matrix,
// X12 = -T X22  (NN, k up to the tile's last column) - the pairs of a level in ONE batched launch each.  Same n^3 / 3 flops
// as the triangular-aware solve of the identity, but the dependent chain is log2(n / 512) levels of large products instead
// of n / 512 block kernels with an update each (32 + 30 launches, the block kernels latency-bound: 32 ms -> see DESIGN).
static int trinv_recursive(const double* L, int n, int ldl, double* X, int ldx, double* T, int ldt, hipStream_t stream) {
    const int leaf = 512;
    int rc = trinv_blocks_fused2(L, n, ldl, X, ldx, leaf / 64, stream);
    for (int s = leaf; s < n && !rc; s *= 2) {
        const int full = n / (2 * s);   // pairs with two whole blocks
        const long long sx = 2LL * s * (ldx + 1), sl = 2LL * s * (ldl + 1), st = 2LL * s * (ldt + 1);
        if (full > 0) {
            rc = gemm_launch(0, 1, s, s, s, 1.0, X, ldx, L + (size_t)s * ldl, ldl, 0.0, T + s, ldt, GPAR_GEMM_K_FROM_ROW, stream, 0, full, sx, sl, st);
            if (!rc) rc = gemm_launch(0, 0, s, s, s, -1.0, T + s, ldt, X + (size_t)s * (ldx + 1), ldx, 0.0, X + s, ldx, GPAR_GEMM_K_TO_COL, stream, 0, full, st, sx, sx);
        }
        const int o = 2 * s * full, s2 = n - (o + s);   // a last pair whose second block is short (a multiple of 512)
        if (!rc && s2 > 0) {
            const double* X11 = X + (size_t)o * (ldx + 1);
            double* Tp = T + (size_t)o * ldt + o + s;
            rc = gemm_launch(0, 1, s, s2, s, 1.0, X11, ldx, L + (size_t)(o + s) * ldl + o, ldl, 0.0, Tp, ldt, GPAR_GEMM_K_FROM_ROW, stream);
            if (!rc) rc = gemm_launch(0, 0, s, s2, s2, -1.0, Tp, ldt, X + (size_t)(o + s) * (ldx + 1), ldx, 0.0, X + (size_t)o * ldx + o + s, ldx,
                                      GPAR_GEMM_K_TO_COL, stream);
        }
    }
    return rc;
}

// Identity in the diagonal blk x blk blocks only.  The recursive inversion and the product that follows never read X outside
// what they have written themselves - their K ranges stop at the diagonal 128 x 128 tiles - except for the zeros BELOW the
// diagonal inside the diagonal blocks: 67 MB to initialise instead of 2.1 GB at n = 16384 (0.03 against 1.1 ms).  The
// strictly lower off-diagonal blocks of X are left as they were (scratch).
__global__ __launch_bounds__(256) void set_block_identity_kernel(double* __restrict__ X, int n, int ldx, int blk) {
    const int r = blockIdx.y;
    const int c = (r / blk) * blk + blockIdx.x * 256 + threadIdx.x;
    if (c < n && c < (r / blk + 1) * blk) X[(size_t)r * ldx + c] = (r == c) ? 1.0 : 0.0;
}

static int chol_inverse_run(const double* L, int n, int ldl, double* X, int ldx, double* Kinv, int ldk, hipStream_t stream) {
    if (n <= 0) return 0;
    const bool recursive = env_int("GPAR_INVERSE_RECURSIVE", 1) && n >= 1024 && n % 512 == 0 && gpar_aligned16(L) &&
                           gpar_aligned16(X) && gpar_aligned16(Kinv) && ldl % 2 == 0 && ldx % 2 == 0 && ldk % 2 == 0;
    if (recursive) hipLaunchKernelGGL(set_block_identity_kernel, dim3(2, n), dim3(256), 0, stream, X, n, ldx, 512);
    else hipLaunchKernelGGL(set_identity_kernel, dim3(gpar_ceil_div(n, 256), n), dim3(256), 0, stream, X, n, ldx);
    int rc = recursive ? trinv_recursive(L, n, ldl, X, ldx, Kinv, ldk, stream) : trsm_rlt_run2(L, n, ldl, X, n, ldx, 1, stream);
    if (rc) return rc;
    return gemm_launch(0, 1, n, n, n, 1.0, X, ldx, X, ldx, 0.0, Kinv, ldk, GPAR_GEMM_C_LOWER | GPAR_GEMM_K_FROM_ROW, stream);
}

// ---------------------------------------------------------------------------------------------------
// x L = b for ONE row vector (alpha^T = z^T L^-1, gpar/model.py:298-301 via stheno's posterior mean): backward
// substitution.  The general path (64-row strips + GEMMs) spends 2 launches per 64 columns on a single row: 512
// dependent launches, 9.6 ms at n = 16384 for 1 GB of reads.  Here: column blocks of 512 from last to first, one
// single-workgroup kernel per block (64-wide triangular solves by one wave, lanes = columns, pivots broadcast by
// v_readlane; in-block updates spread over the four waves) and one memory-bound GEMV-shaped update of everything
// to the left of the block (one 64-column slice per workgroup, the 512 rows split over its waves).
constexpr int TRSV_NB = 512;
constexpr int TRSV_LD = 65;

// One 512-column block, eight 64-column sub-blocks from last to first, one workgroup of four waves, each on its own SIMD.
//   wave 0, the solver: sub-block s entirely in registers - lane = column, the 64 rows of the triangle it needs (L[i][lane], zero
//     above the diagonal), scaled by the reciprocal pivot up front - and a 64-step chain of {v_readlane x 2, multiply-add}.
//     Two register sets: the triangle of the next sub-block is in flight while one is in use (a third set would push the
//     kernel past 512 registers - the updaters hold three - and into scratch: 34 -> 66 us).
//   waves 1-3, the updaters: task (s, c), c < s, is  x_c -= x_s L[s][c]  (64 x 64 tile, lane = column; x_s read once per lane and
//     broadcast through scalar registers).  Target sub-block c belongs to wave 1 + c % 3 - one writer per target, a fixed
//     summation order - and a wave works TARGET-major: target c gets the sources s = 7 .. c + 1 in that order, the last of which
//     is the only task on the critical path (x_{c+1} has just been solved, x_c is next); the first tasks of its NEXT target
//     (sources solved long ago) are a backlog that runs under the other waves' critical steps.  Three register sets: the tiles
//     of the next two tasks are in flight while one is used - tile addresses are static, nothing waits for memory that it
//     could have asked for earlier.
// Hand-offs are words in LDS, not barriers: ready[s] (x_s is in LDS) and done[c] (updates applied to target c so far); a wave
// polls the one word it needs, so nobody waits for a wave that is busy with work off the critical path.
// History: (1) triangle staged through LDS, a row read per step inside the chain: 106 us per block; (2) triangle in registers,
// ALL updates of source s between the chains of s and s - 1, second and third tile of a wave requested only after the first:
// 55 us; (3) this one.  What bounds it: the chain (64 dependent steps of ~30 ns per sub-block) and what one compute unit can
// pull from L2 / HBM with nine tiles in flight (1.1 MB per block).
#ifdef GPAR_TRSV_STAMPS
__device__ long long g_trsv_stamps[4][64];
#define TRSV_STAMP(k) do { if (lane == 0 && (k) < 64) g_trsv_stamps[w][(k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRSV_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(256) void trsv_block_kernel(const double* __restrict__ L, int ldl, double* __restrict__ b, int c0, int c1) {
    __shared__ double xb[TRSV_NB];
    __shared__ int ready[TRSV_NB / 64], done[TRSV_NB / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    TRSV_STAMP(0);
    const int W = c1 - c0;
    for (int i = t; i < TRSV_NB; i += 256) xb[i] = i < W ? b[c0 + i] : 0.0;   // (zeros beyond a ragged edge: x there stays zero)
    if (t < TRSV_NB / 64) { ready[t] = 0; done[t] = 0; }
    const int nsub = (W + 63) / 64;
    auto cols_of = [&](int s) { const int cs = c0 + 64 * s; return (c1 - cs < 64) ? c1 - cs : 64; };
    auto wait_word = [&](int* word, int value) {   // (all lanes read the same word: a broadcast; the loop is wave-uniform)
        while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < value) __builtin_amdgcn_s_sleep(1);
    };
    auto post_word = [&](int* word, int value) {   // every lane's LDS writes of this wave are ordered before the word
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // (row pointers are wave-uniform - scalar arithmetic -, the lane only adds its column: one address instruction per load
    // instead of a 64-bit multiply-add each, which made requesting a tile cost as much as using it)
    auto load_rows = [&](const double* base, int cb, int col, double (&lv)[64]) {
        if (cb == 64) {
#pragma unroll
            for (int i = 0; i < 64; ++i) lv[i] = (base + (size_t)i * ldl)[col];
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) lv[i] = (base + (size_t)(i < cb ? i : cb - 1) * ldl)[col];   // clamped, never behind a branch
        }
    };
    if (w == 0) {
        double tri[2][64], dg[2];
        auto load_triangle = [&](int s, double (&lv)[64], double& d) {
            const int cs = c0 + 64 * s, cb = cols_of(s);
            const int lc = lane < cb ? lane : cb - 1;
            const double* base = L + (size_t)cs * ldl + cs;
            load_rows(base, cb, lc, lv);
            d = (base + (size_t)lc * ldl)[lc];
        };
        auto solve = [&](int s, double (&lv)[64], double d) {
            const int cb = cols_of(s);
            // The column is scaled by its reciprocal pivot up front (64 independent multiplies, off the chain): lane j then carries
            // x_j-to-be, (b_j - sum_{i > j} x_i L[i][j]) / L[j][j], and a chain step is {v_readlane x 2, multiply-add}.  Entries above
            // the diagonal (row i < column) and beyond a ragged edge are zero: a lane's value stops changing once its own step
            // has passed, and is read off after the loop.
            const double rinv = lane < cb ? 1.0 / d : 0.0;
#pragma unroll
            for (int i = 0; i < 64; ++i) lv[i] = (lane < i && i < cb) ? lv[i] * rinv : 0.0;
            wait_word(&done[s], nsub - 1 - s);   // everything owed to sub-block s has landed
            TRSV_STAMP(1 + 2 * (nsub - 1 - s));
            double xj = lane < cb ? xb[64 * s + lane] * rinv : 0.0;
#pragma unroll
            for (int i = 63; i >= 0; --i) {
                const double xi = gpar_readlane_f64(xj, i);   // final for lane i: every step above it has been applied (zero for i >= cb)
                xj = fma(-xi, lv[i], xj);
            }
            if (lane < cb) xb[64 * s + lane] = xj;
            post_word(&ready[s], 1);
            TRSV_STAMP(2 + 2 * (nsub - 1 - s));
        };
        load_triangle(nsub - 1, tri[0], dg[0]);
        if (nsub > 1) load_triangle(nsub - 2, tri[1], dg[1]);
        __syncthreads();   // xb and the hand-off words are initialised
        for (int s = nsub - 1; s >= 0; s -= 2) {
            solve(s, tri[0], dg[0]);
            if (s > 1) load_triangle(s - 2, tri[0], dg[0]);   // (its registers are free again)
            if (s == 0) break;
            solve(s - 1, tri[1], dg[1]);
            if (s > 2) load_triangle(s - 3, tri[1], dg[1]);
        }
    } else {
        int gs = nsub - 1, gc = nsub - 2;   // generator of this wave's task sequence: targets it owns, downwards; sources nsub - 1 .. c + 1
        while (gc >= 0 && gc % 3 != w - 1) --gc;
        auto take = [&](int& s_, int& c_) {
            s_ = gs; c_ = gc;
            if (gc < 0) return;
            if (gs > gc + 1) { --gs; return; }
            gc -= 3;
            gs = nsub - 1;
        };
        auto load_tile = [&](int s, int c, double (&lv)[64]) {
            load_rows(L + (size_t)(c0 + 64 * s) * ldl + c0 + 64 * c, cols_of(s), lane, lv);
        };
        int ntask = 0;
        auto compute = [&](int s, int c, const double (&lv)[64]) {
            wait_word(&ready[s], 1);   // x_s exists
            TRSV_STAMP(1 + 2 * ntask);
            const double xs = xb[64 * s + lane];   // (zero beyond a ragged edge, the clamped rows finite)
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 64; ++i) acc[i & 3] = fma(gpar_readlane_f64(xs, i), lv[i], acc[i & 3]);
            xb[64 * c + lane] -= (acc[0] + acc[1]) + (acc[2] + acc[3]);
            post_word(&done[c], nsub - s);   // sources nsub - 1 .. s applied: nsub - s of them
            TRSV_STAMP(2 + 2 * ntask);
            ++ntask;
        };
        double tile[3][64];
        int ts[3], tc[3];
        take(ts[0], tc[0]);
        take(ts[1], tc[1]);
        if (tc[0] >= 0) load_tile(ts[0], tc[0], tile[0]);
        if (tc[1] >= 0) load_tile(ts[1], tc[1], tile[1]);
        __syncthreads();   // xb and the hand-off words are initialised
        for (;;) {
            take(ts[2], tc[2]);
            if (tc[2] >= 0) load_tile(ts[2], tc[2], tile[2]);
            if (tc[0] < 0) break;
            compute(ts[0], tc[0], tile[0]);
            take(ts[0], tc[0]);
            if (tc[0] >= 0) load_tile(ts[0], tc[0], tile[0]);
            if (tc[1] < 0) break;
            compute(ts[1], tc[1], tile[1]);
            take(ts[1], tc[1]);
            if (tc[1] >= 0) load_tile(ts[1], tc[1], tile[1]);
            if (tc[2] < 0) break;
            compute(ts[2], tc[2], tile[2]);
        }
    }
    __syncthreads();   // every update has landed
    TRSV_STAMP(63);
    for (int i = t; i < W; i += 256) b[c0 + i] = xb[i];
}

// b[j] -= sum_k x[k] L[c0 + k][j] for the columns j in [j0, j1), j1 <= c0 (x = b[c0 .. c1), just solved): one 64-column slice per workgroup
__global__ __launch_bounds__(256) void trsv_update_kernel(const double* __restrict__ L, int ldl, double* __restrict__ b, int c0, int c1, int j0, int j1) {
    __shared__ double xs[TRSV_NB];
    __shared__ double part[4][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int W = c1 - c0;
    for (int i = t; i < W; i += 256) xs[i] = b[c0 + i];
    __syncthreads();
    const int j = j0 + 64 * blockIdx.x + lane;
    const int jc = j < j1 ? j : j1 - 1;
    const int kw = (W + 3) / 4, k0 = w * kw, k1 = (k0 + kw < W) ? k0 + kw : W;
    const double* Lp = L + (size_t)c0 * ldl + jc;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int kb = k0; kb < k1; kb += 32) {
        double lv[32];   // 32 loads in flight per lane
#pragma unroll
        for (int i = 0; i < 32; ++i) lv[i] = Lp[(size_t)(kb + i < k1 ? kb + i : k1 - 1) * ldl];
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
            a0 = fma(kb + i < k1 ? xs[kb + i] : 0.0, lv[i], a0);
            a1 = fma(kb + i + 1 < k1 ? xs[kb + i + 1] : 0.0, lv[i + 1], a1);
            a2 = fma(kb + i + 2 < k1 ? xs[kb + i + 2] : 0.0, lv[i + 2], a2);
            a3 = fma(kb + i + 3 < k1 ? xs[kb + i + 3] : 0.0, lv[i + 3], a3);
        }
    }
    part[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && j < j1) b[j] -= (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// (Tried: after block k only the next block's 512 columns updated on the caller's stream, everything further left on the look-ahead
// side stream under block k + 1's chain - three event operations and an extra launch per block cost more than the overlap gains:
// n = 16384 1.52 -> 1.87 ms.)
static int trsv_rln_run(const double* L, int n, int ldl, double* b, hipStream_t stream) {
    const int nblk = gpar_ceil_div(n, TRSV_NB);
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int c0 = blk * TRSV_NB;
        const int c1 = (c0 + TRSV_NB < n) ? c0 + TRSV_NB : n;
        hipLaunchKernelGGL(trsv_block_kernel, dim3(1), dim3(256), 0, stream, L, ldl, b, c0, c1);
        if (c0 > 0) hipLaunchKernelGGL(trsv_update_kernel, dim3(gpar_ceil_div(c0, 64)), dim3(256), 0, stream, L, ldl, b, c0, c1, 0, c0);
    }
    GPAR_LAUNCH_CHECK();
    return 0;
}

// B <- B L^-1 : backward over column blocks.  B[:, c:c+cb] solved against L_cc, then
// B[:, :c] -= X_c L[c:c+cb, :c]  (NN GEMM, K = cb)
static int trsm_rln_run(const double* L, int n, int ldl, double* B, int nrows, int ldb, hipStream_t stream) {
    if (nrows <= 0) return 0;
    if (nrows == 1 && n >= 128 && env_int("GPAR_TRSV", 1)) return trsv_rln_run(L, n, ldl, B, stream);
    // From n = 128 on (round 2: 1024): 512-column blocks from the last to the first, each solved by the fused backward block kernel
    // (panel2.h) and followed by ONE K = 512 NN update of everything to its left - instead of 64-column strips with a K = 64
    // update each (16 + 16 launches per 1024 columns, the updates at a fifth of the rate).  A ragged tail (n not a multiple of
    // 64) goes first, by the strip path.
    const bool fusable = env_int("GPAR_TRSM_FUSED", 1) && n >= 128 && nrows >= 2 &&
                         gpar_aligned16(L) && gpar_aligned16(B) && (ldl % 2 == 0) && (ldb % 2 == 0);
    int ntop = n;   // columns [0, ntop) still to be solved
    if (fusable) {
        const int rag = n % 64;
        if (rag) {
            const int c = n - rag;
            launch_strip<false>(L + (size_t)c * ldl + c, ldl, rag, B + c, ldb, nrows, stream);
            int rc = gemm_launch(0, 0, nrows, c, rag, -1.0, B + c, ldb, L + (size_t)c * ldl, ldl, 1.0, B, ldb, 0, stream);
            if (rc) return rc;
            ntop = c;
        }
        while (ntop > 0) {
            const int w = ntop % 512 ? ntop % 512 : 512;   // the (possibly short) last block first: the others are whole
            const int c0 = ntop - w;
            int rc = trsm_block_back_fused2(L, n, ldl, B, nrows, ldb, c0, w / 64, stream);
            if (!rc && c0 > 0) rc = gemm_launch(0, 0, nrows, c0, w, -1.0, B + c0, ldb, L + (size_t)c0 * ldl, ldl, 1.0, B, ldb, 0, stream);
            if (rc) return rc;
            ntop = c0;
        }
        GPAR_LAUNCH_CHECK();
        return 0;
    }
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
