// Panel factorisation and fused triangular-solve blocks, second generation: LEFT-LOOKING row-block tasks on the matrix cores.
//
// A W-column panel (W = 64 S) is a 64 S x 64 S diagonal block (S row blocks: the "team") followed by row blocks that only
// need  X = A21 L11^-T  ("bulk" rows).  One workgroup per 64-row block, block index = row block, so that every wait is
// on a LOWER-indexed workgroup (the dispatcher launches blocks in index order: whatever is resident can always finish,
// the grid may be larger than the chip, and several of these kernels can be in flight from different streams).
//
// Row block rb, for every column block c (64 columns) it has to solve:
//     acc  = sum_{u<c} X[rb][u] L[c][u]^T         one K = 64 chunk per u on v_mfma_f64_16x16x4, operands through LDS, the
//                                                 next chunk's two tiles in flight in registers; C never leaves registers
//     T    = A[rb][c] - acc
//     X    = T L[c][c]^-T                         "strip": again on the matrix cores (see p2_strip)
// and a team row block (rb < S) keeps  D = sum_u X[rb][u] X[rb][u]^T  in registers, factors  A[rb][rb] - D  (pnl_diag) and
// publishes L[rb][rb].  The first version of the fused panel kernel (panel.h) updated A[rb][c] right-looking, one
// read-modify-write of a 64 x 64 tile in global memory per K = 64, and solved strips by substitution on the vector ALUs
// (19.5 k cycles per 64 x 64 strip, one wave doing the serial part): both the hand-off chain (60-65 k cycles per 64
// columns) and the bulk work per row block (~480 k cycles per 512-column panel) ran at ~10 % of the matrix-core rate.
//
// Data layout in registers ("T layout"): wave w owns rows 16 w .. 16 w + 15 of the row block and all 64 columns of the
// column block, as four TRANSPOSED 16 x 16 tiles: register v of tile mi in lane l holds element
//     (column 16 mi + (l >> 4) + 4 v,  row 16 w + (l & 15)).
// This is the D layout of v_mfma_f64_16x16x4 for the product  L[c][u] X^T  (M = columns, N = rows) - and, because D's
// register v holds rows 4 v .. 4 v + 3 of the tile, register k4 of a tile IS the B operand of K-step k4 of the next
// product: the strip's chain  X_jb^T = W_jb T_jb,  T_jb' -= L[jb'][jb] X_jb^T  runs from accumulators to operands with
// no shuffle, no LDS round trip and no barrier (each wave solves its own 16 rows).
//
// Strip arithmetic: the 64 x 64 triangle L[c][c] is used in 16 x 16 blocks; off-diagonal blocks enter as plain products
// (substitution), the four diagonal 16 x 16 blocks through their explicit inverses W_jb (four waves, one block each, 136
// fused multiply-adds per lane: in the factorisation once per diagonal tile by its owner, which publishes them in four
// strictly upper 16 x 16 blocks of the tile - scratch by the ABI's convention; in a triangular solve once per strip).  Those blocks are diagonal blocks of a Cholesky factor
// of K + noise: their condition number is the square root of that of a 16 x 16 principal block of the Schur complement.
#pragma once
#include "common.h"
#include "panel.h"

namespace gpar {

constexpr int P2_LDS_BYTES = 2 * PNL_TILE * 8 + 1024;   // operand tile (L) + row-block tile (X / A) + progress cache, reciprocals

// ---- tile movement: 64 x 64 doubles, 256 threads, 8 x 16 bytes per thread, coalesced ---------------------------------
// (The thread index is passed through an empty volatile asm in each of these: otherwise the compiler hoists the eight
// 64-bit row addresses of every tile kind out of the column loop - loop invariants - and, out of registers, spills them;
// the reloads then sit in front of the stores on the hand-off chain.  Recomputing them is a dozen integer instructions.)
__device__ __forceinline__ int p2_opaque(int t) {
    asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ void p2_gload(const double* __restrict__ M, int ld, int nrows, int r0, int c0, int t, pan_d2 (&v)[8]) {
    t = p2_opaque(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q;
        const int r = c >> 5, cc = (c & 31) * 2;
        const int rr = min(r0 + r, nrows - 1);   // clamped, never behind a branch
        v[q] = *reinterpret_cast<const pan_d2*>(M + (size_t)rr * ld + c0 + cc);
    }
}

__device__ __forceinline__ void p2_sstore(double* __restrict__ dst, int t, const pan_d2 (&v)[8]) {
    t = p2_opaque(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q;
        const int r = c >> 5, cc = (c & 31) * 2;
        *reinterpret_cast<pan_d2*>(dst + r * PNL_LD + cc) = v[q];
    }
}

// LDS tile -> global, rows beyond nrows skipped; `through`: write-through (sc1) stores for tiles other workgroups read.
__device__ __forceinline__ void p2_gstore(double* __restrict__ M, int ld, int nrows, int r0, int c0, const double* __restrict__ src,
                                          int t, bool through) {
    t = p2_opaque(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q;
        const int r = c >> 5, cc = (c & 31) * 2;
        if (r0 + r < nrows) {
            const pan_d2 v = *reinterpret_cast<const pan_d2*>(src + r * PNL_LD + cc);
            double* dst = M + (size_t)(r0 + r) * ld + c0 + cc;
            if (through) {
                // one 16-byte write-through store (an 8-byte sc1 store costs a fabric write of its own: 2.7x per byte).  Every asm
                // store of more than 8 bytes in this file ends with `s_nop 1`: the data registers are read over several cycles and
                // the compiler, which does not look inside asm, may otherwise overwrite them with its next instruction (seen in
                // round 5: the low word of one double in ~1e5 stores zeroed - 1e-7 relative -, only in some launches).
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v) : "memory");
            } else {
                *reinterpret_cast<pan_d2*>(dst) = v;
            }
        }
    }
}

// ---- one K = 64 chunk:  acc[mi] (column tile mi x this wave's 16 rows) += L[.][k] X[.][k]^T --------------------------
// The operand fragments of K-step k4 + 1 are requested from LDS before the four products of step k4 are issued (two
// register sets, the loop fully unrolled): with one wave per SIMD nothing else covers the LDS latency, and reads issued
// right before the products that need them left the matrix core idle a third of the time (110 instead of 64 cycles per
// v_mfma_f64_16x16x4: tools/time_panel2.hip).
__device__ __forceinline__ void p2_frag(const double* __restrict__ Ls, const double* __restrict__ Xs, int w, int l15, int kk,
                                        double (&a)[4], double& b) {
    b = Xs[(16 * w + l15) * PNL_LD + kk];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a[mi] = Ls[(16 * mi + l15) * PNL_LD + kk];
}

__device__ __forceinline__ void p2_chunk(const double* __restrict__ Ls, const double* __restrict__ Xs, pan_d4 (&acc)[4], int w,
                                         int l15, int lk) {
    double a0[4], a1[4], b0, b1;
    p2_frag(Ls, Xs, w, l15, lk, a0, b0);
    // (scheduling barriers: left alone, the compiler sinks every read to just before its product)
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 2) {
        p2_frag(Ls, Xs, w, l15, 4 * (k4 + 1) + lk, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[mi], b0, acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 2 < 16) p2_frag(Ls, Xs, w, l15, 4 * (k4 + 2) + lk, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[mi], b1, acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- the look-ahead update of a latency-bound step ------------------------------------------------------------------------
// C[kend.., kend .. kend + 64 nc) -= P P^T restricted to the next panel's columns (P = the columns [k0, kend) just factored), one
// 64 x 64 output tile per workgroup: K / 64 chunks of the panel kernel's own product (p2_chunk), operands double-buffered through
// registers, the C tile arriving behind the last chunk.  In the tail of a factorisation (and throughout a small one) this update
// has fewer tiles than the chip has slots, and its duration is ONE tile's: 43-100 us in the 128 x 64 GEMM tile (K = 512 in 32
// dependent stages at one workgroup per compute unit), ~20 us here - on the critical path of every step.  Diagonal tiles store
// their lower triangle only (the strict upper triangle holds the next panel's hand-off flags).
struct LaUpdateArgs {
    double* A;
    int N, lda, k0, kend, nc;   // nc: 64-column blocks of the next panel
    long long batch_a;
};

__global__ __launch_bounds__(256, 2) void potrf_la_update_kernel(LaUpdateArgs a) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    const int nc = a.nc, tri = nc * (nc + 1) / 2;
    const int tile = blockIdx.x;
    int ti, tj;
    if (tile < tri) {   // the top nc x nc block of tiles: lower triangle
        ti = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
        while (ti * (ti + 1) / 2 > tile) --ti;
        tj = tile - ti * (ti + 1) / 2;
    } else {
        const int r = tile - tri;
        ti = nc + r / nc;
        tj = r - (r / nc) * nc;
    }
    double* A = a.A + (size_t)blockIdx.y * a.batch_a;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    const int r0 = a.kend + 64 * ti, c0 = a.kend + 64 * tj;
    const int nchunks = (a.kend - a.k0) / 64;
    pan_d4 acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0.0, 0.0, 0.0, 0.0};
    pan_d2 xa[8], la[8];
    p2_gload(A, a.lda, a.N, r0, a.k0, t, xa);
    p2_gload(A, a.lda, a.N, c0, a.k0, t, la);
    for (int u = 0; u < nchunks; ++u) {
        __syncthreads();   // the previous chunk's operand reads are done
        p2_sstore(Cs, t, la);
        p2_sstore(Xs, t, xa);
        __syncthreads();
        p2_gload(A, a.lda, a.N, c0, a.k0 + 64 * min(u + 1, nchunks - 1), t, la);
        if (u + 1 < nchunks) p2_gload(A, a.lda, a.N, r0, a.k0 + 64 * (u + 1), t, xa);
        else p2_gload(A, a.lda, a.N, r0, c0, t, xa);   // behind the last chunk: the output tile itself
        __builtin_amdgcn_sched_barrier(0);
        p2_chunk(Cs, Xs, acc, w, l15, lk);
    }
    __syncthreads();
    p2_sstore(Xs, t, xa);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] -= acc[mi][v];
    __syncthreads();
    if (ti != tj) {
        p2_gstore(A, a.lda, a.N, r0, c0, Xs, t, false);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = t + 256 * q;
            const int r = c >> 5, cc = (c & 31) * 2;
            if (r0 + r < a.N && cc <= r) {
                double* dst = A + (size_t)(r0 + r) * a.lda + c0 + cc;
                if (cc + 1 <= r) *reinterpret_cast<pan_d2*>(dst) = *reinterpret_cast<const pan_d2*>(Xs + r * PNL_LD + cc);
                else dst[0] = Xs[r * PNL_LD + cc];
            }
        }
    }
}

static int potrf_la_update_small(double* A, int N, int lda, int k0, int kend, int ncols, hipStream_t stream, int batch, long long batch_a) {
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&potrf_la_update_kernel), P2_LDS_BYTES));
    const int nc = ncols / 64, tr = (N - kend + 63) / 64;
    const int tiles = nc * (nc + 1) / 2 + (tr - nc) * nc;
    LaUpdateArgs a{A, N, lda, k0, kend, nc, batch_a};
    hipLaunchKernelGGL(potrf_la_update_kernel, dim3(tiles, batch), dim3(256), P2_LDS_BYTES, stream, a);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// Position of W_jb (inverse of the jb-th diagonal 16 x 16 block) inside the strictly upper 16 x 16 blocks of the LDS tile.
__device__ __forceinline__ int p2_wblock(int jb) {
    // jb: 0 -> block (0, 2), 1 -> (0, 3), 2 -> (1, 2), 3 -> (1, 3): columns >= 32, clear of the progress words that the
    // panel's first diagonal block keeps in its row 0 (columns 8 .. 23)
    return 16 * (jb >> 1) * PNL_LD + 16 * (2 + (jb & 1));
}

// Refinement flags.  A strip solves against the diagonal 16 x 16 blocks of the tile through their explicit inverses W:
// x0 = W t is not backward stable - its residual t - L x0 is of order eps * cond(L_bb) |t| where substitution leaves eps |L| |x|.
// For K + noise that is invisible (pivots of a block within a factor ~10 of each other); for K_zz + 1e-12 of many inducing inputs
// on one axis - numerically rank-deficient by construction - the block in which the pivots fall from 1 to 1e-6 has cond ~ 1e6,
// the factorisation's backward error grew from 2e-15 (LAPACK, and the unfused path here) to 1e-13, and one n = 2048 case failed
// outright where LAPACK does not (tools/diag_illcond_potrf.py).  So whoever inverts a block also records whether its largest
// pivot exceeds P2_REFINE_RATIO times its smallest, and the strips give such blocks one step of iterative refinement:
// x1 = x0 + W (t - L_bb x0), eight more matrix-core products for that block - backward stable while eps * cond << 1, and paid
// only where it is needed.  The four flags of a tile sit in row 1, columns 16 .. 19 of the tile (strictly upper: scratch by the
// ABI's convention, clear of the progress words in row 0 and of the inverses from column 32 on).
#ifndef GPAR_P2_REFINE_RATIO
#define GPAR_P2_REFINE_RATIO 32.0
#endif
constexpr double P2_REFINE_RATIO = GPAR_P2_REFINE_RATIO;
__device__ __forceinline__ int p2_flag_slot(int jb) { return PNL_LD + 16 + jb; }

// Wave w inverts the w-th diagonal 16 x 16 block  [La 0; Lba Lb]  of the lower-triangular tile Cs into the 16 x 16 block
// W at p2_wblock(w):  W = [Wa 0; -Wb Lba Wa, Wb].
//   (1) the two 8 x 8 inverses by substitution on 16 lanes (lane = 8 h + j solves L_h x = e_j; all 36 coefficients of its
//       triangle requested up front, reciprocal pivots by v_rcp_f64 + two Newton steps);
//   (2) the coupling block on the matrix cores: P = Lba Wa, then -Wb P - four v_mfma_f64_16x16x4 with zero padding, the
//       result going from the accumulators of the first product straight into the B operand of the second.
// A 16-step substitution on the vector ALUs needs 120 wave-uniform LDS reads per wave (every one a full LDS instruction):
// 2 us with four waves at it, on the critical chain of every 64 columns; this form takes a quarter of that.
// (the two phases are separate functions because the panel's diagonal-tile owner runs them in different rounds of p3_diag)
__device__ __forceinline__ void p2_inverse_sub(double* __restrict__ Cs, int w, int lane) {
    const int l15 = lane & 15;
    const int base = 16 * w;
    double* W = Cs + p2_wblock(w);
    {
        const int h = l15 >> 3, j = l15 & 7, b8 = base + 8 * h;
        double c[8][8];   // c[i][k], k < i: strictly lower coefficients; c[i][i]: pivots
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k <= i; ++k) c[i][k] = Cs[(b8 + i) * PNL_LD + b8 + k];
        {   // refinement flag of this block: spread of its sixteen pivots (each lane holds the eight of its half)
            double mn = c[0][0], mx = c[0][0];
#pragma unroll
            for (int k = 1; k < 8; ++k) { mn = fmin(mn, c[k][k]); mx = fmax(mx, c[k][k]); }
            mn = fmin(mn, __shfl_xor(mn, 8, 64));
            mx = fmax(mx, __shfl_xor(mx, 8, 64));
            if (lane == 0) Cs[p2_flag_slot(w)] = (mx > P2_REFINE_RATIO * mn) ? 1.0 : 0.0;
        }
        double x[8], rp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (i == j) ? 1.0 : 0.0;
        // the eight reciprocal pivots ahead of the substitution (independent of it: eight pipelined v_rcp_f64 + Newton steps instead
        // of one dependent chain per step; same operations, same bits)
#pragma unroll
        for (int k = 0; k < 8; ++k) rp[k] = __builtin_amdgcn_rcp(c[k][k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) rp[k] = fma(fma(-c[k][k], rp[k], 1.0), rp[k], rp[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) rp[k] = fma(fma(-c[k][k], rp[k], 1.0), rp[k], rp[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double r = rp[k];
            const double xk = x[k] * r;
            x[k] = xk;
#pragma unroll
            for (int i = k + 1; i < 8; ++i) x[i] = fma(-c[i][k], xk, x[i]);
        }
        // diagonal 8 x 8 blocks of W (zeros above their diagonals included) and the zero upper-right 8 x 8 block
#pragma unroll
        for (int i = 0; i < 8; ++i) W[(8 * h + i) * PNL_LD + 8 * h + j] = x[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) W[i * PNL_LD + 8 + (l15 & 7)] = 0.0;
    }
}

__device__ __forceinline__ void p2_inverse_couple(double* __restrict__ Cs, int w, int lane) {
    const int l15 = lane & 15, lk = lane >> 4;
    const int base = 16 * w;
    double* W = Cs + p2_wblock(w);
    // P = Lba Wa: rows 8 .. 15 (registers 2, 3 of the D layout), columns 0 .. 7
    pan_d4 P = pan_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 2; ++k4) {
        const int k = 4 * k4 + lk;
        const double av = Cs[(base + l15) * PNL_LD + base + k];   // L16[m][k], used for m >= 8 only
        const double bv = W[k * PNL_LD + l15];                    // Wa[k][n], n < 8
        P = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 >= 8 ? av : 0.0, l15 < 8 ? bv : 0.0, P, 0, 0, 0);
    }
    pan_d4 Q = pan_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 2; ++k4) {
        const double av = W[l15 * PNL_LD + 8 + 4 * k4 + lk];      // Wb[m - 8][k - 8] = W[m][8 + ...], m >= 8
        Q = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 >= 8 ? av : 0.0, P[2 + k4], Q, 0, 0, 0);
    }
    if (l15 < 8) {
        W[(8 + lk) * PNL_LD + l15] = -Q[2];
        W[(12 + lk) * PNL_LD + l15] = -Q[3];
    }
}

__device__ __forceinline__ void p2_inverse_blocks(double* __restrict__ Cs, int w, int lane) {
    p2_inverse_sub(Cs, w, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    p2_inverse_couple(Cs, w, lane);
}

// T (this wave's 16 rows x 64 columns, T layout) <- X^T with X L^T = T, L the lower-triangular tile in Cs whose diagonal
// 16 x 16 blocks have their inverses at p2_wblock().
// (one 16-column block of the strip, JB = 0 .. 3 in this order; a team row that follows the diagonal tile's owner block by block -
// p2_row_block, progressive hand-off - calls the steps one at a time)
template <int JB>
__device__ __forceinline__ void p2_strip_step(const double* __restrict__ Cs, pan_d4 (&T)[4], int l15, int lk) {
    constexpr int jb = JB;
    const double* W = Cs + p2_wblock(jb);
    pan_d4 x = pan_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(W[l15 * PNL_LD + 4 * k4 + lk], T[jb][k4], x, 0, 0, 0);
    if (Cs[p2_flag_slot(jb)] != 0.0) {   // (wave-uniform) an ill-conditioned block: x += W (t - L_bb x)
        pan_d4 r = T[jb];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int k = 4 * k4 + lk;
            const double l = Cs[(16 * jb + l15) * PNL_LD + 16 * jb + k];   // L_bb[m = l15][k]; above its diagonal the tile holds scratch
            r = __builtin_amdgcn_mfma_f64_16x16x4f64(k <= l15 ? -l : 0.0, x[k4], r, 0, 0, 0);
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(W[l15 * PNL_LD + 4 * k4 + lk], r[k4], x, 0, 0, 0);
    }
    T[jb] = x;
#pragma unroll
    for (int j2 = jb + 1; j2 < 4; ++j2) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
            T[j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Cs[(16 * j2 + l15) * PNL_LD + 16 * jb + 4 * k4 + lk], x[k4], T[j2], 0, 0, 0);
    }
}

__device__ __forceinline__ void p2_strip(const double* __restrict__ Cs, pan_d4 (&T)[4], int l15, int lk) {
    p2_strip_step<0>(Cs, T, l15, lk);
    p2_strip_step<1>(Cs, T, l15, lk);
    p2_strip_step<2>(Cs, T, l15, lk);
    p2_strip_step<3>(Cs, T, l15, lk);
}

// ---- the diagonal-tile factorisation: wave-uniform coefficients through DPP ------------------------------------------
// A lone wave issues one instruction every ~4.3 cycles whatever it is (tools/ubench_dp_latency.hip: dependent v_fma_f64
// 4.3 cycles, v_rsq_f64 16), so the pivot chain of pnl_diag (panel.h) is bound by its INSTRUCTION COUNT (~34 per pivot,
// ~60 with the rank-8 update of the block's own columns; trimming its bookkeeping and interleaving that update gained 5 %), and a third of those are v_readlane pairs and LDS broadcast reads that only
// move wave-uniform coefficients.  Here they come with the arithmetic: the 16 lanes of a DPP row are
//     lanes 0-7:  rows 8 jb .. 8 jb + 7 of the tile (the block's own diagonal rows, replicated in every DPP row),
//     lanes 8-15: eight other rows (wave w < 2, DPP row R: rows 32 w + 8 R .. + 7),
// and  v_fmac_f64_dpp acc, -x row_newbcast:k, y  multiplies by lane k's x of the same DPP row - the coefficient
// L[8 jb + k][.] - in the instruction that uses it (DP-ALU DPP supports exactly this control).  Two waves factor (both
// carry the replicated rows, so they never talk), the other two apply the previous block's rank-8 update to the columns to
// the right on the matrix cores (16 x 16 blocks, K = 8).  Per pivot: v_rsq_f64 + one cubic correction (5) + scaling +
// (7 - j) multiply-adds = ~16 issue slots; the rank-8 update of the block's own columns is 64 more, without LDS reads.
// The bodies are generated (tools/gen_diag_asm.py) and kept as one asm statement per block so that the hazards of DPP
// (two wait states after a vector write of any register it reads) and of transcendental results (one) are under control.
// GENERATED by tools/gen_diag_asm.py -------------------------------------------------------------------------------
#define P3_ASM_FIRST \
    "s_nop 1\n" \
    "v_mov_b64_dpp %8, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %0, %0, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %1, -%0, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%0, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%0, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%0, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%0, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %1, %1, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %2, -%1, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%1, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%1, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%1, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %2, %2, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %3, -%2, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%2, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%2, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%2, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%2, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %3, %3, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %4, -%3, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%3, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%3, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%3, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %4, %4, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %5, -%4, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%4, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%4, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %5, %5, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %6, -%5, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%5, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 0\n" \
    "v_mov_b64_dpp %8, %6 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %6, %6, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %7, -%6, %6 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 1\n" \
    "v_mov_b64_dpp %8, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %13, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %7, %7, %9\n"

#define P3_ASM_UPDATE \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %0, -%13, %13 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%13, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%13, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%13, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%13, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%13, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%13, %13 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%13, %13 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%14, %14 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%14, %14 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%14, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%14, %14 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%14, %14 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%14, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%14, %14 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%14, %14 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%15, %15 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%15, %15 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%15, %15 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%15, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%15, %15 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%15, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%15, %15 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%15, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%16, %16 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%16, %16 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%16, %16 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%16, %16 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%16, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%16, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%16, %16 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%16, %16 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%17, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%17, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%17, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%17, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%17, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%17, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%17, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%17, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%18, %18 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%18, %18 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%18, %18 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%18, %18 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%18, %18 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%18, %18 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%18, %18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%18, %18 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%19, %19 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%19, %19 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%19, %19 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%19, %19 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%19, %19 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%19, %19 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%19, %19 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%19, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %0, -%20, %20 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %1, -%20, %20 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%20, %20 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%20, %20 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%20, %20 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%20, %20 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%20, %20 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%20, %20 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %0, %0, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %1, -%0, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %2, -%0, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%0, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%0, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%0, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %1, %1, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %2, -%1, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %3, -%1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%1, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%1, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%1, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %2, %2, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %3, -%2, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %4, -%2, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%2, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%2, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%2, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %3, %3, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %4, -%3, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %5, -%3, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%3, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%3, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %4, %4, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %5, -%4, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %6, -%4, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%4, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b64_dpp %8, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %5, %5, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %6, -%5, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %7, -%5, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 0\n" \
    "v_mov_b64_dpp %8, %6 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %6, %6, %9\n" \
    "s_nop 1\n" \
    "v_fmac_f64_dpp %7, -%6, %6 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "s_nop 1\n" \
    "v_mov_b64_dpp %8, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
    "v_rsq_f64 %9, %8\n" \
    "s_nop 0\n" \
    "v_mul_f64 %11, %8, %9\n" \
    "v_fma_f64 %10, -%11, %9, 1.0\n" \
    "v_fma_f64 %12, %10, %21, 0.5\n" \
    "v_mul_f64 %11, %9, %10\n" \
    "v_fma_f64 %9, %11, %12, %9\n" \
    "v_mul_f64 %7, %7, %9\n"

// END GENERATED -----------------------------------------------------------------------------------------------------

template <bool UPDATE>
__device__ __forceinline__ void p3_diag_block(double (&acc)[8], const double (&prev)[8]) {
    double D, R, E, T1, T2;
    const double c375 = 0.375;
    if (UPDATE) {
        asm volatile(P3_ASM_UPDATE
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                       "=&v"(D), "=&v"(R), "=&v"(E), "=&v"(T1), "=&v"(T2)
                     : "v"(prev[0]), "v"(prev[1]), "v"(prev[2]), "v"(prev[3]), "v"(prev[4]), "v"(prev[5]), "v"(prev[6]), "v"(prev[7]),
                       "v"(c375));
    } else {
        asm volatile(P3_ASM_FIRST
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                       "=&v"(D), "=&v"(R), "=&v"(E), "=&v"(T1), "=&v"(T2)
                     : "v"(c375));
    }
}

// ---- inverse blocks for free, and the progressive hand-off of the diagonal tile (panel kernels) --------------------------------
// (1) The strips need the inverses W_b of the four diagonal 16 x 16 blocks of the tile (p2_strip).  They used to be computed after
// the factorisation (p2_inverse_blocks: 8 x 8 substitutions + a coupling product, ~1 us on the chain of every 64 columns).  Now
// they fall out of the factorisation itself: a row e_i appended to the tile is turned by the very column operations of the
// factorisation into  e_i L^-T = column i of L^-1,  and wave 0 has idle lanes to carry such rows - lanes 8-15 of its DPP row R hold
// tile rows 8 R .. 8 R + 7, which are dead (above the block, or the replicated rows themselves) from round R on.  Block b (rounds
// 2 b and 2 b + 1) uses DPP rows 2 (b & 1) ("A": e_0 .. e_7, carried through both rounds) and 2 (b & 1) + 1 ("B": e_8 .. e_15,
// second round only); at the end of round 2 b + 1 they write W_b.  Same instruction stream, no extra instructions on the chain.
// (2) The team row that factors NEXT needs this tile for the last strip of its row block, and that strip consumes the tile in four
// 16-column blocks (p2_strip_step).  Block column b - rows 16 b .. 63 of columns 16 b .. 16 b + 15 with W_b and its refinement flag -
// is complete in LDS at the end of round 2 b + 1: wave 3 stores it (write-through) in round 2 b + 2 and announces it in round
// 2 b + 3, while the later columns are still being factored; the follower's strip, its share of X X^T and the hand-off latency run
// beside the factorisation instead of behind it.  The progress word counts QUARTERS of a column block for that reason.
// (The replicated rows of a round reach the tile in LDS one round late - see below; wave 0 also leaves them in the 8 x 8 scratch
// block S right away, which is where wave 3 takes the last diagonal 8 x 8 block of a block column from.)
constexpr int P3_ELD = 18;                 // row pitch of an identity block (even: 16-byte row loads; 16 rows x 18 doubles)
constexpr int P3_EBLK = 16 * P3_ELD;       // doubles per block; four blocks: 9 KB of the row block's (dead) second LDS tile

// one wave: W_b (rows = rows of the inverse) from E_b = W_b^T
__device__ __forceinline__ void p3_extract_inverse(double* __restrict__ T, const double* __restrict__ E, int b, int lane) {
    const double* Eb = E + b * P3_EBLK;
    double* Wb = T + p2_wblock(b);
    const int i = lane & 15, j0 = lane >> 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) Wb[(j0 + 4 * v) * PNL_LD + i] = Eb[i * P3_ELD + j0 + 4 * v];
}

struct P3Publish {
    double* dst = nullptr;             // the tile in global memory (row r0, column of the diagonal block); nullptr: nothing is published here
    int ld = 0, rows = 0;              // leading dimension; valid rows from r0
    unsigned long long* word = nullptr;
    unsigned long long base = 0ull;    // word value that means "every strip of this row block is out" (4 * trow)
    bool add = false;                  // split team: the word is a sum fed by several workgroups - announce by +1 instead of a value
};

// rotate a double by `ROR` lanes within every row of 16 lanes (two 32-bit DPP moves: no LDS crossbar round trip)
template <int ROR>
__device__ __forceinline__ double p3_row_ror(double v) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, 0x120 + ROR, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), 0x120 + ROR, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// refinement flag of diagonal block b (see P2_REFINE_RATIO): one wave; S != nullptr: its last eight pivots are still on their way
// to the tile and are read from the scratch block
__device__ __forceinline__ void p3_block_flag(double* __restrict__ T, const double* __restrict__ S, int b, int lane) {
    const int l = lane & 15;
    double d = (S && l >= 8) ? S[(l - 8) * 8 + (l - 8)] : T[(16 * b + l) * PNL_LD + 16 * b + l];
    double mn = d, mx = d;
    mn = fmin(mn, p3_row_ror<8>(mn)); mx = fmax(mx, p3_row_ror<8>(mx));
    mn = fmin(mn, p3_row_ror<4>(mn)); mx = fmax(mx, p3_row_ror<4>(mx));
    mn = fmin(mn, p3_row_ror<2>(mn)); mx = fmax(mx, p3_row_ror<2>(mx));
    mn = fmin(mn, p3_row_ror<1>(mn)); mx = fmax(mx, p3_row_ror<1>(mx));
    if (lane == 0) T[p2_flag_slot(b)] = (mx > P2_REFINE_RATIO * mn) ? 1.0 : 0.0;
}

// wave-wide (64 lanes): block column b of the lower-triangular tile T, the inverse block W_b and its refinement flag, write-through
// (a team row's diagonal tile is whole: no row test; rows of the diagonal 16 x 16 block store their pairs up to the diagonal)
__device__ __forceinline__ void p3_store_column(const double* __restrict__ T, const double* __restrict__ S, int b, int lane, const P3Publish& pub) {
    const int r8 = lane >> 3, c2 = (lane & 7) * 2;
    pan_d2 v[8], wv[2];
    const int cc = 16 * b + c2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = min(16 * b + 8 * i + r8, 63);
        v[i] = *reinterpret_cast<const pan_d2*>(T + r * PNL_LD + cc);
    }
    if (S && c2 >= 8) v[1] = *reinterpret_cast<const pan_d2*>(S + r8 * 8 + (c2 - 8));   // rows 16 b + 8 .., columns 16 b + 8 ..: the late block
#pragma unroll
    for (int q = 0; q < 2; ++q) wv[q] = *reinterpret_cast<const pan_d2*>(T + (16 * (b >> 1) + 8 * q + r8) * PNL_LD + 32 + 16 * (b & 1) + c2);
    const double fl = T[p2_flag_slot(b)];
    double* dst = pub.dst + (size_t)(16 * b + r8) * pub.ld + cc;
    const size_t step = (size_t)8 * pub.ld;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (16 * b + 8 * i < 64 && (i >= 2 || c2 <= 8 * i + r8))
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v[i]) : "memory");
        dst += step;
    }
    double* wd = pub.dst + (size_t)(16 * (b >> 1) + r8) * pub.ld + 32 + 16 * (b & 1) + c2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(wd), "v"(wv[q]) : "memory");
        wd += step;
    }
    if (lane == 0) {
        double* fd = pub.dst + (size_t)pub.ld + 16 + b;
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(fd), "v"(fl) : "memory");
    }
}

// The rank-8 updates of round JB of p3_diag:  T[mi][ni] -= L[mi][JB - 1] L[ni][JB - 1]^T  (K = 8) for the 16 x 16 blocks that reach
// columns >= 8 JB + 8 - 6, 6, 3, 3, 1, 1, 0 of them for JB = 1 .. 7, in the order (1,1) (2,1) (3,1) (2,2) (3,2) (3,3) (the lists of the
// later rounds are tails of it).  This instance takes the blocks FIRST, FIRST + STRIDE, ...; everything about the blocks is a
// compile-time constant - as run-time loops over a block table this section was ~200 instructions and several exec-mask loops per
// block, and it, not the factoring waves (0.65 us per round), set the length of rounds 1 - 4 (1.1 us with three blocks per wave).
// Up to three blocks are loaded together, multiplied, stored: one LDS round trip per batch.
template <int JB, int FIRST, int STRIDE>
__device__ __forceinline__ void p3_update_round(double* __restrict__ T, int l15, int lk) {
    constexpr int c0 = 8 * JB + 8, n0 = c0 >> 4, nblk = (4 - n0) * (5 - n0) / 2, g0 = n0 == 1 ? 0 : (n0 == 2 ? 3 : 5);
    constexpr int mine = FIRST < nblk ? (nblk - FIRST + STRIDE - 1) / STRIDE : 0;
    const int kb = 8 * (JB - 1) + lk;
#pragma unroll
    for (int bt = 0; bt < (mine + 2) / 3; ++bt) {
        pan_d4 c[3];
        double a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int k = 3 * bt + sl;
            if (k < mine) {
                const int g = g0 + FIRST + STRIDE * k, mi = (4025 >> (2 * g)) & 3, ni = (3733 >> (2 * g)) & 3;
                const double* C = &T[(16 * mi + lk) * PNL_LD + 16 * ni + l15];
                c[sl] = pan_d4{C[0], C[4 * PNL_LD], C[8 * PNL_LD], C[12 * PNL_LD]};
                const double* a = &T[(16 * mi + l15) * PNL_LD + kb];
                const double* b = &T[(16 * ni + l15) * PNL_LD + kb];
                a0[sl] = -a[0]; a1[sl] = -a[4]; b0[sl] = b[0]; b1[sl] = b[4];
            }
        }
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int k = 3 * bt + sl;
            if (k < mine) {
                c[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[sl], b0[sl], c[sl], 0, 0, 0);
                c[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[sl], b1[sl], c[sl], 0, 0, 0);
            }
        }
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int k = 3 * bt + sl;
            if (k < mine) {
                const int g = g0 + FIRST + STRIDE * k, mi = (4025 >> (2 * g)) & 3, ni = (3733 >> (2 * g)) & 3;
                double* C = &T[(16 * mi + lk) * PNL_LD + 16 * ni + l15];
                if (16 * ni >= c0 || 16 * ni + l15 >= c0) { C[0] = c[sl][0]; C[4 * PNL_LD] = c[sl][1]; C[8 * PNL_LD] = c[sl][2]; C[12 * PNL_LD] = c[sl][3]; }
            }
        }
    }
}

// wave `w` (2 or 3) in round jb; solo: wave 2 takes every block (wave 3 is publishing)
__device__ __forceinline__ void p3_update(double* __restrict__ T, int jb, int w, bool solo, int l15, int lk) {
#define P3_UPDATE_CASE(JB)                                                       \
    case JB:                                                                     \
        if (solo) { if (w == 2) p3_update_round<JB, 0, 1>(T, l15, lk); }         \
        else if (w == 2) p3_update_round<JB, 0, 2>(T, l15, lk);                  \
        else p3_update_round<JB, 1, 2>(T, l15, lk);                              \
        break;
    switch (jb) {
        P3_UPDATE_CASE(1) P3_UPDATE_CASE(2) P3_UPDATE_CASE(3) P3_UPDATE_CASE(4) P3_UPDATE_CASE(5) P3_UPDATE_CASE(6)
        default: break;
    }
#undef P3_UPDATE_CASE
}

template <bool STAMP = false>
__device__ __forceinline__ void p3_diag(double* __restrict__ T, int col0, const PanelArgs& p, int t, long long* __restrict__ st = nullptr,
                                        double* __restrict__ S = nullptr, const P3Publish pub = P3Publish(), double* __restrict__ E = nullptr) {
    // (the wave index as a SCALAR: every `if (w == ..)` below is then a scalar branch, and the block tables of the rank-8 updates
    // scalar arithmetic - as a vector value the compiler wrapped them in exec-mask loops, ~200 instructions per 16 x 16 block)
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), l15 = lane & 15, lk = lane >> 4;
    const bool progressive = pub.dst != nullptr && S != nullptr;
    // dev aid (tools/time_panel2.hip): column 9 = the publishing wave's phases, column 10 = end of every round
#define P3_STAMP(col, k)                                                                                             \
    do {                                                                                                              \
        if (p.stamps) p.stamps[((size_t)(col0 / 64 % 16) * 17 + (col)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
    double held[8];   // threads 0-7: the replicated rows' results of the previous round, not yet in the tile (see below)
#pragma unroll
    for (int q = 0; q < 8; ++q) held[q] = 0.0;
    // the identity rows (see (1) above) live in E: four 16 x 16 blocks, E_b = I before block b's two rounds, W_b^T after them
    if (E) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + 256 * q, b = e >> 8, i = (e >> 4) & 15, j = e & 15;
            E[b * P3_EBLK + i * P3_ELD + j] = (i == j) ? 1.0 : 0.0;
        }
        __syncthreads();
    }
    for (int jb = 0; jb < 8; ++jb) {
        // (wave 1 owns rows 32-63: nothing of it is left below the last block)
        if (w == 0 || (w == 1 && jb < 7)) {
            if (STAMP && t == 0) st[3 * jb] = (long long)__builtin_readcyclecounter();
            const int row = l15 < 8 ? 8 * jb + l15 : 32 * w + 8 * lk + (l15 - 8);
            // where this lane's eight values of column block jb (src) and jb - 1 (psrc) are, and whether it writes them back: a row
            // of the tile - or, for wave 0's lanes 8-15 of DPP rows ga (both rounds of block bq) and ga + 1 (second round), which
            // hold dead rows by then, row e of E_bq (psrc in the first round: a row whose first half is and stays zero)
            const int bq = jb >> 1, half = jb & 1, ga = 2 * (bq & 1), ii = l15 - 8;
            const bool ident = E != nullptr && w == 0 && l15 >= 8 && (lk == ga || (half && lk == ga + 1));
            const int erow = (lk == ga ? 0 : 8) + ii;
            const double* src = &T[row * PNL_LD + 8 * jb];
            const double* psrc = &T[row * PNL_LD + 8 * (jb - 1)];
            bool wr = l15 >= 8 ? (row > 8 * jb + 7) : (t < 8 && jb == 7);
            if (ident) {
                const double* Eb = E + bq * P3_EBLK;
                src = Eb + erow * P3_ELD + 8 * half;
                psrc = half ? Eb + erow * P3_ELD : Eb + (8 + ii) * P3_ELD;
                wr = true;
            }
            double acc[8], prev[8];
            {
                const pan_d2* sv = reinterpret_cast<const pan_d2*>(src);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const pan_d2 v = sv[q]; acc[2 * q] = v[0]; acc[2 * q + 1] = v[1]; }
            }
            if (jb > 0) {
                const pan_d2* ps = reinterpret_cast<const pan_d2*>(psrc);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const pan_d2 v = ps[q]; prev[2 * q] = v[0]; prev[2 * q + 1] = v[1]; }
                // The replicated rows of round jb - 1 go to LDS only now.  The two factoring waves do not synchronise inside a
                // round, and beside a trailing-update workgroup one of them can fall a whole round behind: written at the end
                // of round jb - 1, these rows reached LDS before wave 1 had read them at the START of that round (observed:
                // one tile in ~2000, only with a SYRK co-resident).  Nobody reads them during round jb.
                if (t < 8) {
                    pan_d2* hd = reinterpret_cast<pan_d2*>(&T[(8 * (jb - 1) + t) * PNL_LD + 8 * (jb - 1)]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) hd[q] = pan_d2{held[2 * q], held[2 * q + 1]};
                }
                p3_diag_block<true>(acc, prev);
            } else {
                p3_diag_block<false>(acc, prev);
            }
            // rows below the block: written by their only holder (rows above it carry the strict upper triangle's scratch
            // through the same arithmetic and are dropped); the replicated rows: held by wave 0's first DPP row - in the last
            // round, which wave 1 sits out, written at once; identity rows: back into E
            if (wr) {
                pan_d2* dst = reinterpret_cast<pan_d2*>(const_cast<double*>(src));
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = pan_d2{acc[2 * q], acc[2 * q + 1]};
            }
            if (t < 8 && S && half) {   // (the publishing wave reads the block's last diagonal 8 x 8 block from here in the NEXT round;
                                        // written in odd rounds only, so that it is not overwritten while being read)
                pan_d2* sd = reinterpret_cast<pan_d2*>(&S[t * 8]);
#pragma unroll
                for (int q = 0; q < 4; ++q) sd[q] = pan_d2{acc[2 * q], acc[2 * q + 1]};
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) held[q] = acc[q];
            if (STAMP && t == 0) st[3 * jb + 2] = (long long)__builtin_readcyclecounter();
        } else if (w >= 2) {
            if (progressive && w == 3 && jb >= 2 && (jb & 1) == 0) {   // the publishing wave: block column (jb - 2) / 2 goes out
                if (lane == 0) P3_STAMP(9, jb - 2);
                const int b = (jb - 2) >> 1;
                p3_extract_inverse(T, E, b, lane);
                p3_block_flag(T, S, b, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                p3_store_column(T, S, b, lane, pub);
            }
            p3_update(T, jb, w, progressive && jb >= 2 && (jb & 1) == 0, l15, lk);
            if (progressive && w == 3 && jb >= 3 && (jb & 1)) {
                // the block column stored in the round before: its stores are acknowledged by now (the wait sits at the END of the
                // round: at its start it stalled the round's barrier for ~0.7 us)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    if (pub.add) __hip_atomic_fetch_add(pub.word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_store(pub.word, pub.base + (unsigned long long)((jb - 1) >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (lane == 0 && w >= 1) P3_STAMP(10 + w, jb);   // (dev aid: when waves 1, 2, 3 reach the round's barrier)
        __syncthreads();
        if (t == 0) P3_STAMP(10, jb);
    }
#undef P3_STAMP
}

// log-determinant share and LAPACK-style info of a factored diagonal tile (wave 0; off the hand-off chain: called after the tile
// has been published)
__device__ __forceinline__ void p3_diag_logdet(const double* __restrict__ T, int col0, const PanelArgs& p, int t) {
    if (t < 64) {
        const int lane = t;
        const double mydiag = T[lane * PNL_LD + lane];
        const unsigned long long badmask = __ballot(!(mydiag > 0.0));
        double ld = 2.0 * log(mydiag);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
        if (t == 0) {
            if (p.logdet) atomicAdd(p.logdet, ld);
            if (badmask && p.info) atomicCAS(p.info, 0, col0 + __builtin_ctzll(badmask) + 1);
        }
    }
}

// Tiles (mi, ni), mi >= ni, of a 64 x 64 lower triangle dealt over the four waves: slot q of wave w.
//   w0: (0,0) (1,0) (2,0)    w1: (1,1) (2,1) (3,1)    w2: (2,2) (3,2)    w3: (3,3) (3,0)
// Every tile belongs to a wave that holds one of its two row groups (wave w owns rows 16 w .. 16 w + 15 of the row block): in the
// progressive last strip that operand comes straight from the strip's registers (the T layout IS an MFMA operand), the diagonal
// tile (w, w) needs no LDS read at all.  Waves 0 - 2 hold the N side (ni = w), wave 3's second tile the M side (mi = 3).
__device__ __forceinline__ int p2_dtile_m(int w, int q) { return q == 0 ? w : (w == 0 ? q : (w == 1 ? q + 1 : 3)); }
__device__ __forceinline__ int p2_dtile_n(int w, int q) { return q == 0 ? w : (w == 3 ? 0 : w); }

// ---- progress words: prog[t] = 4 x the number of column blocks team row t has completed and published (u strips, then the
// diagonal block: 4 (t + 1) means L[t][t] is out; 4 t + b, b = 1 .. 3: the first b block columns of the diagonal tile with their
// inverse blocks are out - the progressive hand-off of p3_diag).  They live where the first generation kept its flags (pnl_flag).
__device__ __forceinline__ void p2_publish(const PanelArgs& p, int trow, unsigned long long value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pnl_flag(p, trow), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns once prog[trow] >= need.  `seen` (LDS) caches the last value read per team row, so a satisfied wait costs one
// LDS read and one barrier.  All threads call it; the barrier(s) inside also separate the LDS phases around it.
__device__ __forceinline__ void p2_wait(const PanelArgs& p, unsigned long long* __restrict__ seen, int trow, unsigned long long need) {
    const bool ok = seen[trow] >= need;
    __syncthreads();   // everybody has read the cached value before anybody may overwrite it
    if (ok) return;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long v;
        while ((v = __hip_atomic_load(pnl_flag(p, trow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > PNL_SPIN_LIMIT) {
                if (p.info) atomicCAS(p.info, 0, -77);
                v = need;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        seen[trow] = v;
    }
    __syncthreads();
}

// ---- the split team (progressive mode, S <= 8) ------------------------------------------------------------------------------------
// A team row t used to be ONE workgroup working through its columns 0 .. t - 1 left to right: the sums of column c are c tile
// products that can only start when the row is done with column c - 1, so a late row (t = 6, 7) enters its LAST column - the strip
// the next diagonal tile waits for - with five or six products (3 us each) still to do, and from the fifth step on those, not the
// diagonal tile, paced the chain (tools/time_panel2: steps 14, 14, 16, 16, 19, 21, 24 us).  In a latency-bound panel the chip is
// idle, so the team is split by TILE: tile (t, c), c <= t - 2, is a workgroup of its own (p2_team_tile: product u as soon as
// X[t][u] and L[c][u] exist - one per step of the chain -, then the strip on diag(c)), and row t's CHAIN workgroup keeps the last
// column and the diagonal tile: per step one product for column t - 1 and one for D = X X^T from the same staged operand.  Nobody
// has more than two tile products per step.  Dispatch order, column by column: chain(c), then tiles (c + 2 .. S - 1, c) - every
// wait is on a lower index, as before.  Words: tile (t, c) raises flag tf(t, c) when X[t][c] is out, for the tile-level waits; the
// row's progress word is the SUM of +4 per published strip and +1 per published block column of the diagonal tile (atomic adds:
// several workgroups feed it), which is what the bulk row blocks wait on, unchanged.
__device__ __forceinline__ int p2_team_count(int S, int split) { return split ? S + (S - 1) * (S - 2) / 2 : S; }

// team workgroup tw -> chain of row t (c = -1) or tile (t, c)
__device__ __forceinline__ void p2_team_decode(int S, int tw, int& t, int& c) {
    int col = 0, off = tw;
    while (off >= 1 + max(0, S - 2 - col)) { off -= 1 + max(0, S - 2 - col); ++col; }
    if (off == 0) { t = col; c = -1; }
    else { t = col + 1 + off; c = col; }
}

// flag of tile (t, c), c < t <= 7: rows 2 and 3 of the panel's first diagonal tile, columns 9 .. 31 (strictly upper; zeroed with
// the progress words by potrf_zero_flags; clear of the progress words in row 0, the refinement flags in row 1, the words of
// potrf_group_kernel in column 8 and the inverse blocks from column 32 on)
__device__ __forceinline__ unsigned long long* p2_tile_flag(const PanelArgs& p, int t, int c) {
    const int q = t * (t - 1) / 2 + c;
    const int r = q < 23 ? 2 : 3, col = q < 23 ? 9 + q : 9 + q - 23;
    return reinterpret_cast<unsigned long long*>(p.A + (size_t)(p.k0 + r) * p.lda + p.k0 + col);
}

// all threads; returns once *word >= need (bounded like p2_wait) with this compute unit's stale lines dropped
__device__ __forceinline__ void grp_wait(unsigned long long* word, unsigned long long need, int* info) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > PNL_SPIN_LIMIT) {
                if (info) atomicCAS(info, 0, -77);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// all threads; returns once both words are >= need, with the smaller of the two values seen (the same for every thread: through
// `sh`, an LDS word nobody else writes before the workgroup's next barrier)
__device__ __forceinline__ unsigned long long grp_wait2(unsigned long long* wi, unsigned long long* wj, unsigned long long need, int* info,
                                                        unsigned long long* sh) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long a, b;
        while ((a = __hip_atomic_load(wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > PNL_SPIN_LIMIT) {
                if (info) atomicCAS(info, 0, -77);
                a = need;
                break;
            }
        }
        b = a;
        if (wj != wi)
            while ((b = __hip_atomic_load(wj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > PNL_SPIN_LIMIT) {
                    if (info) atomicCAS(info, 0, -77);
                    b = need;
                    break;
                }
            }
        *sh = a < b ? a : b;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return *sh;
}

// tile (t, c) of the split team: X[t][c] = (A[t][c] - sum_{u<c} X[t][u] L[c][u]^T) L[c][c]^-T, published
__device__ __forceinline__ void p2_team_tile(const PanelArgs& p, int trow, int c, double* __restrict__ psm) {
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    double* A = p.A;
    const int r0 = p.k0 + 64 * trow, l0 = p.k0 + 64 * c;
    pan_d4 acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0.0, 0.0, 0.0, 0.0};
    pan_d2 ct[8];   // the tile itself (last written by an earlier launch on the stream, or by the update tiles of a fused launch that
                    // this row has been waiting for)
    p2_gload(A, p.lda, p.N, r0, l0, t, ct);
    for (int u = 0; u < c; ++u) {
        grp_wait(p2_tile_flag(p, trow, u), 1ull, p.info);
        grp_wait(p2_tile_flag(p, c, u), 1ull, p.info);
        pan_d2 xa[8], la[8];
        p2_gload(A, p.lda, p.N, r0, p.k0 + 64 * u, t, xa);
        p2_gload(A, p.lda, p.N, l0, p.k0 + 64 * u, t, la);
        p2_sstore(Cs, t, la);
        p2_sstore(Xs, t, xa);
        __syncthreads();
        p2_chunk(Cs, Xs, acc, w, l15, lk);
        __syncthreads();
    }
    p2_sstore(Xs, t, ct);
    __syncthreads();
    pan_d4 T[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int v = 0; v < 4; ++v) T[mi][v] = Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] - acc[mi][v];
    grp_wait(pnl_flag(p, c), 4ull * (unsigned long long)c + 4ull, p.info);   // L[c][c] with its inverse blocks and flags
    {
        pan_d2 lt[8];
        p2_gload(A, p.lda, p.N, l0, l0, t, lt);
        p2_sstore(Cs, t, lt);
    }
    __syncthreads();
    p2_strip(Cs, T, l15, lk);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] = T[mi][v];
    __syncthreads();
    p2_gstore(A, p.lda, p.N, r0, l0, Xs, t, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_fetch_add(pnl_flag(p, trow), 4ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p2_tile_flag(p, trow, c), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// p2_wait without the acquire fence: for data that is then read with L1-bypassing (sc1) loads only - the block columns of a
// diagonal tile in the progressive last strip; their producer stored them write-through
__device__ __forceinline__ void p2_wait_nofence(const PanelArgs& p, unsigned long long* __restrict__ seen, int trow, unsigned long long need) {
    const bool ok = seen[trow] >= need;
    __syncthreads();
    if (ok) return;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        unsigned long long v;
        while ((v = __hip_atomic_load(pnl_flag(p, trow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PNL_SPIN_LIMIT) {
                if (p.info) atomicCAS(p.info, 0, -77);
                v = need;
                break;
            }
        }
        seen[trow] = v;
    }
    __syncthreads();
}

// 16 bytes past this compute unit's L1 (two 8-byte relaxed agent-scope loads = global_load_dwordx2 sc1, which the compiler counts)
__device__ __forceinline__ pan_d2 p2_load_sc1(const double* __restrict__ src) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(src);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return pan_d2{__longlong_as_double((long long)a), __longlong_as_double((long long)b)};
}

// Tile (rb, c) of a BULK row block that is a team row of the NEXT panel of a fused launch (potrf_group_kernel): the same task as
// p2_team_tile for a row below the diagonal block.  As one workgroup per row block these rows do their left-looking sums column by
// column and finish ~25 us behind the chain - and the next panel's team is exactly these rows: its chain started ~45 us after the
// previous one's ended.  By tile, product u runs as soon as X[rb][u] and L[c][u] exist (one per step of the chain) and the row is
// complete one strip after the last diagonal tile.  The row's progress word keeps its meaning - column blocks published IN
// ORDER - because a tile announces itself only once its left neighbour has (which, one step of the chain earlier, it has).
__device__ __forceinline__ void p2_bulk_tile(const PanelArgs& p, int rb, int c, unsigned long long* word, unsigned long long base,
                                             double* __restrict__ psm) {
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    double* A = p.A;
    const int r0 = p.k0 + 64 * rb, l0 = p.k0 + 64 * c;
    pan_d4 acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0.0, 0.0, 0.0, 0.0};
    pan_d2 ct[8];
    p2_gload(A, p.lda, p.N, r0, l0, t, ct);
    for (int u = 0; u < c; ++u) {
        grp_wait(word, base + (unsigned long long)u + 1ull, p.info);                       // X[rb][u]
        if (p.split) grp_wait(p2_tile_flag(p, c, u), 1ull, p.info);                       // L[c][u] (split team: by tile)
        else grp_wait(pnl_flag(p, c), 4ull * (unsigned long long)u + 4ull, p.info);       // (one workgroup per team row: in order)
        pan_d2 xa[8], la[8];
        p2_gload(A, p.lda, p.N, r0, p.k0 + 64 * u, t, xa);
        p2_gload(A, p.lda, p.N, l0, p.k0 + 64 * u, t, la);
        p2_sstore(Cs, t, la);
        p2_sstore(Xs, t, xa);
        __syncthreads();
        p2_chunk(Cs, Xs, acc, w, l15, lk);
        __syncthreads();
    }
    p2_sstore(Xs, t, ct);
    __syncthreads();
    pan_d4 T[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int v = 0; v < 4; ++v) T[mi][v] = Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] - acc[mi][v];
    grp_wait(pnl_flag(p, c), 4ull * (unsigned long long)c + 4ull, p.info);
    {
        pan_d2 lt[8];
        p2_gload(A, p.lda, p.N, l0, l0, t, lt);
        p2_sstore(Cs, t, lt);
    }
    __syncthreads();
    p2_strip(Cs, T, l15, lk);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] = T[mi][v];
    __syncthreads();
    p2_gstore(A, p.lda, p.N, r0, l0, Xs, t, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c > 0) grp_wait(word, base + (unsigned long long)c, p.info);   // the left neighbour has announced itself (barrier inside)
    else __syncthreads();
    if (t == 0) __hip_atomic_store(word, base + (unsigned long long)c + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One row block of the panel / of a triangular-solve block.
//   L (ldl, lrows rows): the triangular factor's tiles, read at rows lr0 + 64 c, columns lc0 + 64 u;
//   B (ldb, brows rows): the right-hand sides / the panel's own rows, read and overwritten at rows r0, columns bc0 + 64 c.
// FLAGS: L is being produced by the team workgroups of the same launch (wait on progress words); TEAM: this row block is
// one of them (row block index trow): publishes its strips, accumulates its diagonal block and factors it.
template <bool FLAGS, bool TEAM>
__device__ __forceinline__ void p2_row_block(const PanelArgs& p, const double* __restrict__ L, int ldl, int lrows, int lr0, int lc0,
                                             double* __restrict__ B, int ldb, int brows, int r0, int bc0, int ncol, int ufirst,
                                             int trow, double* __restrict__ psm) {
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    unsigned long long* seen = reinterpret_cast<unsigned long long*>(psm + 2 * PNL_TILE);   // 16 words
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    if (FLAGS) {
        if (t < 16) seen[t] = 0ull;
        __syncthreads();
    }
    // TEAM: the ten 16 x 16 tiles on and below the diagonal of this row block's diagonal tile, dealt 3 / 3 / 2 / 2 over the
    // waves; slot q of wave w holds tile (mi, ni) = (p2_dtile_m, p2_dtile_n)
    pan_d4 dacc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) dacc[q] = pan_d4{0.0, 0.0, 0.0, 0.0};

    // dev aid (tools/time_panel2.hip): 100 MHz wall-clock stamps of the first 16 row blocks, normally off
#define P2_STAMP(col, k)                                                                                              \
    do {                                                                                                              \
        if (FLAGS && TEAM && p.stamps && t == 0)                                                                      \
            p.stamps[((size_t)trow * 17 + (col)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime();            \
    } while (0)
    const bool progressive = FLAGS && TEAM && p.progressive;
    const bool split = progressive && p.split;   // this workgroup is row trow's CHAIN workgroup: last column + diagonal tile only
    bool parked = false;   // TEAM: the row block's diagonal tile is in Cs and the last X is on its way out (progressive last strip)
    // D += X X^T for the tiles on and below the diagonal of this row block's diagonal tile, X = the tile in Xs
    auto daccum_from_xs = [&]() {
        const double* xa0 = Xs + (16 * p2_dtile_m(w, 0) + l15) * PNL_LD + lk;
        const double* xa1 = Xs + (16 * p2_dtile_m(w, 1) + l15) * PNL_LD + lk;
        const double* xa2 = Xs + (16 * p2_dtile_m(w, 2) + l15) * PNL_LD + lk;
        const double* xb0 = Xs + (16 * p2_dtile_n(w, 0) + l15) * PNL_LD + lk;
        const double* xb1 = Xs + (16 * p2_dtile_n(w, 1) + l15) * PNL_LD + lk;
        const double* xb2 = Xs + (16 * p2_dtile_n(w, 2) + l15) * PNL_LD + lk;
        double fa[2][3], fb[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { fa[0][q] = (q == 0 ? xa0 : q == 1 ? xa1 : xa2)[0]; fb[0][q] = (q == 0 ? xb0 : q == 1 ? xb1 : xb2)[0]; }
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const int cur = k4 & 1, nxt = cur ^ 1;
            if (k4 + 1 < 16) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    fa[nxt][q] = (q == 0 ? xa0 : q == 1 ? xa1 : xa2)[4 * (k4 + 1)];
                    fb[nxt][q] = (q == 0 ? xb0 : q == 1 ? xb1 : xb2)[4 * (k4 + 1)];
                }
            }
            dacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0], fb[cur][0], dacc[0], 0, 0, 0);
            dacc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1], fb[cur][1], dacc[1], 0, 0, 0);
            if (w < 2) dacc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][2], fb[cur][2], dacc[2], 0, 0, 0);
        }
    };
    for (int c = split ? max(ufirst, ncol - 1) : ufirst; c < ncol; ++c) {
        P2_STAMP(c, 0);
        pan_d4 acc[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0.0, 0.0, 0.0, 0.0};
        pan_d2 xa[8];   // the row block's own tile of column block u (operand of chunk u), finally of column block c itself
        if (split) {
            // chain workgroup of the split team: product u for the last column and for D as soon as tiles (trow, u) and (c, u) are out
            // (X[trow][u] is out ~3 us after diag(u); L[c][u], u = c - 1, is the last strip of the row before and arrives with the
            // start of diag(c): the D product runs while that is awaited, only the product for the last column behind it)
            pan_d2 xt[8];   // the tile to be solved: requested ahead of the products (nobody else writes it in this launch)
            p2_gload(B, ldb, brows, r0, bc0 + 64 * c, t, xt);
            for (int u = 0; u < c; ++u) {
                grp_wait(p2_tile_flag(p, trow, u), 1ull, p.info);
                p2_gload(B, ldb, brows, r0, bc0 + 64 * u, t, xa);
                p2_sstore(Xs, t, xa);
                __syncthreads();
                daccum_from_xs();
                grp_wait(p2_tile_flag(p, c, u), 1ull, p.info);
                pan_d2 la[8];
                p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * u, t, la);
                p2_sstore(Cs, t, la);
                __syncthreads();
                p2_chunk(Cs, Xs, acc, w, l15, lk);
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) xa[q] = xt[q];
        } else {
        p2_gload(B, ldb, brows, r0, bc0 + 64 * ufirst, t, xa);
        if (c > ufirst) {
            if (FLAGS) p2_wait(p, seen, c, 4ull * (unsigned long long)c);   // every L[c][u], u < c, is out
            pan_d2 la[8];
            p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * ufirst, t, la);
            for (int u = ufirst; u < c; ++u) {
                __syncthreads();   // the previous chunk's operand reads are done
                p2_sstore(Cs, t, la);
                p2_sstore(Xs, t, xa);
                __syncthreads();
                // next chunk's tiles in flight under this chunk's products; after the last chunk the row block's tile of
                // column block c (what is to be solved) arrives the same way (the L index is clamped: a harmless repeat)
                p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * min(u + 1, c - 1), t, la);
                p2_gload(B, ldb, brows, r0, bc0 + 64 * (u + 1), t, xa);
                __builtin_amdgcn_sched_barrier(0);   // the requests go out before the products, not in the middle of them
                p2_chunk(Cs, Xs, acc, w, l15, lk);
            }
        }
        }
        __syncthreads();
        P2_STAMP(c, 1);   // chunks done
        p2_sstore(Xs, t, xa);
        __syncthreads();
        pan_d4 T[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) T[mi][v] = Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] - acc[mi][v];
        const bool last = TEAM && c == ncol - 1;
        if (last && progressive) {
            // ---- the strip that the NEXT diagonal tile waits for, block column by block column behind the owner of L[c][c] (see
            // P3Publish): per 16-column block  wait -> its rows of the triangle + its inverse block into Cs -> strip step -> the
            // block's columns of X into Xs and out -> its K = 16 share of D = X X^T, operands from the strip's registers.
            // Same products in the same order as the one-piece path below: the same bits.
            pan_d2 dt[8];   // the row block's own diagonal tile: requested ahead of everything, rides in 32 registers
            p2_gload(B, ldb, brows, r0, bc0 + 64 * trow, t, dt);
            const double* Lt = L + (size_t)(lr0 + 64 * c) * ldl + lc0 + 64 * c;
            const int lvalid = lrows - (lr0 + 64 * c);   // valid rows of that tile (a team row's diagonal tile: 64)
            // rows 16 jb .. 63 of columns 16 jb .. 16 jb + 15 (512 pairs at most), W_jb (128 pairs), the refinement flag: requested
            // (fetch_issue, past the L1: the owner stored them write-through, so no acquire fence is needed) - possibly ahead of time,
            // under the strip steps of the blocks before, when the owner is that far ahead - and put into Cs (fetch_land).  The strip
            // takes the tile in two HALVES of two block columns: per step one wait, one landing, three barriers - taken block by block
            // (1.6 us per block, of which 0.5 arithmetic) the follower, not the factorisation, paced the chain.
            pan_d2 fv[2][3];
            double ffl[2] = {0.0, 0.0};
            auto fetch_issue = [&](int jb, int slot) {
                const int e0 = t, e1 = t + 256;
                const int ra = 16 * jb + (e0 >> 3), rb_ = 16 * jb + (e1 >> 3), cc = 16 * jb + (e0 & 7) * 2;
                fv[slot][0] = p2_load_sc1(Lt + (size_t)min(min(ra, 63), lvalid - 1) * ldl + cc);
                fv[slot][1] = p2_load_sc1(Lt + (size_t)min(min(rb_, 63), lvalid - 1) * ldl + cc);
                const int wr = 16 * (jb >> 1) + ((t & 127) >> 3), wc = 32 + 16 * (jb & 1) + (t & 7) * 2;
                fv[slot][2] = p2_load_sc1(Lt + (size_t)min(wr, lvalid - 1) * ldl + wc);
                if (t == 255)
                    ffl[slot] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(Lt + (size_t)min(1, lvalid - 1) * ldl + 16 + jb),
                                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            };
            auto fetch_land = [&](int jb, int slot) {
                const int e0 = t, e1 = t + 256;
                const int ra = 16 * jb + (e0 >> 3), rb_ = 16 * jb + (e1 >> 3), cc = 16 * jb + (e0 & 7) * 2;
                const int wr = 16 * (jb >> 1) + ((t & 127) >> 3), wc = 32 + 16 * (jb & 1) + (t & 7) * 2;
                if (ra < 64) *reinterpret_cast<pan_d2*>(Cs + ra * PNL_LD + cc) = fv[slot][0];
                if (rb_ < 64) *reinterpret_cast<pan_d2*>(Cs + rb_ * PNL_LD + cc) = fv[slot][1];
                if (t < 128) *reinterpret_cast<pan_d2*>(Cs + wr * PNL_LD + wc) = fv[slot][2];
                if (t == 255) Cs[p2_flag_slot(jb)] = ffl[slot];
            };
            bool ahead = false;   // the next half's data have been requested already
            auto xout = [&](int jb) {   // columns 16 jb .. + 15 of X: own rows -> Xs (all rows -> global after the next barrier)
#pragma unroll
                for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * jb + lk + 4 * v] = T[jb][v];
            };
            auto xstore = [&](int jb) {   // 64 rows x 16 columns, write-through: 512 pairs
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = t + 256 * q, r = e >> 3, cc = 16 * jb + (e & 7) * 2;
                    if (r0 + r < brows) {
                        const pan_d2 v = *reinterpret_cast<const pan_d2*>(Xs + r * PNL_LD + cc);
                        double* dst = B + (size_t)(r0 + r) * ldb + bc0 + 64 * c + cc;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v) : "memory");
                    }
                }
            };
            auto dself = [&](int jb) {   // the wave's diagonal tile (w, w): both operands are the strip's own registers
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) dacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(T[jb][k4], T[jb][k4], dacc[0], 0, 0, 0);
            };
            auto dcross = [&](int jb) {   // the other tiles: one operand from registers, the other wave's rows from Xs
#pragma unroll
                for (int q = 1; q < 3; ++q) {
                    if (q == 2 && w >= 2) continue;
                    const int other = (w == 3) ? p2_dtile_n(w, q) : p2_dtile_m(w, q);
                    const double* xo = Xs + (16 * other + l15) * PNL_LD + 16 * jb + lk;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const double o = xo[4 * k4];
                        dacc[q] = (w == 3) ? __builtin_amdgcn_mfma_f64_16x16x4f64(T[jb][k4], o, dacc[q], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f64_16x16x4f64(o, T[jb][k4], dacc[q], 0, 0, 0);
                    }
                }
            };
#define P2_PROGRESSIVE_STEP(JA)                                                                                  \
            if (!ahead) {                                                                                        \
                p2_wait_nofence(p, seen, c, 4ull * (unsigned long long)c + (unsigned long long)(JA) + 2ull);     \
                fetch_issue(JA, 0);                                                                              \
                fetch_issue((JA) + 1, 1);                                                                        \
            }                                                                                                    \
            if ((JA) == 0) P2_STAMP(c, 2);                                                                       \
            if ((JA) == 2) P2_STAMP(c, 3);                                                                       \
            fetch_land(JA, 0);                                                                                   \
            fetch_land((JA) + 1, 1);                                                                             \
            __syncthreads();                                                                                     \
            ahead = (JA) == 0 && seen[c] >= 4ull * (unsigned long long)c + 4ull;                                 \
            if (ahead) {                                                                                         \
                fetch_issue(2, 0);                                                                               \
                fetch_issue(3, 1);                                                                               \
            }                                                                                                    \
            p2_strip_step<JA>(Cs, T, l15, lk);                                                                   \
            p2_strip_step<(JA) + 1>(Cs, T, l15, lk);                                                             \
            xout(JA);                                                                                            \
            xout((JA) + 1);                                                                                      \
            dself(JA);                                                                                           \
            dself((JA) + 1);                                                                                     \
            __syncthreads();                                                                                     \
            if ((JA) == 2) P2_STAMP(c, 4);                                                                       \
            xstore(JA);                                                                                          \
            xstore((JA) + 1);                                                                                    \
            dcross(JA);                                                                                          \
            dcross((JA) + 1);
            P2_PROGRESSIVE_STEP(0)
            P2_PROGRESSIVE_STEP(2)
#undef P2_PROGRESSIVE_STEP
            P2_STAMP(c, 5);   // diagonal-tile accumulation done
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // X is out (the last half's stores rode out its cross products) ...
            __syncthreads();  // ... and every wave is done with the triangle in Cs
            if (t == 0) {     // announced at once: the next row's chain workgroup takes X as the operand of its last product
                if (split) {
                    __hip_atomic_fetch_add(pnl_flag(p, trow), 4ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p2_tile_flag(p, trow, ncol - 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    __hip_atomic_store(pnl_flag(p, trow), 4ull * (unsigned long long)ncol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            p2_sstore(Cs, t, dt);
            parked = true;
            P2_STAMP(c, 6);
            continue;
        }
        // the triangle to solve against
        if (FLAGS) p2_wait(p, seen, c, 4ull * (unsigned long long)c + 4ull);
        else __syncthreads();
        P2_STAMP(c, 2);   // triangle available
        {
            pan_d2 lt[8];
            p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * c, t, lt);
            p2_sstore(Cs, t, lt);
        }
        __syncthreads();
        if (!FLAGS) {   // a team workgroup publishes the inverses with its diagonal tile (they arrived with the load above)
            p2_inverse_blocks(Cs, w, lane);
            __syncthreads();
        }
        P2_STAMP(c, 3);   // triangle in LDS
        p2_strip(Cs, T, l15, lk);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] = T[mi][v];
        __syncthreads();
        P2_STAMP(c, 4);   // strip done
        // A team row block's last strip: its diagonal tile is requested BEFORE the write-through stores of X (memory
        // operations return in order: requested behind them it would arrive with their acknowledgement) and rides out the
        // accumulation below in 32 registers.
        pan_d2 dt[8];
        if (last) p2_gload(B, ldb, brows, r0, bc0 + 64 * trow, t, dt);
        p2_gstore(B, ldb, brows, r0, bc0 + 64 * c, Xs, t, TEAM);
        if (TEAM) {
            daccum_from_xs();
            P2_STAMP(c, 5);   // diagonal-tile accumulation done
            if (!last) {
                p2_publish(p, trow, 4ull * (unsigned long long)c + 4ull);
            } else {
                // the tile is parked in Cs (the triangle is dead: every wave is past the barrier that ended the strip); X is
                // published after the barrier that the assembly below needs anyway
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                p2_sstore(Cs, t, dt);
            }
            P2_STAMP(c, 6);   // published
        }
    }
    if (TEAM) {
        P2_STAMP(trow, 0);
        if (ncol == 0) {   // (otherwise the tile is in Cs already)
            pan_d2 dt[8];
            p2_gload(B, ldb, brows, r0, bc0 + 64 * trow, t, dt);
            p2_sstore(Cs, t, dt);
        }
        if (!parked) {
            __syncthreads();
            if (ncol > 0 && t == 0)   // every wave has drained its write-through stores of the last X
                __hip_atomic_store(pnl_flag(p, trow), 4ull * (unsigned long long)ncol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < 2 || w < 2) {
                const int mi = p2_dtile_m(w, q), ni = p2_dtile_n(w, q);
#pragma unroll
                for (int v = 0; v < 4; ++v) Cs[(16 * mi + lk + 4 * v) * PNL_LD + 16 * ni + l15] -= dacc[q][v];
            }
        __syncthreads();
        P2_STAMP(trow, 1);   // diagonal tile assembled
        P3Publish pub;
        pub.dst = B + (size_t)r0 * ldb + bc0 + 64 * trow;
        pub.ld = ldb;
        pub.rows = brows - r0;
        pub.word = pnl_flag(p, trow);
        pub.base = 4ull * (unsigned long long)trow;
        pub.add = split;
        double* Sd = psm + 2 * PNL_TILE + 16;   // 8 x 8 scratch block behind the progress cache
        p3_diag(Cs, r0, p, t, nullptr, progressive ? Sd : nullptr, pub, Xs);
        // (every round ends with a barrier: the tile is complete in LDS here, the inverse blocks - those not yet taken out by the
        // publishing wave - transposed in Xs)
        P2_STAMP(trow, 2);   // factored
        if (!progressive) {
            p3_extract_inverse(Cs, Xs, w, lane);
            p3_block_flag(Cs, nullptr, w, lane);
            __syncthreads();
            {   // the lower triangle in 16-byte write-through stores (the pair that holds the diagonal element of an even row also
                // writes its right-hand neighbour: scratch - the progress words sit in row 0 from column 8 on, the inverses from
                // column 32 on in rows 0 .. 31).
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = t + 256 * q;
                    const int r = e >> 5, cc = (e & 31) * 2;
                    if (cc <= r && r0 + r < brows) {
                        const pan_d2 v = *reinterpret_cast<const pan_d2*>(Cs + r * PNL_LD + cc);
                        double* dst = B + (size_t)(r0 + r) * ldb + bc0 + 64 * trow + cc;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v) : "memory");
                    }
                }
            }
            {   // the four inverse blocks: 1024 doubles, two 16-byte write-through stores per thread
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = t + 256 * q;                  // 512 pairs
                    const int jb = e >> 7, i = (e >> 3) & 15, jj = (e & 7) * 2;
                    const int off = p2_wblock(jb) + i * PNL_LD + jj;
                    const pan_d2 v = *reinterpret_cast<const pan_d2*>(Cs + off);
                    double* dst = B + (size_t)(r0 + (off / PNL_LD)) * ldb + bc0 + 64 * trow + (off % PNL_LD);
                    if (r0 + (off / PNL_LD) < brows) asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v) : "memory");
                }
            }
            if (t < 2 && r0 + 1 < brows) {   // the four refinement flags: row 1, columns 16 .. 19
                const pan_d2 v = *reinterpret_cast<const pan_d2*>(Cs + p2_flag_slot(2 * t));
                double* dst = B + (size_t)(r0 + 1) * ldb + bc0 + 64 * trow + 16 + 2 * t;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
        } else {
            // block columns 0 .. 2 went out during the factorisation; the last one now (wave 3: a handful of stores)
            if (w == 3) {
                p3_extract_inverse(Cs, Xs, 3, lane);
                p3_block_flag(Cs, nullptr, 3, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                p3_store_column(Cs, nullptr, 3, lane, pub);
            }
        }
        P2_STAMP(trow, 3);   // stores issued
        if (split) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(pnl_flag(p, trow), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            p2_publish(p, trow, 4ull * (unsigned long long)trow + 4ull);
        }
        P2_STAMP(trow, 4);   // published
        p3_diag_logdet(Cs, r0, p, t);
    }
#undef P2_STAMP
}

// The row-block task of a triangular solve (FLAGS = false: L is final) or of a panel's bulk rows (FLAGS = true: L arrives from the
// team rows of the same launch) with the column blocks taken in PAIRS (c, c + 1): the row
// block's tiles X[rb][u], u < c, are staged once per pair and used against L[c][u] and L[c + 1][u] in turn, and the product
// (c + 1, c) takes X[rb][c] from the LDS tile the strip of column c has just written - 56 tile movements per 512-column block
// instead of 72 (the bulk work of this task is bound by operand bytes: lesson 31).  Same sums in the same order: same bits.
// (two panels in one launch, potrf_group_kernel below)  cstart > ufirst: the column blocks [ufirst, cstart) of this row block are
// SOLVED already - the previous panel's columns - and only enter the sums; column block c is column block c - cstart of the panel
// whose team the progress words belong to.  pub != nullptr: after every column block the row block's own progress word receives
// pub_base + (number of column blocks solved), behind write-through stores of the tile.
template <bool FLAGS>
__device__ __forceinline__ void p2_row_block_pairs(const PanelArgs& p, const double* __restrict__ L, int ldl, int lrows, int lr0, int lc0,
                                                   double* __restrict__ B, int ldb, int brows, int r0, int bc0, int ncol, double* __restrict__ psm,
                                                   int ufirst = 0,   // (ufirst: column blocks before it are zero and stay zero - an upper-triangular right-hand side)
                                                   int cstart = -1, unsigned long long* pub = nullptr, unsigned long long pub_base = 0ull) {
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    unsigned long long* seen = reinterpret_cast<unsigned long long*>(psm + 2 * PNL_TILE);   // 16 words (FLAGS: progress cache)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    if (cstart < 0) cstart = ufirst;
    const int fo = FLAGS ? cstart : 0;   // progress words count the column blocks of the panel that starts at column block cstart
    if (FLAGS) {
        if (t < 16) seen[t] = 0ull;
        __syncthreads();
    }
    // Xs holds the row block's tile of column block c (what is to be solved), acc the products owed to it: solve, leave X in Xs, store
    auto solve_column = [&](int c, pan_d4 (&acc)[4]) {
        pan_d4 T[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) T[mi][v] = Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] - acc[mi][v];
        if (FLAGS) p2_wait(p, seen, c - fo, 4ull * (unsigned long long)(c - fo) + 4ull);   // the triangle (with its inverse blocks and flags) is out
        else __syncthreads();
        {
            pan_d2 lt[8];
            p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * c, t, lt);
            p2_sstore(Cs, t, lt);
        }
        __syncthreads();
        if (!FLAGS) {
            p2_inverse_blocks(Cs, w, lane);
            __syncthreads();
        }
        p2_strip(Cs, T, l15, lk);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] = T[mi][v];
        __syncthreads();
        p2_gstore(B, ldb, brows, r0, bc0 + 64 * c, Xs, t, pub != nullptr);
        if (pub) {   // (wave-uniform) the tile is out before the word says so
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(pub, pub_base + (unsigned long long)(c - cstart) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    for (int c = cstart; c < ncol; c += 2) {
        const bool pair = c + 1 < ncol;
        pan_d4 acc0[4], acc1[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { acc0[mi] = pan_d4{0.0, 0.0, 0.0, 0.0}; acc1[mi] = pan_d4{0.0, 0.0, 0.0, 0.0}; }
        pan_d2 xa[8];
        p2_gload(B, ldb, brows, r0, bc0 + 64 * ufirst, t, xa);   // X[rb][ufirst] - or, for c = ufirst, the tile to be solved itself
        if (c > ufirst) {
            if (FLAGS) {   // every L[c][u] and L[c + 1][u], u < c, is out
                p2_wait(p, seen, c - fo, 4ull * (unsigned long long)(c - fo));
                if (pair) p2_wait(p, seen, c + 1 - fo, 4ull * (unsigned long long)(c - fo));
            }
            pan_d2 la0[8], la1[8];
            p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * ufirst, t, la0);
            p2_gload(L, ldl, lrows, lr0 + 64 * (pair ? c + 1 : c), lc0 + 64 * ufirst, t, la1);
            for (int u = ufirst; u < c; ++u) {
                __syncthreads();   // the previous chunk's operand reads are done
                p2_sstore(Cs, t, la0);
                p2_sstore(Xs, t, xa);
                __syncthreads();
                const int un = min(u + 1, c - 1);
                p2_gload(L, ldl, lrows, lr0 + 64 * c, lc0 + 64 * un, t, la0);
                p2_gload(B, ldb, brows, r0, bc0 + 64 * (u + 1), t, xa);   // (after the last chunk: the tile of column block c)
                __builtin_amdgcn_sched_barrier(0);
                p2_chunk(Cs, Xs, acc0, w, l15, lk);
                if (pair) {
                    __syncthreads();
                    p2_sstore(Cs, t, la1);   // X[rb][u] stays where it is
                    __syncthreads();
                    p2_gload(L, ldl, lrows, lr0 + 64 * (c + 1), lc0 + 64 * un, t, la1);
                    __builtin_amdgcn_sched_barrier(0);
                    p2_chunk(Cs, Xs, acc1, w, l15, lk);
                }
            }
        }
        __syncthreads();
        p2_sstore(Xs, t, xa);
        __syncthreads();
        solve_column(c, acc0);
        if (pair) {
            // the product (c + 1, c): X[rb][c] is in Xs (the strip has just left it there), L[c + 1][c] and the next tile to solve arrive now
            if (FLAGS) p2_wait(p, seen, c + 1 - fo, 4ull * (unsigned long long)(c - fo) + 4ull);   // team row c + 1 has solved (and published) its strip of column c
            pan_d2 lc[8];
            p2_gload(L, ldl, lrows, lr0 + 64 * (c + 1), lc0 + 64 * c, t, lc);
            p2_gload(B, ldb, brows, r0, bc0 + 64 * (c + 1), t, xa);
            __syncthreads();   // the strip's reads of Cs and the global store's reads of Xs are done
            p2_sstore(Cs, t, lc);
            __syncthreads();
            p2_chunk(Cs, Xs, acc1, w, l15, lk);
            __syncthreads();
            p2_sstore(Xs, t, xa);
            __syncthreads();
            solve_column(c + 1, acc1);
        }
    }
}

// launch bounds (256, 2): at most 256 unified registers, so that a wave fits beside a trailing-update wave (see panel.h)
__global__ __launch_bounds__(256, 2) void potrf_panel2_kernel(PanelArgs p) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    // Batched launch (gridDim.y matrices).  Workgroups are dispatched x first, then y; the linear dispatch index is dealt so that
    // the TEAMS of all matrices come first (team row rb of matrix b at index rb * batch + b), then the bulk row blocks, matrices
    // interleaved: every chain starts at once instead of one matrix's bulk rows holding the slots the next matrix's team needs
    // (same-box A/B: C3 189.8 -> 187.9 ms, C5 60.4 -> 59.7 ms per evaluation).  A row block still only waits for team rows of its
    // own matrix with a smaller row index, i.e. a smaller dispatch index: any number of matrices is safe.
    const int batch = gridDim.y, R = gridDim.x;
    const int lin = blockIdx.x + R * blockIdx.y;
    const int NT = p2_team_count(p.S, p.split);   // team workgroups per matrix (split team: chains + tiles)
    int rb, b, tile_c = -1;
    if (lin < NT * batch) {
        rb = lin / batch;
        b = lin - rb * batch;
        if (p.split) p2_team_decode(p.S, rb, rb, tile_c);
    } else {
        const int idx = lin - NT * batch;
        rb = p.S + idx / batch;
        b = idx % batch;
    }
    p.A += (size_t)b * p.batch_a;
    if (p.logdet) p.logdet += b;
    if (p.info) p.info += b;
    const int r0 = p.k0 + 64 * rb;
    if (tile_c >= 0) {
        __builtin_amdgcn_s_setprio(2);
        p2_team_tile(p, rb, tile_c, psm);
    } else if (rb < p.S) {
        __builtin_amdgcn_s_setprio(3);   // the chain: never lose an issue arbitration to bulk work on the same compute unit
        p2_row_block<true, true>(p, p.A, p.lda, p.N, p.k0, p.k0, p.A, p.lda, p.N, r0, p.k0, rb, 0, rb, psm);
    } else if (p.pairs) {
        p2_row_block_pairs<true>(p, p.A, p.lda, p.N, p.k0, p.k0, p.A, p.lda, p.N, r0, p.k0, p.S, psm);
    } else {
        p2_row_block<true, false>(p, p.A, p.lda, p.N, p.k0, p.k0, p.A, p.lda, p.N, r0, p.k0, p.S, 0, 0, psm);
    }
}

static int potrf_panel_fused2(double* A, int N, int lda, int k0, int W, double* logdet, int* info, hipStream_t stream,
                              bool prezeroed = false, int batch, long long batch_a) {
    PanelArgs p{A, N, lda, k0, W / 64, logdet, info, nullptr};
    p.batch_a = batch_a;
    p.pairs = env_int("GPAR_PANEL_PAIRS", 1);
    p.progressive = env_int("GPAR_PANEL_PROGRESSIVE", 1);
    p.split = p.progressive && p.S <= 8 && env_int("GPAR_PANEL_SPLIT", 1);
    if (p.S > PNL_MAX_S) return GPAR_ARG_ERROR(5);
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&potrf_panel2_kernel), P2_LDS_BYTES));
    if (!(prezeroed && potrf_flags_prezeroed(N, k0)))
        for (int b = 0; b < batch; ++b) {
            GPAR_HIP_TRY(hipMemsetAsync(A + (size_t)b * batch_a + (size_t)k0 * lda + k0 + 8, 0, PNL_FLAG_SLOTS * sizeof(double), stream));
            if (p.split)   // the tile flags: rows 2 and 3, columns 9 .. 31 (rows that exist: a team row block is whole)
                for (int r = 2; r < 4; ++r)
                    GPAR_HIP_TRY(hipMemsetAsync(A + (size_t)b * batch_a + (size_t)(k0 + r) * lda + k0 + 9, 0, 23 * sizeof(double), stream));
        }
    const int R = (N - k0 + 63) / 64 - p.S + (p.split ? p.S + (p.S - 1) * (p.S - 2) / 2 : p.S);
    if (int rc = spin_chain_enter(stream, (long long)R * batch)) return rc;
    hipLaunchKernelGGL(potrf_panel2_kernel, dim3(R, batch), dim3(256), P2_LDS_BYTES, stream, p);
    spin_chain_leave(stream, (long long)R * batch);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// SEVERAL consecutive panels in one launch (the latency-bound tail of a factorisation, and all of a small one).
//
// Between two panel kernels the schedule of potrf_run has a launch boundary, the update of the next panel's columns (a launch of
// its own, 20-100 us even when it is a single round of tiles) and another boundary: a 64-column step of the chain costs 12.9 us
// inside a panel and ~26 us averaged over a latency-bound step.  Here panel q + 1's chain starts ~one tile product after panel
// q's ends.  Workgroups in dispatch order (every wait is on a lower index, as in potrf_panel2_kernel):
//   (0) the row blocks of panel 0 [k0, k1): as above, except that bulk row blocks publish a progress word per column block;
//   then for every further panel q = 1 .. G - 1 of the launch, [kq, kq + 512):
//   (a) the 36 tiles of its diagonal block, C -= X_i X_j^T over the 8 q column blocks of the launch's earlier panels: one
//       workgroup per tile, column block u as soon as rows i and j have it, the C tile in registers from the start - behind the
//       last column block of panel q - 1 there is one tile product and one store left; each tile then counts itself into its
//       row's word;
//   (b) its team rows: each waits for its row's tiles of (a), then as above;
//   (c) the tiles of its columns BELOW the diagonal block, like (a) (folded into the bulk row blocks' left-looking sums - 64 more
//       tile products in one workgroup, ~150 us - they outlast the panel's chain: n = 1024 0.33 -> 0.43 ms);
//   (d) its bulk row blocks: each waits for its eight tiles of (c), and publishes like (0) unless the panel is the last.
// Words (all in the strict upper triangle of diagonal tiles - scratch by the ABI's convention - and zeroed by potrf_zero_flags):
// row block b keeps its progress word in row 2, its tile count in row 3, column 8 of the diagonal tile of row block b - 1
// (its own diagonal tile may have a single row: the augmented row).  Progress counts column blocks of the whole matrix, the
// tile count accumulates over the fused launches of a factorisation (8 per panel after a launch's first for every row below
// them): no word is ever reset.
struct GroupArgs {
    PanelArgs p;                  // panel 0
    int G;                        // panels in this launch
    unsigned long long la_base;   // tiles every row block below k0 has counted in earlier fused launches of this factorisation
};

__device__ __forceinline__ unsigned long long* grp_word(double* A, int lda, int arb, int which) {
    return reinterpret_cast<unsigned long long*>(A + (size_t)(64 * (arb - 1) + 2 + which) * lda + 64 * (arb - 1) + 8);
}

// one 64 x 64 tile (ti, tj) of panel q's columns (kq = its first column, q >= 1): C -= X_i X_j^T over the 8 q column blocks of the
// launch's earlier panels [k0, kq) - column block u as soon as rows i and j have it -, then the tile counts itself into row ti's word
__device__ __forceinline__ void grp_la_tile(const PanelArgs& p, int kq, int ti, int tj, double* __restrict__ psm) {
    const int t = threadIdx.x;
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    const int lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    const int r0 = kq + 64 * ti, c0 = kq + 64 * tj;
    const int kb0 = p.k0 / 64, nchunks = (kq - p.k0) / 64;
    unsigned long long* wi = grp_word(p.A, p.lda, kq / 64 + ti, 0);
    unsigned long long* wj = grp_word(p.A, p.lda, kq / 64 + tj, 0);
    pan_d4 acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0.0, 0.0, 0.0, 0.0};
    pan_d2 ct[8];   // the output tile: last written by an earlier launch on the stream
    p2_gload(p.A, p.lda, p.N, r0, c0, t, ct);
    // Column blocks both rows have published are taken without polling, the next one's operands fetched under the current
    // product (a tile of a late panel starts with 8 (q - 1) column blocks of earlier panels behind it: at one poll, one
    // round trip to memory and two barriers per column block - 6 us for 0.9 us of matrix-core work - those outlasted the chain
    // they should hide behind, and the next team waited for them).  At the frontier - a column block per ~12 us - nothing changes.
    unsigned long long* seen = reinterpret_cast<unsigned long long*>(psm + 2 * PNL_TILE);
    int avail = 0;   // column blocks of this launch known to be published by both rows
    bool loaded = false;
    pan_d2 xa[8], la[8];
    for (int u = 0; u < nchunks; ++u) {
        if (!loaded) {
            if (u >= avail) {
                const unsigned long long m = grp_wait2(wi, wj, (unsigned long long)(kb0 + u + 1), p.info, seen);
                const long long have = (long long)m - kb0;
                avail = have > nchunks ? nchunks : (int)have;
            }
            p2_gload(p.A, p.lda, p.N, r0, p.k0 + 64 * u, t, xa);
            p2_gload(p.A, p.lda, p.N, c0, p.k0 + 64 * u, t, la);
        }
        p2_sstore(Cs, t, la);
        p2_sstore(Xs, t, xa);
        __syncthreads();
        loaded = u + 1 < avail;   // (avail <= nchunks)
        if (loaded) {
            p2_gload(p.A, p.lda, p.N, r0, p.k0 + 64 * (u + 1), t, xa);
            p2_gload(p.A, p.lda, p.N, c0, p.k0 + 64 * (u + 1), t, la);
        }
        p2_chunk(Cs, Xs, acc, w, l15, lk);
        __syncthreads();   // the operand reads are done before the next chunk's tiles (or the output tile) land
    }
    p2_sstore(Xs, t, ct);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] -= acc[mi][v];
    __syncthreads();
    if (ti != tj) {
        p2_gstore(p.A, p.lda, p.N, r0, c0, Xs, t, true);
    } else {   // lower triangle only: the strict upper triangle holds the panel's words
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = t + 256 * q;
            const int r = c >> 5, cc = (c & 31) * 2;
            if (r0 + r < p.N && cc <= r) {
                double* dst = p.A + (size_t)(r0 + r) * p.lda + c0 + cc;
                if (cc + 1 <= r) {
                    const pan_d2 v = *reinterpret_cast<const pan_d2*>(Xs + r * PNL_LD + cc);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" ::"v"(dst), "v"(v) : "memory");
                } else {
                    const double v = Xs[r * PNL_LD + cc];
                    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(grp_word(p.A, p.lda, kq / 64 + ti, 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256, 2) void potrf_group_kernel(GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    PanelArgs p = g.p;
    const int batch = gridDim.y;
    const int S = p.S;                         // 8
    const int R0 = (p.N - p.k0 + 63) / 64;     // row blocks from k0
    const int TD = S * (S + 1) / 2;            // tiles of a panel's diagonal block
    const int kb0 = p.k0 / 64;
    const int NT = p2_team_count(S, p.split);   // team workgroups per matrix and panel (split team: chains + tiles)
    int lin = blockIdx.x + gridDim.x * blockIdx.y;
    int rb, b, tile_c = -1;
    // ---- which segment: panel 0's rows, then per panel q >= 1 (a) diagonal-block tiles, (b) team rows, (c) tiles below, (d) bulk rows
    int q = 0, seg = 0;
    const int bulk0 = (p.tile_rows && g.G > 1) ? S * S + (R0 - 2 * S) : R0 - S;   // panel 0's bulk segment (see (d) below)
    if (lin >= (NT + bulk0) * batch) {
        lin -= (NT + bulk0) * batch;
        for (q = 1; q < g.G; ++q) {
            const int Rq = R0 - S * q;
            // (d): the rows that form the next panel's team by tile (S * S workgroups), the others one workgroup per row block
            const int bulk = (p.tile_rows && q < g.G - 1) ? S * S + (Rq - 2 * S) : Rq - S;
            const int sizes[4] = {TD * batch, NT * batch, (Rq - S) * S * batch, bulk * batch};
            for (seg = 1; seg <= 4; ++seg) {
                if (lin < sizes[seg - 1]) break;
                lin -= sizes[seg - 1];
            }
            if (seg <= 4) break;
        }
    }
    const int kq = p.k0 + 64 * S * q;
    const bool last = q == g.G - 1;
    int bulk_c = -1;   // >= 0: this workgroup is tile (rb, bulk_c) of a next-team row block
    auto bulk_decode = [&](int idx) {
        if (p.tile_rows && !last) {
            if (idx < S * S * batch) {
                const int tl = idx / batch;
                b = idx - tl * batch;
                rb = S + tl / S;
                bulk_c = tl - (tl / S) * S;
                return;
            }
            idx -= S * S * batch;
            rb = 2 * S + idx / batch;
            b = idx % batch;
            return;
        }
        rb = S + idx / batch;
        b = idx % batch;
    };
    if (seg == 0) {
        // teams of all matrices first, as in potrf_panel2_kernel
        if (lin < NT * batch) {
            rb = lin / batch;
            b = lin - rb * batch;
            if (p.split) p2_team_decode(S, rb, rb, tile_c);
        } else {
            bulk_decode(lin - NT * batch);
        }
    } else if (seg == 1 || seg == 3) {
        rb = lin / batch;   // tile index
        b = lin - rb * batch;
    } else if (seg == 2) {
        rb = lin / batch;
        b = lin - rb * batch;
        if (p.split) p2_team_decode(S, rb, rb, tile_c);
    } else {
        bulk_decode(lin);
    }
    p.A += (size_t)b * p.batch_a;
    if (p.logdet) p.logdet += b;
    if (p.info) p.info += b;
    if (seg == 1) {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= rb) ++ti;
        __builtin_amdgcn_s_setprio(2);
        grp_la_tile(p, kq, ti, rb - ti * (ti + 1) / 2, psm);
        return;
    }
    if (seg == 3) {
        const int ti = S + rb / S;
        grp_la_tile(p, kq, ti, rb - (ti - S) * S, psm);
        return;
    }
    // ---- a row block of panel q
    const int r0 = kq + 64 * rb;
    const unsigned long long counted = g.la_base + (unsigned long long)(S * (q - 1));   // tiles this row has counted before panel q's
    if (rb < S) {
        if (tile_c >= 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
        if (q > 0) grp_wait(grp_word(p.A, p.lda, kq / 64 + rb, 1), counted + (unsigned long long)rb + 1ull, p.info);
        p.k0 = kq;
        if (tile_c >= 0) p2_team_tile(p, rb, tile_c, psm);
        else p2_row_block<true, true>(p, p.A, p.lda, p.N, kq, kq, p.A, p.lda, p.N, r0, kq, rb, 0, rb, psm);
    } else if (bulk_c >= 0) {
        __builtin_amdgcn_s_setprio(1);
        if (q > 0) grp_wait(grp_word(p.A, p.lda, kq / 64 + rb, 1), counted + (unsigned long long)S, p.info);
        p.k0 = kq;
        p2_bulk_tile(p, rb, bulk_c, grp_word(p.A, p.lda, kq / 64 + rb, 0), (unsigned long long)(kq / 64), psm);
    } else {
        if (q > 0) grp_wait(grp_word(p.A, p.lda, kq / 64 + rb, 1), counted + (unsigned long long)S, p.info);
        p.k0 = kq;
        // (every panel but the last: the row block publishes its column blocks for the tiles of the panels to come)
        p2_row_block_pairs<true>(p, p.A, p.lda, p.N, kq, kq, p.A, p.lda, p.N, r0, kq, S, psm, 0, 0,
                                 last ? nullptr : grp_word(p.A, p.lda, kq / 64 + rb, 0), (unsigned long long)(kq / 64));
    }
}

static long long potrf_group_workgroups(int N, int k0, int S, int G, int split, int tile_rows) {
    const int R0 = (N - k0 + 63) / 64;
    const int NT = split ? S + (S - 1) * (S - 2) / 2 : S;
    long long per = NT + ((tile_rows && G > 1) ? S * S + (R0 - 2 * S) : R0 - S);
    for (int q = 1; q < G; ++q) {
        const int Rq = R0 - S * q;
        per += S * (S + 1) / 2 + NT + (long long)(Rq - S) * S + ((tile_rows && q < G - 1) ? S * S + (Rq - 2 * S) : Rq - S);
    }
    return per;
}

static int potrf_group_fused(double* A, int N, int lda, int k0, int W, int G, double* logdet, int* info, hipStream_t stream, int batch,
                             long long batch_a, unsigned long long la_base) {
    GroupArgs g;
    g.p = PanelArgs{A, N, lda, k0, W / 64, logdet, info, nullptr};
    g.p.batch_a = batch_a;
    g.p.pairs = 1;
    g.p.progressive = env_int("GPAR_PANEL_PROGRESSIVE", 1);
    g.p.split = g.p.progressive && g.p.S <= 8 && env_int("GPAR_PANEL_SPLIT", 1);
    // The split team and the next team's rows by tile are latency devices that cost compute-unit slots: per matrix and panel 29 + 64
    // workgroups that mostly wait, beside its bulk rows.  Where those of all matrices of a lock-step batch no longer fit the chip's 512
    // slots they crowd out the launch's update tiles, and the launch is bound by those (four matrices of 4096 rows: 2.80 -> 2.67 ms
    // without, eight of 2048: 1.41 -> 1.34; four of 2048 - 440 such workgroups - 0.97 -> 1.02 the other way; profiles/r05_exp_batch_fuse.txt).
    if (g.p.split && env_int("GPAR_PANEL_SPLIT", 1) == 1) {
        const int R0 = (N - k0 + 63) / 64, S = g.p.S;
        const long long waiting = (long long)batch * (S + (S - 1) * (S - 2) / 2 + S * S + (R0 > 2 * S ? R0 - 2 * S : 0));
        if (waiting > 512) g.p.split = 0;
    }
    g.p.tile_rows = g.p.split && env_int("GPAR_PANEL_TILE_ROWS", 1);
    g.G = G;
    g.la_base = la_base;
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&potrf_group_kernel), P2_LDS_BYTES));
    const long long wgs = potrf_group_workgroups(N, k0, W / 64, G, g.p.split, g.p.tile_rows);
    if (int rc = spin_chain_enter(stream, wgs * batch)) return rc;
    hipLaunchKernelGGL(potrf_group_kernel, dim3((unsigned)wgs, batch), dim3(256), P2_LDS_BYTES, stream, g);
    spin_chain_leave(stream, wgs * batch);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Fused block of the forward triangular solve  X L^T = B  (gpar_trsm_rlt): one workgroup per 64-row block of B carries
// it through the S column blocks [c0, c0 + 64 S) - the same left-looking row-block task, without hand-offs.
__global__ __launch_bounds__(256, 2) void trsm_block2_kernel(TrsmBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    if (gpar_pred_skip(a.pred, a.pred_sense)) return;
    const int r0 = 64 * blockIdx.x;
    // upper_tri: row r has nothing left of column r, so column blocks that end before r0 are still zero - and stay zero
    int ufirst = 0;
    if (a.upper_tri) {
        while (ufirst < a.S && a.c0 + 64 * ufirst + 63 < r0) ++ufirst;
    }
    if (a.pairs) {
        const PanelArgs nothing{nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr};
        p2_row_block_pairs<false>(nothing, a.L, a.ldl, a.n, a.c0, a.c0, a.B, a.ldb, a.nrows, r0, a.c0, a.S, psm, ufirst);
        return;
    }
    const PanelArgs none{nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr};
    p2_row_block<false, false>(none, a.L, a.ldl, a.n, a.c0, a.c0, a.B, a.ldb, a.nrows, r0, a.c0, a.S, ufirst, 0, psm);
}

static int trsm_block_fused2(const double* L, int n, int ldl, double* B, int nrows, int ldb, int c0, int S, int upper_tri,
                             hipStream_t stream) {
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&trsm_block2_kernel), P2_LDS_BYTES));
    TrsmBlockArgs a{L, n, ldl, B, nrows, ldb, c0, S, upper_tri};
    a.pairs = env_int("GPAR_TRSM_PAIRS", 1);
    a.pred = g_pred.flag; a.pred_sense = g_pred.sense;
    hipLaunchKernelGGL(trsm_block2_kernel, dim3(gpar_ceil_div(nrows, 64)), dim3(256), P2_LDS_BYTES, stream, a);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// The mirrored task: one 64-row block of B carried through the S column blocks [c0, c0 + 64 S) of the BACKWARD solve
// X L = B  (gpar_trsm_rln, many rows: the W_fu L_z^-1 of the inducing-point gradient), last column block first:
//     acc  = sum_{u>c} X[rb][u] L[u][c]        the L tile enters UNTRANSPOSED: the A fragment of column m and K index k is
//                                              L[u][c][k][m], a transposed read of the LDS tile
//     X    = (B[rb][c] - acc) L[c][c]^-1       16 x 16 blocks from the last to the first:  x_j = W_j^T-applied, then
//                                              T_j' -= L[j][j']^T-products for j' < j - the chain of p2_strip run backwards
// Same T layout, same LDS tiles, same register-to-operand chaining.
__device__ __forceinline__ void p2_chunk_back(const double* __restrict__ Ls, const double* __restrict__ Xs, pan_d4 (&acc)[4], int w,
                                              int l15, int lk) {
    double a0[4], a1[4], b0, b1;
    auto frag = [&](int kk, double (&a)[4], double& b) {
        b = Xs[(16 * w + l15) * PNL_LD + kk];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[mi] = Ls[kk * PNL_LD + 16 * mi + l15];
    };
    frag(lk, a0, b0);
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 2) {
        frag(4 * (k4 + 1) + lk, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[mi], b0, acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 2 < 16) frag(4 * (k4 + 2) + lk, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[mi], b1, acc[mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// T (this wave's 16 rows x 64 columns, T layout) <- X^T with X L = T, L the lower-triangular tile in Cs whose diagonal
// 16 x 16 blocks have their inverses at p2_wblock().
__device__ __forceinline__ void p2_strip_back(const double* __restrict__ Cs, pan_d4 (&T)[4], int l15, int lk) {
#pragma unroll
    for (int jb = 3; jb >= 0; --jb) {
        const double* W = Cs + p2_wblock(jb);
        pan_d4 x = pan_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)   // x^T = W^T t^T:  A[m][k] = W[k][m]
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(4 * k4 + lk) * PNL_LD + l15], T[jb][k4], x, 0, 0, 0);
        if (Cs[p2_flag_slot(jb)] != 0.0) {   // (wave-uniform) an ill-conditioned block: x += W^T (t - L_bb^T x)
            pan_d4 r = T[jb];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int k = 4 * k4 + lk;
                const double l = Cs[(16 * jb + k) * PNL_LD + 16 * jb + l15];   // L_bb^T[m = l15][k] = L_bb[k][m]: zero for m > k
                r = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 <= k ? -l : 0.0, x[k4], r, 0, 0, 0);
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(4 * k4 + lk) * PNL_LD + l15], r[k4], x, 0, 0, 0);
        }
        T[jb] = x;
#pragma unroll
        for (int j2 = jb - 1; j2 >= 0; --j2) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)   // t_j2^T -= L[jb][j2]^T x^T:  A[m][k] = L[16 jb + k][16 j2 + m]
                T[j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Cs[(16 * jb + 4 * k4 + lk) * PNL_LD + 16 * j2 + l15], x[k4], T[j2], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void p2_row_block_back(const double* __restrict__ L, int ldl, int lrows, int c0, double* __restrict__ B, int ldb,
                                                  int brows, int r0, int ncol, double* __restrict__ psm) {
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, lk = lane >> 4;
    for (int c = ncol - 1; c >= 0; --c) {
        pan_d4 acc[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = pan_d4{0.0, 0.0, 0.0, 0.0};
        pan_d2 xa[8];   // the row block's own tile of column block u (operand of chunk u), finally of column block c itself
        p2_gload(B, ldb, brows, r0, c0 + 64 * (ncol - 1), t, xa);
        if (c < ncol - 1) {
            pan_d2 la[8];
            p2_gload(L, ldl, lrows, c0 + 64 * (ncol - 1), c0 + 64 * c, t, la);
            for (int u = ncol - 1; u > c; --u) {
                __syncthreads();   // the previous chunk's operand reads are done
                p2_sstore(Cs, t, la);
                p2_sstore(Xs, t, xa);
                __syncthreads();
                // next chunk's tiles in flight under this chunk's products; after the last chunk the row block's tile of
                // column block c arrives the same way (the L index is clamped: a harmless repeat)
                p2_gload(L, ldl, lrows, c0 + 64 * max(u - 1, c + 1), c0 + 64 * c, t, la);
                p2_gload(B, ldb, brows, r0, c0 + 64 * (u - 1), t, xa);
                __builtin_amdgcn_sched_barrier(0);
                p2_chunk_back(Cs, Xs, acc, w, l15, lk);
            }
        }
        __syncthreads();
        p2_sstore(Xs, t, xa);
        {
            pan_d2 lt[8];
            p2_gload(L, ldl, lrows, c0 + 64 * c, c0 + 64 * c, t, lt);
            p2_sstore(Cs, t, lt);
        }
        __syncthreads();
        pan_d4 T[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) T[mi][v] = Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] - acc[mi][v];
        p2_inverse_blocks(Cs, w, lane);
        __syncthreads();
        p2_strip_back(Cs, T, l15, lk);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) Xs[(16 * w + l15) * PNL_LD + 16 * mi + lk + 4 * v] = T[mi][v];
        __syncthreads();
        p2_gstore(B, ldb, brows, r0, c0 + 64 * c, Xs, t, false);
    }
}

__global__ __launch_bounds__(256, 2) void trsm_block2_back_kernel(TrsmBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    p2_row_block_back(a.L, a.ldl, a.n, a.c0, a.B, a.ldb, a.nrows, 64 * (int)blockIdx.x, a.S, psm);
}

static int trsm_block_back_fused2(const double* L, int n, int ldl, double* B, int nrows, int ldb, int c0, int S, hipStream_t stream) {
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&trsm_block2_back_kernel), P2_LDS_BYTES));
    TrsmBlockArgs a{L, n, ldl, B, nrows, ldb, c0, S, 0};
    hipLaunchKernelGGL(trsm_block2_back_kernel, dim3(gpar_ceil_div(nrows, 64)), dim3(256), P2_LDS_BYTES, stream, a);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// X_bb = L_bb^-T for ALL 64 S-column diagonal blocks of a factor in one launch (X holds the identity on entry): the
// triangular-solve row-block task on the identity, blockIdx.y = diagonal block.  Leaves of the recursive inversion in
// chol_inverse_run (potrf.h).
__global__ __launch_bounds__(256, 2) void trinv_blocks2_kernel(TrsmBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    const int c0 = 64 * a.S * blockIdx.y;
    const int r0 = c0 + 64 * blockIdx.x;   // row block blockIdx.x of the block: zero left of its own column block
    const PanelArgs none{nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr};
    p2_row_block<false, false>(none, a.L, a.ldl, a.n, c0, c0, a.B, a.ldb, a.nrows, r0, c0, a.S, (int)blockIdx.x, 0, psm);
}

static int trinv_blocks_fused2(const double* L, int n, int ldl, double* X, int ldx, int S, hipStream_t stream) {
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&trinv_blocks2_kernel), P2_LDS_BYTES));
    TrsmBlockArgs a{L, n, ldl, X, n, ldx, 0, S, 1};
    hipLaunchKernelGGL(trinv_blocks2_kernel, dim3(S, n / (64 * S)), dim3(256), P2_LDS_BYTES, stream, a);
    GPAR_LAUNCH_CHECK();
    return 0;
}

}  // namespace gpar
