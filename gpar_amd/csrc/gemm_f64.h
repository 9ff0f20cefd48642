// fp64 GEMM / SYRK on the gfx950 matrix cores.
//
//   C <- alpha * op(A) op(B) + beta * C,   op(A): m x k, op(B): k x n, all row-major.
//
// Instruction: v_mfma_f64_16x16x4_f64.  With its accumulators pinned in registers it issues every ~64 cycles:
// 72-75 TFLOP/s from one wave per SIMD, 78 from two (tools/probe_mfma_f64_16x16x4.hip; spec fp64 matrix peak 78.6);
// rocBLAS' Tensile kernels for these shapes use it too (MI16x16x4x1, 76.6 TFLOP/s at 8192^3).  [The first version of
// this kernel was built on v_mfma_f64_4x4x4_4b_f64 because an earlier micro-benchmark had "measured" 16x16x4 at 47-49
// TFLOP/s: that benchmark let the compiler shuttle the 8-register accumulators between AGPRs and VGPRs on every
// iteration and so timed the shuttle, not the instruction.  The 4x4x4 form tops out at 71 and needs 2.5x the LDS reads.]
// Lane maps (probed, profiles/r01_probe_mfma_f64_16x16x4.txt): A-operand lane l = (row l & 15, k = l >> 4);
// B-operand lane l = (column l & 15, k = l >> 4); register v of D holds row (l >> 4) + 4 v, column l & 15.
//
// (A BM = 64 instantiation - half a tile per workgroup - serves launches with fewer tiles than compute units; see the kernel.)
// Work decomposition: 256 threads = 4 waves (2 x 2), workgroup tile 128 x 128, wave tile 64 x 64 =
// 4 x 4 accumulators of 16 x 16 = 64 f64 = 128 VGPRs (arch VGPRs: no AGPR traffic); BK = 16 per stage, two LDS
// stages (73.7 KB) so two workgroups share a CU (2 waves per SIMD) and cover each other's barrier/staging.
// Operand tiles are staged global -> registers -> LDS (the loads for stage s+1 are issued before the
// MFMAs of stage s), in one of two padded, bank-conflict-free LDS images:
//   KC  [128][18]  for operands stored with k contiguous   (read: 18*r + k  distinct mod 32 per half wave)
//   MC  [16][144]  for operands stored with m/n contiguous (read: 144*k + r)
#pragma once
#include "common.h"
#include <stdlib.h>

namespace gpar {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 16;
constexpr int GEMM_LDKC = 18;
constexpr int GEMM_LDMC = 144;
constexpr int GEMM_TILE = 2304;  // doubles per operand stage (128*18 == 16*144)
constexpr int GEMM_LDS_BYTES = 4 * GEMM_TILE * 8;
// Build-time switches of the tile's tail (A/B'd with tools/time_gemm_phases.hip and tools/ab_variants.sh; round 2: the
// epilogue's share of a K = 512 tile went from 12 to 5 us, most of which the co-resident workgroup's K loop takes back -
// a compute unit finishes two tiles per ~125 us either way, against 109 us of pure MFMA issue):
//   EPI_PRIO     issue priority of a wave in its epilogue (the K loop's MFMA phase runs at 1)
//   NT_STORE     the tile of C is written with non-temporal stores (it is next read a whole trailing update later)
//   PRELOAD_C    interior tiles of C <- C -+ op(A) op(B) start their accumulators from C (see gemm_mainloop_pf2)
#ifndef GPAR_GEMM_EPI_PRIO
#define GPAR_GEMM_EPI_PRIO 3
#endif
#ifndef GPAR_GEMM_NT_STORE
#define GPAR_GEMM_NT_STORE 1
#endif
#ifndef GPAR_GEMM_PRELOAD_C
#define GPAR_GEMM_PRELOAD_C 1
#endif
constexpr int GEMM_LDT = 66;     // row pitch of the epilogue's transposition buffer: 16 x 66 doubles per wave

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int m, n, k;
    int lda, ldb, ldc;
    double alpha, beta;
    int flags;
    int tiles_m, tiles_n;
    int fastA, fastB, fastC;
    int ksplit;          // > 0: blockIdx.y selects the K range [y * ksplit, (y + 1) * ksplit) and the output slab y
    long long part_stride;   // elements between consecutive partial slabs of C (split-K)
    long long batch_a, batch_b, batch_c;   // blockIdx.z selects problem z of a batch: operands advance by these many elements
    int nfull, ntail;    // MIXED launches (see gemm_f64_kernel): blocks [0, nfull) take whole tiles, the 2 * ntail blocks behind them half tiles
    long long* stamps;   // dev aid (tools/time_gemm_phases.hip): per-block s_memrealtime stamps + hardware ids, normally null
    long long* stage_stamps;   // dev aid (tools/time_gemm_stages.hip, compiled with GPAR_GEMM_STAGE_STAMPS): 3 stamps per K stage
    const int* pred;     // role-0 launches inside a predicated solve (common.h: GparPredicate): return at once unless (*pred != 0) == pred_sense
    int pred_sense;
};

typedef double gpar_d2 __attribute__((ext_vector_type(2)));
typedef double gpar_d4 __attribute__((ext_vector_type(4)));

// Global -> register stage of one operand tile (4 x 16-byte chunks per thread).
//   KC: tile element (r, kk) lives at g[(r0 + r) * ld + k0 + kk]     (r < 128, kk < 16)
//   MC: tile element (r, kk) lives at g[(k0 + kk) * ld + r0 + r]
// `lower`: entries with global k > global r are treated as zero (triangular op(A)).
// FAST: 0 = general, 1 = interior tile, 2 = tile that overhangs the operand's last row, k-contiguous operand: the same
// 16-byte loads from row indices clamped to the last valid row (the duplicated rows only feed accumulators that the
// epilogue never stores).  Without it the one-row overhang of the augmented matrix [[K, .], [y^T, c]] sent a whole row of
// tiles per trailing update down the general path - and those tiles are the last ones dispatched.
template <bool KC, int FAST, int ROWS = 128>
__device__ __forceinline__ void gemm_gload(const double* __restrict__ g, int ld, int r0, int rmax, int k0,
                                           int kmax, bool lower, int t, gpar_d2 (&reg)[ROWS / 32]) {
    if (FAST) {
        // interior tile, aligned operand: branch-free 16-byte loads (any branch around a load makes hipcc fence it
        // with vmcnt(0), serialising the A and B requests and exposing a memory round trip per stage)
#pragma unroll
        for (int q = 0; q < ROWS / 32; ++q) {
            const int c = t + 256 * q;
            if (KC) {
                const int r = c >> 3, kk = (c & 7) * 2;
                const int row = (FAST == 2) ? min(r0 + r, rmax - 1) : r0 + r;
                reg[q] = *reinterpret_cast<const gpar_d2*>(g + (size_t)row * ld + k0 + kk);
            } else {
                const int kk = c / (ROWS / 2), r = (c % (ROWS / 2)) * 2;
                reg[q] = *reinterpret_cast<const gpar_d2*>(g + (size_t)(k0 + kk) * ld + r0 + r);
            }
        }
    } else {
        // edge tile / unaligned operand / triangular operand: clamped (always valid) scalar loads, masked afterwards
#pragma unroll
        for (int q = 0; q < ROWS / 32; ++q) {
            const int c = t + 256 * q;
            int r, kk, r1, kk1;
            if (KC) { r = r0 + (c >> 3); kk = k0 + (c & 7) * 2; r1 = r; kk1 = kk + 1; }
            else { kk = k0 + c / (ROWS / 2); r = r0 + (c % (ROWS / 2)) * 2; r1 = r + 1; kk1 = kk; }
            const bool ok0 = r < rmax && kk < kmax && (!lower || kk <= r);
            const bool ok1 = r1 < rmax && kk1 < kmax && (!lower || kk1 <= r1);
            const int rc = min(r, rmax - 1), kc = min(kk, kmax - 1), rc1 = min(r1, rmax - 1), kc1 = min(kk1, kmax - 1);
            const double v0 = KC ? g[(size_t)rc * ld + kc] : g[(size_t)kc * ld + rc];
            const double v1 = KC ? g[(size_t)rc1 * ld + kc1] : g[(size_t)kc1 * ld + rc1];
            reg[q] = gpar_d2{ok0 ? v0 : 0.0, ok1 ? v1 : 0.0};
        }
    }
}

template <bool KC, int ROWS = 128>
__device__ __forceinline__ void gemm_sstore(double* __restrict__ s, int t, const gpar_d2 (&reg)[ROWS / 32]) {
#pragma unroll
    for (int q = 0; q < ROWS / 32; ++q) {
        const int c = t + 256 * q;
        if (KC) {
            const int r = c >> 3, kk = (c & 7) * 2;
            *reinterpret_cast<gpar_d2*>(s + r * GEMM_LDKC + kk) = reg[q];
        } else {
            const int kk = c / (ROWS / 2), r = (c % (ROWS / 2)) * 2;
            *reinterpret_cast<gpar_d2*>(s + kk * GEMM_LDMC + r) = reg[q];
        }
    }
}

// K loop of one 128 x 128 tile.  FAST (interior tile, aligned operands, k % 16 == 0) is branch-free so the loads
// of stage s+1 stay in flight under the MFMAs of stage s; the other instantiation handles every edge case.
template <bool A_KC, bool B_KC, int FAST, int BM>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& p, double* smem, gpar_d4 (&acc)[BM / 32][4], int m0, int n0,
                                              int kbeg, int kend, int nk, bool a_lower, int t, int lane, int wm, int wn) {
    gpar_d2 ra[BM / 32], rb[4];
    if (nk > 0) {
        gemm_gload<A_KC, FAST, BM>(p.A, p.lda, m0, p.m, kbeg, kend, a_lower, t, ra);
        gemm_gload<B_KC, FAST>(p.B, p.ldb, n0, p.n, kbeg, kend, false, t, rb);
        gemm_sstore<A_KC, BM>(smem, t, ra);
        gemm_sstore<B_KC>(smem + GEMM_TILE, t, rb);
    }
    __syncthreads();

    const int l15 = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const double* As = smem + (kt & 1) * 2 * GEMM_TILE;
        const double* Bs = As + GEMM_TILE;
        const bool more = kt + 1 < nk;
        // the wave that is in its MFMA phase wins issue arbitration over its neighbour's staging instructions (measured:
        // 68.5 -> 70.3 TFLOP/s at 8192^3, 63.1 -> 64.3 at the K = 512 trailing-update shape; priority 3 is no better than 1)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const int kk = k4 * 4 + lk;
            double af[BM / 32], bf[4];   // lane l: row / column l & 15 of each 16-wide block, k = 4 k4 + (l >> 4)
#pragma unroll
            for (int mi = 0; mi < BM / 32; ++mi) {
                const int r = wm * (BM / 2) + 16 * mi + l15;
                af[mi] = A_KC ? As[r * GEMM_LDKC + kk] : As[kk * GEMM_LDMC + r];
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int c = wn * 64 + 16 * nj + l15;
                bf[nj] = B_KC ? Bs[c * GEMM_LDKC + kk] : Bs[kk * GEMM_LDMC + c];
            }
#pragma unroll
            for (int mi = 0; mi < BM / 32; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
                    acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
            if (k4 == 0 && more) {
                // the next stage's global loads go out after the first quarter of the MFMAs, not before them: their
                // address arithmetic no longer delays the start of the MFMA phase (70.3 -> 72.3 TFLOP/s at 8192^3)
                gemm_gload<A_KC, FAST, BM>(p.A, p.lda, m0, p.m, kbeg + (kt + 1) * GEMM_BK, kend, a_lower, t, ra);
                gemm_gload<B_KC, FAST>(p.B, p.ldb, n0, p.n, kbeg + (kt + 1) * GEMM_BK, kend, false, t, rb);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) {
            double* An = smem + ((kt + 1) & 1) * 2 * GEMM_TILE;
            gemm_sstore<A_KC, BM>(An, t, ra);
            gemm_sstore<B_KC>(An + GEMM_TILE, t, rb);
        }
        __syncthreads();
    }
}

// One 16-deep stage of the K loop out of LDS images As / Bs; after the first quarter of its MFMAs the global loads of a
// LATER stage (k offset `knext`) are issued into (ra, rb).
template <bool A_KC, bool B_KC, int FAST, int BM>
__device__ __forceinline__ void gemm_stage(const GemmArgs& p, const double* __restrict__ As, const double* __restrict__ Bs,
                                           gpar_d4 (&acc)[BM / 32][4], int m0, int n0, int knext, int kend, int t, int l15, int lk,
                                           int wm, int wn, gpar_d2 (&ra)[BM / 32], gpar_d2 (&rb)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        const int kk = k4 * 4 + lk;
        double af[BM / 32], bf[4];
#pragma unroll
        for (int mi = 0; mi < BM / 32; ++mi) {
            const int r = wm * (BM / 2) + 16 * mi + l15;
            af[mi] = A_KC ? As[r * GEMM_LDKC + kk] : As[kk * GEMM_LDMC + r];
        }
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const int c = wn * 64 + 16 * nj + l15;
            bf[nj] = B_KC ? Bs[c * GEMM_LDKC + kk] : Bs[kk * GEMM_LDMC + c];
        }
#pragma unroll
        for (int mi = 0; mi < BM / 32; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
            }
        if (k4 == 0) {
            gemm_gload<A_KC, FAST, BM>(p.A, p.lda, m0, p.m, knext, kend, false, t, ra);
            gemm_gload<B_KC, FAST>(p.B, p.ldb, n0, p.n, knext, kend, false, t, rb);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// K loop with the global loads issued TWO stages ahead (two register sets, the loop unrolled by two so that both are
// statically named): one stage of MFMAs (4096 cycles for a wave that has its SIMD to itself) is shorter than a trip to
// HBM, so with a one-stage distance a workgroup that is alone on its CU - every launch with fewer tiles than CUs: the
// look-ahead slices and the whole tail of a factorisation - waited for memory in every stage.  Fast modes only (the
// prefetch index is clamped to the last stage instead of being guarded, which needs branch-free loads).
//
// PRELOAD (interior tiles of an update C <- C -+ op(A) op(B), i.e. beta = 1, alpha = -+1): the accumulators start from
// alpha * C instead of zero - its 64 loads per lane, straight into the MFMA D layout (8-byte accesses in full 128-byte
// runs), leave with the operand loads of the first two stages and share their round trip - and the epilogue only stores
// alpha * acc: the tile's read of C is off the tail of the workgroup, where nothing covered it.
template <bool A_KC, bool B_KC, int FAST, int BM, bool PRELOAD = false>
__device__ __forceinline__ void gemm_mainloop_pf2(const GemmArgs& p, double* smem, gpar_d4 (&acc)[BM / 32][4], int m0, int n0,
                                                  int kbeg, int kend, int nk, int t, int lane, int wm, int wn) {
    static_assert(FAST != 0, "branch-free loads only");
    if (nk <= 0) return;
    gpar_d2 ra0[BM / 32], rb0[4], ra1[BM / 32], rb1[4];
    double* buf0 = smem;
    double* buf1 = smem + 2 * GEMM_TILE;
    const int klast = kbeg + (nk - 1) * GEMM_BK;
    gemm_gload<A_KC, FAST, BM>(p.A, p.lda, m0, p.m, kbeg, kend, false, t, ra0);
    gemm_gload<B_KC, FAST>(p.B, p.ldb, n0, p.n, kbeg, kend, false, t, rb0);
    gemm_gload<A_KC, FAST, BM>(p.A, p.lda, m0, p.m, min(kbeg + GEMM_BK, klast), kend, false, t, ra1);
    gemm_gload<B_KC, FAST>(p.B, p.ldb, n0, p.n, min(kbeg + GEMM_BK, klast), kend, false, t, rb1);
    if (PRELOAD) {
        // wave-uniform base (scalar registers) + one 32-bit lane offset: no per-row address registers beside the accumulators
        const int wmu = __builtin_amdgcn_readfirstlane(wm), wnu = __builtin_amdgcn_readfirstlane(wn);
        const double* cw = p.C + (size_t)(m0 + wmu * (BM / 2)) * p.ldc + n0 + wnu * 64;
        const unsigned voff = ((unsigned)(lane >> 4) * (unsigned)p.ldc + (unsigned)(lane & 15)) * 8u;   // bytes; < 2^32: 3 rows of C
        const double sgn = p.alpha;   // +-1: alpha * C == C / alpha exactly
#pragma unroll
        for (int mi = 0; mi < BM / 32; ++mi)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const char* cr = reinterpret_cast<const char*>(cw + (size_t)(16 * mi + 4 * v) * p.ldc);
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[mi][nj][v] = sgn * *reinterpret_cast<const double*>(cr + voff + 128 * nj);
            }
    }
    gemm_sstore<A_KC, BM>(buf0, t, ra0);
    gemm_sstore<B_KC>(buf0 + GEMM_TILE, t, rb0);
    __syncthreads();
    const int l15 = lane & 15, lk = lane >> 4;
#ifdef GPAR_GEMM_STAGE_STAMPS
#define GEMM_STAGE_STAMP(stage, which)                                                                                     \
    do {                                                                                                                   \
        if (p.stage_stamps && t == 0 && blockIdx.x >= 1024 && blockIdx.x < 1088 && (stage) < 40)                           \
            p.stage_stamps[((size_t)(blockIdx.x - 1024) * 40 + (stage)) * 3 + (which)] =                                     \
                (GPAR_GEMM_STAGE_STAMPS == 2) ? (long long)__builtin_amdgcn_s_memtime() : (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GEMM_STAGE_STAMP(stage, which) do {} while (0)
#endif
    for (int kt = 0; kt < nk; kt += 2) {
        // even stage: compute buf0; (ra1, rb1) carry stage kt + 1; stage kt + 2 is requested into (ra0, rb0)
        GEMM_STAGE_STAMP(kt, 0);
        gemm_stage<A_KC, B_KC, FAST, BM>(p, buf0, buf0 + GEMM_TILE, acc, m0, n0, min(kbeg + (kt + 2) * GEMM_BK, klast), kend, t, l15,
                                     lk, wm, wn, ra0, rb0);
        GEMM_STAGE_STAMP(kt, 1);
        if (kt + 1 >= nk) break;
        gemm_sstore<A_KC, BM>(buf1, t, ra1);
        gemm_sstore<B_KC>(buf1 + GEMM_TILE, t, rb1);
        GEMM_STAGE_STAMP(kt, 2);
        __syncthreads();
        // odd stage: compute buf1; (ra0, rb0) carry stage kt + 2; stage kt + 3 is requested into (ra1, rb1)
        GEMM_STAGE_STAMP(kt + 1, 0);
        gemm_stage<A_KC, B_KC, FAST, BM>(p, buf1, buf1 + GEMM_TILE, acc, m0, n0, min(kbeg + (kt + 3) * GEMM_BK, klast), kend, t, l15,
                                     lk, wm, wn, ra1, rb1);
        GEMM_STAGE_STAMP(kt + 1, 1);
        if (kt + 2 >= nk) break;
        gemm_sstore<A_KC, BM>(buf0, t, ra0);
        gemm_sstore<B_KC>(buf0 + GEMM_TILE, t, rb0);
        GEMM_STAGE_STAMP(kt + 1, 2);
        __syncthreads();
    }
    __syncthreads();   // the epilogue reuses the stage buffers
}

// Epilogue of an interior tile with a 16-byte aligned C: the accumulators go through LDS (the operand stages are dead after
// the K loop; every wave owns a private slice, so no workgroup barrier is needed) so that each lane ends up with two
// ADJACENT columns of one row: 16-byte accesses, two full 512-byte row segments per wave instruction, instead of 8-byte
// accesses in 128-byte runs straight from the MFMA layout.  Four quarters of 16 rows (one mi each); the C values of quarter
// h + 1 are requested before quarter h is processed.
template <int MI, bool HAS_BETA>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& p, double* smem, gpar_d4 (&acc)[MI][4], int row0, int col0,
                                                  double alpha, double beta, int w, int lane) {
    const int l15 = lane & 15, lk = lane >> 4;
    double* S = smem + w * GEMM_TILE;                     // [16][GEMM_LDT] doubles, private to this wave
    const int rrow = lane >> 5, rcol = (lane & 31) * 2;   // read-back: row 2 q + rrow, columns rcol, rcol + 1
    double* cbase = p.C + (size_t)(row0 + rrow) * p.ldc + col0 + rcol;
    gpar_d2 cv[2][8];
    if (HAS_BETA) {
#pragma unroll
        for (int q = 0; q < 8; ++q) cv[0][q] = *reinterpret_cast<const gpar_d2*>(cbase + (size_t)(2 * q) * p.ldc);
    }
#pragma unroll
    for (int h = 0; h < MI; ++h) {
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int v = 0; v < 4; ++v) S[(lk + 4 * v) * GEMM_LDT + 16 * nj + l15] = acc[h][nj][v];
        if (HAS_BETA && h < MI - 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                cv[(h + 1) & 1][q] = *reinterpret_cast<const gpar_d2*>(cbase + (size_t)(16 * (h + 1) + 2 * q) * p.ldc);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        gpar_d2 v2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v2[q] = *reinterpret_cast<const gpar_d2*>(S + (2 * q + rrow) * GEMM_LDT + rcol);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the next quarter overwrites S
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            gpar_d2 o = v2[q] * alpha;
            if (HAS_BETA) o = gpar_d2{fma(beta, cv[h & 1][q][0], o[0]), fma(beta, cv[h & 1][q][1], o[1])};
#if GPAR_GEMM_NT_STORE
            __builtin_nontemporal_store(o, reinterpret_cast<gpar_d2*>(cbase + (size_t)(16 * h + 2 * q) * p.ldc));
#else
            *reinterpret_cast<gpar_d2*>(cbase + (size_t)(16 * h + 2 * q) * p.ldc) = o;
#endif
        }
    }
}

// `blk` of `nblk`: this workgroup's position among the launch's blocks OF ITS TILE SHAPE (whole tiles, or half tiles: two
// consecutive positions per tile); `tile0`: index of the first tile those blocks cover.
template <bool TA, bool TB, int BM>
__device__ __forceinline__ void gemm_tile_body(GemmArgs& p, double* smem, const int blk, const int nblk, const int tile0) {
    constexpr int MI = BM / 32;   // 16-row blocks per wave
    constexpr bool A_KC = !TA;
    constexpr bool B_KC = TB;

    // XCD-aware, bijective remap of the block index: the dispatcher places block b on XCD b % 8; give each
    // XCD a contiguous run of tiles so neighbouring tiles (shared operand panels) hit the same L2.
    const long long t_start = p.stamps ? (long long)__builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz, chip-wide
    p.A += (size_t)blockIdx.z * p.batch_a;
    p.B += (size_t)blockIdx.z * p.batch_b;
    p.C += (size_t)blockIdx.z * p.batch_c;
    int idx = 0, tm = -1, tn = -1;
    const bool tri_rect = !(p.flags & GPAR_GEMM_C_LOWER) && BM == GEMM_BM;
    if (tri_rect && (p.flags & GPAR_GEMM_K_TO_COL) && !(p.flags & (GPAR_GEMM_K_FROM_ROW | GPAR_GEMM_A_LOWER)) && (p.tiles_n & 7) == 0) {
        // K grows with the tile COLUMN (upper-triangular op(B)): the tiles of one column share their B panel and their K
        // length.  XCD x (the dispatcher puts block b on XCD b % 8) works through whole columns x, x + 8, ... of the
        // longest-first order: equal work per XCD, the shared panel stays in that XCD's L2, short tiles end the launch.
        const int b = blk, xcd = b & 7, j = b >> 3;
        tn = p.tiles_n - 1 - ((j / p.tiles_m) * 8 + xcd);
        tm = j % p.tiles_m;
    } else if (p.flags & (GPAR_GEMM_K_FROM_ROW | GPAR_GEMM_A_LOWER | GPAR_GEMM_K_TO_COL)) {
        // tiles differ in K length by up to n / 128 x (long ones first in the enumeration): contiguous runs per XCD would
        // hand one XCD all the long tiles (the triangular-aware inverse ran at 28 TF that way); deal them round-robin.  (K_FROM_ROW
        // without C_LOWER, grouped by rows like the columns above: 8.6 -> 9.2 ms at 8192^3 / 2 - the row-major enumeration
        // already starts with the longest tiles.)
        idx = tile0 + blk / (GEMM_BM / BM);
    } else {
        const int nb = nblk / (GEMM_BM / BM), b = blk / (GEMM_BM / BM);
        const int xcd = b & 7, q = nb >> 3, r = nb & 7;
        idx = tile0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    if (tm >= 0) {
        // (mapped above)
    } else if (p.flags & GPAR_GEMM_C_LOWER) {
        // lower-trapezoid enumeration: rows tm < tiles_n hold tm+1 tiles, the rest hold tiles_n tiles
        const int tri = p.tiles_n * (p.tiles_n + 1) / 2;
        if (idx < tri) {
            tm = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
            while ((tm + 1) * (tm + 2) / 2 <= idx) ++tm;
            while (tm * (tm + 1) / 2 > idx) --tm;
            tn = idx - tm * (tm + 1) / 2;
        } else {
            const int r = idx - tri;
            tm = p.tiles_n + r / p.tiles_n;
            tn = r % p.tiles_n;
        }
    } else {
        tm = idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int m0 = tm * GEMM_BM + (BM == GEMM_BM ? 0 : (int)(blk & 1) * BM), n0 = tn * GEMM_BN;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const bool a_lower = (p.flags & GPAR_GEMM_A_LOWER) != 0;
    // with a triangular op(A) nothing beyond k = m0 + 127 contributes to this tile
    int kend = a_lower ? min(p.k, m0 + BM) : p.k;
    if (p.flags & GPAR_GEMM_K_TO_COL) kend = min(kend, n0 + GEMM_BN);   // upper-triangular op(B): nothing below the tile's last column
    // K_FROM_ROW: both operands vanish for k < their row (upper-triangular factors): with col <= row nothing
    // before k = m0 contributes to this tile
    int kbeg = (p.flags & GPAR_GEMM_K_FROM_ROW) ? min(m0, kend) : 0;
    if (p.ksplit > 0) {   // split-K: this block's slice of K, accumulated into its own slab of the workspace
        kbeg = max(kbeg, (int)blockIdx.y * p.ksplit);
        kend = min(kend, ((int)blockIdx.y + 1) * p.ksplit);
        if (kend < kbeg) kend = kbeg;
        p.C += (size_t)blockIdx.y * p.part_stride;
    }
    const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;

    gpar_d4 acc[MI][4];   // acc[mi][nj][v]: row 16 mi + (lane >> 4) + 4 v, column 16 nj + (lane & 15) of the wave tile
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = gpar_d4{0.0, 0.0, 0.0, 0.0};

    const bool fastk = p.fastA && p.fastB && !a_lower && ((kend - kbeg) % GEMM_BK == 0);
    const bool inner = (m0 + BM <= p.m) && (n0 + GEMM_BN <= p.n);
    // (see PRELOAD above; not with split-K, whose slabs are partial sums, nor for diagonal tiles of a lower-triangular C)
    const bool c_lower = (p.flags & GPAR_GEMM_C_LOWER) != 0;
    const bool preload = GPAR_GEMM_PRELOAD_C && fastk && inner && nk > 0 && p.ksplit <= 0 && p.beta == 1.0 &&
                         (p.alpha == 1.0 || p.alpha == -1.0) && (!c_lower || n0 + GEMM_BN - 1 <= m0);
    if (preload) gemm_mainloop_pf2<A_KC, B_KC, 1, BM, true>(p, smem, acc, m0, n0, kbeg, kend, nk, t, lane, wm, wn);
    else if (fastk && inner) gemm_mainloop_pf2<A_KC, B_KC, 1, BM>(p, smem, acc, m0, n0, kbeg, kend, nk, t, lane, wm, wn);
    else if (fastk && A_KC && B_KC) gemm_mainloop_pf2<A_KC, B_KC, 2, BM>(p, smem, acc, m0, n0, kbeg, kend, nk, t, lane, wm, wn);
    else gemm_mainloop<A_KC, B_KC, 0, BM>(p, smem, acc, m0, n0, kbeg, kend, nk, a_lower, t, lane, wm, wn);
    const long long t_main = p.stamps ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#if GPAR_GEMM_EPI_PRIO
    __builtin_amdgcn_s_setprio(GPAR_GEMM_EPI_PRIO);
#endif
    const int l15 = lane & 15, lk = lane >> 4;

    // epilogue: register v of acc[mi][nj] holds C[16 mi + (lane >> 4) + 4 v][16 nj + (lane & 15)] of the wave tile
    const int colw = n0 + wn * 64 + l15;
    const int roww = m0 + wm * (BM / 2) + lk;
    const double alpha = p.alpha, beta = preload ? 0.0 : p.beta;
    // Loads of C are never placed behind a per-element condition (hipcc would fence each with vmcnt(0): 64 serial
    // memory round trips per tile, measured as a fixed ~20 us per tile): interior tiles use plain loads/stores, edge
    // tiles load from clamped (always valid) addresses and only the stores are predicated.
    const bool interior = (m0 + BM <= p.m) && (n0 + GEMM_BN <= p.n) && (!c_lower || n0 + GEMM_BN - 1 <= m0);
    if (interior && p.fastC) {
        // (beta == 0 / != 0 are two instantiations: with the test inside, hipcc's counter bookkeeping assumed at every use of a
        // prefetched C value that the NEXT quarter's loads might not have been issued and waited for those too - the prefetch
        // bought nothing and each quarter exposed a trip to HBM: 12 us per tile, `profiles/r01_gemm_phases.txt`)
        if (beta != 0.0) gemm_epilogue_lds<MI, true>(p, smem, acc, m0 + wm * (BM / 2), n0 + wn * 64, alpha, beta, w, lane);
        else gemm_epilogue_lds<MI, false>(p, smem, acc, m0 + wm * (BM / 2), n0 + wn * 64, alpha, beta, w, lane);
    } else if (interior) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            double cv[4][4];
            if (beta != 0.0) {
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        cv[nj][v] = p.C[(size_t)(roww + 16 * mi + 4 * v) * p.ldc + colw + 16 * nj];
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double x = alpha * acc[mi][nj][v];
                    if (beta != 0.0) x = fma(beta, cv[nj][v], x);
                    p.C[(size_t)(roww + 16 * mi + 4 * v) * p.ldc + colw + 16 * nj] = x;
                }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // wave-uniform: 16-row blocks entirely below the last row (all but one block of the one-row overhang of an
            // augmented matrix) or entirely above the diagonal have nothing to store
            const int blk0 = m0 + wm * (BM / 2) + 16 * mi;
            if (blk0 >= p.m || (c_lower && n0 + wn * 64 > blk0 + 15)) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = roww + 16 * mi + 4 * v;
                const int rowc = min(row, p.m - 1);
                double cv[4] = {0.0, 0.0, 0.0, 0.0};
                if (beta != 0.0) {
#pragma unroll
                    for (int nj = 0; nj < 4; ++nj) cv[nj] = p.C[(size_t)rowc * p.ldc + min(colw + 16 * nj, p.n - 1)];
                }
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) {
                    const int col = colw + 16 * nj;
                    const bool ok = row < p.m && col < p.n && (!c_lower || col <= row);
                    double x = alpha * acc[mi][nj][v];
                    if (beta != 0.0) x = fma(beta, cv[nj], x);
                    if (ok) p.C[(size_t)row * p.ldc + col] = x;
                }
            }
        }
    }
    if (p.stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.stamps[blockIdx.x * 4 + 0] = t_start;
        p.stamps[blockIdx.x * 4 + 1] = t_main;
        p.stamps[blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
        // where it ran: HW_ID (register 4: cu_id bits 11:8, sh_id 12, se_id 15:13) and XCC_ID (register 20, bits 3:0)
        p.stamps[blockIdx.x * 4 + 3] = (long long)(__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4) | (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) << 16));
    }
}


// TA: A stored k x m (op(A) = A^T);  TB: B stored n x k (op(B) = B^T).
// ROLE only gives the trailing SYRK of gpar_potrf (ROLE = 1) its own kernel symbol, so that profilers report the
// dominant kernel separately from the small panel-internal updates that share the code.
// BM = 64: the workgroup computes one 64-row half of a 128 x 128 tile (blockIdx.x = 2 * tile + half; wave tile 32 x 64).
// For launches with fewer tiles than compute units: a lone workgroup on a CU has one wave per SIMD and nothing to cover
// its barrier / LDS latencies with (58 % MFMA issue); two half-tile workgroups per CU cover each other's.
// MIXED (whole-tile instantiations of ROLE 1): the chip holds 512 workgroups of this kernel; a launch of T tiles runs T / 512
// rounds of them, and a last round that is less than half full leaves most compute units idle for a whole tile time (n = 16384:
// 2145 tiles, 4.19 rounds, 60.8 TFLOP/s against 65-66 for launches that end on a full round).  With p.ntail > 0 the tiles of that
// last round are computed as half tiles by twice as many workgroups: blocks [0, nfull) take whole tiles, blocks nfull + 2 j,
// nfull + 2 j + 1 the halves of tile nfull + j.  Same arithmetic per element: a tile's rows do not interact.
template <bool TA, bool TB, int ROLE, int BM = 128>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (ROLE == 0 && gpar_pred_skip(p.pred, p.pred_sense)) return;   // (the trailing update of a factorisation, ROLE 1, is never predicated)
    if (ROLE == 1 && BM == GEMM_BM && p.ntail > 0 && (int)blockIdx.x >= p.nfull)
        gemm_tile_body<TA, TB, 64>(p, smem, (int)blockIdx.x - p.nfull, 2 * p.ntail, p.nfull);
    else
        gemm_tile_body<TA, TB, BM>(p, smem, (int)blockIdx.x, ROLE == 1 && BM == GEMM_BM && p.ntail > 0 ? p.nfull : (int)gridDim.x, 0);
}

inline int gemm_num_tiles(int tiles_m, int tiles_n, int flags) {
    if (flags & GPAR_GEMM_C_LOWER) {
        const int tn = tiles_n < tiles_m ? tiles_n : tiles_m;
        return tn * (tn + 1) / 2 + (tiles_m - tn) * tn;
    }
    return tiles_m * tiles_n;
}

// `batch` > 1: that many independent problems of the same shape in one launch, problem z reading A + z * batch_a etc.
static int gemm_launch(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda,
                       const double* B, int ldb, double beta, double* C, int ldc, int flags, hipStream_t stream,
                       int role = 0, int batch = 1, long long batch_a = 0, long long batch_b = 0, long long batch_c = 0) {
    if (m <= 0 || n <= 0 || batch <= 0) return 0;
    GemmArgs p;
    p.batch_a = batch_a; p.batch_b = batch_b; p.batch_c = batch_c;
    p.A = A; p.B = B; p.C = C;
    p.m = m; p.n = n; p.k = k;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta;
    p.flags = flags;
    p.tiles_m = gpar_ceil_div(m, GEMM_BM);
    p.tiles_n = gpar_ceil_div(n, GEMM_BN);
    if (flags & GPAR_GEMM_C_LOWER) {
        // tiles strictly above the diagonal are never needed
        if (p.tiles_n > p.tiles_m) p.tiles_n = p.tiles_m;
    }
    p.fastA = gpar_aligned16(A) && (lda % 2 == 0);
    p.fastB = gpar_aligned16(B) && (ldb % 2 == 0);
    p.fastC = gpar_aligned16(C) && (ldc % 2 == 0);
    p.stamps = nullptr; p.stage_stamps = nullptr;
    p.pred = role == 0 ? g_pred.flag : nullptr; p.pred_sense = g_pred.sense;
    p.ksplit = 0;
    p.part_stride = 0;
    const int ntiles = gemm_num_tiles(p.tiles_m, p.tiles_n, flags);
    // (A super-block tile order for the trailing update - GPAR_GEMM_TILE_BLOCK, rounds 3-4 - cut its counted fetches by a third and
    // did not make it faster: the update is not bound by them.  Retired in round 5; the record is NOTES.md section 7.)
    for (const void* fn : {reinterpret_cast<const void*>(&gemm_f64_kernel<false, false, 0>), reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 0>),
                           reinterpret_cast<const void*>(&gemm_f64_kernel<true, false, 0>), reinterpret_cast<const void*>(&gemm_f64_kernel<true, true, 0>),
                           reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1>), reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 0, 64>),
                           reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 1, 64>), reinterpret_cast<const void*>(&gemm_f64_kernel<false, false, 0, 64>)})
        GPAR_HIP_TRY(gpar_set_max_lds(fn, 160 * 1024));
    // half tiles (two workgroups per 128 x 128 tile) while a launch has fewer tiles than the chip has compute units
    static int half_tiles = -1;
    if (half_tiles < 0) { const char* e = getenv("GPAR_GEMM_HALF_TILES"); half_tiles = e ? atoi(e) : 256; }
    // (triangular-aware products included since round 5 - the K range of a tile follows its own first row / last column whatever its height:
    // the inverse's few-tile products at small n are one K-deep tile each, half as long per workgroup this way: fit(iters=20), one thread,
    // n = 400 103-120 -> 97-100 ms, n = 1024 136-153 -> 118-121)
    static int half_flags = -1;
    if (half_flags < 0) { const char* e = getenv("GPAR_GEMM_HALF_TILES_TRIANGULAR"); half_flags = e ? atoi(e) : 1; }
    // (round 6: the NN form too - the second product of every level of the recursive inversion, X12 = -T X22, is a few-tile launch
    // with K up to the block size: n = 2048 / 4096 ... us per evaluation of a training objective)
    const bool half = !ta && ntiles * batch <= half_tiles && k >= 64 &&
                      (half_flags || !(flags & (GPAR_GEMM_A_LOWER | GPAR_GEMM_K_FROM_ROW | GPAR_GEMM_K_TO_COL)));
    const int GEMM_LDS_REQ = GEMM_LDS_BYTES;   // (padding the request so that ONE workgroup fits a compute unit - a hole for a panel workgroup on every unit - changed nothing: lesson 34)
    // whole tiles, except for a last round that would be at most half full (MIXED, see the kernel): the trailing update alone
    p.nfull = ntiles; p.ntail = 0;
    static int mixed_tail = -1;
    if (mixed_tail < 0) { const char* e = getenv("GPAR_GEMM_MIXED_TAIL"); mixed_tail = e ? atoi(e) : 1; }
    if (mixed_tail && !half && role == 1 && !ta && tb && batch == 1 && k >= 64 && ntiles > 512 && ntiles % 512 != 0 && ntiles % 512 <= 256 &&
        !(flags & (GPAR_GEMM_A_LOWER | GPAR_GEMM_K_FROM_ROW | GPAR_GEMM_K_TO_COL))) {
        p.ntail = ntiles % 512;
        p.nfull = ntiles - p.ntail;
    }
    dim3 grid(half ? 2 * ntiles : p.nfull + 2 * p.ntail, 1, batch), block(256);
    if (half && role == 1) hipLaunchKernelGGL((gemm_f64_kernel<false, true, 1, 64>), grid, block, GEMM_LDS_REQ, stream, p);
    else if (half && tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true, 0, 64>), grid, block, GEMM_LDS_REQ, stream, p);
    else if (half) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 0, 64>), grid, block, GEMM_LDS_REQ, stream, p);
    else if (role == 1 && !ta && tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true, 1>), grid, block, GEMM_LDS_REQ, stream, p);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 0>), grid, block, GEMM_LDS_REQ, stream, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true, 0>), grid, block, GEMM_LDS_REQ, stream, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<true, false, 0>), grid, block, GEMM_LDS_REQ, stream, p);
    else hipLaunchKernelGGL((gemm_f64_kernel<true, true, 0>), grid, block, GEMM_LDS_REQ, stream, p);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// ---- split-K: for outputs with few tiles and a very long K (e.g. the VFE matrix B D^-1 B^T: 1024 x 1024 from
// K = 65536) one tile per workgroup leaves most of the chip idle.  The K range is cut into `splits` slices, each
// producing a partial C in a caller-provided workspace (deterministic: slabs are summed in order by a second kernel).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const double* __restrict__ ws, long long part_stride, int splits,
                                                                 int m, int n, double alpha, double beta, double* __restrict__ C,
                                                                 int ldc, int lower) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= n || row >= m || (lower && col > row)) return;
    double s = 0.0;
    for (int q = 0; q < splits; ++q) s += ws[(size_t)q * part_stride + (size_t)row * n + col];
    double* dst = C + (size_t)row * ldc + col;
    *dst = (beta != 0.0) ? fma(beta, *dst, alpha * s) : alpha * s;
}

static int gemm_splitk_launch(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B,
                              int ldb, double beta, double* C, int ldc, int flags, int splits, double* workspace,
                              hipStream_t stream) {
    if (m <= 0 || n <= 0) return 0;
    if (splits <= 1 || !workspace) return gemm_launch(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, flags, stream);
    GemmArgs p;
    p.batch_a = p.batch_b = p.batch_c = 0;
    p.A = A; p.B = B; p.C = workspace;
    p.m = m; p.n = n; p.k = k;
    p.lda = lda; p.ldb = ldb; p.ldc = n;
    p.alpha = 1.0; p.beta = 0.0;
    p.flags = flags;
    p.tiles_m = gpar_ceil_div(m, GEMM_BM);
    p.tiles_n = gpar_ceil_div(n, GEMM_BN);
    if ((flags & GPAR_GEMM_C_LOWER) && p.tiles_n > p.tiles_m) p.tiles_n = p.tiles_m;
    p.fastA = gpar_aligned16(A) && (lda % 2 == 0);
    p.fastB = gpar_aligned16(B) && (ldb % 2 == 0);
    p.fastC = gpar_aligned16(workspace) && (n % 2 == 0) && (((long long)m * n) % 2 == 0);
    p.stamps = nullptr; p.stage_stamps = nullptr;
    p.pred = nullptr; p.pred_sense = 0;
    const int len = gpar_ceil_div(gpar_ceil_div(k, splits), GEMM_BK) * GEMM_BK;
    p.ksplit = len;
    p.nfull = 0; p.ntail = 0;
    p.part_stride = (long long)m * n;
    const int nsl = gpar_ceil_div(k, len);
    const int ntiles = gemm_num_tiles(p.tiles_m, p.tiles_n, flags);
    for (const void* fn : {reinterpret_cast<const void*>(&gemm_f64_kernel<false, false, 0>), reinterpret_cast<const void*>(&gemm_f64_kernel<false, true, 0>),
                           reinterpret_cast<const void*>(&gemm_f64_kernel<true, false, 0>), reinterpret_cast<const void*>(&gemm_f64_kernel<true, true, 0>)})
        GPAR_HIP_TRY(gpar_set_max_lds(fn, 160 * 1024));
    dim3 grid(ntiles, nsl), block(256);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<false, false, 0>), grid, block, GEMM_LDS_BYTES, stream, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true, 0>), grid, block, GEMM_LDS_BYTES, stream, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<true, false, 0>), grid, block, GEMM_LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((gemm_f64_kernel<true, true, 0>), grid, block, GEMM_LDS_BYTES, stream, p);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(gpar_ceil_div(n, 256), m), dim3(256), 0, stream, (const double*)workspace,
                       p.part_stride, nsl, m, n, alpha, beta, C, ldc, (flags & GPAR_GEMM_C_LOWER) ? 1 : 0);
    GPAR_LAUNCH_CHECK();
    return 0;
}

}  // namespace gpar
