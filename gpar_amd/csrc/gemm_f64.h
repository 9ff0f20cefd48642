// fp64 GEMM / SYRK on the gfx950 matrix cores.
//
//   C <- alpha * op(A) op(B) + beta * C,   op(A): m x k, op(B): k x n, all row-major.
//
// Instruction choice (measured on MI355X, tools/ubench_mfma_f64.hip, profiles/r01_ubench_mfma_f64.txt):
// v_mfma_f64_16x16x4_f64 saturates at 47-49 TFLOP/s on gfx950 (~100+ cycles per instruction), whereas
// v_mfma_f64_4x4x4_4b_f64 issues every ~18 cycles: 64 TFLOP/s from one wave per SIMD, 71 TFLOP/s with
// four (spec fp64 matrix peak: 78.6).  The kernel is therefore built on the 4x4x4 4-block form.
//
// Lane maps of v_mfma_f64_4x4x4_4b_f64 (probed, tools/probe_mfma_f64_4x4x4.hip; CBSZ/ABID broadcast is
// ignored for this opcode):   A-operand lane l: k = l>>4, x = l&15;   B-operand lane l: k = l>>4, y = l&15;
//   D lane l (block b=(l>>2)&3, i=l>>4, j=l&3) = sum_k A[x=4b+i][k] * B[y=4b+j][k]   (block-diagonal only).
// We feed the N-side operand (16 distinct columns) as "A" and the M-side operand (4 rows, replicated over
// the four blocks via an LDS broadcast read) as "B", so one instruction yields a 4(row) x 16(col) patch of C
// with lane l holding C[row = l&3][col = 4*((l>>2)&3) + (l>>4)]: each row is a full 128-byte line on store.
//
// Work decomposition: 256 threads = 4 waves (2 x 2), workgroup tile 128 x 128, wave tile 64 x 64 =
// 16 (row patches) x 4 (column patches) accumulators = 64 f64 = 128 VGPRs; BK = 16 per stage, two LDS
// stages (73.7 KB) so two workgroups share a CU (2 waves per SIMD) and cover each other's barrier/staging.
// Operand tiles are staged global -> registers -> LDS (the loads for stage s+1 are issued before the
// MFMAs of stage s), in one of two padded, bank-conflict-free LDS images:
//   KC  [128][18]  for operands stored with k contiguous   (read: 18*r + k  distinct mod 32 per half wave)
//   MC  [16][144]  for operands stored with m/n contiguous (read: 144*k + r)
#pragma once
#include "common.h"

namespace gpar {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 16;
constexpr int GEMM_LDKC = 18;
constexpr int GEMM_LDMC = 144;
constexpr int GEMM_TILE = 2304;  // doubles per operand stage (128*18 == 16*144)
constexpr int GEMM_LDS_BYTES = 4 * GEMM_TILE * 8;

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int m, n, k;
    int lda, ldb, ldc;
    double alpha, beta;
    int flags;
    int tiles_m, tiles_n;
    int fastA, fastB;
};

typedef double gpar_d2 __attribute__((ext_vector_type(2)));

// Global -> register stage of one operand tile (4 x 16-byte chunks per thread).
//   KC: tile element (r, kk) lives at g[(r0 + r) * ld + k0 + kk]     (r < 128, kk < 16)
//   MC: tile element (r, kk) lives at g[(k0 + kk) * ld + r0 + r]
// `lower`: entries with global k > global r are treated as zero (triangular op(A)).
template <bool KC>
__device__ __forceinline__ void gemm_gload(const double* __restrict__ g, int ld, int r0, int rmax, int k0,
                                           int kmax, bool vec, bool lower, int t, gpar_d2 (&reg)[4]) {
    const bool full = vec && (r0 + GEMM_BM <= rmax) && (k0 + GEMM_BK <= kmax) && (!lower || k0 + GEMM_BK - 1 <= r0);
    if (full) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = t + 256 * q;
            if (KC) {
                const int r = c >> 3, kk = (c & 7) * 2;
                reg[q] = *reinterpret_cast<const gpar_d2*>(g + (size_t)(r0 + r) * ld + k0 + kk);
            } else {
                const int kk = c >> 6, r = (c & 63) * 2;
                reg[q] = *reinterpret_cast<const gpar_d2*>(g + (size_t)(k0 + kk) * ld + r0 + r);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = t + 256 * q;
            double v0 = 0.0, v1 = 0.0;
            if (KC) {
                const int r = r0 + (c >> 3), kk = k0 + (c & 7) * 2;
                if (r < rmax) {
                    if (kk < kmax && (!lower || kk <= r)) v0 = g[(size_t)r * ld + kk];
                    if (kk + 1 < kmax && (!lower || kk + 1 <= r)) v1 = g[(size_t)r * ld + kk + 1];
                }
            } else {
                const int kk = k0 + (c >> 6), r = r0 + (c & 63) * 2;
                if (kk < kmax) {
                    if (r < rmax && (!lower || kk <= r)) v0 = g[(size_t)kk * ld + r];
                    if (r + 1 < rmax && (!lower || kk <= r + 1)) v1 = g[(size_t)kk * ld + r + 1];
                }
            }
            reg[q] = gpar_d2{v0, v1};
        }
    }
}

template <bool KC>
__device__ __forceinline__ void gemm_sstore(double* __restrict__ s, int t, const gpar_d2 (&reg)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = t + 256 * q;
        if (KC) {
            const int r = c >> 3, kk = (c & 7) * 2;
            *reinterpret_cast<gpar_d2*>(s + r * GEMM_LDKC + kk) = reg[q];
        } else {
            const int kk = c >> 6, r = (c & 63) * 2;
            *reinterpret_cast<gpar_d2*>(s + kk * GEMM_LDMC + r) = reg[q];
        }
    }
}

// TA: A stored k x m (op(A) = A^T);  TB: B stored n x k (op(B) = B^T).
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool A_KC = !TA;
    constexpr bool B_KC = TB;

    // XCD-aware, bijective remap of the block index: the dispatcher places block b on XCD b % 8; give each
    // XCD a contiguous run of tiles so neighbouring tiles (shared operand panels) hit the same L2.
    int idx;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nb >> 3, r = nb & 7;
        idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int tm, tn;
    if (p.flags & GPAR_GEMM_C_LOWER) {
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
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const bool a_lower = (p.flags & GPAR_GEMM_A_LOWER) != 0;
    // with a triangular op(A) nothing beyond k = m0 + 127 contributes to this tile
    const int kend = a_lower ? min(p.k, m0 + GEMM_BM) : p.k;
    const int nk = (kend + GEMM_BK - 1) / GEMM_BK;

    double acc[16][4];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;

    gpar_d2 ra[4], rb[4];
    if (nk > 0) {
        gemm_gload<A_KC>(p.A, p.lda, m0, p.m, 0, kend, p.fastA, a_lower, t, ra);
        gemm_gload<B_KC>(p.B, p.ldb, n0, p.n, 0, kend, p.fastB, false, t, rb);
        gemm_sstore<A_KC>(smem, t, ra);
        gemm_sstore<B_KC>(smem + GEMM_TILE, t, rb);
    }
    __syncthreads();

    const int l3 = lane & 3, l15 = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const double* As = smem + (kt & 1) * 2 * GEMM_TILE;
        const double* Bs = As + GEMM_TILE;
        const bool more = kt + 1 < nk;
        if (more) {
            gemm_gload<A_KC>(p.A, p.lda, m0, p.m, (kt + 1) * GEMM_BK, kend, p.fastA, a_lower, t, ra);
            gemm_gload<B_KC>(p.B, p.ldb, n0, p.n, (kt + 1) * GEMM_BK, kend, p.fastB, false, t, rb);
        }
#pragma unroll 1
        for (int k4 = 0; k4 < 4; ++k4) {
            const int kk = k4 * 4 + lk;
            double pf[16], qf[4];
#pragma unroll
            for (int mi = 0; mi < 16; ++mi) {
                const int r = wm * 64 + 4 * mi + l3;
                pf[mi] = A_KC ? As[r * GEMM_LDKC + kk] : As[kk * GEMM_LDMC + r];
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int c = wn * 64 + 16 * nj + l15;
                qf[nj] = B_KC ? Bs[c * GEMM_LDKC + kk] : Bs[kk * GEMM_LDMC + c];
            }
#pragma unroll
            for (int mi = 0; mi < 16; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj)
                    acc[mi][nj] = __builtin_amdgcn_mfma_f64_4x4x4f64(qf[nj], pf[mi], acc[mi][nj], 0, 0, 0);
        }
        if (more) {
            double* An = smem + ((kt + 1) & 1) * 2 * GEMM_TILE;
            gemm_sstore<A_KC>(An, t, ra);
            gemm_sstore<B_KC>(An + GEMM_TILE, t, rb);
        }
        __syncthreads();
    }

    // epilogue: lane l of acc[mi][nj] holds C[4*mi + (l&3)][16*nj + 4*((l>>2)&3) + (l>>4)] of the wave tile
    const bool c_lower = (p.flags & GPAR_GEMM_C_LOWER) != 0;
    const int colw = n0 + wn * 64 + 4 * ((lane >> 2) & 3) + lk;
    const int roww = m0 + wm * 64 + l3;
    const double alpha = p.alpha, beta = p.beta;
#pragma unroll
    for (int mi = 0; mi < 16; ++mi) {
        const int row = roww + 4 * mi;
        double* crow = p.C + (size_t)row * p.ldc;
        if (beta != 0.0) {
            double cv[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int col = colw + 16 * nj;
                const bool ok = row < p.m && col < p.n && (!c_lower || col <= row);
                cv[nj] = ok ? crow[col] : 0.0;
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int col = colw + 16 * nj;
                const bool ok = row < p.m && col < p.n && (!c_lower || col <= row);
                if (ok) crow[col] = alpha * acc[mi][nj] + beta * cv[nj];
            }
        } else {
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int col = colw + 16 * nj;
                const bool ok = row < p.m && col < p.n && (!c_lower || col <= row);
                if (ok) crow[col] = alpha * acc[mi][nj];
            }
        }
    }
}

inline int gemm_num_tiles(int tiles_m, int tiles_n, int flags) {
    if (flags & GPAR_GEMM_C_LOWER) {
        const int tn = tiles_n < tiles_m ? tiles_n : tiles_m;
        return tn * (tn + 1) / 2 + (tiles_m - tn) * tn;
    }
    return tiles_m * tiles_n;
}

static int gemm_launch(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda,
                       const double* B, int ldb, double beta, double* C, int ldc, int flags, hipStream_t stream) {
    if (m <= 0 || n <= 0) return 0;
    GemmArgs p;
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
    const int ntiles = gemm_num_tiles(p.tiles_m, p.tiles_n, flags);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        attr_done = true;
    }
    dim3 grid(ntiles), block(256);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, block, GEMM_LDS_BYTES, stream, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, block, GEMM_LDS_BYTES, stream, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, block, GEMM_LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, block, GEMM_LDS_BYTES, stream, p);
    GPAR_LAUNCH_CHECK();
    return 0;
}

}  // namespace gpar
