// Feature map and fused Gram-matrix construction for GPAR's composite layer kernels.
//
// The layer kernel (gpar/regression.py:92-180 in the reference) is evaluated in ONE pass over the output:
// every term (EQ / RQ / linear / locally periodic / constant, input and output parts) and the noise
// diagonal are fused, so the n x n matrix is written once and never re-read (the reference materialises
// one n x n temporary per kernel term).  Algorithmic HBM traffic: 8 bytes per stored entry.
//
// Tiling: 64 x 64 outputs per 256-thread workgroup, 4 x 4 per thread; the two 64-row feature panels are
// staged transposed in LDS ([dz][68]) so a thread's 4 rows / 4 columns are one 32-byte LDS read per
// feature, and each output row of the tile is written as 512 contiguous bytes per 16 lanes.
#pragma once
#include "common.h"

namespace gpar {

constexpr int GRAM_T = 64;
constexpr int GRAM_LD = 68;

__global__ __launch_bounds__(256) void featurize_kernel(gpar_fspec_t fs, const double* __restrict__ x, int n, int ldx,
                                                        double* __restrict__ z, int ldz) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dz = fs.dz;
    if (idx >= n * dz) return;
    const int r = idx / dz, q = idx - r * dz;
    const double v = x[(size_t)r * ldx + fs.col[q]];
    double e;
    if (fs.embed[q] == GPAR_EMBED_SIN) e = sin(v * fs.freq[q]);
    else if (fs.embed[q] == GPAR_EMBED_COS) e = cos(v * fs.freq[q]);
    else e = v;
    z[(size_t)r * ldz + q] = e * fs.inv_scale[q];
}

__device__ __forceinline__ double gram_nonlin(int type, double s, double alpha) {
    if (type == GPAR_K_EQ) return exp(-0.5 * s);
    if (type == GPAR_K_RQ) return exp(-alpha * log1p(s / (2.0 * alpha)));
    return s;
}

__global__ __launch_bounds__(256) void gram_kernel(gpar_kspec_t ks, const double* __restrict__ z1, int n1, int ldz1,
                                                   const double* __restrict__ z2, int n2, int ldz2, int dz,
                                                   double* __restrict__ K, int ldk, int flags,
                                                   const double* __restrict__ diag_add, double diag_const, int sym) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int bm = blockIdx.y, bn = blockIdx.x;
    if ((flags & GPAR_GRAM_LOWER) && bn > bm) return;
    double* Za = gsm;
    double* Zb = gsm + (size_t)dz * GRAM_LD;
    const int t = threadIdx.x;
    const int row0 = bm * GRAM_T, col0 = bn * GRAM_T;
    for (int idx = t; idx < GRAM_T * dz; idx += 256) {
        const int r = idx / dz, d = idx - r * dz;
        Za[d * GRAM_LD + r] = (row0 + r < n1) ? z1[(size_t)(row0 + r) * ldz1 + d] : 0.0;
        Zb[d * GRAM_LD + r] = (col0 + r < n2) ? z2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
    }
    __syncthreads();
    const int tx = t & 15, ty = t >> 4;
    double total[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) total[i][j] = 0.0;

    int f = 0;
    for (int term = 0; term < ks.nterms; ++term) {
        double prod[4][4];
        const double coef = ks.coef[term];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) prod[i][j] = coef;
        while (f < ks.nfactors && ks.factor[f].term == term) {
            const int type = ks.factor[f].type, off = ks.factor[f].off, nd = ks.factor[f].nd;
            const double alpha = ks.factor[f].alpha;
            double s[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[i][j] = 0.0;
            for (int d = off; d < off + nd; ++d) {
                double za[4], zb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) za[i] = Za[d * GRAM_LD + 4 * ty + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) zb[j] = Zb[d * GRAM_LD + 4 * tx + j];
                if (type == GPAR_K_LINEAR) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[i][j] = fma(za[i], zb[j], s[i][j]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const double df = za[i] - zb[j];
                            s[i][j] = fma(df, df, s[i][j]);
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) prod[i][j] *= gram_nonlin(type, s[i][j], alpha);
            ++f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) total[i][j] += prod[i][j];
    }

    const bool vec = ((ldk & 1) == 0) && ((((uintptr_t)K) & 15u) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * ty + i;
        if (row >= n1) continue;
        const int col = col0 + 4 * tx;
        if (sym) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col + j == row) total[i][j] += (diag_add ? diag_add[row] : 0.0) + diag_const;
        }
        double* out = K + (size_t)row * ldk + col;
        if (vec && col + 3 < n2) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2*>(out) = d2{total[i][0], total[i][1]};
            *reinterpret_cast<d2*>(out + 2) = d2{total[i][2], total[i][3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (col + j < n2) out[j] = total[i][j];
        }
    }
}

__global__ __launch_bounds__(256) void gram_diag_kernel(gpar_kspec_t ks, const double* __restrict__ z, int n, int ldz,
                                                        double* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double total = 0.0;
    int f = 0;
    for (int term = 0; term < ks.nterms; ++term) {
        double prod = ks.coef[term];
        while (f < ks.nfactors && ks.factor[f].term == term) {
            double s = 0.0;
            if (ks.factor[f].type == GPAR_K_LINEAR) {
                for (int d = ks.factor[f].off; d < ks.factor[f].off + ks.factor[f].nd; ++d) {
                    const double v = z[(size_t)r * ldz + d];
                    s = fma(v, v, s);
                }
            }
            prod *= gram_nonlin(ks.factor[f].type, s, ks.factor[f].alpha);
            ++f;
        }
        total += prod;
    }
    out[r] = total;
}

}  // namespace gpar
