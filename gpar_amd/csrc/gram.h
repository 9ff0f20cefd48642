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

// shared with the generated kernels (gram_jit.h), one text for both: typedefs, GRAM_T / GRAM_LD, gram_accum*, gram_exph8, gram_rqh8
#define GPAR_DEVICE_CODE(...) __VA_ARGS__
#include "gram_math.inc"
#undef GPAR_DEVICE_CODE

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

// Two passes of a 4 x 2 micro-tile per thread.  Per product term the exponents of its EQ / RQ factors are SUMMED and
// exponentiated once (a locally periodic term is exp(-(s_per + s_dec) / 2): one exponential, not two), linear factors
// multiply into a separate product; the factor type is wave-uniform and dispatched once per factor.
__global__ __launch_bounds__(256) void gram_kernel(gpar_kspec_t ks, const double* __restrict__ z1, int n1, int ldz1,
                                                   const double* __restrict__ z2, int n2, int ldz2, int dz,
                                                   double* __restrict__ K, int ldk, int flags,
                                                   const double* __restrict__ diag_add, double diag_const,
                                                   const double* __restrict__ row_scale, int sym, long long batch_z = 0,
                                                   long long batch_k = 0) {
    extern __shared__ __attribute__((aligned(32))) double gsm[];
    // batched launch (gpar_gram_batch): blockIdx.z picks one of several input sets of one size and its output matrix
    z1 += (size_t)blockIdx.z * batch_z;
    z2 += (size_t)blockIdx.z * batch_z;
    K += (size_t)blockIdx.z * batch_k;
    int bm = blockIdx.y, bn = blockIdx.x;
    if (flags & GPAR_GRAM_LOWER) {
        // 1-D grid over the tiles of the lower triangle (a 2-D grid would launch as many empty workgroups again)
        const int tile = blockIdx.x;
        bm = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
        while ((bm + 1) * (bm + 2) / 2 <= tile) ++bm;
        while (bm * (bm + 1) / 2 > tile) --bm;
        bn = tile - bm * (bm + 1) / 2;
    }
    const double* tab = gsm;                         // exponential / logarithm tables (gram_math.inc), then the two feature panels
    double* Za = gsm + GRAM_TAB_DOUBLES;
    double* Zb = Za + (size_t)dz * GRAM_LD;
    const int t = threadIdx.x;
    const int row0 = bm * GRAM_T, col0 = bn * GRAM_T;
    gram_load_tables(gsm, t);
    for (int idx = t; idx < GRAM_T * dz; idx += 256) {
        const int r = idx / dz, d = idx - r * dz;
        Za[d * GRAM_LD + r] = (row0 + r < n1) ? z1[(size_t)(row0 + r) * ldz1 + d] : 0.0;
        Zb[d * GRAM_LD + r] = (col0 + r < n2) ? z2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
    }
    __syncthreads();
    const int tx = t & 15, ty = t >> 4;
    const bool vec = ((ldk & 1) == 0) && ((((uintptr_t)K) & 15u) == 0);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int cb = 32 * h + 2 * tx;   // first of this thread's two columns within the tile
        double total[8];                  // entry (i, j) of the 4 x 2 micro-tile at 2 i + j
#pragma unroll
        for (int e = 0; e < 8; ++e) total[e] = 0.0;
        int f = 0;
        for (int term = 0; term < ks.nterms; ++term) {
            double expo[8], lin[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { expo[e] = 0.0; lin[e] = ks.coef[term]; }
            bool any_exp = false;
            while (f < ks.nfactors && ks.factor[f].term == term) {
                const int type = ks.factor[f].type, off = ks.factor[f].off, nd = ks.factor[f].nd;
                double s[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = 0.0;
                if (type == GPAR_K_LINEAR) {
                    gram_accum_dims<true>(Za, Zb, off, nd, ty, cb, s);
#pragma unroll
                    for (int e = 0; e < 8; ++e) lin[e] *= s[e];
                } else {
                    gram_accum_dims<false>(Za, Zb, off, nd, ty, cb, s);
                    any_exp = true;
                    if (type == GPAR_K_EQ) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) expo[e] += s[e];   // doubled exponent: gram_exph8 takes exp(-E / 2)
                    } else {   // RQ: (1 + s / 2 alpha)^-alpha = exp(-alpha log1p(s / 2 alpha))
                        gram_rqh8(s, ks.factor[f].alpha, expo, tab);
                    }
                }
                ++f;
            }
            if (any_exp) {
                gram_exph8(expo, tab);
#pragma unroll
                for (int e = 0; e < 8; ++e) total[e] = fma(lin[e], expo[e], total[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) total[e] += lin[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * ty + i;
            if (row >= n1) continue;
            const int col = col0 + cb;
            double v0 = total[2 * i], v1 = total[2 * i + 1];
            if (row_scale) { const double rs = row_scale[row]; v0 *= rs; v1 *= rs; }
            if (sym && col0 == row0) {   // (the noise diagonal only exists in diagonal tiles: a workgroup-uniform branch)
                const double dadd = (diag_add ? diag_add[row] : 0.0) + diag_const;
                if (col == row) v0 += dadd;
                if (col + 1 == row) v1 += dadd;
            }
            double* out = K + (size_t)row * ldk + col;
            if (vec && col + 1 < n2) {
                *reinterpret_cast<g_d2*>(out) = g_d2{v0, v1};
            } else {
                if (col < n2) out[0] = v0;
                if (col + 1 < n2) out[1] = v1;
            }
        }
    }
}

// ---- one launch builds the augmented matrices of up to LS_CHUNK layers of a lock-step evaluation (gpar_logpdf_lockstep) ----------
// Layer b = blockIdx.z: features straight from the design matrix (what featurize_kernel computes, per tile instead of per launch),
// the lower triangle of k_b + diag(noise_b / w) + jitter I by the SAME term loop as gram_kernel (same helpers, same order: same
// bits), the observations into row n, corner / log-determinant / info word zeroed.  The specifications travel BY VALUE in the
// kernel arguments (a compact feature map: at most LS_MAXDZ dims, columns < 256), so nothing is copied to the device and a
// small evaluation's build is one launch instead of 2 p + 1.
constexpr int LS_CHUNK = 4;
constexpr int LS_MAXDZ = 24;
struct LockstepFeat {
    int dz, pad_;
    unsigned char col[LS_MAXDZ];
    unsigned char embed[LS_MAXDZ];
    double inv_scale[LS_MAXDZ];
    double freq[LS_MAXDZ];
};
struct LockstepSpecs {
    gpar_kspec_t ks[LS_CHUNK];
    LockstepFeat fs[LS_CHUNK];
    double noise[LS_CHUNK];
    int ycol[LS_CHUNK];
};
static_assert(sizeof(LockstepSpecs) <= 3600, "the specifications of a chunk must fit the kernel-argument segment");

__device__ __forceinline__ double lockstep_feature(const LockstepFeat& fs, int q, const double* __restrict__ xrow) {
    const double v = xrow[fs.col[q]];
    double e;
    if (fs.embed[q] == GPAR_EMBED_SIN) e = sin(v * fs.freq[q]);
    else if (fs.embed[q] == GPAR_EMBED_COS) e = cos(v * fs.freq[q]);
    else e = v;
    return e * fs.inv_scale[q];
}

__global__ __launch_bounds__(256) void lockstep_build_kernel(LockstepSpecs sp, const double* __restrict__ x, int n, int ldx,
                                                             const double* __restrict__ y, int ldy, const double* __restrict__ w, int ldw,
                                                             double jitter, double* __restrict__ A, int lda, long long stride_a,
                                                             double* __restrict__ logdet, int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(32))) double gsm[];
    const int b = blockIdx.z;
    const gpar_kspec_t& ks = sp.ks[b];
    const LockstepFeat& fs = sp.fs[b];
    const int dz = fs.dz;
    double* K = A + (size_t)b * stride_a;
    const int tile = blockIdx.x;
    int bm = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
    while ((bm + 1) * (bm + 2) / 2 <= tile) ++bm;
    while (bm * (bm + 1) / 2 > tile) --bm;
    const int bn = tile - bm * (bm + 1) / 2;
    const double* tab = gsm;
    double* Za = gsm + GRAM_TAB_DOUBLES;
    double* Zb = Za + (size_t)(dz > 0 ? dz : 1) * GRAM_LD;
    const int t = threadIdx.x;
    const int row0 = bm * GRAM_T, col0 = bn * GRAM_T;
    gram_load_tables(gsm, t);
    for (int idx = t; idx < GRAM_T * dz; idx += 256) {
        const int r = idx / dz, d = idx - r * dz;
        Za[d * GRAM_LD + r] = (row0 + r < n) ? lockstep_feature(fs, d, x + (size_t)(row0 + r) * ldx) : 0.0;
        Zb[d * GRAM_LD + r] = (col0 + r < n) ? lockstep_feature(fs, d, x + (size_t)(col0 + r) * ldx) : 0.0;
    }
    const int ycol = sp.ycol[b];
    if (bm == bn) {   // the diagonal tiles carry the observations of their columns into row n; the first one the scalars
        if (t < GRAM_T && col0 + t < n) K[(size_t)n * lda + col0 + t] = y[(size_t)(col0 + t) * ldy + ycol];
        if (tile == 0 && t == 0) {
            K[(size_t)n * lda + n] = 0.0;
            logdet[b] = 0.0;
            info[b] = 0;
        }
    }
    __syncthreads();
    const int tx = t & 15, ty = t >> 4;
    const bool vec = ((lda & 1) == 0) && ((((uintptr_t)K) & 15u) == 0);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int cb = 32 * h + 2 * tx;
        double total[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) total[e] = 0.0;
        int f = 0;
        for (int term = 0; term < ks.nterms; ++term) {
            double expo[8], lin[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { expo[e] = 0.0; lin[e] = ks.coef[term]; }
            bool any_exp = false;
            while (f < ks.nfactors && ks.factor[f].term == term) {
                const int type = ks.factor[f].type, off = ks.factor[f].off, nd = ks.factor[f].nd;
                double s[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = 0.0;
                if (type == GPAR_K_LINEAR) {
                    gram_accum_dims<true>(Za, Zb, off, nd, ty, cb, s);
#pragma unroll
                    for (int e = 0; e < 8; ++e) lin[e] *= s[e];
                } else {
                    gram_accum_dims<false>(Za, Zb, off, nd, ty, cb, s);
                    any_exp = true;
                    if (type == GPAR_K_EQ) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) expo[e] += s[e];
                    } else {
                        gram_rqh8(s, ks.factor[f].alpha, expo, tab);
                    }
                }
                ++f;
            }
            if (any_exp) {
                gram_exph8(expo, tab);
#pragma unroll
                for (int e = 0; e < 8; ++e) total[e] = fma(lin[e], expo[e], total[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) total[e] += lin[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * ty + i;
            if (row >= n) continue;
            const int col = col0 + cb;
            double v0 = total[2 * i], v1 = total[2 * i + 1];
            if (col0 == row0) {
                // noise_b / w: an IEEE division, then + jitter - what the per-layer route adds as diag_add[row] + diag_const
                const double nd_ = w ? sp.noise[b] / w[(size_t)row * ldw + ycol] : sp.noise[b];
                const double dadd = nd_ + jitter;
                if (col == row) v0 += dadd;
                if (col + 1 == row) v1 += dadd;
            }
            double* out = K + (size_t)row * lda + col;
            if (vec && col + 1 < n) {
                *reinterpret_cast<g_d2*>(out) = g_d2{v0, v1};
            } else {
                if (col < n) out[0] = v0;
                if (col + 1 < n) out[1] = v1;
            }
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


// ---------------------------------------------------------------------------------------------------
// Gradient of the log marginal likelihood with respect to every kernel parameter, one fused pass over
// W = alpha alpha^T - K^-1 (lower triangle; symmetric):   d/dtheta = 1/2 sum_ab W_ab dK_ab/dtheta.
// Instead of one n x n derivative matrix per parameter (what autograd materialises in the reference), the
// pass accumulates a fixed vector of moment sums from which the host forms every derivative:
//   C_t  = sum W * prod_f phi_f                               (term coefficient)
//   Al_f = sum W * rest_f * phi_f * ((s/2a)/(1+s/2a) - log1p(s/2a))        (RQ alpha)
//   A_q  = sum W * g_f * m_q,   m_q = (z_aq - z_bq)^2 (EQ/RQ)  or  z_aq z_bq (linear)    (length scales)
//   P_q  = sum W * g_f * (z_aq - z_bq)(z'_aq - z'_bq),  z' = dz/dfreq                      (periods)
// with rest_f = coef * prod_{f' != f} phi_f' and g_f = rest_f * dphi_f/ds.  Algorithmic HBM traffic: the lower
// triangle of W once (4 n^2 bytes).  Deterministic: per-wave LDS slots, per-block partials, ordered reduce.
constexpr int GRAD_OFF_C = 0;
constexpr int GRAD_OFF_AL = GPAR_MAX_TERMS;
constexpr int GRAD_OFF_A = GPAR_MAX_TERMS + GPAR_MAX_FACTORS;
constexpr int GRAD_OFF_P = GRAD_OFF_A + GPAR_MAX_DIMS;
constexpr int GRAD_NACC = GRAD_OFF_P + GPAR_MAX_DIMS;
constexpr int GRAD_MAXF = 4;  // factors per product term handled by the gradient pass

__global__ __launch_bounds__(256) void featurize_dfreq_kernel(gpar_fspec_t fs, const double* __restrict__ x, int n, int ldx,
                                                              double* __restrict__ zd, int ldz) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dz = fs.dz;
    if (idx >= n * dz) return;
    const int r = idx / dz, q = idx - r * dz;
    const double v = x[(size_t)r * ldx + fs.col[q]];
    double e = 0.0;
    if (fs.embed[q] == GPAR_EMBED_SIN) e = v * cos(v * fs.freq[q]);
    else if (fs.embed[q] == GPAR_EMBED_COS) e = -v * sin(v * fs.freq[q]);
    zd[(size_t)r * ldz + q] = e * fs.inv_scale[q];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// mode GPAR_GRAD_SYM : one point set (z2 == z1), W symmetric n1 x n1 given by its lower triangle (sum over ALL pairs);
//      GPAR_GRAD_RECT: two point sets, W a full n1 x n2 matrix (cross-Gram weights of the inducing-point bound);
//      GPAR_GRAD_DIAG: one point set, W a vector: only the pairs (a, a) (derivative of the prior variances k(x_a, x_a)).
__global__ __launch_bounds__(256) void gram_grad_kernel(gpar_kspec_t ks, const double* __restrict__ z, const double* __restrict__ zd,
                                                        int n, int ldz, const double* __restrict__ z2,
                                                        const double* __restrict__ zd2, int n2, int ldz2, int dz,
                                                        const double* __restrict__ W, int ldw, int mode,
                                                        double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int dzl = dz > 0 ? dz : 1;
    double* Za = gsm;
    double* Zb = Za + (size_t)dzl * GRAM_LD;
    double* Zda = Zb + (size_t)dzl * GRAM_LD;
    double* Zdb = Zda + (size_t)dzl * GRAM_LD;
    double* acc = Zdb + (size_t)dzl * GRAM_LD;  // [4][GRAD_NACC]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < 4 * GRAD_NACC; i += 256) acc[i] = 0.0;
    double* myacc = acc + wv * GRAD_NACC;
    const int tx = t & 15, ty = t >> 4;
    const int nt = (n + GRAM_T - 1) / GRAM_T, nt2 = (n2 + GRAM_T - 1) / GRAM_T;
    const int ntiles = mode == GPAR_GRAD_SYM ? nt * (nt + 1) / 2 : (mode == GPAR_GRAD_RECT ? nt * nt2 : nt);
    const bool has_zd = zd != nullptr;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bm, bn;
        if (mode == GPAR_GRAD_SYM) {
            bm = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
            while ((bm + 1) * (bm + 2) / 2 <= tile) ++bm;
            while (bm * (bm + 1) / 2 > tile) --bm;
            bn = tile - bm * (bm + 1) / 2;
        } else if (mode == GPAR_GRAD_RECT) {
            bm = tile / nt2;
            bn = tile - bm * nt2;
        } else {
            bm = bn = tile;
        }
        const int row0 = bm * GRAM_T, col0 = bn * GRAM_T;
        __syncthreads();
        for (int idx = t; idx < GRAM_T * dz; idx += 256) {
            const int r = idx / dz, d = idx - r * dz;
            const bool ra = row0 + r < n, rb = col0 + r < n2;
            Za[d * GRAM_LD + r] = ra ? z[(size_t)(row0 + r) * ldz + d] : 0.0;
            Zb[d * GRAM_LD + r] = rb ? z2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
            if (has_zd) {
                Zda[d * GRAM_LD + r] = ra ? zd[(size_t)(row0 + r) * ldz + d] : 0.0;
                Zdb[d * GRAM_LD + r] = rb ? zd2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
            }
        }
        __syncthreads();
        // symmetric weights: strictly-lower tiles stand for their mirror image too
        double w[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + 4 * ty + i, col = col0 + 4 * tx + j;
                double v = 0.0;
                if (row < n && col < n2) {
                    if (mode == GPAR_GRAD_SYM) {
                        if (bm == bn) v = (col <= row) ? W[(size_t)row * ldw + col] : W[(size_t)col * ldw + row];
                        else v = 2.0 * W[(size_t)row * ldw + col];
                    } else if (mode == GPAR_GRAD_RECT) {
                        v = W[(size_t)row * ldw + col];
                    } else {
                        v = (row == col) ? W[row] : 0.0;
                    }
                }
                w[i][j] = v;
            }

        int f0 = 0;
        for (int term = 0; term < ks.nterms; ++term) {
            int nf = 0;
            while (f0 + nf < ks.nfactors && ks.factor[f0 + nf].term == term) ++nf;
            const double coef = ks.coef[term];
            double phi[GRAD_MAXF][4][4], sv[GRAD_MAXF][4][4];
#pragma unroll
            for (int ff = 0; ff < GRAD_MAXF; ++ff) {
                if (ff < nf) {
                    const gpar_factor_t fa = ks.factor[f0 + ff];
                    double s[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[i][j] = 0.0;
                    for (int d = fa.off; d < fa.off + fa.nd; ++d) {
                        double za[4], zb[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) za[i] = Za[d * GRAM_LD + 4 * ty + i];
#pragma unroll
                        for (int j = 0; j < 4; ++j) zb[j] = Zb[d * GRAM_LD + 4 * tx + j];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (fa.type == GPAR_K_LINEAR) s[i][j] = fma(za[i], zb[j], s[i][j]);
                                else { const double df = za[i] - zb[j]; s[i][j] = fma(df, df, s[i][j]); }
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { sv[ff][i][j] = s[i][j]; phi[ff][i][j] = gram_nonlin(fa.type, s[i][j], fa.alpha); }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { sv[ff][i][j] = 0.0; phi[ff][i][j] = 1.0; }
                }
            }
            // coefficient
            {
                double c = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c = fma(w[i][j], phi[0][i][j] * phi[1][i][j] * phi[2][i][j] * phi[3][i][j], c);
                c = wave_sum(c);
                if (lane == 0) myacc[GRAD_OFF_C + term] += c;
            }
#pragma unroll
            for (int ff = 0; ff < GRAD_MAXF; ++ff) {
                if (ff >= nf) continue;
                const gpar_factor_t fa = ks.factor[f0 + ff];
                double g[4][4];
                double al = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double rest = coef;
#pragma unroll
                        for (int f2 = 0; f2 < GRAD_MAXF; ++f2)
                            if (f2 != ff) rest *= phi[f2][i][j];
                        if (fa.type == GPAR_K_EQ) g[i][j] = w[i][j] * rest * (-0.5 * phi[ff][i][j]);
                        else if (fa.type == GPAR_K_RQ) {
                            const double tq = sv[ff][i][j] / (2.0 * fa.alpha), base = 1.0 + tq;
                            g[i][j] = w[i][j] * rest * (-0.5 * phi[ff][i][j] / base);
                            al = fma(w[i][j] * rest * phi[ff][i][j], tq / base - log1p(tq), al);
                        } else g[i][j] = w[i][j] * rest;
                    }
                if (fa.type == GPAR_K_RQ) {
                    al = wave_sum(al);
                    if (lane == 0) myacc[GRAD_OFF_AL + f0 + ff] += al;
                }
                for (int d = fa.off; d < fa.off + fa.nd; ++d) {
                    double za[4], zb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) za[i] = Za[d * GRAM_LD + 4 * ty + i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) zb[j] = Zb[d * GRAM_LD + 4 * tx + j];
                    double a = 0.0, pp = 0.0;
                    if (fa.type == GPAR_K_LINEAR) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) a = fma(g[i][j], za[i] * zb[j], a);
                    } else {
                        double zda[4] = {0, 0, 0, 0}, zdb[4] = {0, 0, 0, 0};
                        if (has_zd) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) zda[i] = Zda[d * GRAM_LD + 4 * ty + i];
#pragma unroll
                            for (int j = 0; j < 4; ++j) zdb[j] = Zdb[d * GRAM_LD + 4 * tx + j];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const double df = za[i] - zb[j];
                                a = fma(g[i][j], df * df, a);
                                pp = fma(g[i][j], df * (zda[i] - zdb[j]), pp);
                            }
                    }
                    a = wave_sum(a);
                    if (lane == 0) myacc[GRAD_OFF_A + d] += a;
                    if (has_zd && fa.type != GPAR_K_LINEAR) {
                        pp = wave_sum(pp);
                        if (lane == 0) myacc[GRAD_OFF_P + d] += pp;
                    }
                }
            }
            f0 += nf;
        }
    }
    __syncthreads();
    for (int k = t; k < GRAD_NACC; k += 256)
        partial[(size_t)blockIdx.x * GRAD_NACC + k] = ((acc[k] + acc[GRAD_NACC + k]) + acc[2 * GRAD_NACC + k]) + acc[3 * GRAD_NACC + k];
}

// ---------------------------------------------------------------------------------------------------
// Gradient with respect to the INPUTS (in feature space): for a weight matrix W over pairs (a, b),
//     out[a][q] = sum_b W(a, b) * d k(z1_a, z2_b) / d z1_a[q]
//               = sum_b W(a, b) * rest_f(a, b) * dphi_f/ds(a, b) * 2 (z1_a[q] - z2_b[q])     (EQ / RQ factor f holding feature q)
//               = sum_b W(a, b) * rest_f(a, b) * z2_b[q]                                      (linear factor)
// with rest_f = coef * prod_{f' != f} phi_f'.  What torch autograd computes in the reference when a layer's inputs are
// themselves functions of hyper-parameters - posterior means fed forward by `replace` / `impute` / inducing points under
// `fit(fix=False)` (gpar/regression.py:447-456, gpar/model.py:291-322) - and what moving inducing points needs.
//   mode GPAR_GRAD_RECT: W is n1 x n2;   GPAR_GRAD_SYM: z2 == z1 and W is symmetric, given by its lower triangle.
// One workgroup per (64-row block, column split): it walks the 64-column tiles of its split, every thread a 4 x 4 patch;
// row sums are reduced over the 16 lanes that share the rows and accumulated in LDS [64][dz]; splits are summed in order.
__global__ __launch_bounds__(256) void gram_input_grad_kernel(gpar_kspec_t ks, const double* __restrict__ z1, int n1, int ldz1,
                                                              const double* __restrict__ z2, int n2, int ldz2, int dz,
                                                              const double* __restrict__ W, int ldw, int mode, int nsplit,
                                                              double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int dzl = dz > 0 ? dz : 1;
    double* Za = gsm;
    double* Zb = Za + (size_t)dzl * GRAM_LD;
    double* acc = Zb + (size_t)dzl * GRAM_LD;   // [64][dzl]
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    const int bm = blockIdx.x, split = blockIdx.y;
    const int row0 = bm * GRAM_T;
    const int nt2 = (n2 + GRAM_T - 1) / GRAM_T;
    for (int i = t; i < GRAM_T * dzl; i += 256) acc[i] = 0.0;
    for (int idx = t; idx < GRAM_T * dz; idx += 256) {
        const int r = idx / dz, d = idx - r * dz;
        Za[d * GRAM_LD + r] = (row0 + r < n1) ? z1[(size_t)(row0 + r) * ldz1 + d] : 0.0;
    }
    for (int bn = split; bn < nt2; bn += nsplit) {
        const int col0 = bn * GRAM_T;
        __syncthreads();
        for (int idx = t; idx < GRAM_T * dz; idx += 256) {
            const int r = idx / dz, d = idx - r * dz;
            Zb[d * GRAM_LD + r] = (col0 + r < n2) ? z2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
        }
        __syncthreads();
        double w[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + 4 * ty + i, col = col0 + 4 * tx + j;
                double v = 0.0;
                if (row < n1 && col < n2) {
                    if (mode == GPAR_GRAD_SYM) v = (col <= row) ? W[(size_t)row * ldw + col] : W[(size_t)col * ldw + row];
                    else v = W[(size_t)row * ldw + col];
                }
                w[i][j] = v;
            }
        int f0 = 0;
        for (int term = 0; term < ks.nterms; ++term) {
            int nf = 0;
            while (f0 + nf < ks.nfactors && ks.factor[f0 + nf].term == term) ++nf;
            const double coef = ks.coef[term];
            double phi[GRAD_MAXF][4][4], sv[GRAD_MAXF][4][4];
#pragma unroll
            for (int ff = 0; ff < GRAD_MAXF; ++ff) {
                if (ff < nf) {
                    const gpar_factor_t fa = ks.factor[f0 + ff];
                    double s[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[i][j] = 0.0;
                    for (int d = fa.off; d < fa.off + fa.nd; ++d) {
                        double za[4], zb[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) za[i] = Za[d * GRAM_LD + 4 * ty + i];
#pragma unroll
                        for (int j = 0; j < 4; ++j) zb[j] = Zb[d * GRAM_LD + 4 * tx + j];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (fa.type == GPAR_K_LINEAR) s[i][j] = fma(za[i], zb[j], s[i][j]);
                                else { const double df = za[i] - zb[j]; s[i][j] = fma(df, df, s[i][j]); }
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { sv[ff][i][j] = s[i][j]; phi[ff][i][j] = gram_nonlin(fa.type, s[i][j], fa.alpha); }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { sv[ff][i][j] = 0.0; phi[ff][i][j] = 1.0; }
                }
            }
#pragma unroll
            for (int ff = 0; ff < GRAD_MAXF; ++ff) {
                if (ff >= nf) continue;
                const gpar_factor_t fa = ks.factor[f0 + ff];
                double g[4][4];   // W * rest * (d phi / d s * 2  for EQ / RQ,  1 for linear)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        double rest = coef;
#pragma unroll
                        for (int f2 = 0; f2 < GRAD_MAXF; ++f2)
                            if (f2 != ff) rest *= phi[f2][i][j];
                        if (fa.type == GPAR_K_EQ) g[i][j] = -w[i][j] * rest * phi[ff][i][j];
                        else if (fa.type == GPAR_K_RQ) g[i][j] = -w[i][j] * rest * phi[ff][i][j] / (1.0 + sv[ff][i][j] / (2.0 * fa.alpha));
                        else g[i][j] = w[i][j] * rest;
                    }
                for (int d = fa.off; d < fa.off + fa.nd; ++d) {
                    double za[4], zb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) za[i] = Za[d * GRAM_LD + 4 * ty + i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) zb[j] = Zb[d * GRAM_LD + 4 * tx + j];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double p = 0.0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) p = fma(g[i][j], fa.type == GPAR_K_LINEAR ? zb[j] : za[i] - zb[j], p);
                        // the 16 lanes tx = 0 .. 15 of one ty are consecutive lanes: reduce within the 16-lane row
#pragma unroll
                        for (int off = 8; off > 0; off >>= 1) p += __shfl_xor(p, off, 16);
                        if (tx == 0) acc[(4 * ty + i) * dzl + d] += p;
                    }
                }
            }
            f0 += nf;
        }
    }
    __syncthreads();
    for (int i = t; i < GRAM_T * dz; i += 256) {
        const int r = i / dz, d = i - r * dz;
        if (row0 + r < n1) partial[((size_t)split * n1 + row0 + r) * dz + d] = acc[r * dzl + d];
    }
}

__global__ __launch_bounds__(256) void gram_input_grad_reduce_kernel(const double* __restrict__ partial, int nsplit, int n1, int dz,
                                                                     double* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n1 * dz) return;
    const int r = (int)(i / dz), d = (int)(i - (long long)r * dz);
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += partial[((size_t)q * n1 + r) * dz + d];
    out[(size_t)r * ldo + d] = s;
}

// Sum of the per-workgroup partial sums, one wave per moment: lane l adds blocks l, l + 64, ... in order, the 64 lane sums are
// combined by a fixed butterfly (deterministic).  (One thread per moment walking all blocks took 155 us for 1024 blocks - a chain
// of dependent loads - which was more than the generated gradient kernel itself.)
__global__ __launch_bounds__(64) void gram_grad_reduce_kernel(const double* __restrict__ partial, int nblocks,
                                                              double* __restrict__ out) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= GRAD_NACC) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * GRAD_NACC + k];
    s = wave_sum(s);
    if (lane == 0) out[k] = s;
}

}  // namespace gpar
