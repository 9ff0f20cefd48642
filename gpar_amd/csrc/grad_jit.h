// Source generators of the per-specification GRADIENT kernels (see jit.h): the weighted sums of kernel derivatives behind the
// training gradient (gram_grad_kernel in gram.h) and behind input / inducing-point gradients (gram_input_grad_kernel).
//
// The ahead-of-time kernels keep, for every product term, the values and squared distances of up to four factors of a 4 x 4
// patch in registers while they loop over a run-time factor / dim list: 397 and 380 registers (accumulator-file spills, one wave
// per SIMD), and every per-dim sum is reduced over the wave - six cross-lane steps - once per tile.  With the structure known
// at compile time: a 4 x 2 micro-tile, arrays sized by the term's actual factor count, every moment sum in a register of its own
// ACROSS all tiles of the workgroup (static names: no dynamic indexing), one cross-lane reduction per sum per kernel.
// Same sums as the interpreter (summation order differs: agreement to rounding, not to the bit; tests compare both with the numpy
// oracle).  The exponential / logarithm are the table-based ones of gram_math.inc.
#pragma once
#include <string>

#include "common.h"
#include "gram_jit.h"
#include "jit.h"

namespace gpar {

// Code that evaluates term `t` on the micro-tile: for each of its factors f the squared distances / inner products s<f>[8] and
// the factor values phi<f>[8]; for RQ factors also tq<f>[8] = s / 2 alpha and lg<f>[8] = log1p(tq).
static std::string grad_jit_term_values(const gpar_kspec_t& ks, int t, int f0, int nf) {
    std::string o;
    for (int k = 0; k < nf; ++k) {
        const gpar_factor_t& fa = ks.factor[f0 + k];
        const std::string F = std::to_string(f0 + k), off = std::to_string(fa.off), nd = std::to_string(fa.nd);
        o += "            double s" + F + "[8], phi" + F + "[8];\n";
        o += "            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) s" + F + "[e] = 0.0;\n";
        o += "            gram_accum_static<" + off + ", " + nd + ", " + (fa.type == GPAR_K_LINEAR ? "true" : "false") + ">(Za, Zb, ty, cb, s" + F + ");\n";
        if (fa.type == GPAR_K_LINEAR) {
            o += "            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) phi" + F + "[e] = s" + F + "[e];\n";
        } else if (fa.type == GPAR_K_EQ) {
            o += "            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) phi" + F + "[e] = -0.5 * s" + F + "[e];\n";
            o += "            gram_exp8(phi" + F + ", tab);\n";
        } else {
            o += "            double tq" + F + "[8], lg" + F + "[8];\n";
            o += "            {\n                const double alpha = ks.factor[" + F + "].alpha, h2a = 0.5 / alpha;\n";
            o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) { tq" + F + "[e] = s" + F + "[e] * h2a; lg" + F + "[e] = gram_log1p_pos(tq" + F +
                 "[e], tab); phi" + F + "[e] = -alpha * lg" + F + "[e]; }\n";
            o += "                gram_exp8(phi" + F + ", tab);\n            }\n";
        }
    }
    (void)t;
    return o;
}

// rest[e] = coef_t * product of the OTHER factors' values
static std::string grad_jit_rest(const gpar_kspec_t& ks, int t, int f0, int nf, int k) {
    std::string o = "                double rest[8];\n                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) { rest[e] = ks.coef[" + std::to_string(t) + "];";
    for (int k2 = 0; k2 < nf; ++k2)
        if (k2 != k) o += " rest[e] *= phi" + std::to_string(f0 + k2) + "[e];";
    o += " }\n";
    (void)ks;
    return o;
}

// ---- parameter gradients ------------------------------------------------------------------------------------------------------
static std::string grad_jit_source(const gpar_kspec_t& ks, int dz, int mode, bool has_zd) {
    std::string o = GRAM_JIT_PRELUDE;
    o += GRAM_MATH_SRC;
    const int DZ = dz > 0 ? dz : 1;
    o += "\nconstexpr int DZ = " + std::to_string(DZ) + ";\nconstexpr int DZ_LOAD = " + std::to_string(dz) + ";\n";
    o += "constexpr int NT = " + std::to_string(ks.nterms > 0 ? ks.nterms : 1) + ";\nconstexpr int NF = " + std::to_string(ks.nfactors > 0 ? ks.nfactors : 1) + ";\n";
    o += "constexpr int MODE = " + std::to_string(mode) + ";   // 0: symmetric W (lower triangle given), 1: rectangular W\n";
    o += std::string("constexpr bool HAS_ZD = ") + (has_zd ? "true" : "false") + ";\n";
    o += "constexpr int OFF_C = 0, OFF_AL = " + std::to_string(GPAR_MAX_TERMS) + ", OFF_A = " + std::to_string(GPAR_MAX_TERMS + GPAR_MAX_FACTORS) +
         ", OFF_P = " + std::to_string(GPAR_MAX_TERMS + GPAR_MAX_FACTORS + GPAR_MAX_DIMS) + ", NACC = " + std::to_string(GPAR_GRAD_NACC) + ";\n";
    o += R"GJ(
__device__ __forceinline__ double gj_wave_sum(double v) {
    _Pragma("unroll")
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

extern "C" __global__ __launch_bounds__(256, 2) void gram_grad_jit(gj_kspec ks, const double* __restrict__ z, const double* __restrict__ zd,
                                                                   int n, int ldz, const double* __restrict__ z2,
                                                                   const double* __restrict__ zd2, int n2, int ldz2,
                                                                   const double* __restrict__ W, int ldw, double* __restrict__ partial) {
    __shared__ __attribute__((aligned(32))) double gsm[(HAS_ZD ? 4 : 2) * DZ * GRAM_LD];
    __shared__ __attribute__((aligned(32))) double tab[GRAM_TAB_DOUBLES];
    __shared__ double red[4][NACC];
    double* Za = gsm;
    double* Zb = Za + DZ * GRAM_LD;
    double* Zda = HAS_ZD ? Zb + DZ * GRAM_LD : Za;
    double* Zdb = HAS_ZD ? Zda + DZ * GRAM_LD : Zb;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int tx = t & 15, ty = t >> 4;
    gram_load_tables(tab, t);
    for (int i = t; i < 4 * NACC; i += 256) (&red[0][0])[i] = 0.0;
    const int nt = (n + GRAM_T - 1) / GRAM_T, nt2 = (n2 + GRAM_T - 1) / GRAM_T;
    const int ntiles = MODE == 0 ? nt * (nt + 1) / 2 : nt * nt2;
    double accC[NT], accAl[NF], accA[DZ], accP[DZ];
    _Pragma("unroll") for (int i = 0; i < NT; ++i) accC[i] = 0.0;
    _Pragma("unroll") for (int i = 0; i < NF; ++i) accAl[i] = 0.0;
    _Pragma("unroll") for (int i = 0; i < DZ; ++i) { accA[i] = 0.0; accP[i] = 0.0; }
    const bool vecw = ((ldw & 1) == 0) && ((((size_t)W) & 15u) == 0);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bm, bn;
        if (MODE == 0) {
            bm = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
            while ((bm + 1) * (bm + 2) / 2 <= tile) ++bm;
            while (bm * (bm + 1) / 2 > tile) --bm;
            bn = tile - bm * (bm + 1) / 2;
        } else {
            bm = tile / nt2;
            bn = tile - bm * nt2;
        }
        const int row0 = bm * GRAM_T, col0 = bn * GRAM_T;
        __syncthreads();
        for (int idx = t; idx < GRAM_T * DZ_LOAD; idx += 256) {
            const int r = idx / DZ, d = idx - r * DZ;
            const bool ra = row0 + r < n, rb = col0 + r < n2;
            Za[d * GRAM_LD + r] = ra ? z[(size_t)(row0 + r) * ldz + d] : 0.0;
            Zb[d * GRAM_LD + r] = rb ? z2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
            if (HAS_ZD) {
                Zda[d * GRAM_LD + r] = ra ? zd[(size_t)(row0 + r) * ldz + d] : 0.0;
                Zdb[d * GRAM_LD + r] = rb ? zd2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
            }
        }
        __syncthreads();
        _Pragma("unroll 1")
        for (int h = 0; h < 2; ++h) {
            const int cb = 32 * h + 2 * tx;
            // weights of the micro-tile: entry (row i, column j) at 2 i + j.  Symmetric W: a strictly-lower tile stands for its mirror
            // image too (factor 2); in a diagonal tile the entries above the diagonal are read from their mirror positions.
            double w[8];
            _Pragma("unroll")
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * ty + i, col = col0 + cb;
                double v0 = 0.0, v1 = 0.0;
                if (MODE == 0 && bm == bn) {
                    if (row < n && col < n2) v0 = (col <= row) ? W[(size_t)row * ldw + col] : W[(size_t)col * ldw + row];
                    if (row < n && col + 1 < n2) v1 = (col + 1 <= row) ? W[(size_t)row * ldw + col + 1] : W[(size_t)(col + 1) * ldw + row];
                } else if (row < n) {
                    if (vecw && col + 1 < n2) {
                        const g_d2 v = *reinterpret_cast<const g_d2*>(W + (size_t)row * ldw + col);
                        v0 = v[0]; v1 = v[1];
                    } else {
                        if (col < n2) v0 = W[(size_t)row * ldw + col];
                        if (col + 1 < n2) v1 = W[(size_t)row * ldw + col + 1];
                    }
                    if (MODE == 0) { v0 *= 2.0; v1 *= 2.0; }
                }
                w[2 * i] = v0;
                w[2 * i + 1] = v1;
            }
)GJ";
    int f0 = 0;
    for (int t = 0; t < ks.nterms; ++t) {
        int nf = 0;
        while (f0 + nf < ks.nfactors && ks.factor[f0 + nf].term == t) ++nf;
        const std::string T = std::to_string(t);
        o += "            {   // term " + T + "\n";
        o += grad_jit_term_values(ks, t, f0, nf);
        // coefficient moment: sum w * product of factor values
        o += "            {\n                double c = accC[" + T + "];\n                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) { double pr = w[e];";
        for (int k = 0; k < nf; ++k) o += " pr *= phi" + std::to_string(f0 + k) + "[e];";
        o += " c += pr; }\n                accC[" + T + "] = c;\n            }\n";
        for (int k = 0; k < nf; ++k) {
            const gpar_factor_t& fa = ks.factor[f0 + k];
            const std::string F = std::to_string(f0 + k), off = std::to_string(fa.off), nd = std::to_string(fa.nd);
            o += "            {   // factor " + F + "\n";
            o += grad_jit_rest(ks, t, f0, nf, k);
            o += "                double g[8];\n";
            if (fa.type == GPAR_K_EQ) {
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) g[e] = w[e] * rest[e] * (-0.5 * phi" + F + "[e]);\n";
            } else if (fa.type == GPAR_K_RQ) {
                o += "                {\n                    double al = accAl[" + F + "];\n";
                o += "                    _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) {\n";
                o += "                        const double base = 1.0 + tq" + F + "[e], wr = w[e] * rest[e] * phi" + F + "[e], ib = 1.0 / base;\n";
                o += "                        g[e] = wr * (-0.5 * ib);\n";
                o += "                        al = fma(wr, tq" + F + "[e] * ib - lg" + F + "[e], al);\n";
                o += "                    }\n                    accAl[" + F + "] = al;\n                }\n";
            } else {
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) g[e] = w[e] * rest[e];\n";
            }
            o += "                grad_dims_static<" + off + ", " + nd + ", " + (fa.type == GPAR_K_LINEAR ? "true" : "false") +
                 ", HAS_ZD, DZ>(Za, Zb, Zda, Zdb, ty, cb, g, accA, accP);\n";
            o += "            }\n";
        }
        o += "            }\n";
        f0 += nf;
    }
    o += R"GJ(
        }
    }
    // one cross-lane reduction per sum; the four waves' values are added in a fixed order
    __syncthreads();
)GJ";
    for (int t = 0; t < ks.nterms; ++t)
        o += "    { const double v = gj_wave_sum(accC[" + std::to_string(t) + "]); if (lane == 0) red[wv][OFF_C + " + std::to_string(t) + "] = v; }\n";
    for (int f = 0; f < ks.nfactors; ++f) {
        const gpar_factor_t& fa = ks.factor[f];
        if (fa.type == GPAR_K_RQ) o += "    { const double v = gj_wave_sum(accAl[" + std::to_string(f) + "]); if (lane == 0) red[wv][OFF_AL + " + std::to_string(f) + "] = v; }\n";
        for (int d = fa.off; d < fa.off + fa.nd; ++d) {
            const std::string D = std::to_string(d);
            o += "    { const double v = gj_wave_sum(accA[" + D + "]); if (lane == 0) red[wv][OFF_A + " + D + "] = v; }\n";
            if (has_zd && fa.type != GPAR_K_LINEAR)
                o += "    { const double v = gj_wave_sum(accP[" + D + "]); if (lane == 0) red[wv][OFF_P + " + D + "] = v; }\n";
        }
    }
    o += R"GJ(
    __syncthreads();
    for (int k = t; k < NACC; k += 256)
        partial[(size_t)blockIdx.x * NACC + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
}
)GJ";
    return o;
}

struct GradJitArgs {
    gpar_kspec_t ks;
    const double* z; const double* zd; int n, ldz;
    const double* z2; const double* zd2; int n2, ldz2;
    const double* W; int ldw;
    double* partial;
};

// Launches of at least this many weight entries take the generated kernel (GPAR_GRAD_JIT_MIN_ENTRIES; 0: always, negative: never).
// The generated passes are 4-5x faster than the interpreter (no accumulator spills), ~1.1 ms per evaluation and layer at 2^24
// entries (n = 4096): the reference's default training run (1000 iterations) repays 0.5 s of compilation several times over from
// there on; at n = 1024 it would cost a 0.06 s fit 0.5 s.
static long long grad_jit_min_entries() {
    const char* e = getenv("GPAR_GRAD_JIT_MIN_ENTRIES");
    return e ? atoll(e) : (1LL << 24);
}

static bool grad_jit_launch(const gpar_kspec_t* ks, const double* z1, const double* zd1, int n1, int ldz1, const double* z2, const double* zd2,
                            int n2, int ldz2, int dz, const double* W, int ldw, int mode, double* workspace, int nblocks, hipStream_t stream) {
    const long long min_entries = grad_jit_min_entries();
    if (mode == GPAR_GRAD_DIAG || min_entries < 0) return false;
    const bool has_zd = zd1 != nullptr;
    const size_t lds = ((size_t)(has_zd ? 4 : 2) * (dz > 0 ? dz : 1) * GRAM_LD + GRAM_TAB_DOUBLES + 4 * GPAR_GRAD_NACC) * sizeof(double);
    if (lds > 64 * 1024) return false;   // static LDS limit of a generated kernel: very wide kernels stay on the interpreter
    const int extra = 100 + mode * 2 + (has_zd ? 1 : 0);
    if ((long long)n1 * n2 < min_entries && ((long long)n1 * n2 < aot_min_entries() || !aot_has(JIT_GRAD, *ks, dz, extra))) return false;   // small: only a build-time compiled kernel
    hipFunction_t fn = jit_get(JIT_GRAD, *ks, dz, extra, "gram_grad_jit", [&]() { return grad_jit_source(*ks, dz, mode, has_zd); });
    if (!fn) return false;
    GradJitArgs a{*ks, z1, zd1, n1, ldz1, z2, zd2, n2, ldz2, W, ldw, workspace};
    void* params[] = {&a.ks, &a.z, &a.zd, &a.n, &a.ldz, &a.z2, &a.zd2, &a.n2, &a.ldz2, &a.W, &a.ldw, &a.partial};
    return hipModuleLaunchKernel(fn, nblocks, 1, 1, 256, 1, 1, 0, stream, params, nullptr) == hipSuccess;
}

// ---- input gradients ----------------------------------------------------------------------------------------------------------
// out[a][q] = sum_b W(a, b) d k(z1_a, z2_b) / d z1_a[q].  One workgroup per (64-row block, column split) as in the interpreter; a
// thread's four rows are fixed for the whole workgroup, so its 4 x dz partial sums stay in registers across all column tiles and
// are reduced over the 16 lanes that share the rows once, at the end (the interpreter: four cross-lane steps per row, dim and tile).
static std::string input_grad_jit_source(const gpar_kspec_t& ks, int dz, int mode) {
    std::string o = GRAM_JIT_PRELUDE;
    o += GRAM_MATH_SRC;
    const int DZ = dz > 0 ? dz : 1;
    o += "\nconstexpr int DZ = " + std::to_string(DZ) + ";\nconstexpr int DZ_LOAD = " + std::to_string(dz) + ";\n";
    o += "constexpr int MODE = " + std::to_string(mode) + ";\n";
    o += R"GJ(
extern "C" __global__ __launch_bounds__(256, 2) void gram_input_grad_jit(gj_kspec ks, const double* __restrict__ z1, int n1, int ldz1,
                                                                         const double* __restrict__ z2, int n2, int ldz2,
                                                                         const double* __restrict__ W, int ldw, int nsplit,
                                                                         double* __restrict__ partial) {
    __shared__ __attribute__((aligned(32))) double gsm[2 * DZ * GRAM_LD];
    __shared__ __attribute__((aligned(32))) double tab[GRAM_TAB_DOUBLES];
    double* Za = gsm;
    double* Zb = Za + DZ * GRAM_LD;
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    const int bm = blockIdx.x, split = blockIdx.y;
    const int row0 = bm * GRAM_T;
    const int nt2 = (n2 + GRAM_T - 1) / GRAM_T;
    gram_load_tables(tab, t);
    for (int idx = t; idx < GRAM_T * DZ_LOAD; idx += 256) {
        const int r = idx / DZ, d = idx - r * DZ;
        Za[d * GRAM_LD + r] = (row0 + r < n1) ? z1[(size_t)(row0 + r) * ldz1 + d] : 0.0;
    }
    double accX[4][DZ];
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { _Pragma("unroll") for (int d = 0; d < DZ; ++d) accX[i][d] = 0.0; }
    const bool vecw = ((ldw & 1) == 0) && ((((size_t)W) & 15u) == 0);
    for (int bn = split; bn < nt2; bn += nsplit) {
        const int col0 = bn * GRAM_T;
        __syncthreads();
        for (int idx = t; idx < GRAM_T * DZ_LOAD; idx += 256) {
            const int r = idx / DZ, d = idx - r * DZ;
            Zb[d * GRAM_LD + r] = (col0 + r < n2) ? z2[(size_t)(col0 + r) * ldz2 + d] : 0.0;
        }
        __syncthreads();
        _Pragma("unroll 1")
        for (int h = 0; h < 2; ++h) {
            const int cb = 32 * h + 2 * tx;
            double w[8];
            _Pragma("unroll")
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * ty + i, col = col0 + cb;
                double v0 = 0.0, v1 = 0.0;
                if (MODE == 0 && col + 1 > row) {   // (symmetric W, an entry on or above the diagonal in this pair: mirror positions)
                    if (row < n1 && col < n2) v0 = (col <= row) ? W[(size_t)row * ldw + col] : W[(size_t)col * ldw + row];
                    if (row < n1 && col + 1 < n2) v1 = W[(size_t)(col + 1) * ldw + row];
                } else if (row < n1) {
                    if (vecw && col + 1 < n2) {
                        const g_d2 v = *reinterpret_cast<const g_d2*>(W + (size_t)row * ldw + col);
                        v0 = v[0]; v1 = v[1];
                    } else {
                        if (col < n2) v0 = W[(size_t)row * ldw + col];
                        if (col + 1 < n2) v1 = W[(size_t)row * ldw + col + 1];
                    }
                }
                w[2 * i] = v0;
                w[2 * i + 1] = v1;
            }
)GJ";
    int f0 = 0;
    for (int t = 0; t < ks.nterms; ++t) {
        int nf = 0;
        while (f0 + nf < ks.nfactors && ks.factor[f0 + nf].term == t) ++nf;
        o += "            {   // term " + std::to_string(t) + "\n";
        o += grad_jit_term_values(ks, t, f0, nf);
        for (int k = 0; k < nf; ++k) {
            const gpar_factor_t& fa = ks.factor[f0 + k];
            const std::string F = std::to_string(f0 + k), off = std::to_string(fa.off), nd = std::to_string(fa.nd);
            o += "            {   // factor " + F + "\n";
            o += grad_jit_rest(ks, t, f0, nf, k);
            o += "                double g[8];   // W * rest * (2 d phi / d s for EQ / RQ, 1 for linear)\n";
            if (fa.type == GPAR_K_EQ)
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) g[e] = -w[e] * rest[e] * phi" + F + "[e];\n";
            else if (fa.type == GPAR_K_RQ)
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) g[e] = -w[e] * rest[e] * phi" + F + "[e] / (1.0 + tq" + F + "[e]);\n";
            else
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) g[e] = w[e] * rest[e];\n";
            o += "                grad_rows_static<" + off + ", " + nd + ", " + (fa.type == GPAR_K_LINEAR ? "true" : "false") + ", DZ>(Za, Zb, ty, cb, g, accX);\n";
            o += "            }\n";
        }
        o += "            }\n";
        f0 += nf;
    }
    o += R"GJ(
        }
    }
    // the 16 lanes tx = 0 .. 15 of one ty are consecutive lanes and share their four rows: one reduction per (row, dim)
    _Pragma("unroll")
    for (int i = 0; i < 4; ++i) {
        _Pragma("unroll")
        for (int d = 0; d < DZ_LOAD; ++d) {
            double p = accX[i][d];
            _Pragma("unroll")
            for (int off = 8; off > 0; off >>= 1) p += __shfl_xor(p, off, 16);
            const int row = row0 + 4 * ty + i;
            if (tx == 0 && row < n1) partial[((size_t)split * n1 + row) * DZ_LOAD + d] = p;
        }
    }
}
)GJ";
    return o;
}

struct InputGradJitArgs {
    gpar_kspec_t ks;
    const double* z1; int n1, ldz1;
    const double* z2; int n2, ldz2;
    const double* W; int ldw, nsplit;
    double* partial;
};

static bool input_grad_jit_launch(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2, int dz,
                                  const double* W, int ldw, int mode, int nsplit, double* workspace, hipStream_t stream) {
    const long long min_entries = grad_jit_min_entries();
    // (4 x dz partial sums per thread in registers: up to 20 dims)
    if (min_entries < 0 || dz < 1 || dz > 20) return false;
    if ((long long)n1 * n2 < min_entries && ((long long)n1 * n2 < aot_min_entries() || !aot_has(JIT_INPUT_GRAD, *ks, dz, 200 + mode))) return false;
    hipFunction_t fn = jit_get(JIT_INPUT_GRAD, *ks, dz, 200 + mode, "gram_input_grad_jit", [&]() { return input_grad_jit_source(*ks, dz, mode); });
    if (!fn) return false;
    InputGradJitArgs a{*ks, z1, n1, ldz1, z2, n2, ldz2, W, ldw, nsplit, workspace};
    void* params[] = {&a.ks, &a.z1, &a.n1, &a.ldz1, &a.z2, &a.n2, &a.ldz2, &a.W, &a.ldw, &a.nsplit, &a.partial};
    return hipModuleLaunchKernel(fn, gpar_ceil_div(n1, GRAM_T), nsplit, 1, 256, 1, 1, 0, stream, params, nullptr) == hipSuccess;
}

}  // namespace gpar
