// Source generator of the per-specification Gram kernel (see jit.h).  Same tiling, arguments and arithmetic as gram_kernel
// (gram.h) - 64 x 64 outputs per 256-thread workgroup, two passes of a 4 x 2 micro-tile per thread, feature panels staged
// transposed in LDS - with the term list unrolled into straight-line code: no loops over terms / factors, no type dispatch,
// dim counts as template arguments of the shared accumulation helpers.
#pragma once
#include <string>

#include "common.h"
#include "jit.h"

namespace gpar {

#define GPAR_DEVICE_CODE(...) #__VA_ARGS__
static const char* const GRAM_MATH_SRC =
#include "gram_math.inc"
    ;
#undef GPAR_DEVICE_CODE

// Mirror of the two ABI structs for the generated translation unit (hiprtc sees no project headers); the static_asserts tie
// it to include/gpar_hip.h.
static const char* const GRAM_JIT_PRELUDE = R"GJ(
struct gj_factor { int type; int term; int off; int nd; double alpha; };
struct gj_kspec { int nterms; int nfactors; double coef[8]; gj_factor factor[12]; };
#define GPAR_GRAM_LOWER 1
)GJ";
static_assert(GPAR_MAX_TERMS == 8 && GPAR_MAX_FACTORS == 12, "gram_jit.h mirrors gpar_kspec_t");
static_assert(sizeof(gpar_factor_t) == 24 && sizeof(gpar_kspec_t) == 8 + 8 * 8 + 12 * 24, "gram_jit.h mirrors gpar_kspec_t");

// Straight-line evaluation of all terms into total[8] for the micro-tile (ty, cb).  Mirrors the interpreter's order of
// operations exactly: per term expo = DOUBLED sum of factor exponents (EQ: expo += s - the first one a plain copy, 0 + s being s -; RQ: gram_rqh8), lin = coef * product of
// linear factors, one gram_exph8 per term that has a nonlinear factor, total = fma(lin, expo, total) or total += lin.
static std::string gram_jit_terms(const gpar_kspec_t& ks) {
    std::string o;
    int f = 0;
    for (int t = 0; t < ks.nterms; ++t) {
        const std::string ts = std::to_string(t);
        o += "        {   // term " + ts + "\n";
        o += "            double expo[8], lin[8];\n";
        o += "            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) { expo[e] = 0.0; lin[e] = ks.coef[" + ts + "]; }\n";
        bool any_exp = false, first_exp = true;
        while (f < ks.nfactors && ks.factor[f].term == t) {
            const gpar_factor_t& fa = ks.factor[f];
            const std::string off = std::to_string(fa.off), nd = std::to_string(fa.nd), fs = std::to_string(f);
            o += "            {\n                double s[8];\n                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) s[e] = 0.0;\n";
            if (fa.type == GPAR_K_LINEAR) {
                o += "                gram_accum_static<" + off + ", " + nd + ", true>(Za, Zb, ty, cb, s);\n";
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) lin[e] *= s[e];\n";
            } else {
                any_exp = true;
                o += "                gram_accum_static<" + off + ", " + nd + ", false>(Za, Zb, ty, cb, s);\n";
                if (fa.type == GPAR_K_EQ) o += std::string("                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) expo[e] ") + (first_exp ? "= s[e];\n" : "+= s[e];\n");
                else o += "                gram_rqh8(s, ks.factor[" + fs + "].alpha, expo, tab);\n";
                first_exp = false;
            }
            o += "            }\n";
            ++f;
        }
        if (any_exp) {
            o += "            gram_exph8(expo, tab);\n";
            o += "            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) total[e] = fma(lin[e], expo[e], total[e]);\n";
        } else {
            o += "            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) total[e] += lin[e];\n";
        }
        o += "        }\n";
    }
    return o;
}

// NARROW structures (at most GRAM_JIT_MAX_DZ feature dims) take the strip kernel below: every dim loop unrolled, the row features
// in registers across a strip of tiles.  A WIDE one (C5's last layer: 42 dims, rq + eq.eq + linear + rq) gains nothing from that
// form - fully unrolled it ran 0.303 ms against the interpreter's 0.295 ms at n = 8192 (profiles/r03_gram_configs*.jsonl): with a
// 4 x 2 micro-tile every dim costs two 16-byte LDS reads per 16 arithmetic instructions, and four SIMDs at that rate ask the
// compute unit's LDS for exactly its 128 bytes per clock - so it takes gram_jit_wide_source (round 5): a 4 x 4 micro-tile (three
// reads per 32 instructions), dims in ROLLED loops of four.  Same arithmetic per entry, same bits.
constexpr int GRAM_JIT_MAX_DZ = 16;
static int gram_jit_smax(int dz) { return dz <= 9 ? 8 : 4; }   // (1 + SMAX) panels of dz x 68 doubles: <= 44 KB

// Column tiles per workgroup: a workgroup walks a strip of up to `strip` consecutive 64 x 64 tiles of one tile row (a kernel
// argument; at most SMAX, which is part of the generated source and a function of dz).
static int gram_jit_strip(long long tiles, int dz) {
    if (dz > GRAM_JIT_MAX_DZ) return dz <= 48 ? 1 : 0;   // wide: one tile per workgroup (two panels of dz x 68 doubles: <= 52 KB)
    if (const char* e = getenv("GPAR_GRAM_JIT_STRIP")) { const int v = atoi(e); if (v >= 1 && v <= 64) return v < gram_jit_smax(dz) ? v : gram_jit_smax(dz); }
    // as long as ~2000 workgroups remain (three rounds of the chip's 768 slots).  Measured with the first version of the strip
    // kernel (ms; strip 1 / 2 / 4 / 8): C3 lower triangle n = 16384, 8 dims 0.449 / 0.415 / 0.398 / 0.394; C4 cross 65536 x 1024,
    // 14 dims 0.283 / 0.245 / 0.227 / 0.228; C2 n = 4096, 5 dims 0.031 / 0.033 / 0.035 / 0.042.
    int strip = gram_jit_smax(dz);
    while (strip > 1 && tiles / strip < 2000) strip /= 2;
    return strip;
}

// ---- wide structures: 4 x 4 micro-tile, rolled dim loops ------------------------------------------------------------------------
static std::string gram_jit_wide_terms(const gpar_kspec_t& ks) {
    // Per term: the distance / inner-product sums of ALL its factors for the thread's 16 entries first (the LDS-fed part, 4 x 4
    // micro-tile), then the elementary functions eight entries at a time - and the second eight only after the first: the two halves
    // are independent, and left to itself the compiler interleaves them (and the two calls of every helper) for instruction-level
    // parallelism, which needs 450 registers (spills: 0.55 ms against the interpreter's 0.28 at C5).  An empty asm statement that
    // redefines the second half's sums and reads the first half's results orders them; it emits nothing.
    // The values are those of the interpreter: lin = coef * product of linear factors (coef itself without any), expo = doubled
    // sum of the exponents (the first one a plain copy), total = fma(lin, exp(-expo / 2), total) or total + lin.
    std::string o;
    int f = 0;
    for (int t = 0; t < ks.nterms; ++t) {
        const std::string ts = std::to_string(t);
        const int f0 = f;
        int f1 = f0;
        bool has_lin = false, has_exp = false;
        while (f1 < ks.nfactors && ks.factor[f1].term == t) { (ks.factor[f1].type == GPAR_K_LINEAR ? has_lin : has_exp) = true; ++f1; }
        o += "        {   // term " + ts + "\n";
        for (int g = f0; g < f1; ++g) {
            const gpar_factor_t& fa = ks.factor[g];
            const std::string gs = std::to_string(g);
            o += "            double s" + gs + "[2][8];\n            _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) s" + gs + "[0][e] = s" + gs + "[1][e] = 0.0;\n";
            o += std::string("            gw_accum_dims<") + (fa.type == GPAR_K_LINEAR ? "true" : "false") + ">(Za, Zb, " + std::to_string(fa.off) + ", " +
                 std::to_string(fa.nd) + ", ty, tx, s" + gs + ");\n";
        }
        for (int h = 0; h < 2; ++h) {
            const std::string hs = std::to_string(h);
            if (h == 1 && f1 > f0) {
                for (int g = f0; g < f1; ++g) {
                    const std::string gs = std::to_string(g);
                    o += "            asm volatile(\"\" : \"+v\"(s" + gs + "[1][0]), \"+v\"(s" + gs + "[1][1]), \"+v\"(s" + gs + "[1][2]), \"+v\"(s" + gs + "[1][3]), \"+v\"(s" + gs +
                         "[1][4]), \"+v\"(s" + gs + "[1][5]), \"+v\"(s" + gs + "[1][6]), \"+v\"(s" + gs + "[1][7]) : \"v\"(total[0][0]), \"v\"(total[0][1]), \"v\"(total[0][2]), "
                         "\"v\"(total[0][3]), \"v\"(total[0][4]), \"v\"(total[0][5]), \"v\"(total[0][6]), \"v\"(total[0][7]));\n";
                }
            }
            o += "            {\n";
            if (has_exp) o += "                double expo[8];\n";
            if (has_lin) o += "                double lin[8];\n                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) lin[e] = ks.coef[" + ts + "];\n";
            bool first_exp = true;
            for (int g = f0; g < f1; ++g) {
                const gpar_factor_t& fa = ks.factor[g];
                const std::string gs = std::to_string(g);
                if (fa.type == GPAR_K_LINEAR) {
                    o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) lin[e] *= s" + gs + "[" + hs + "][e];\n";
                } else if (fa.type == GPAR_K_EQ) {
                    o += std::string("                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) expo[e] ") + (first_exp ? "= s" : "+= s") + gs + "[" + hs + "][e];\n";
                    first_exp = false;
                } else {
                    if (first_exp) o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) expo[e] = 0.0;\n";
                    o += "                gram_rqh8(s" + gs + "[" + hs + "], ks.factor[" + gs + "].alpha, expo, tab);\n";
                    first_exp = false;
                }
            }
            const std::string l = has_lin ? "lin[e]" : "ks.coef[" + ts + "]";
            if (has_exp) {
                o += "                gram_exph8(expo, tab);\n";
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) total[" + hs + "][e] = fma(" + l + ", expo[e], total[" + hs + "][e]);\n";
            } else {
                o += "                _Pragma(\"unroll\") for (int e = 0; e < 8; ++e) total[" + hs + "][e] += " + l + ";\n";
            }
            o += "            }\n";
        }
        o += "        }\n";
        f = f1;
    }
    return o;
}

static std::string gram_jit_wide_source(const gpar_kspec_t& ks, int dz) {
    std::string o = GRAM_JIT_PRELUDE;
    o += GRAM_MATH_SRC;
    o += "\nconstexpr int DZ = " + std::to_string(dz) + ";\n";
    o += R"GJ(
// One workgroup = one 64 x 64 tile; thread (tx, ty) = t & 15, t >> 4 holds the 4 x 4 entries of rows 4 ty .. + 3, columns 4 tx .. + 3
// as two arrays of eight (rows 4 ty, 4 ty + 1 | rows 4 ty + 2, 4 ty + 3: entry (i, j) is [i >> 1][4 (i & 1) + j]) - the shape the
// shared exp / log helpers take.  Per entry the same operations in the same order as every other Gram kernel of the library.
constexpr int PANEL = GRAM_T * DZ;
constexpr int PER_THREAD = (PANEL + 255) / 256;

template <int ND, bool LINEAR>
__device__ __forceinline__ void gw_accum(const double* __restrict__ Za, const double* __restrict__ Zb, int d0, int ty, int tx, double (&s)[2][8]) {
    g_d4 za[ND > 0 ? ND : 1], zb[ND > 0 ? ND : 1];
    _Pragma("unroll")
    for (int q = 0; q < ND; ++q) {
        za[q] = *reinterpret_cast<const g_d4*>(&Za[(d0 + q) * GRAM_LD + 4 * ty]);
        zb[q] = *reinterpret_cast<const g_d4*>(&Zb[(d0 + q) * GRAM_LD + 4 * tx]);
    }
    _Pragma("unroll")
    for (int q = 0; q < ND; ++q) {
        _Pragma("unroll")
        for (int i = 0; i < 4; ++i) {
            _Pragma("unroll")
            for (int j = 0; j < 4; ++j) {
                double& acc = s[i >> 1][4 * (i & 1) + j];
                if (LINEAR) {
                    acc = fma(za[q][i], zb[q][j], acc);
                } else {
                    const double d_ = za[q][i] - zb[q][j];
                    acc = fma(d_, d_, acc);
                }
            }
        }
    }
}

// dims [off, off + nd): a ROLLED loop over groups of four (a compile-time trip count, not unrolled: the body is 128 multiply-adds)
template <bool LINEAR>
__device__ __forceinline__ void gw_accum_dims(const double* __restrict__ Za, const double* __restrict__ Zb, int off, int nd, int ty, int tx,
                                              double (&s)[2][8]) {
    int d = off;
    _Pragma("unroll 1")
    for (; d + 4 <= off + nd; d += 4) {
        gw_accum<4, LINEAR>(Za, Zb, d, ty, tx, s);
        // (as in gram_accum_static: left alone, the LDS reads of every later group - and term - are hoisted to the front and the
        // multiply-adds sunk behind them, 450 registers; the empty statements pin the sums and fence the reads, they emit nothing)
        _Pragma("unroll")
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(s[0][e]), "+v"(s[1][e]) : : "memory");
    }
    switch (off + nd - d) {
        case 3: gw_accum<3, LINEAR>(Za, Zb, d, ty, tx, s); break;
        case 2: gw_accum<2, LINEAR>(Za, Zb, d, ty, tx, s); break;
        case 1: gw_accum<1, LINEAR>(Za, Zb, d, ty, tx, s); break;
        default: break;
    }
    _Pragma("unroll")
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(s[0][e]), "+v"(s[1][e]) : : "memory");
}

__device__ __forceinline__ void gw_panel(const double* __restrict__ z, int n, int ldz, int r0, int t, double* __restrict__ Z) {
    double reg[PER_THREAD];
    _Pragma("unroll")
    for (int q = 0; q < PER_THREAD; ++q) {
        const int idx = t + 256 * q;
        const int r = idx / DZ, d = idx - r * DZ;
        const bool ok = idx < PANEL && r0 + r < n;
        reg[q] = ok ? z[(size_t)(ok ? r0 + r : 0) * ldz + (ok ? d : 0)] : 0.0;
    }
    _Pragma("unroll")
    for (int q = 0; q < PER_THREAD; ++q) {
        const int idx = t + 256 * q;
        const int r = idx / DZ, d = idx - r * DZ;
        if (idx < PANEL) Z[d * GRAM_LD + r] = reg[q];
    }
}

extern "C" __global__ __launch_bounds__(256, 2) void gram_jit(gj_kspec ks, const double* __restrict__ z1, int n1, int ldz1,
                                                              const double* __restrict__ z2, int n2, int ldz2,
                                                              double* __restrict__ K, int ldk, int flags,
                                                              const double* __restrict__ diag_add, double diag_const,
                                                              const double* __restrict__ row_scale, int sym, long long batch_z,
                                                              long long batch_k, int strip) {
    __shared__ __attribute__((aligned(32))) double gsm[2 * DZ * GRAM_LD];
    __shared__ __attribute__((aligned(32))) double tab[GRAM_TAB_DOUBLES];
    (void)strip;
    int bm = blockIdx.y, bn = blockIdx.x;
    if (flags & GPAR_GRAM_LOWER) {   // 1-D grid over the tiles of the lower triangle, row by row
        const int L = blockIdx.x;
        bm = (int)((sqrt(8.0 * (double)L + 1.0) - 1.0) * 0.5);
        while ((bm + 1) * (bm + 2) / 2 <= L) ++bm;
        while (bm * (bm + 1) / 2 > L) --bm;
        bn = L - bm * (bm + 1) / 2;
    }
    const int row0 = bm * GRAM_T, col0 = bn * GRAM_T;
    if (row0 >= n1 || col0 >= n2) return;
    z1 += (size_t)blockIdx.z * batch_z;
    z2 += (size_t)blockIdx.z * batch_z;
    K += (size_t)blockIdx.z * batch_k;
    double* Za = gsm;
    double* Zb = gsm + DZ * GRAM_LD;
    const int t = threadIdx.x;
    gram_load_tables(tab, t);
    gw_panel(z1, n1, ldz1, row0, t, Za);
    gw_panel(z2, n2, ldz2, col0, t, Zb);
    __syncthreads();
    const int tx = t & 15, ty = t >> 4;
    double total[2][8];
    _Pragma("unroll") for (int e = 0; e < 8; ++e) total[0][e] = total[1][e] = 0.0;
)GJ";
    o += gram_jit_wide_terms(ks);
    o += R"GJ(
    const bool vec = ((ldk & 1) == 0) && ((((size_t)K) & 15u) == 0);
    const bool diag_tile = sym && col0 == row0;
    _Pragma("unroll")
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * ty + i;
        if (row >= n1) continue;
        const int col = col0 + 4 * tx;
        double v[4];
        _Pragma("unroll") for (int j = 0; j < 4; ++j) v[j] = total[i >> 1][4 * (i & 1) + j];
        if (row_scale) { const double rs = row_scale[row]; _Pragma("unroll") for (int j = 0; j < 4; ++j) v[j] *= rs; }
        if (diag_tile) {
            const double dadd = (diag_add ? diag_add[row] : 0.0) + diag_const;
            _Pragma("unroll") for (int j = 0; j < 4; ++j) if (col + j == row) v[j] += dadd;
        }
        double* out = K + (size_t)row * ldk + col;
        if (vec && col + 3 < n2) {
            *reinterpret_cast<g_d2*>(out) = g_d2{v[0], v[1]};
            *reinterpret_cast<g_d2*>(out + 2) = g_d2{v[2], v[3]};
        } else {
            _Pragma("unroll") for (int j = 0; j < 4; ++j) if (col + j < n2) out[j] = v[j];
        }
    }
}
)GJ";
    return o;
}

static std::string gram_jit_source(const gpar_kspec_t& ks, int dz, int strip) {
    if (strip <= 0) return std::string();   // (too wide even for the wide form: the interpreter)
    if (dz > GRAM_JIT_MAX_DZ) return gram_jit_wide_source(ks, dz);
    std::string o = GRAM_JIT_PRELUDE;
    o += GRAM_MATH_SRC;
    o += "\nconstexpr int DZ = " + std::to_string(dz > 0 ? dz : 1) + ";\nconstexpr int DZ_LOAD = " + std::to_string(dz) + ";\n";
    // (the strip length is a kernel argument: it only bounds a loop, and as a constant every problem size that picks another
    // length would compile the structure again - 0.4 s each, which showed in the first predict / fit of a run)
    (void)strip;
    o += "#define STRIP strip\n";
    o += "constexpr int SMAX = " + std::to_string(gram_jit_smax(dz)) + ";\n";
    o += R"GJ(
// One workgroup = one strip of up to STRIP <= SMAX consecutive 64 x 64 tiles of a tile row (blockIdx.y = tile row, blockIdx.x = strip;
// lower-triangular builds enumerate their strips in one dimension).  The tables, the row panel Za and ALL column panels of the
// strip are staged up front with every load in flight at once (<= 44 KB of LDS: three workgroups per compute unit, which the
// registers allow anyway); the tile loop then holds no load, no barrier and no wait - waves drift apart, so one wave's stores
// drain under another's arithmetic.  (The first version double-buffered one column panel per tile: a barrier per tile and a
// full vmcnt(0) drain of the tile's stores at each loop head; tools/exp_gram: 0.312 -> 0.301 ms at C3.)
constexpr int PANEL = GRAM_T * DZ_LOAD;             // doubles per feature panel
constexpr int PER_THREAD = (PANEL + 255) / 256;

__device__ __forceinline__ void gj_panel_load(const double* __restrict__ z, int n, int ldz, int r0, int t, double (&reg)[PER_THREAD > 0 ? PER_THREAD : 1]) {
    _Pragma("unroll")
    for (int q = 0; q < PER_THREAD; ++q) {
        const int idx = t + 256 * q;
        const int r = idx / DZ, d = idx - r * DZ;
        const bool ok = idx < PANEL && r0 + r < n;
        reg[q] = ok ? z[(size_t)(ok ? r0 + r : 0) * ldz + (ok ? d : 0)] : 0.0;
    }
}
__device__ __forceinline__ void gj_panel_store(double* __restrict__ Z, int t, const double (&reg)[PER_THREAD > 0 ? PER_THREAD : 1]) {
    _Pragma("unroll")
    for (int q = 0; q < PER_THREAD; ++q) {
        const int idx = t + 256 * q;
        const int r = idx / DZ, d = idx - r * DZ;
        if (idx < PANEL) Z[d * GRAM_LD + r] = reg[q];
    }
}

extern "C" __global__ __launch_bounds__(256, 2) void gram_jit(gj_kspec ks, const double* __restrict__ z1, int n1, int ldz1,
                                                              const double* __restrict__ z2, int n2, int ldz2,
                                                              double* __restrict__ K, int ldk, int flags,
                                                              const double* __restrict__ diag_add, double diag_const,
                                                              const double* __restrict__ row_scale, int sym, long long batch_z,
                                                              long long batch_k, int strip) {
    __shared__ __attribute__((aligned(32))) double gsm[(1 + SMAX) * DZ * GRAM_LD];
    __shared__ __attribute__((aligned(32))) double tab[GRAM_TAB_DOUBLES];
    const int nt2 = (n2 + GRAM_T - 1) / GRAM_T;
    int bm = blockIdx.y, bn0 = blockIdx.x * STRIP;
    if (flags & GPAR_GRAM_LOWER) {
        // 1-D grid over the strips of the lower triangle (empty workgroups above the diagonal are not free: ~25 ns of dispatch
        // each, as many again as useful ones).  Tile rows come in groups of STRIP rows with g + 1 strips each, g = 0, 1, ...:
        // STRIP g (g + 1) / 2 strips precede group g.
        const int L = blockIdx.x;
        int g = (int)((sqrt(8.0 * (double)L / (double)STRIP + 1.0) - 1.0) * 0.5);
        while (STRIP * (g + 1) * (g + 2) / 2 <= L) ++g;
        while (STRIP * g * (g + 1) / 2 > L) --g;
        const int rem = L - STRIP * g * (g + 1) / 2;
        bm = g * STRIP + rem / (g + 1);
        bn0 = (rem % (g + 1)) * STRIP;
    }
    int bn1 = bn0 + STRIP < nt2 ? bn0 + STRIP : nt2;                 // tiles [bn0, bn1)
    if ((flags & GPAR_GRAM_LOWER) && bn1 > bm + 1) bn1 = bm + 1;     // lower triangle: nothing right of the diagonal tile
    if (bn0 >= bn1 || bm * GRAM_T >= n1) return;
    z1 += (size_t)blockIdx.z * batch_z;
    z2 += (size_t)blockIdx.z * batch_z;
    K += (size_t)blockIdx.z * batch_k;
    double* Za = gsm;
    const int t = threadIdx.x;
    const int row0 = bm * GRAM_T;
    gram_load_tables(tab, t);
    {
        double preg[1 + SMAX][PER_THREAD > 0 ? PER_THREAD : 1];
        gj_panel_load(z1, n1, ldz1, row0, t, preg[0]);
        _Pragma("unroll") for (int q = 0; q < SMAX; ++q) if (q < bn1 - bn0) gj_panel_load(z2, n2, ldz2, (bn0 + q) * GRAM_T, t, preg[1 + q]);
        gj_panel_store(Za, t, preg[0]);
        _Pragma("unroll") for (int q = 0; q < SMAX; ++q) if (q < bn1 - bn0) gj_panel_store(gsm + (1 + q) * DZ * GRAM_LD, t, preg[1 + q]);
    }
    __syncthreads();
    const int tx = t & 15, ty = t >> 4;
    const bool vec = ((ldk & 1) == 0) && ((((size_t)K) & 15u) == 0);
    _Pragma("unroll 1")
    for (int bn = bn0; bn < bn1; ++bn) {
        const double* Zb = gsm + (1 + (bn - bn0)) * DZ * GRAM_LD;
        const int col0 = bn * GRAM_T;
    _Pragma("unroll 1")
    for (int h = 0; h < 2; ++h) {
        const int cb = 32 * h + 2 * tx;
        double total[8];
        _Pragma("unroll") for (int e = 0; e < 8; ++e) total[e] = 0.0;
)GJ";
    // (the row features - Za at 4 ty - do not depend on the pass h or on the tile: up to 16 dims stay in registers across the strip)
    o += gram_jit_terms(ks);
    o += R"GJ(
        // (the noise diagonal only exists in the diagonal tiles of a symmetric build: everywhere else its row load, two compares
        // and two selects per pair of entries are skipped by a workgroup-uniform branch)
        const bool diag_tile = sym && col0 == row0;
        _Pragma("unroll")
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * ty + i;
            if (row >= n1) continue;
            const int col = col0 + cb;
            double v0 = total[2 * i], v1 = total[2 * i + 1];
            if (row_scale) { const double rs = row_scale[row]; v0 *= rs; v1 *= rs; }
            if (diag_tile) {
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
}
)GJ";
    return o;
}

// Arguments of gram_jit in declaration order (hipModuleLaunchKernel takes an array of pointers to them).
struct GramJitArgs {
    gpar_kspec_t ks;
    const double* z1; int n1, ldz1;
    const double* z2; int n2, ldz2;
    double* K; int ldk, flags;
    const double* diag_add; double diag_const;
    const double* row_scale; int sym;
    long long batch_z, batch_k;
    int strip;
};

// Launches of at least this many entries take the generated kernel (GPAR_GRAM_JIT_MIN_ENTRIES; 0: always, negative: never).
// A structure costs 0.3-0.6 s of hiprtc time once per process; per launch the generated kernel saves ~30 % of a Gram build that
// is 0.03 ms at n = 4096 and 0.46 ms at n = 16384.  From 2^26 entries (n = 8192 symmetric, C4's 65536 x 1024 cross-Gram) a few
// hundred evaluations - one training run - repay it; below, the interpreter's microseconds are not worth half a second.
static long long gram_jit_min_entries() {
    const char* e = getenv("GPAR_GRAM_JIT_MIN_ENTRIES");
    return e ? atoll(e) : (1LL << 26);
}

// Launch the generated kernel if there is (or can be) one for this structure; false: the caller falls back to the interpreter.
static bool gram_jit_launch(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2, int dz,
                            double* K, int ldk, int flags, const double* diag_add, double diag_const, const double* row_scale, int sym,
                            dim3 grid, hipStream_t stream, long long batch_z, long long batch_k) {
    const long long min_entries = gram_jit_min_entries();
    if (min_entries < 0) return false;
    // (below the threshold only a kernel compiled at build time is used: it costs no compilation)
    if ((long long)n1 * n2 * grid.z < min_entries && ((long long)n1 * n2 * grid.z < aot_min_entries() || !aot_has(JIT_GRAM, *ks, dz, 1)))
        return false;
    // (the interpreter's grid enumerates tiles; this kernel takes tile rows x strips of column tiles)
    const int nt1 = gpar_ceil_div(n1, GRAM_T), nt2 = gpar_ceil_div(n2, GRAM_T);
    const long long tiles = ((flags & GPAR_GRAM_LOWER) ? (long long)nt1 * (nt1 + 1) / 2 : (long long)nt1 * nt2) * grid.z;
    const int strip = gram_jit_strip(tiles, dz);
    if (strip <= 0) return false;   // more than 48 dims: the interpreter
    hipFunction_t fn = jit_get(JIT_GRAM, *ks, dz, 1, "gram_jit", [&]() { return gram_jit_source(*ks, dz, strip); });
    if (!fn) return false;
    if (flags & GPAR_GRAM_LOWER) {
        // strips of the lower triangle: full groups of `strip` rows with g + 1 strips per row, then the rows of a last, partial group
        const int full = nt1 / strip, rest = nt1 - full * strip;
        grid = dim3((unsigned)((long long)strip * full * (full + 1) / 2 + (long long)rest * (full + 1)), 1, grid.z);
    } else {
        grid = dim3(gpar_ceil_div(nt2, strip), nt1, grid.z);
    }
    GramJitArgs a{*ks, z1, n1, ldz1, z2, n2, ldz2, K, ldk, flags, diag_add, diag_const, row_scale, sym, batch_z, batch_k, strip};
    void* params[] = {&a.ks, &a.z1, &a.n1, &a.ldz1, &a.z2, &a.n2, &a.ldz2, &a.K, &a.ldk, &a.flags, &a.diag_add, &a.diag_const,
                      &a.row_scale, &a.sym, &a.batch_z, &a.batch_k, &a.strip};
    return hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, 256, 1, 1, 0, stream, params, nullptr) == hipSuccess;
}

}  // namespace gpar
