// Fused panel factorisation: ONE persistent launch factors a whole W-column panel (W = 64 * S, S <= 16) for all
// rows below it, instead of S x {diag kernel, strip kernel, rank-64 GEMM} = 3 S dependent launches.
//
// Why: the blocked Cholesky's critical path is the chain  diag(s) -> strip(s) -> update(s) -> diag(s+1) ...; as
// separate kernels each link costs a launch boundary plus two or three cold memory round trips (26 + 21 + ~40 us
// per 64 columns, profiles/r01_bench_kernel_stats.txt), 21 of the 48 ms of an n = 16384 factorisation.  Here the
// links are in-launch hand-offs between workgroups.
//
// Decomposition: 64-row blocks; workgroup g owns row blocks g, g + G, ... (G = grid size <= resident capacity, so
// every workgroup is resident and spinning is safe; every spin is bounded and reports through `info`).
// For step s (column block s):
//   * the owner of row block s factors the 64 x 64 diagonal block (right-looking in 8-column blocks, pivots on wave 0,
//     rank-8 updates pipelined over the other waves: pnl_diag) and publishes it (flag1[s]);
//   * every workgroup, for each of its row blocks rb > s: waits for flag1[s], solves its 64 rows against L_ss
//     (the same 8-column pipeline: pnl_strip), writes X = L[rb][s], publishes it if rb < S (flag2[s][rb]: those rows are
//     the B operands of everybody's updates), then applies  A[rb][c] -= X L[c][s]^T  for c = s+1 .. min(rb, S-1)
//     on the matrix cores (v_mfma_f64_16x16x4; X and L[c][s] in LDS, C read-modify-written in global memory).
// Hand-offs follow the agent-scope recipe (cdna_hip_programming.md Guideline 16 / R1): write-through (sc1) payload
// stores -> every wave drains vmcnt -> barrier -> one lane: relaxed agent-scope flag store;
// consumer: one lane polls relaxed, one acquire fence, barrier, plain loads.  The flag words live in the strict
// upper triangle of the panel's diagonal block (scratch by the ABI's convention) and are zeroed by a memset node
// on the stream before every launch.
#pragma once
#include "common.h"
#include "potrf.h"

namespace gpar {

constexpr int PNL_LD = 66;
constexpr int PNL_TILE = 64 * PNL_LD;                  // doubles
constexpr int PNL_LDT = 34;                            // row pitch of the update's wave-private transposition buffers
constexpr int PNL_LDS_BYTES = 2 * PNL_TILE * 8 + 512 + 4 * 4 * PNL_LDT * 8;  // Cs/T (aliased by the update operand), Xs, reciprocal
                                                       // pivots, 4 transposition buffers: 70.8 KB - it must fit the
                                                       // 73.7 KB hole a retiring SYRK workgroup leaves (see pnl_update)
constexpr int PNL_FLAG_SLOTS = 56;                     // usable scratch words per row of the diagonal block
constexpr int PNL_MAX_S = 16;                          // S + S^2 <= 8 rows x 56 slots
constexpr unsigned PNL_SPIN_LIMIT = 1u << 22;

// Kernels whose workgroups WAIT for one another (the three panel kernels) and the compute-unit slots they hold.  The chip hands
// workgroup i of a launch to XCD i % 8 and each XCD starts its share in order: inside ONE launch the lowest unfinished workgroup
// always runs (every wait targets a lower index), but two launches on two streams can fill each other's XCDs with waiters whose
// producers then find no slot - a cycle that only the bounded spin breaks (-77; three streams at n = 8192 hit it once in two runs
// when a launch carried thousands of tile workgroups, round 5).  A launch A can be starved only if the OTHER waiting launches in
// flight fill an XCD by themselves (64 slots: 32 compute units x 2 workgroups of this LDS size; update kernels never wait and always
// retire).  So:
//   * BIG launches (more than SPIN_SMALL_WGS workgroups) are never in flight together: one on a stream other than the previous big
//     one's waits for it (one event, re-recorded after every big launch);
//   * of SMALL launches at most two are in flight on different streams (2 x 192 / 8 = 48 < 64 slots per XCD beside one big launch):
//     a ring of two events, a launch waits for the older entry unless its own stream made it.  An inducing-point layer's two small
//     factorisations (K_zz on a side stream beside the bound's matrix on the caller's) alternate slots and never wait.
// Every waiting launch records its end on ITS OWN stream (one hipEventRecord per panel launch, ~1 us of host time): the chain never
// touches a stream handle it merely remembers - a caller may have destroyed it - and needs no "second stream seen" state that would
// have to decay (round 5 recorded nothing while one stream was in use and marked the first stream's tail when a second showed up).
// A launch on the stream that made the entry it would wait for does not wait (stream order already holds); waiting for an event
// whose launch has long finished costs the device nothing.  GPAR_SPIN_CHAIN=0 switches all of it off (the round-4 behaviour).
// The state is PER PROCESS (and per device): two processes sharing one GPU are ordered by nothing - there the bounded spin, the -77
// status and the caller's unfused retry are what remains.
constexpr int SPIN_SMALL_WGS = 192;
struct SpinChain {
    hipEvent_t ev = nullptr;          // the last big launch
    hipEvent_t sev[2] = {nullptr, nullptr};   // the last two small launches
    hipStream_t last = nullptr;       // stream of the last big launch (compared, never used)
    hipStream_t sst[2] = {nullptr, nullptr};
    bool big = false, small[2] = {false, false};   // is there such a launch
    int snext = 0;
    bool created = false;
};
static SpinChain g_spin_chain[16];
static SpinChain& spin_chain() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return g_spin_chain[dev % 16];
}
static bool spin_chain_init() {
    SpinChain& c = spin_chain();
    if (c.created) return true;
    if (hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c.sev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c.sev[1], hipEventDisableTiming) != hipSuccess)
        return false;
    c.created = true;
    return true;
}
static inline bool spin_chain_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { GPAR_HIP_IGNORE(hipGetLastError()); return false; }
    return st != hipStreamCaptureStatusNone;
}
// Callers hold the library mutex.  `enter` before the launch of `wgs` workgroups, `leave` after it.
static int spin_chain_enter(hipStream_t s, long long wgs) {
    SpinChain& c = spin_chain();
    if (!(c.big || c.small[0] || c.small[1]) || !env_int("GPAR_SPIN_CHAIN", 1)) return 0;
    if (spin_chain_capturing(s)) return 0;   // (a graph being captured is ordered by whoever replays it)
    if (wgs > SPIN_SMALL_WGS) {
        if (c.big && c.last != s) GPAR_HIP_TRY(hipStreamWaitEvent(s, c.ev, 0));
    } else {
        const int slot = c.snext;   // the older of the two
        if (c.small[slot] && c.sst[slot] != s) GPAR_HIP_TRY(hipStreamWaitEvent(s, c.sev[slot], 0));
    }
    return 0;
}
static void spin_chain_leave(hipStream_t s, long long wgs) {
    SpinChain& c = spin_chain();
    if (!env_int("GPAR_SPIN_CHAIN", 1) || spin_chain_capturing(s) || !spin_chain_init()) return;
    if (wgs > SPIN_SMALL_WGS) {
        if (hipEventRecord(c.ev, s) != hipSuccess) { GPAR_HIP_IGNORE(hipGetLastError()); return; }
        c.last = s;
        c.big = true;
    } else {
        if (hipEventRecord(c.sev[c.snext], s) != hipSuccess) { GPAR_HIP_IGNORE(hipGetLastError()); return; }
        c.sst[c.snext] = s;
        c.small[c.snext] = true;
        c.snext ^= 1;
    }
}

struct PanelArgs {
    double* A;
    int N, lda, k0, S;   // panel columns [k0, k0 + 64 S)
    double* logdet;
    int* info;
    long long* stamps;   // dev aid (tools/time_panel.hip): cycle stamps of the critical chain, normally null
    long long batch_a = 0;   // batched launch (panel2.h, gridDim.y matrices of the same shape): matrix y starts at A + y * batch_a,
                             // its logdet / info words are logdet[y], info[y]
    int pairs = 0;           // panel2.h: bulk row blocks take their column blocks in pairs (p2_row_block_pairs)
    int progressive = 0;     // panel2.h: a diagonal tile is handed to the next team row in four block columns while it is being factored (P3Publish)
    int split = 0;           // panel2.h: the team is split by tile (p2_team_tile + chain workgroups); needs progressive, S <= 8
    int tile_rows = 0;       // panel2.h, fused launches: the rows that form the NEXT team are taken tile by tile (p2_bulk_tile); needs split
};

__device__ __forceinline__ unsigned long long* pnl_flag(const PanelArgs& p, int f) {
    // flag f lives at A[k0 + f / 56][k0 + 8 + f % 56]: strictly above the diagonal
    return reinterpret_cast<unsigned long long*>(p.A + (size_t)(p.k0 + f / PNL_FLAG_SLOTS) * p.lda + p.k0 + 8 + f % PNL_FLAG_SLOTS);
}

// All threads of the workgroup have issued WRITE-THROUGH (sc1) stores of the payload (pnl_store_tile with
// `publish`): every wave drains them, then one lane raises the flag.  No release fence is needed (recipe R1): a
// 32 KB tile published with plain stores + buffer_wbl2 costs ~6 us per hand-off, two of which sit on the
// critical path of every 64-column step.
__device__ __forceinline__ void pnl_publish(const PanelArgs& p, int f) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pnl_flag(p, f), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns after flag f is set and this CU's stale lines are dropped.  Bounded: on timeout the error is recorded and
// the kernel carries on (results are garbage, info says so) - it never hangs.
__device__ __forceinline__ void pnl_wait(const PanelArgs& p, int f) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(pnl_flag(p, f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > PNL_SPIN_LIMIT) {
                if (p.info) atomicCAS(p.info, 0, -77);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// 64 x 64 tile: global (rows r0.., cols c0.., clamped to valid rows) -> LDS [64][PNL_LD]; 256 threads, 16-byte loads.
__device__ __forceinline__ void pnl_load_tile(const PanelArgs& p, int r0, int c0, double* __restrict__ dst, int t) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q;          // 2048 chunks of 2 doubles
        const int r = c >> 5, cc = (c & 31) * 2;
        const int rr = min(r0 + r, p.N - 1);
        v[q] = *reinterpret_cast<const d2*>(p.A + (size_t)rr * p.lda + c0 + cc);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q;
        const int r = c >> 5, cc = (c & 31) * 2;
        *reinterpret_cast<d2*>(dst + r * PNL_LD + cc) = v[q];
    }
}

// LDS tile -> global; `lower_only`: store only entries with col <= row (diagonal block); rows beyond N are skipped.
// `publish`: the tile will be handed to other workgroups -> relaxed agent-scope atomic stores (global_store ... sc1,
// write-through to memory) instead of plain stores.
__device__ __forceinline__ void pnl_store_tile(const PanelArgs& p, int r0, int c0, const double* __restrict__ src, int t,
                                               bool lower_only, bool publish) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = t + 256 * q;          // 4096 elements
        const int r = e >> 6, c = e & 63;
        if (r0 + r < p.N && (!lower_only || c <= r)) {
            double* dst = p.A + (size_t)(r0 + r) * p.lda + c0 + c;
            if (publish)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(src[r * PNL_LD + c]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *dst = src[r * PNL_LD + c];
        }
    }
}

// Rank-8 right-looking update shared by the diagonal factorisation and the strip solve:
//   D[i][k] -= sum_{j<8} D[i][8 jb + j] * Cf[k][8 jb + j]      for k = kbeg, kbeg + kstep, ... < 64,
// lane i = row.  The row's own 8 values are one lane-private read; the 8 coefficients of column k are a
// wave-uniform broadcast read.
__device__ __forceinline__ void pnl_rank8(double* __restrict__ D, const double* __restrict__ Cf, int jb, int i, int kbeg, int kstep) {
    double mine[8];
    const pan_d2* ms = reinterpret_cast<const pan_d2*>(&D[i * PNL_LD + 8 * jb]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const pan_d2 v = ms[q]; mine[2 * q] = v[0]; mine[2 * q + 1] = v[1]; }
#pragma unroll 2
    for (int k = kbeg; k < 64; k += kstep) {
        const pan_d2* cs = reinterpret_cast<const pan_d2*>(&Cf[k * PNL_LD + 8 * jb]);
        const pan_d2 c0 = cs[0], c1 = cs[1], c2 = cs[2], c3 = cs[3];
        double v0 = D[i * PNL_LD + k], v1 = 0.0;
        v0 = fma(-mine[0], c0[0], v0); v1 = fma(-mine[1], c0[1], v1);
        v0 = fma(-mine[2], c1[0], v0); v1 = fma(-mine[3], c1[1], v1);
        v0 = fma(-mine[4], c2[0], v0); v1 = fma(-mine[5], c2[1], v1);
        v0 = fma(-mine[6], c3[0], v0); v1 = fma(-mine[7], c3[1], v1);
        D[i * PNL_LD + k] = v0 + v1;
    }
}

// The 8 values D[i][8 jb .. 8 jb + 7] of lane i's row with the rank-8 update of block jb - 1 applied, in registers
// (wave 0's share of the update: exactly the columns it is about to factor / solve).
__device__ __forceinline__ void pnl_rank8_next(const double* __restrict__ D, const double* __restrict__ Cf, int jb, int i, double (&acc)[8]) {
    const pan_d2* src = reinterpret_cast<const pan_d2*>(&D[i * PNL_LD + 8 * jb]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const pan_d2 v = src[q]; acc[2 * q] = v[0]; acc[2 * q + 1] = v[1]; }
    if (jb == 0) return;
    double mine[8];
    const pan_d2* ms = reinterpret_cast<const pan_d2*>(&D[i * PNL_LD + 8 * (jb - 1)]);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const pan_d2 v = ms[q]; mine[2 * q] = v[0]; mine[2 * q + 1] = v[1]; }
    // four columns at a time: their 16 broadcast reads first, then the FMAs (independent chains) - all eight at once cost
    // 64 VGPRs and pushed the kernel over the register budget it has to keep (see potrf_panel_kernel)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        pan_d2 c[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const pan_d2* cs = reinterpret_cast<const pan_d2*>(&Cf[(8 * jb + 4 * kh + k) * PNL_LD + 8 * (jb - 1)]);
#pragma unroll
            for (int q = 0; q < 4; ++q) c[k][q] = cs[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * kh + k] = fma(-mine[2 * q], c[k][q][0], acc[4 * kh + k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[4 * kh + k] = fma(-mine[2 * q + 1], c[k][q][1], acc[4 * kh + k]);
        }
    }
}

// Right-looking Cholesky of the 64 x 64 tile T (lower) in 8-column blocks, software-pipelined over the waves: in round
// jb wave 0 (lane = row) applies the rank-8 update of block jb - 1 to the 8 columns of block jb only and factors them
// (pivots broadcast by v_readlane, reciprocal pivots by v_rsq_f64 + Newton) while waves 1-3 apply the same update to all
// columns to the right of block jb - one barrier per round, and the serial pivot work no longer waits for the bulk of
// the update.  All waves take every barrier.
__device__ __forceinline__ void pnl_diag(double* __restrict__ T, int col0, const PanelArgs& p, int t) {
    const int i = t & 63, w = t >> 6;
    double mydiag = 1.0;
    int bad = 0;
    for (int jb = 0; jb < 8; ++jb) {
        if (w == 0) {
            double acc[8];
            pnl_rank8_next(T, T, jb, i, acc);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = 8 * jb + j;
                const double d = gpar_readlane_f64(acc[j], col);
                if (!(d > 0.0) && bad == 0) bad = col0 + col + 1;
                // pivot: rinv = d^-1/2 by v_rsq_f64 + two Newton steps, sd = d * rinv with one correction (the
                // library sqrt + divide are ~50 dependent fp64 instructions on this serial chain); 1 ulp agreement
                double rinv = __builtin_amdgcn_rsq(d);
                rinv = rinv * fma(-0.5 * d * rinv, rinv, 1.5);
                rinv = rinv * fma(-0.5 * d * rinv, rinv, 1.5);
                double sd = d * rinv;
                sd = fma(fma(-sd, sd, d), 0.5 * rinv, sd);
                const double lij = (i == col) ? sd : acc[j] * rinv;
                if (i == col) mydiag = sd;
                acc[j] = lij;
#pragma unroll
                for (int j2 = j + 1; j2 < 8; ++j2) acc[j2] = fma(-lij, gpar_readlane_f64(lij, 8 * jb + j2), acc[j2]);
            }
            pan_d2* dst = reinterpret_cast<pan_d2*>(&T[i * PNL_LD + 8 * jb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = pan_d2{acc[2 * q], acc[2 * q + 1]};
        } else if (jb > 0) {
            pnl_rank8(T, T, jb - 1, i, 8 * jb + 8 + (w - 1), 3);
        }
        __syncthreads();
    }
    if (t < 64) {
        double ld = 2.0 * log(mydiag);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
        if (t == 0) {
            if (p.logdet) atomicAdd(p.logdet, ld);
            if (bad && p.info) atomicCAS(p.info, 0, bad);
        }
    }
}

// X L^T = B for the 64 rows in Xs against the lower-triangular tile Cs, right-looking in 8-column blocks with the same
// pipeline: in round jb wave 0 (lane = row) brings the 8 columns of block jb up to date with block jb - 1 and solves
// their 8 x 8 diagonal part, waves 1-3 eliminate block jb - 1 from the columns to the right.  All waves take every barrier.
__device__ __forceinline__ void pnl_strip(const double* __restrict__ Cs, double* __restrict__ Xs, const double* __restrict__ rinvs, int t) {
    const int lane = t & 63, w = t >> 6;
    for (int jb = 0; jb < 8; ++jb) {
        if (w == 0) {
            double acc[8];
            pnl_rank8_next(Xs, Cs, jb, lane, acc);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double* crow = &Cs[(8 * jb + j) * PNL_LD + 8 * jb];
                double sacc = acc[j];
#pragma unroll
                for (int k = 0; k < j; ++k) sacc = fma(-acc[k], crow[k], sacc);
                acc[j] = sacc * rinvs[8 * jb + j];
            }
            pan_d2* dst = reinterpret_cast<pan_d2*>(&Xs[lane * PNL_LD + 8 * jb]);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = pan_d2{acc[2 * q], acc[2 * q + 1]};
        } else if (jb > 0) {
            pnl_rank8(Xs, Cs, jb - 1, lane, 8 * jb + 8 + (w - 1), 3);
        }
        __syncthreads();
    }
}

// C (64 x 64 block at rows r0, cols c0 of A) -= Xs Bs^T, both LDS tiles [64][PNL_LD] with k contiguous.
// Wave w owns rows 16 w .. 16 w + 15 x all 64 columns: four 16 x 16 v_mfma_f64_16x16x4 accumulators.
// C is read and written row-contiguously (16 bytes per lane, 256-byte row segments: the accesses of the first version,
// 8 bytes per lane straight from a 4x4x4 MFMA layout in which neighbouring lanes held different rows, were most of this
// routine's time) and the accumulators are transposed into that layout, half a row patch (4 x 32) at a time, through a
// small wave-private LDS buffer Tw ([4][PNL_LDT]).  The buffer is kept that small on purpose: the
// whole workgroup must stay below the 73.7 KB of a SYRK workgroup, because LDS is allocated contiguously and the
// hole a retiring SYRK workgroup leaves is exactly that big (a 76 KB version of this kernel was not dispatched
// until the co-running trailing update had drained: tools/time_panel.hip).
__device__ __forceinline__ void pnl_update(const PanelArgs& p, int r0, int c0, const double* __restrict__ Xs,
                                           const double* __restrict__ Bs, double* __restrict__ Tw, int t, bool lower_only) {
    const int lane = t & 63, w = t >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int rrow = lane >> 4, rcol = (lane & 15) * 2;
    // the C values of this lane are requested first so their memory latency runs under the MFMA loop
    pan_d2 cv[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int row = min(r0 + 16 * w + 4 * mi + rrow, p.N - 1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) cv[mi][hf] = *reinterpret_cast<const pan_d2*>(p.A + (size_t)row * p.lda + c0 + 32 * hf + rcol);
    }
    // v_mfma_f64_16x16x4: A lane l = (row l & 15, k = l >> 4), B lane l = (column l & 15, k = l >> 4); register v of the
    // result holds row (l >> 4) + 4 v, column l & 15 (probed: profiles/r01_probe_mfma_f64_16x16x4.txt)
    pan_d4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = pan_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k4 = 0; k4 < 16; ++k4) {
        const int kk = 4 * k4 + lk;
        const double af = Xs[(16 * w + l15) * PNL_LD + kk];
        double bf[4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) bf[nj] = Bs[(16 * nj + l15) * PNL_LD + kk];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) acc[nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf[nj], acc[nj], 0, 0, 0);
    }
    // rows 4 v .. 4 v + 3 of this wave's 16 sit in register v: one 4 x 64 patch per v, transposed half (4 x 32) at a time
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int rloc = 16 * w + 4 * mi + rrow;
        const int row = r0 + rloc;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) Tw[lk * PNL_LDT + 16 * nj + l15] = acc[2 * hf + nj][mi];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const pan_d2 v = *reinterpret_cast<const pan_d2*>(Tw + rrow * PNL_LDT + rcol);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the next half patch overwrites Tw
            const int cloc = 32 * hf + rcol;
            double* dst = p.A + (size_t)row * p.lda + c0 + cloc;
            const pan_d2 o = cv[mi][hf] - v;
            if (row < p.N) {
                if (!lower_only || cloc + 1 <= rloc) *reinterpret_cast<pan_d2*>(dst) = o;
                else if (cloc == rloc) dst[0] = o[0];
            }
        }
    }
}

// launch bounds (256, 2): at most 256 unified registers per wave.  The kernel must fit beside a trailing-update wave (240
// registers of the 512 per SIMD lane): a build that used 256 + 54 AGPRs was not dispatched until the update had drained.
__global__ __launch_bounds__(256, 2) void potrf_panel_kernel(PanelArgs p) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    double* Cs = psm;                    // L_ss (strip coefficients) / diagonal work tile
    double* Xs = psm + PNL_TILE;         // this row block's X
    double* Bs = psm;                    // L[c][s] operand of the update: ALIASES Cs (dead after the strip)
    double* rinvs = psm + 2 * PNL_TILE;  // reciprocal pivots of L_ss
    double* Tw = psm + 2 * PNL_TILE + 64 + (threadIdx.x >> 6) * 4 * PNL_LDT;   // this wave's transposition buffer
    const int t = threadIdx.x;
    const int G = gridDim.x, g = blockIdx.x;
    const int R = (p.N - p.k0 + 63) / 64;   // row blocks below (and including) the panel's first row
    const int S = p.S;
    // Under look-ahead this latency-bound kernel shares every CU with a trailing-update workgroup whose waves issue
    // fp64 MFMAs back to back on the same double-precision pipes, and its serial pivot chain runs ~3x slower than alone
    // (tools/time_panel.hip).  Raised issue priority is kept because these few, mostly waiting waves should never lose an
    // arbitration - but measured it changes little: an MFMA that has issued holds the pipe for its 16 cycles regardless.
    __builtin_amdgcn_s_setprio(3);
    if (p.stamps && t == 0) p.stamps[64 + g] = (long long)__builtin_amdgcn_s_memrealtime();   // dev aid: arrival (100 MHz wall clock)

    for (int s = 0; s < S; ++s) {
        const int cs = p.k0 + 64 * s;       // first column of column block s; row block s starts at the same index
        if (s % G == g) {
            // ---- owner of the diagonal block
            if (p.stamps && t == 0) p.stamps[s * 8 + 0] = (long long)__builtin_readcyclecounter();
            pnl_load_tile(p, cs, cs, Cs, t);
            __syncthreads();
            if (p.stamps && t == 0) p.stamps[s * 8 + 1] = (long long)__builtin_readcyclecounter();
            pnl_diag(Cs, cs, p, t);
            if (p.stamps && t == 0) p.stamps[s * 8 + 2] = (long long)__builtin_readcyclecounter();
            pnl_store_tile(p, cs, cs, Cs, t, true, true);
            pnl_publish(p, s);
            if (p.stamps && t == 0) p.stamps[s * 8 + 3] = (long long)__builtin_readcyclecounter();
        }
        bool have_lss = false;
        int first = s + 1;                  // first owned row block above s
        first += ((g - first) % G + G) % G;
        for (int rb = first; rb < R; rb += G) {
            const int r0 = p.k0 + 64 * rb;
            const bool crit = p.stamps && t == 0 && rb == s + 1;
            if (crit) p.stamps[s * 8 + 4] = (long long)__builtin_readcyclecounter();
            if (!have_lss) {
                pnl_wait(p, s);
                if (crit) p.stamps[s * 8 + 5] = (long long)__builtin_readcyclecounter();
                have_lss = true;
            }
            __syncthreads();                // previous iteration done with Xs / Bs
            pnl_load_tile(p, cs, cs, Cs, t); // (re)load L_ss: the update operand of the previous row block overwrote it
            pnl_load_tile(p, r0, cs, Xs, t);
            __syncthreads();
            if (t < 64) rinvs[t] = 1.0 / Cs[t * PNL_LD + t];
            __syncthreads();
            pnl_strip(Cs, Xs, rinvs, t);
            __syncthreads();
            if (crit) p.stamps[s * 8 + 6] = (long long)__builtin_readcyclecounter();
            pnl_store_tile(p, r0, cs, Xs, t, false, rb < S);
            if (rb < S) pnl_publish(p, S + s * S + rb);
            const int cmax = rb < S - 1 ? rb : S - 1;
            for (int c = s + 1; c <= cmax; ++c) {
                const double* Bt = Xs;
                if (c != rb) {
                    __syncthreads();        // strip / earlier update done with the Cs = Bs tile
                    pnl_wait(p, S + s * S + c);
                    pnl_load_tile(p, p.k0 + 64 * c, cs, Bs, t);
                    __syncthreads();
                    Bt = Bs;
                }
                pnl_update(p, r0, p.k0 + 64 * c, Xs, Bt, Tw, t, c == rb);
            }
            if (crit) p.stamps[s * 8 + 7] = (long long)__builtin_readcyclecounter();
        }
        // a workgroup that owns the next diagonal block must see its own updates of that block: same CU, plain
        // stores then plain loads through the same L1/L2 -> ordered by the vmcnt drain + barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p.stamps && t == 0) p.stamps[64 + 256 * (1 + s) + g] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// Number of workgroups that can be co-resident (1 per CU at this LDS size): queried once.
static int panel_grid_cap() {
    static int caps[64];   // per device; 0 = not queried yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64;
    if (caps[dev] == 0) {
        int cus = 0;
        int cap = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 64;
        const char* e = getenv("GPAR_PANEL_GRID");   // experiment knob: fewer, busier panel workgroups
        if (e && atoi(e) > 0 && atoi(e) < cap) cap = atoi(e);
        caps[dev] = cap;
    }
    return caps[dev];
}

// The flag words of every 64-aligned diagonal block, zeroed in ONE launch at the start of a factorisation: nothing in
// gpar_potrf writes the strict upper triangle of a diagonal block before that block's own panel kernel does, and two
// hipMemsetAsync per panel were ~12 us of idle chip on the serial chain (32 panels at n = 16384).
constexpr int PNL_FLAG_ROWS = (PNL_MAX_S + PNL_MAX_S * PNL_MAX_S + PNL_FLAG_SLOTS - 1) / PNL_FLAG_SLOTS;
__global__ __launch_bounds__(256) void potrf_zero_flags_kernel(double* __restrict__ A, int lda, long long batch_a) {
    A += (size_t)blockIdx.y * batch_a;
    const int k0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < PNL_FLAG_ROWS * PNL_FLAG_SLOTS; i += blockDim.x)
        A[(size_t)(k0 + i / PNL_FLAG_SLOTS) * lda + k0 + 8 + i % PNL_FLAG_SLOTS] = 0.0;
}
// true if the flags of a panel starting at k0 are covered by potrf_zero_flags(A, N, ...)
static inline bool potrf_flags_prezeroed(int N, int k0) { return k0 % 64 == 0 && k0 + 64 <= N; }
static void potrf_zero_flags(double* A, int N, int lda, hipStream_t stream, int batch = 1, long long batch_a = 0) {
    if (N >= 64) hipLaunchKernelGGL(potrf_zero_flags_kernel, dim3(N / 64, batch), dim3(256), 0, stream, A, lda, batch_a);
}

static int potrf_panel_fused(double* A, int N, int lda, int k0, int W, double* logdet, int* info, hipStream_t stream,
                             bool prezeroed = false) {
    PanelArgs p{A, N, lda, k0, W / 64, logdet, info, nullptr};
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&potrf_panel_kernel), PNL_LDS_BYTES));
    // zero the flag words (S + S^2 of them, 56 per scratch row in the strict upper triangle of the first diagonal
    // block: rows 0..7 have columns 8..63 strictly above the diagonal, which bounds S at 16)
    const int nflags = p.S + p.S * p.S;
    if (p.S > PNL_MAX_S) return GPAR_ARG_ERROR(5);
    if (!(prezeroed && potrf_flags_prezeroed(N, k0)))
        for (int r = 0; r * PNL_FLAG_SLOTS < nflags; ++r)
            GPAR_HIP_TRY(hipMemsetAsync(A + (size_t)(k0 + r) * lda + k0 + 8, 0, PNL_FLAG_SLOTS * sizeof(double), stream));
    const int R = (N - k0 + 63) / 64;
    int G = R < panel_grid_cap() ? R : panel_grid_cap();
    if (int rc = spin_chain_enter(stream, G)) return rc;
    hipLaunchKernelGGL(potrf_panel_kernel, dim3(G), dim3(256), PNL_LDS_BYTES, stream, p);
    spin_chain_leave(stream, G);
    GPAR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Fused block of the forward triangular solve  X L^T = B  (gpar_trsm_rlt): ONE launch carries a 64-row block of B
// through S consecutive 64-column steps of L - strip solve against L_ss, then the rank-64 updates of the block's
// remaining columns - instead of S strip launches + (S - 1) GEMM launches.  Row blocks are independent, so unlike
// the panel factorisation there are no hand-offs between workgroups: it is the panel kernel's strip / update code
// with the diagonal factorisation and the flags removed.
struct TrsmBlockArgs {
    const double* L;   // n x n lower-triangular factor (strict upper triangle never read)
    int n, ldl;
    double* B;         // nrows x n right-hand sides, overwritten by X
    int nrows, ldb;
    int c0, S;         // columns [c0, c0 + 64 S)
    int upper_tri;     // B is upper triangular on entry: row r has nothing left of column r -> whole steps are skipped
    int pairs = 0;     // panel2.h: column blocks taken in pairs (p2_row_block_pairs)
    const int* pred = nullptr;   // predicated solve (common.h: GparPredicate): return at once unless (*pred != 0) == pred_sense
    int pred_sense = 0;
};

__global__ __launch_bounds__(256) void trsm_block_kernel(TrsmBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) double psm[];
    if (gpar_pred_skip(a.pred, a.pred_sense)) return;
    double* Cs = psm;
    double* Xs = psm + PNL_TILE;
    double* Bs = psm;
    double* rinvs = psm + 2 * PNL_TILE;
    double* Tw = psm + 2 * PNL_TILE + 64 + (threadIdx.x >> 6) * 4 * PNL_LDT;
    const int t = threadIdx.x;
    const int r0 = 64 * blockIdx.x;
    const PanelArgs pL{const_cast<double*>(a.L), a.n, a.ldl, 0, 0, nullptr, nullptr, nullptr};
    const PanelArgs pB{a.B, a.nrows, a.ldb, 0, 0, nullptr, nullptr, nullptr};
    for (int s = 0; s < a.S; ++s) {
        const int cs = a.c0 + 64 * s;
        if (a.upper_tri && r0 >= cs + 64) continue;   // these rows are still zero in every column <= cs + 63
        __syncthreads();                              // previous step done with Xs / Bs
        pnl_load_tile(pL, cs, cs, Cs, t);
        pnl_load_tile(pB, r0, cs, Xs, t);
        __syncthreads();
        if (t < 64) rinvs[t] = 1.0 / Cs[t * PNL_LD + t];
        __syncthreads();
        pnl_strip(Cs, Xs, rinvs, t);
        __syncthreads();
        pnl_store_tile(pB, r0, cs, Xs, t, false, false);
        for (int c = s + 1; c < a.S; ++c) {
            __syncthreads();                          // strip / earlier update done with the Cs = Bs tile
            pnl_load_tile(pL, a.c0 + 64 * c, cs, Bs, t);
            __syncthreads();
            pnl_update(pB, r0, a.c0 + 64 * c, Xs, Bs, Tw, t, false);
        }
    }
}

static int trsm_block_fused(const double* L, int n, int ldl, double* B, int nrows, int ldb, int c0, int S, int upper_tri,
                            hipStream_t stream) {
    GPAR_HIP_TRY(gpar_set_max_lds(reinterpret_cast<const void*>(&trsm_block_kernel), PNL_LDS_BYTES));
    TrsmBlockArgs a{L, n, ldl, B, nrows, ldb, c0, S, upper_tri};
    a.pred = g_pred.flag; a.pred_sense = g_pred.sense;
    hipLaunchKernelGGL(trsm_block_kernel, dim3(gpar_ceil_div(nrows, 64)), dim3(256), PNL_LDS_BYTES, stream, a);
    GPAR_LAUNCH_CHECK();
    return 0;
}

}  // namespace gpar
