// What the fused panel kernels (panel2.h) share: the geometry constants of a 64 x 64 tile in LDS, the hand-off flag words in the strict
// upper triangle of a panel's first diagonal tile, the ordering of waiting launches across streams (the spin chain), the argument
// structs, and the one launch that zeroes every panel's flag words at the start of a factorisation.
// (The first-generation fused panel kernel and solve block that lived here - one workgroup per row block doing diagonal tile, strip and
// update in turn, round 1; selected by GPAR_PANEL_V=1 until round 5 - were retired in round 6: the second generation has been the
// default since round 2 and the unfused leaf kernels are the fallback; `git show a55bc67:gpar_amd/csrc/panel.h` has the code.)
#pragma once
#include "common.h"
#include "potrf.h"

namespace gpar {

constexpr int PNL_LD = 66;
constexpr int PNL_TILE = 64 * PNL_LD;                  // doubles
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

// Arguments of the fused block of the forward triangular solve  X L^T = B  (gpar_trsm_rlt; kernel: panel2.h, trsm_block2_kernel).
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

}  // namespace gpar
