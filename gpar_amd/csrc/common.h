// Shared helpers for libgpar_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <utility>
#include <vector>

#include "../../include/gpar_hip.h"

#define GPAR_WAVE 64

static inline int gpar_hip_status(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }

#define GPAR_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return -(int)e__;    \
    } while (0)

// A HIP runtime call inside a function that returns the library's int status: a failure becomes -(hipError_t).
#define GPAR_HIP_TRY(call)                          \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return -(int)e__;    \
    } while (0)
// A HIP runtime call whose failure cannot be reported from where it is made and does not affect results (profiling
// events, restoring the caller's device): the status is dropped on purpose.
#define GPAR_HIP_IGNORE(call) ((void)(call))

#define GPAR_ARG_ERROR(code) (-1000 - (code))

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device) - a
// process may drive several GPUs (callers hold the library mutex).
static inline hipError_t gpar_set_max_lds(const void* fn, int bytes) {
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    for (const auto& d : done)
        if (d.first == fn && d.second == dev) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.emplace_back(fn, dev);
    return e;
}

static inline int gpar_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Device-side predication of a solve (gpar_trsm_rlt_if): while `flag` is set, the kernels the triangular-solve path launches
// (block kernels, strips, their role-0 GEMM updates) return at once unless (flag[0] != 0) == (sense != 0) - the decision is
// taken on the device, from a word an earlier kernel of the stream wrote, without a host synchronisation.  Entry points run under
// the library's mutex, so one context serves.
struct GparPredicate {
    const int* flag = nullptr;
    int sense = 0;
};
static GparPredicate g_pred;
__device__ __forceinline__ bool gpar_pred_skip(const int* flag, int sense) { return flag && ((*flag != 0) != (sense != 0)); }

__host__ __device__ static inline bool gpar_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Broadcast lane `src` (compile-time constant after unrolling) of a double to the whole wave through SGPRs
// (v_readlane_b32 x2) instead of the LDS crossbar (ds_bpermute) that __shfl lowers to.
__device__ __forceinline__ double gpar_readlane_f64(double v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
