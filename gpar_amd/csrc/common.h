// Shared helpers for libgpar_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/gpar_hip.h"

#define GPAR_WAVE 64

static inline int gpar_hip_status(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }

#define GPAR_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return -(int)e__;    \
    } while (0)

#define GPAR_ARG_ERROR(code) (-1000 - (code))

static inline int gpar_ceil_div(int a, int b) { return (a + b - 1) / b; }

static inline bool gpar_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
