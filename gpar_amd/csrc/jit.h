// Per-specification kernels, compiled at run time with hiprtc (ROCm's runtime compiler) and cached per device.
//
// A GPAR layer decides its kernel structure once (gpar/regression.py:92-180: which terms, which factor of which type over
// how many feature dims) and then evaluates it thousands of times with changing hyper-parameter VALUES.  The ahead-of-time
// kernels in gram.h interpret the term list at run time - loops over terms / factors / dims with wave-uniform branches,
// 93 vector instructions executed per entry of which 62 are arithmetic.  Here the STRUCTURE (term list, factor types, offsets,
// dim counts) is baked into generated source, so every loop is unrolled, every type test folds away and per-parameter
// accumulators have static register names; the VALUES (coefficients, RQ shapes) stay kernel arguments, so training never
// recompiles.  The interpreter stays as the fallback (small problems, hiprtc unavailable) and as the bit-exactness reference:
// both paths call the same arithmetic helpers in the same order.
#pragma once
#include <hip/hiprtc.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <string>

#include "common.h"

namespace gpar {

enum JitKind { JIT_GRAM = 0, JIT_GRAD = 1, JIT_INPUT_GRAD = 2 };

struct JitEntry {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
};

struct JitState {
    std::map<std::string, JitEntry> cache;   // key: device + kind + structure signature
    std::string last_log;
    int compiled = 0, failures = 0;
};
static JitState g_jit;

// Structure signature of a kernel specification: everything the generated code depends on, nothing that training changes.
static std::string jit_signature(const gpar_kspec_t& ks, int dz, int extra) {
    std::string s = "t" + std::to_string(ks.nterms) + "d" + std::to_string(dz) + "x" + std::to_string(extra);
    for (int f = 0; f < ks.nfactors; ++f) {
        const gpar_factor_t& fa = ks.factor[f];
        s += "|" + std::to_string(fa.type) + "," + std::to_string(fa.term) + "," + std::to_string(fa.off) + "," + std::to_string(fa.nd);
    }
    return s;
}

// Compile `source` (entry point `name`) for `arch`; on success the code object is returned in `code`.
static bool jit_compile(const std::string& source, const char* name, const std::string& arch, std::string& code, std::string& log) {
    if (const char* dump = getenv("GPAR_JIT_DUMP")) {   // development aid: keep the generated source (one file per entry point)
        const std::string path = std::string(dump) + "/" + name + ".hip";
        if (FILE* f = fopen(path.c_str(), "w")) { fputs(source.c_str(), f); fclose(f); }
    }
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, source.c_str(), name, 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    const std::string archopt = "--offload-arch=" + arch;
    // (the ahead-of-time build's options: no fast-math - the generated kernels must round exactly like the ahead-of-time ones)
    const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    size_t logn = 0;
    if (hiprtcGetProgramLogSize(prog, &logn) == HIPRTC_SUCCESS && logn > 1) {
        log.assign(logn, '\0');
        hiprtcGetProgramLog(prog, &log[0]);
    } else {
        log.clear();
    }
    bool ok = r == HIPRTC_SUCCESS;
    if (ok) {
        size_t n = 0;
        ok = hiprtcGetCodeSize(prog, &n) == HIPRTC_SUCCESS && n > 0;
        if (ok) {
            code.assign(n, '\0');
            ok = hiprtcGetCode(prog, &code[0]) == HIPRTC_SUCCESS;
        }
    }
    hiprtcDestroyProgram(&prog);
    return ok;
}

static std::string jit_device_arch() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return "";
    return std::string(prop.gcnArchName);
}

static std::string jit_key(int kind, const gpar_kspec_t& ks, int dz, int extra) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return "";
    return std::to_string(dev) + "#" + std::to_string(kind) + "#" + jit_signature(ks, dz, extra);
}

// Load a compiled code object on the current device and enter it into the cache (callers hold the library mutex).
static hipFunction_t jit_install(const std::string& key, const std::string& code, const char* entry, bool compiled_ok, const std::string& log) {
    JitEntry e;
    if (!compiled_ok || hipModuleLoadData(&e.module, code.data()) != hipSuccess || hipModuleGetFunction(&e.fn, e.module, entry) != hipSuccess) {
        e.failed = true;
        e.fn = nullptr;
        g_jit.failures++;
        g_jit.last_log = log;
    } else {
        g_jit.compiled++;
    }
    g_jit.cache[key] = e;
    return e.failed ? nullptr : e.fn;
}

// The compiled function for (kind, structure) on the current device, or nullptr (never compiled twice: a failure is cached too).
// `make_source` is only called on a cache miss.  Callers hold the library mutex.
template <typename MakeSource>
static hipFunction_t jit_get(int kind, const gpar_kspec_t& ks, int dz, int extra, const char* entry, MakeSource make_source) {
    const std::string key = jit_key(kind, ks, dz, extra);
    if (key.empty()) return nullptr;
    auto it = g_jit.cache.find(key);
    if (it != g_jit.cache.end()) return it->second.failed ? nullptr : it->second.fn;
    std::string code, log;
    const std::string arch = jit_device_arch();
    const bool ok = !arch.empty() && jit_compile(make_source(), entry, arch, code, log);
    return jit_install(key, code, entry, ok, log);
}

}  // namespace gpar
