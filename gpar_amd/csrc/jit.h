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
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

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

// ---- kernels compiled at BUILD time --------------------------------------------------------------------------------------------
// The same generated sources, compiled by the same compiler with the same options when the library is built
// (__graft_entry__.build() -> gpar_amd/aot.py -> gpar_jit_compile) for the layer structures GPARRegressor's common configurations
// produce, and kept as an archive of code objects next to the library (gpar_aot_<arch>.bin).  A structure found there costs a
// hipModuleLoadData (~1 ms) instead of 0.3-0.6 s of hiprtc at first use - so it is used at EVERY problem size, not only where a
// training run repays a compilation: below n = 4096 the interpreting gradient kernels run at 397 / 380 registers with spills.
// Archive: "GPARAOT2", u32 ABI version, u64 generator fingerprint, u32 arch length + arch, u32 count, then per entry u32 key length +
// key ("<kind>#<signature>"), u64 code size + code object.  The fingerprint (aot_fingerprint() in gpar_hip.hip) hashes the sources
// this library GENERATES for a probe structure that exercises every factor type and kernel kind: an archive written by a library
// whose generators (gram_jit.h, grad_jit.h, gram_math.inc) or ABI differ is ignored, whatever the file dates say - its code
// objects could have another argument layout or other arithmetic.  Anything unreadable is ignored too (hiprtc then serves).
static unsigned long long aot_fingerprint();   // gpar_hip.hip (needs the source generators)
struct AotState {
    bool tried = false;
    std::string arch;
    std::vector<char> blob;
    std::map<std::string, std::pair<size_t, size_t>> entries;   // key -> (offset, size) into blob
    int loaded = 0;
    bool stale = false;   // an archive was found and rejected: other ABI version or generator fingerprint
};
static AotState g_aot;

static std::string aot_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&aot_dir), &info) || !info.dli_fname) return ".";
    std::string path(info.dli_fname);
    const size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? "." : path.substr(0, slash);
}

static void aot_init(const std::string& arch) {
    if (g_aot.tried) return;
    g_aot.tried = true;
    if (const char* e = getenv("GPAR_AOT")) { if (atoi(e) == 0) return; }
    std::string base = arch.substr(0, arch.find(':'));   // "gfx950:sramecc+:xnack-" -> "gfx950"
    const std::string path = aot_dir() + "/gpar_aot_" + base + ".bin";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> blob(size > 0 ? (size_t)size : 0);
    const bool read_ok = size > 0 && fread(blob.data(), 1, (size_t)size, f) == (size_t)size;
    fclose(f);
    if (!read_ok || blob.size() < 32 || memcmp(blob.data(), "GPARAOT2", 8) != 0) return;
    size_t at = 8;
    auto u32 = [&](uint32_t& v) { if (at + 4 > blob.size()) return false; memcpy(&v, &blob[at], 4); at += 4; return true; };
    auto u64 = [&](uint64_t& v) { if (at + 8 > blob.size()) return false; memcpy(&v, &blob[at], 8); at += 8; return true; };
    uint32_t abi = 0;
    uint64_t fingerprint = 0;
    if (!u32(abi) || !u64(fingerprint) || abi != (uint32_t)GPAR_ABI_VERSION || fingerprint != (uint64_t)aot_fingerprint()) {
        g_aot.stale = true;
        return;
    }
    uint32_t alen = 0, count = 0;
    if (!u32(alen) || at + alen > blob.size()) return;
    const std::string built_for(&blob[at], alen);
    at += alen;
    if (built_for != base || !u32(count)) return;
    std::map<std::string, std::pair<size_t, size_t>> entries;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t klen = 0;
        uint64_t csize = 0;
        if (!u32(klen) || at + klen > blob.size()) return;
        std::string key(&blob[at], klen);
        at += klen;
        if (!u64(csize) || at + csize > blob.size()) return;
        entries[key] = {at, (size_t)csize};
        at += csize;
    }
    g_aot.arch = base;
    g_aot.blob.swap(blob);
    g_aot.entries.swap(entries);
}

static std::string aot_key(int kind, const gpar_kspec_t& ks, int dz, int extra) { return std::to_string(kind) + "#" + jit_signature(ks, dz, extra); }

static std::string jit_device_arch() {
    static std::string known[16];   // (asked on every small launch: the property query is made once per device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return "";
    if (dev >= 0 && dev < 16 && !known[dev].empty()) return known[dev];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return "";
    const std::string arch(prop.gcnArchName);
    if (dev >= 0 && dev < 16) known[dev] = arch;
    return arch;
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

// Launches below this many entries stay on the interpreter even when the archive holds their structure (GPAR_AOT_MIN_ENTRIES): a
// handful of 64 x 64 tiles is all launch latency either way, and there the interpreter's is the shorter one - fit(iters=20), three
// layers at n = 400: 87 ms interpreted, 104 ms generated; from n = 1024 on the generated kernels win (138 -> 123 ms; 3000: 298 -> 268).
static long long aot_min_entries() {
    static long long v = -1;
    if (v < 0) { const char* e = getenv("GPAR_AOT_MIN_ENTRIES"); v = e ? atoll(e) : (1LL << 19); }
    return v;
}

// Is there a build-time compiled kernel for (kind, structure) on the current device?  (Callers hold the library mutex.)
static bool aot_has(int kind, const gpar_kspec_t& ks, int dz, int extra) {
    const std::string arch = jit_device_arch();
    if (arch.empty()) return false;
    aot_init(arch);
    return !g_aot.entries.empty() && g_aot.entries.count(aot_key(kind, ks, dz, extra)) > 0;
}

// Load the build-time compiled kernel of (kind, structure), if the archive holds one, under cache key `key`.
static hipFunction_t aot_install(int kind, const gpar_kspec_t& ks, int dz, int extra, const std::string& key, const char* entry, const std::string& arch) {
    if (arch.empty()) return nullptr;
    aot_init(arch);
    auto it = g_aot.entries.find(aot_key(kind, ks, dz, extra));
    if (it == g_aot.entries.end()) return nullptr;
    JitEntry e;
    if (hipModuleLoadData(&e.module, g_aot.blob.data() + it->second.first) != hipSuccess || hipModuleGetFunction(&e.fn, e.module, entry) != hipSuccess)
        return nullptr;   // (unusable entry: hiprtc serves)
    g_jit.cache[key] = e;
    g_aot.loaded++;
    return e.fn;
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
    if (hipFunction_t fn = aot_install(kind, ks, dz, extra, key, entry, arch)) return fn;
    const bool ok = !arch.empty() && jit_compile(make_source(), entry, arch, code, log);
    return jit_install(key, code, entry, ok, log);
}

}  // namespace gpar
