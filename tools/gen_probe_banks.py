"""Generates tools/probe_mfma_banks.hip: v_mfma_f64_16x16x4_f64 throughput as a function of the VGPR banks of its A / B operands
(16 independent accumulators v[16:143], explicit registers, 1 or 2 waves per SIMD)."""
VARIANTS = {  # name: (A first reg, B first reg)
    "A0_B2": (0, 2), "A0_B4": (0, 4), "A2_B6": (2, 6), "A0_B0": (0, 0), "A0_B6": (0, 6),
}
ACC0 = {"acc16": 16, "acc18": 18}
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <vector>', '#include <algorithm>']
names = []
for an, a0 in ACC0.items():
    for vn, (ra, rb) in VARIANTS.items():
        name = f"k_{vn}_{an}"
        names.append(name)
        body = ["s_mov_b32 s4, %[iters]"]
        for r in range(0, 16 + 8 * 16 + 4):
            body.append(f"v_mov_b32 v{r}, 0")
        body.append("s_memtime s[6:7]")
        body.append("1:")
        for i in range(16):
            d = a0 + 8 * i
            body.append(f"v_mfma_f64_16x16x4_f64 v[{d}:{d+7}], v[{ra}:{ra+1}], v[{rb}:{rb+1}], v[{d}:{d+7}]")
        body += ["s_sub_u32 s4, s4, 1", "s_cmp_lg_u32 s4, 0", "s_cbranch_scc1 1b", "s_memtime s[8:9]", "s_waitcnt lgkmcnt(0)",
                 "s_sub_u32 s6, s8, s6", "s_subb_u32 s7, s9, s7", "v_mov_b32 %[lo], s6", "v_mov_b32 %[hi], s7"]
        clob = ", ".join(f'"v{r}"' for r in range(0, 148)) + ', "s4", "s6", "s7", "s8", "s9", "scc"'
        asm = "\\n\\t".join(body)
        src.append(f'''__global__ __launch_bounds__(256) void {name}(long long* out, int iters) {{
    unsigned lo, hi;
    asm volatile("{asm}" : [lo] "=v"(lo), [hi] "=v"(hi) : [iters] "s"(iters) : {clob});
    if (threadIdx.x == 0) out[blockIdx.x] = ((long long)hi << 32) | lo;
}}''')
src.append("typedef void (*kern_t)(long long*, int);")
src.append("int main(int argc, char** argv) {\n    const int iters = argc > 1 ? atoi(argv[1]) : 4000;\n    long long* d; (void)hipMalloc(&d, sizeof(long long) * 512);")
src.append("    struct { const char* n; kern_t k; } ks[] = {" + ", ".join(f'{{"{n}", {n}}}' for n in names) + "};")
src.append('''    for (auto& e : ks)
        for (int wps = 1; wps <= 2; ++wps) {
            const int nb = 256 * wps;
            std::vector<long long> h(nb);
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(e.k, dim3(nb), dim3(256), 0, 0, d, iters);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h.data(), d, sizeof(long long) * nb, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("%-14s waves/SIMD %d: %.1f cycles per MFMA and SIMD\\n", e.n, wps, (double)h[nb / 2] / (16.0 * iters * wps));
        }
    return 0;
}''')
open("tools/probe_mfma_banks.hip", "w").write("\n".join(src) + "\n")
