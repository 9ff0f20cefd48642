"""Prints the two inline-asm bodies of p3_diag_block (gpar_amd/csrc/panel2.h): the rank-8 update + eight pivots of one
8-column block of the diagonal-tile factorisation, with every wave-uniform coefficient taken through DPP row_newbcast.

    python tools/gen_diag_asm.py > /tmp/diag_asm.inc      # pasted between the GENERATED markers of panel2.h

Operands: %0-%7 acc[0..7] ("+v"), %8 D, %9 R, %10 E, %11 T1, %12 T2 ("=&v"), then (UPDATE) %13-%20 prev[0..7], and last 0.375.
"""
DPP = " row_newbcast:%d row_mask:0xf bank_mask:0xf"


# Hazards (CDNA3 ISA, manually inserted wait states): a vector write of a register -> a DPP instruction reading it: 2;
# a transcendental result -> its first use: 1.


def body(update):
    c375 = "%21" if update else "%13"
    L = ["s_nop 1"]          # registers the compiler may just have copied
    if update:
        # q outer, k inner: an accumulator is touched again eight instructions later
        for q in range(8):
            for k in range(8):
                L.append("v_fmac_f64_dpp %%%d, -%%%d, %%%d" % (k, 13 + q, 13 + q) + DPP % k)
    for j in range(8):
        a = "%%%d" % j
        # (v_rsq_f64_dpp assembles but returns rsq(0) on gfx950: the broadcast goes through a move)
        L.append("v_mov_b64_dpp %8, " + a + DPP % j)
        L.append("v_rsq_f64 %9, %8")
        L.append("s_nop 0")
        L.append("v_mul_f64 %11, %8, %9")
        L.append("v_fma_f64 %10, -%11, %9, 1.0")
        L.append("v_fma_f64 %12, %10, " + c375 + ", 0.5")
        L.append("v_mul_f64 %11, %9, %10")
        L.append("v_fma_f64 %9, %11, %12, %9")
        L.append("v_mul_f64 " + a + ", " + a + ", %9")
        if j < 7:
            L.append("s_nop 1")
        for j2 in range(j + 1, 8):
            L.append("v_fmac_f64_dpp %%%d, -%s, %s" % (j2, a, a) + DPP % j2)
        # the next pivot's DPP reads acc[j + 1], written by the first multiply-add above
        n_after = 6 - j
        if j < 7 and n_after < 2:
            L.append("s_nop %d" % (1 - n_after))
    return L


for name, upd in (("P3_ASM_FIRST", False), ("P3_ASM_UPDATE", True)):
    lines = body(upd)
    print("#define %s \\" % name)
    for i, l in enumerate(lines):
        print('    "%s\\n"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    print()
