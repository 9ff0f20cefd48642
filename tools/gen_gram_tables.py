"""Tables of the Gram kernels' elementary functions (gpar_amd/csrc/gram_math.inc: GRAM_TAB), from 200-bit arithmetic:
  64 entries  2^(j/64)                                    (gram_exph8)
  128 pairs   { v_j = 1 / c_j rounded, log(2 / v_j) },  c_j = (1 + (j + 1/2) / 128) / 2   (gram_log1p_pos)
Prints the initialiser list; `--check` compares it with what the source holds."""
import re
import sys

import mpmath as mp

mp.mp.prec = 200


def hexd(x):
    return float(x).hex().replace("0x1.", "0x1.").replace("p", "p")


def table():
    out = [float(mp.mpf(2) ** (mp.mpf(j) / 64)) for j in range(64)]
    for j in range(128):
        c = (1 + (mp.mpf(j) + mp.mpf(1) / 2) / 128) / 2
        v = float(1 / c)
        out += [v, float(mp.log(2 / mp.mpf(v)))]
    return out


def literal(v):
    m = re.fullmatch(r"0x1\.([0-9a-f]*)p([+-]\d+)", v.hex())
    return "0x1.%sp%s" % (m.group(1).ljust(13, "0"), m.group(2)) if m else v.hex()


if __name__ == "__main__":
    vals = table()
    if "--check" in sys.argv:
        import os
        src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpar_amd", "csrc", "gram_math.inc")).read()
        body = src[src.index("GRAM_TAB[GRAM_TAB_DOUBLES] = {"):]
        body = body[body.index("{") + 1: body.index("}")]
        have = [float.fromhex(t.strip()) for t in body.split(",") if t.strip()]
        assert have == vals, "table in gram_math.inc differs from the generated one"
        print("ok:", len(have), "entries")
    else:
        lines = [", ".join(literal(v) for v in vals[i:i + 4]) for i in range(0, len(vals), 4)]
        print(",\n".join("        " + line for line in lines))
