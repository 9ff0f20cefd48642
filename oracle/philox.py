"""Oracle (TEST INFRASTRUCTURE): Philox-4x32-10 + Box-Muller standard normals, numpy restatement of the
published algorithm (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).

The reference draws its normals from torch's global CPU generator (lab `B.randn`, reached from
`Normal.sample` at /root/reference/gpar/model.py:235,264,266,270); that stream cannot be reproduced on a
GPU, so the product uses a counter-based generator and this file pins its bits: uniforms are bit-exact,
normals agree to a few ulp of log/sin/cos.
"""
import numpy as np

__all__ = ["philox4x32_10", "randn"]

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & _MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def randn(seed, offset, rows, cols):
    """rows x cols standard normals; element index e = r * cols + c; pair p = e // 2 uses counter
    (p_lo, p_hi, offset_lo, offset_hi) and key (seed_lo, seed_hi)."""
    total = rows * cols
    npairs = (total + 1) // 2
    p = np.arange(npairs, dtype=np.uint64)
    c0 = (p & _MASK).astype(np.uint32)
    c1 = (p >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(npairs, offset & 0xFFFFFFFF, dtype=np.uint32)
    c3 = np.full(npairs, (offset >> 32) & 0xFFFFFFFF, dtype=np.uint32)
    r0, r1, r2, r3 = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = (r1.astype(np.uint64) << np.uint64(32)) | r0.astype(np.uint64)
    b = (r3.astype(np.uint64) << np.uint64(32)) | r2.astype(np.uint64)
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    z = np.empty(npairs * 2)
    z[0::2] = rad * np.cos(ang)
    z[1::2] = rad * np.sin(ang)
    return z[:total].reshape(rows, cols)
