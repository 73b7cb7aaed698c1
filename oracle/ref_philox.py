"""numpy restatement of the counter-based input generator (gsa_fill_inputs) -- test infrastructure only.

Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11;
constants and round function as in Random123's philox.h), checked against that library's known-answer vectors in
tests/test_philox.py; then the same Box-Muller transform as the kernel (fp32; libm's log/cos/sin differ from the
device's in the last bits, so normals are compared with a tolerance, the integer stream exactly).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 array, key: (2,) ints -> (..., 4) uint32."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def fill_normal(n, per_sample, first_index, plane, seed):
    """-> (n, per_sample) fp32, the values fill_normal_kernel writes (up to libm rounding)."""
    quads = per_sample // 4
    ctr = np.zeros((n, quads, 4), np.uint32)
    ctr[..., 0] = np.arange(quads, dtype=np.uint32)[None, :]
    ctr[..., 1] = plane
    idx = np.uint64(first_index) + np.arange(n, dtype=np.uint64)
    ctr[..., 2] = (idx & MASK).astype(np.uint32)[:, None]
    ctr[..., 3] = (idx >> np.uint64(32)).astype(np.uint32)[:, None]
    x = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    out = np.empty((n, quads, 4), np.float32)
    for h in range(2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[..., 2 * h]), dtype=np.float32)
        a = np.float32(6.283185307179586) * u[..., 2 * h + 1]
        out[..., 2 * h] = r * np.cos(a, dtype=np.float32)
        out[..., 2 * h + 1] = r * np.sin(a, dtype=np.float32)
    return out.reshape(n, per_sample)
