"""Oracle (test infrastructure only): numpy restatement of the DEVICE normal stream.

The reference draws its normals from numpy's global/Generator state
(cora/util/nputil.py:104-125, called per l from cora/core/skysim.py:120); that stream is
reproduced bit-for-bit by the host path of the build (parity mode).  The throughput mode
replaces it by a counter-based stream so that the draw can stay on the GPU and be identical
for any number of GPUs ("device counter-based RNG keyed by (seed, l, m, nu, re/im)").
This file is the specification of that stream, checked against the kernels in tests/:

  Philox4x32-10 (Salmon et al., SC'11; known-answer vectors of Random123 in tests/test_oracle.py)
  counter = (m // 2, l * 2F + c * F + nu', 0, 0),  key = (seed & 0xffffffff, seed >> 32)
  k1 = r0 << 21 | r1 >> 11,  k2 = r2 << 21 | r3 >> 11        (two 53-bit integers)
  u1 = (k1 + 0.5) 2^-53,     u2 = (k2 + 0.5) 2^-53            (IEEE double arithmetic)
  normal(l, c, nu', m even) = sqrt(-2 ln u1) cos(2 pi u2),  normal(.., m + 1) = sqrt(-2 ln u1) sin(2 pi u2)

laid out in the reference's stream order: for l: F*(l+1) reals [nu'][m], then F*(l+1) imags.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds; inputs uint32-valued arrays/ints, returns 4 uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & _MASK for v in (c0, c1, c2, c3)]
    c = list(np.broadcast_arrays(*c))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c = [n0, n1, n2, n3]
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in c)


def boxmuller_counter(seed, lo, hi):
    """The two normals of Philox counter (lo, hi, 0, 0) under key = seed (broadcasts over arrays)."""
    r0, r1, r2, r3 = philox4x32_10(lo, hi, 0, 0, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    k1 = (r0.astype(np.uint64) << np.uint64(21)) | (r1.astype(np.uint64) >> np.uint64(11))
    k2 = (r2.astype(np.uint64) << np.uint64(21)) | (r3.astype(np.uint64) >> np.uint64(11))
    u1 = (k1.astype(np.float64) + 0.5) * 2.0**-53
    u2 = (k2.astype(np.float64) + 0.5) * 2.0**-53
    rad = np.sqrt(-2.0 * np.log(u1))
    # octant-exact angle reduction (u2 * 8 is exact) so that the restatement itself is good to ~1e-16
    a = 8.0 * u2
    q = np.floor(a).astype(np.int64)
    f = a - q
    g = np.where(q & 1, 1.0 - f, f) * (np.pi / 4)
    s, co = np.sin(g), np.cos(g)
    swap = ((q + 1) >> 1) & 1
    cc = np.where(swap, s, co)
    ss = np.where(swap, co, s)
    cs = np.where(((q + 2) >> 2) & 1, -cc, cc)
    sn = np.where(q & 4, -ss, ss)
    return rad * cs, rad * sn


def normal_pairs(seed, l, F, c, nup, mpair):
    """The two normals of counter (mpair, l*2F + c*F + nup) - broadcasting over array arguments."""
    hi = (np.asarray(l, dtype=np.uint64) * np.uint64(2 * F) + np.asarray(c, dtype=np.uint64) * np.uint64(F)
          + np.asarray(nup, dtype=np.uint64)) & _MASK
    return boxmuller_counter(seed, mpair, hi)


def device_normals(seed, lmax, F):
    """The whole device stream in the reference's stream order (2 F nalm doubles)."""
    out = []
    for l in range(lmax + 1):
        lp1 = l + 1
        npair = (lp1 + 1) // 2
        blk = np.empty((2, F, 2 * npair))
        cc, nn, mm = np.meshgrid(np.arange(2), np.arange(F), np.arange(npair), indexing="ij")
        a, b = normal_pairs(seed, l, F, cc, nn, mm)
        blk[:, :, 0::2] = a
        blk[:, :, 1::2] = b
        out.append(blk[:, :, :lp1].reshape(-1))
    return np.concatenate(out)
