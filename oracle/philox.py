"""Oracle (test infrastructure only): numpy restatement of the DEVICE normal stream.

The reference draws its normals from numpy's global/Generator state
(cora/util/nputil.py:104-125, called per l from cora/core/skysim.py:120); that stream is
reproduced bit-for-bit by the host path of the build (parity mode).  The throughput mode
replaces it by a counter-based stream so that the draw can stay on the GPU and be identical
for any number of GPUs ("device counter-based RNG keyed by (seed, l, m, nu, re/im)").
This file is the specification of that stream, checked against the kernels in tests/:

  Philox4x32-10 (Salmon et al., SC'11; known-answer vectors of Random123 in tests/test_oracle.py)
  counter = (m, l * F + nu', 0, 0),  key = (seed & 0xffffffff, seed >> 32)
  k = r0 << 20 | r1 >> 12                       (52 bits)   u1 = (k + 1/2) 2^-52        (exact in a double)
  j = r2 >> 24,  w = (r2 & 0xffffff) << 28 | r3 >> 4   (8 + 52 bits)   theta = 2 pi (j + (w + 1/2) 2^-52) / 256
  Re-normal(l, nu', m) = sqrt(-2 ln u1) cos(theta),  Im-normal(l, nu', m) = sqrt(-2 ln u1) sin(theta)
(the uniforms are defined on their bits so that the kernels build them without integer -> double conversions:
cora_amd/csrc/rng_dev.h; mathematically evaluated here, in extended precision for the angle)

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
    r0, r1, r2, r3 = (v.astype(np.uint64) for v in (r0, r1, r2, r3))
    k = (r0 << np.uint64(20)) | (r1 >> np.uint64(12))
    u1 = (k.astype(np.float64) + 0.5) * 2.0**-52                  # exact: 2k + 1 < 2^53
    rad = np.sqrt(-2.0 * np.log1p(-(1.0 - u1)))                   # (1 - u1 exact; log1p keeps u1 -> 1 accurate)
    j = r2 >> np.uint64(24)
    w = ((r2 & np.uint64(0xFFFFFF)) << np.uint64(28)) | (r3 >> np.uint64(4))
    # sector-exact reduction: theta = centre of sector j + x, x = ((w + 1/2) 2^-52 - 1/2) 2 pi / 256 (the offset is
    # exact in a double; cos / sin of the centre and of x in long double, combined there, rounded once at the end)
    ld = np.longdouble
    two_pi = 2 * ld(np.pi) + ld(2.4492935982947064e-16)           # 2 pi to ~1e-32 (pi = double(pi) + 1.2246e-16)
    tw = (w.astype(np.float64) * 2.0**-52 - 0.5) + 2.0**-53       # exact
    x = tw.astype(ld) * (two_pi / 256)
    # octant-exact evaluation of the centre angle (a multiple of pi/256: reduce the integer, not the float)
    jj = (2 * j.astype(np.int64) + 1)                             # centre = jj * pi / 256, jj odd in [1, 511]
    q = jj // 128                                                 # quadrant of the centre (0..3), remainder < pi/2
    rem = (jj - 128 * q).astype(ld) * (two_pi / 512)
    cr, sr = np.cos(rem), np.sin(rem)
    c0 = np.where(q == 0, cr, np.where(q == 1, -sr, np.where(q == 2, -cr, sr)))
    s0 = np.where(q == 0, sr, np.where(q == 1, cr, np.where(q == 2, -sr, -cr)))
    cx, sx = np.cos(x), np.sin(x)
    cs = (c0 * cx - s0 * sx).astype(np.float64)
    sn = (s0 * cx + c0 * sx).astype(np.float64)
    return rad * cs, rad * sn


def normal_pairs(seed, l, F, nup, m):
    """(real, imaginary) normal of (l, nu', m): counter (m, l*F + nup) - broadcasting over array arguments."""
    hi = (np.asarray(l, dtype=np.uint64) * np.uint64(F) + np.asarray(nup, dtype=np.uint64)) & _MASK
    return boxmuller_counter(seed, m, hi)


def device_normals(seed, lmax, F):
    """The whole device stream in the reference's stream order (2 F nalm doubles)."""
    out = []
    for l in range(lmax + 1):
        nn, mm = np.meshgrid(np.arange(F), np.arange(l + 1), indexing="ij")
        re, im = normal_pairs(seed, l, F, nn, mm)
        out.append(re.reshape(-1))
        out.append(im.reshape(-1))
    return np.concatenate(out)
