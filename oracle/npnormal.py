"""Oracle (test infrastructure only): restatement of numpy's seeded normal stream, ``Generator(PCG64).standard_normal``.

The reference draws every normal of ``mkfullsky`` from the caller's ``numpy.random.Generator``
(cora/util/nputil.py:121-125, called per l from cora/core/skysim.py:120; built by ``default_rng(seed)`` in
cora/signal/lss.py:449-450).  numpy is a third-party dependency of the reference (absent from /root/reference; numpy
2.2.6 in this image, ``numpy>=1.24`` in pyproject.toml) - what is restated here is its PUBLISHED algorithm:

  bit generator  PCG64 = pcg_setseq_128_xsl_rr_64 (O'Neill, "PCG: A Family of Simple Fast Space-Efficient
                 Statistically Good Algorithms for Random Number Generation", 2014; numpy/random/src/pcg64/pcg64.h):
                     state <- state * M + inc  (mod 2^128),  M = 0x2360ED051FC65DA44385DF649FCCF645
                     output = rotr64(hi ^ lo, hi >> 58)       (step first, then output)
  sampler        the 256-strip ziggurat of numpy/random/src/distributions/distributions.c
                 (``random_standard_normal``; Marsaglia & Tsang 2000 in Doornik's ZIGNOR form), tables ki / wi / fi
                 (cora_amd/csrc/zig_tab.inc, read out of numpy's own compiled library by tools/gen_zig_tabs.py):
                     r = next64; idx = r & 0xff; r >>= 8; sign = r & 1; rabs = (r >> 1) & (2^52 - 1)
                     x = rabs * wi[idx] (negated if sign);  rabs < ki[idx] -> return x            (99.3 %)
                     idx == 0: tail  loop  xx = -log1p(-U)/R, yy = -log1p(-U);  yy + yy > xx xx -> +-(R + xx),
                               sign bit = (rabs >> 8) & 1
                     else wedge: (fi[idx-1] - fi[idx]) U + fi[idx] < exp(-x x / 2) -> return x, else start again
                     U = (next64 >> 11) 2^-53

PINNED by numpy itself, which is present here and on the GPU box: tests/test_oracle.py compares this restatement with
``np.random.Generator(np.random.PCG64(seed)).standard_normal`` value by value (wedge and tail samples included), the
consumed raw count with the generator's own state afterwards, and ``advance`` with ``bit_generator.advance``.  The device
stream (cora_amd/csrc/npnormal.hip) is compared with numpy directly and with the counts of this file.
"""
import math
import os
import re

import numpy as np

PCG_MULT = (2549297995355413924 << 64) + 4865540595714422341
MASK128 = (1 << 128) - 1
MASK64 = (1 << 64) - 1
ZIG_R = 3.6541528853610087963519472518
ZIG_INV_R = 0.27366123732975827203338247596


def glibc_log1p(x):
    """glibc's ``log1p`` (sysdeps/ieee754/dbl-64/s_log1p.c: fdlibm's algorithm with glibc's split evaluation of the
    polynomial) for -1 < x < 0.41, operation by operation in IEEE double - python floats do not contract.  This is
    the restatement cora_amd/csrc/npnormal.hip runs for the tail samples (``glibc_log1p_neg``); tests/test_oracle.py
    compares it with ``math.log1p`` (libm) bit for bit."""
    import struct

    def hi(v):
        h = struct.unpack("<q", struct.pack("<d", v))[0] >> 32
        return h

    def sethi(v, h):
        lo = struct.unpack("<Q", struct.pack("<d", v))[0] & 0xFFFFFFFF
        return struct.unpack("<d", struct.pack("<Q", ((h & 0xFFFFFFFF) << 32) | lo))[0]

    ln2_hi, ln2_lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    Lp = [0.0, 6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
          1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01]
    hx = hi(x)
    ax = hx & 0x7FFFFFFF
    k, c, f, hu = 1, 0.0, 0.0, 0
    if hx < 0x3FDA827A:
        if ax < 0x3E200000:
            if ax < 0x3C900000:
                return x
            return x - x * x * 0.5
        if hx > 0 or hx <= (0xBFD2BEC3 - 2**32):
            k, f, hu = 0, x, 1
    if k != 0:
        u = 1.0 + x
        hu = hi(u)
        k = (hu >> 20) - 1023
        c = (1.0 - (u - x)) if k > 0 else (x - (u - 1.0))
        c /= u
        hu &= 0x000FFFFF
        if hu < 0x6A09E:
            u = sethi(u, hu | 0x3FF00000)
        else:
            k += 1
            u = sethi(u, hu | 0x3FE00000)
            hu = (0x00100000 - hu) >> 2
        f = u - 1.0
    hfsq = 0.5 * f * f
    if hu == 0:
        if f == 0.0:
            if k == 0:
                return 0.0
            c += k * ln2_lo
            return k * ln2_hi + c
        R = hfsq * (1.0 - 0.66666666666666666 * f)
        if k == 0:
            return f - R
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f)
    s = f / (2.0 + f)
    z = s * s
    R1 = z * Lp[1]
    z2 = z * z
    R2 = Lp[2] + z * Lp[3]
    z4 = z2 * z2
    R3 = Lp[4] + z * Lp[5]
    z6 = z4 * z2
    R4 = Lp[6] + z * Lp[7]
    R = R1 + z2 * R2 + z4 * R3 + z6 * R4
    if k == 0:
        return f - (hfsq - s * (hfsq + R))
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f)


# ---- glibc's exp, FMA build (the wedge test of the ziggurat) -----------------------------------------------------------
def _fma(a, b, c):
    """fma(a, b, c) exactly rounded (python 3.10 has no math.fma): rational arithmetic, one rounding."""
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


_EXPTAB = None


def glibc_exp_tables():
    """(InvLn2N, Shift, NegLn2hiN, NegLn2loN, C[4], T[256] as ints) parsed from the include file the kernels are
    compiled with (cora_amd/csrc/glibc_exp_tab.inc, written by tools/gen_glibc_exp_tab.py from the installed libm)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cora_amd", "csrc", "glibc_exp_tab.inc")
    txt = open(path).read()
    hx = r"-?0x[01]\.[0-9a-f]+p[+-]?\d+"
    one = lambda name: float.fromhex(re.search(r"%s (%s)" % (name, hx), txt).group(1))
    C = [float.fromhex(t) for t in re.findall(hx, re.search(r"GLIBC_EXP_C \{(.*)\}", txt).group(1))]
    T = [int(t, 16) for t in re.findall(r"0x([0-9a-f]{16})ull", txt[txt.index("GLIBC_EXP_T"):])]
    assert len(C) == 4 and len(T) == 256
    return one("GLIBC_EXP_INVLN2N"), one("GLIBC_EXP_SHIFT"), one("GLIBC_EXP_NEGLN2HIN"), one("GLIBC_EXP_NEGLN2LON"), C, T


def glibc_exp_fma(x):
    """glibc's ``exp`` (sysdeps/ieee754/dbl-64/e_exp.c) for 2^-54 <= |x| < 512 - and the ``1 + x`` it returns below
    that -, in the evaluation order of its FMA build (``__exp_fma``, what the loader selects on CPUs with FMA + AVX2 -
    read off the installed libm's code): every fused operation below is fused there, every separate one separate.
    cora_amd/csrc/npnormal.hip runs the same sequence (``glibc_exp_fma``) in the wedge test of numpy's ziggurat;
    tests/test_oracle.py compares this with ``math.exp`` on the host."""
    import struct

    global _EXPTAB
    if _EXPTAB is None:
        _EXPTAB = glibc_exp_tables()
    invln2n, shift, neghi, neglo, C, T = _EXPTAB
    bits = struct.unpack("<Q", struct.pack("<d", x))[0]
    abstop = (bits >> 52) & 0x7FF
    if abstop < 0x3C9:                                  # |x| < 2^-54
        return 1.0 + x
    assert abstop < 0x408, "restated for |x| < 512 only"
    kd = _fma(x, invln2n, shift)                        # z + Shift in one rounding
    ki = struct.unpack("<Q", struct.pack("<d", kd))[0]
    kd = kd - shift
    r = _fma(kd, neglo, _fma(kd, neghi, x))
    idx = 2 * (ki % 128)
    top = (ki << 45) & 0xFFFFFFFFFFFFFFFF
    tail = struct.unpack("<d", struct.pack("<Q", T[idx]))[0]
    sbits = (T[idx + 1] + top) & 0xFFFFFFFFFFFFFFFF
    p23 = _fma(r, C[1], C[0])
    tr = r + tail
    r2 = r * r
    p45 = _fma(r, C[3], C[2])
    t1 = _fma(p23, r2, tr)
    r4 = r2 * r2
    tmp = _fma(r4, p45, t1)
    scale = struct.unpack("<d", struct.pack("<Q", sbits))[0]
    return _fma(scale, tmp, scale)


def tables():
    """(ki, wi, fi) parsed from the generated include file the kernels are compiled with."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cora_amd", "csrc", "zig_tab.inc")
    txt = open(path).read()
    ki = [int(t, 16) for t in re.findall(r"0x([0-9a-f]{16})ull", txt)]
    fl = [float.fromhex(t) for t in re.findall(r"(-?0x[01]\.[0-9a-f]+p[+-]?\d+),", txt)]
    assert len(ki) == 256 and len(fl) == 512
    return ki, fl[:256], fl[256:]


def state_of(rng):
    """(state, inc) of a numpy Generator / BitGenerator built on PCG64."""
    bg = getattr(rng, "bit_generator", rng)
    st = bg.state
    assert st["bit_generator"] == "PCG64"
    return int(st["state"]["state"]), int(st["state"]["inc"])


def step(state, inc):
    return (state * PCG_MULT + inc) & MASK128


def output(state):
    hi, lo = state >> 64, state & MASK64
    x = hi ^ lo
    rot = hi >> 58
    return ((x >> rot) | (x << ((64 - rot) & 63))) & MASK64 if rot else x


def advance(state, inc, delta):
    """state after ``delta`` steps: f^n(s) = M^n s + inc (M^n - 1)/(M - 1), by square-and-multiply on (mult, plus)
    (Brown, "Random number generation with arbitrary strides"; numpy's pcg_advance_lcg_128)."""
    acc_mult, acc_plus = 1, 0
    cur_mult, cur_plus = PCG_MULT, inc
    while delta > 0:
        if delta & 1:
            acc_mult = (acc_mult * cur_mult) & MASK128
            acc_plus = (acc_plus * cur_mult + cur_plus) & MASK128
        cur_plus = ((cur_mult + 1) * cur_plus) & MASK128
        cur_mult = (cur_mult * cur_mult) & MASK128
        delta >>= 1
    return (acc_mult * state + acc_plus) & MASK128


def raw_stream(state, inc, n):
    """The next n 64-bit outputs as a uint64 array (python-int loop: small n only) and the state after them."""
    out = np.empty(n, dtype=np.uint64)
    for i in range(n):
        state = step(state, inc)
        out[i] = output(state)
    return out, state


def standard_normal(state, inc, n, stats=None):
    """The next n values of ``standard_normal`` and the number of raw 64-bit draws they consume."""
    ki, wi, fi = tables()
    vals = np.empty(n, dtype=np.float64)
    nraw = 0

    def nxt():
        nonlocal state, nraw
        state = step(state, inc)
        nraw += 1
        return output(state)

    def nxt_double():
        return (nxt() >> 11) * (1.0 / 9007199254740992.0)

    kinds = {"fast": 0, "wedge_accept": 0, "wedge_reject": 0, "tail": 0, "tail_reject": 0}
    for k in range(n):
        while True:
            r = nxt()
            idx = r & 0xFF
            r >>= 8
            sign = r & 1
            rabs = (r >> 1) & 0x000FFFFFFFFFFFFF
            x = rabs * wi[idx]
            if sign:
                x = -x
            if rabs < ki[idx]:
                kinds["fast"] += 1
                break
            if idx == 0:
                while True:
                    xx = -ZIG_INV_R * math.log1p(-nxt_double())
                    yy = -math.log1p(-nxt_double())
                    if yy + yy > xx * xx:
                        break
                    kinds["tail_reject"] += 1
                x = -(ZIG_R + xx) if (rabs >> 8) & 1 else ZIG_R + xx
                kinds["tail"] += 1
                break
            if (fi[idx - 1] - fi[idx]) * nxt_double() + fi[idx] < math.exp(-0.5 * x * x):
                kinds["wedge_accept"] += 1
                break
            kinds["wedge_reject"] += 1
        vals[k] = x
    if stats is not None:
        stats.update(kinds)
    return vals, nraw
