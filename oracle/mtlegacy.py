"""Oracle (test infrastructure only): numpy's LEGACY normal stream - what ``np.random.standard_normal`` draws when the
reference is called without a generator (cora/util/nputil.py:121-123 ``rng=None``; ``Sky3d.getsky()`` ->
``skysim.mkfullsky(cla, nside)``, cora/core/maps.py:235-237).

numpy is a third-party dependency of the reference; what is restated here is its published algorithm:

  bit generator  MT19937 (Matsumoto & Nishimura 1998; numpy/random/src/mt19937/mt19937.c): window of 624 32-bit words,
                     x[k + 624] = x[k + 397] ^ twist(x[k], x[k + 1]),  twist(u, v) = (y >> 1) ^ (y & 1 ? 0x9908b0df : 0),
                     y = (u & 0x80000000) | (v & 0x7fffffff);  output = tempering of the window word
  uniform        ``mt19937_next_double``: a = next32 >> 5, b = next32 >> 6, (a 2^26 + b) / 2^53
  normal         ``legacy_gauss`` (numpy/random/src/legacy/legacy-distributions.c): Marsaglia's polar method - pairs
                 x1, x2 = 2 u - 1 until 0 < r2 = x1^2 + x2^2 < 1 (acceptance pi / 4), f = sqrt(-2 log(r2) / r2), returns
                 f x2 and KEEPS f x1 for the next call (``has_gauss`` / ``gauss`` of the global state)

An attempt always consumes four 32-bit words, so - unlike the ziggurat - the position of every attempt in the raw
stream is fixed and the device only has to compact the accepted ones; the sequential part is MT19937 itself, which
cora_amd/csrc/mtlegacy.hip cuts into segments with jump-ahead polynomials x^J mod phi(x) over GF(2) (phi = the minimal
polynomial of the recurrence, degree 19937; Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008): the window J words
on is  sum_i g_i W_i  with W_i the window i words on - the functions below compute phi (Berlekamp-Massey on the output
bits), the polynomials, and apply them exactly as the kernel does.

PINNED by numpy itself (tests/test_oracle.py): the sequential restatement against ``np.random.seed(s);
np.random.standard_normal(n)`` bit for bit (python's math.log / sqrt are the libm numpy calls), the state it leaves
against ``np.random.get_state()``, and the polynomial jump against plain stepping.
"""
import math

import numpy as np

N, M = 624, 397
UPPER, LOWER, MATRIX_A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF
DEG = 19937


def next_word(u, v, w):
    """x[k + 624] from u = x[k], v = x[k + 1], w = x[k + 397]."""
    y = (u & UPPER) | (v & LOWER)
    return w ^ (y >> 1) ^ (MATRIX_A if y & 1 else 0)


def temper(y):
    y ^= y >> 11
    y ^= (y << 7) & 0x9D2C5680
    y ^= (y << 15) & 0xEFC60000
    y ^= y >> 18
    return y & 0xFFFFFFFF


def extend(window, count):
    """The window's 624 words followed by the next ``count`` words of the recurrence (untempered)."""
    x = list(int(v) for v in window)
    for k in range(count):
        x.append(next_word(x[k], x[k + 1], x[k + M]))
    return x


class LegacyStream:
    """numpy's global legacy state as a python object: key (624 words), pos, has_gauss, gauss."""

    def __init__(self, state=None):
        st = np.random.get_state(legacy=False) if state is None else state
        self.key = [int(v) for v in st["state"]["key"]]
        self.pos = int(st["state"]["pos"])
        self.has_gauss = int(st["has_gauss"])
        self.gauss = float(st["gauss"])

    def _gen(self):                      # mt19937_gen: the next 624 words, in place
        x = extend(self.key, N)
        self.key = x[N:]
        self.pos = 0

    def next32(self):
        if self.pos == N:
            self._gen()
        y = self.key[self.pos]
        self.pos += 1
        return temper(y)

    def next_double(self):
        a, b = self.next32() >> 5, self.next32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    def gauss_next(self):
        if self.has_gauss:
            t = self.gauss
            self.has_gauss, self.gauss = 0, 0.0
            return t
        while True:
            x1 = 2.0 * self.next_double() - 1.0
            x2 = 2.0 * self.next_double() - 1.0
            r2 = x1 * x1 + x2 * x2
            if not (r2 >= 1.0 or r2 == 0.0):
                break
        f = math.sqrt(-2.0 * math.log(r2) / r2)
        self.gauss, self.has_gauss = f * x1, 1
        return f * x2

    def standard_normal(self, n):
        return np.array([self.gauss_next() for _ in range(n)])

    def state(self):
        return {"bit_generator": "MT19937", "state": {"key": np.array(self.key, dtype=np.uint32), "pos": self.pos},
                "has_gauss": self.has_gauss, "gauss": self.gauss}


# ---- GF(2) polynomials as python ints (bit i = coefficient of x^i) -------------------------------------------------
def berlekamp_massey(bits):
    """Minimal polynomial (connection polynomial, as an int) of a binary sequence (list of 0 / 1)."""
    n = len(bits)
    s = 0
    for i, b in enumerate(bits):
        s |= b << i
    C, B = 1, 1
    L, m = 0, 1
    for i in range(n):
        # discrepancy: sum_j C_j s_{i-j}, j = 0..L
        win = (s >> (i - L)) & ((1 << (L + 1)) - 1) if i >= L else s & ((1 << (i + 1)) - 1)
        # reverse alignment: C_j pairs with s_{i-j}: bit j of C with bit (L - j) of win (when i >= L)
        if i >= L:
            d = bin(_rev(C, L + 1) & win).count("1") & 1
        else:
            d = 0
            for j in range(min(L, i) + 1):
                d ^= ((C >> j) & 1) & ((s >> (i - j)) & 1)
        if d == 0:
            m += 1
        elif 2 * L <= i:
            T = C
            C ^= B << m
            L, B, m = i + 1 - L, T, 1
        else:
            C ^= B << m
            m += 1
    return C, L


def _rev(v, nbits):
    return int(bin(v)[2:].zfill(nbits)[::-1], 2)


def minimal_polynomial():
    """phi(x) of MT19937 as an int with bit i = coefficient of x^i (degree 19937): the recurrence of the output bits
    sum_i phi_i b[t + i] = 0.  From Berlekamp-Massey on 2 x 19937 + 64 bits of one output bit position."""
    rs = np.random.RandomState(4357)
    words = rs.randint(0, 2**32, size=2 * DEG + 200, dtype=np.uint64)          # raw 32-bit outputs
    bits = [int(w) & 1 for w in words]
    C, L = berlekamp_massey(bits)
    assert L == DEG, L
    # BM's connection polynomial: sum_j C_j s_{i-j} = 0; as a recurrence forward in time: phi_i = C_{L-i}
    return _rev(C, L + 1)


def polymod(a, phi, deg=DEG):
    """a mod phi."""
    while a.bit_length() > deg:
        a ^= phi << (a.bit_length() - 1 - deg)
    return a


def polysqr_mod(a, phi):
    # squaring over GF(2): spread the bits
    s = 0
    bs = bin(a)[2:]
    s = int("".join(c + "0" for c in bs)[:-1] if len(bs) else "0", 2)
    return polymod(s, phi)


def polymul_mod(a, b, phi):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return polymod(r, phi)


def x_pow_mod(e, phi):
    """x^e mod phi by square and multiply."""
    result, base = 1, 2
    while e:
        if e & 1:
            result = polymul_mod(result, base, phi)
        base = polysqr_mod(base, phi)
        e >>= 1
    return result


def jump_window(window, g):
    """The window J words on, from the polynomial g = x^J mod phi: sum over the set bits i of g of the window i words on
    (what the kernel does: extend the sequence by 19937 words, XOR the shifted windows)."""
    x = extend(window, DEG)
    out = [0] * N
    i = 0
    gg = g
    while gg:
        if gg & 1:
            for w in range(N):
                out[w] ^= x[i + w]
        gg >>= 1
        i += 1
    return out


def step_window(window, j):
    """The window j words on by plain stepping."""
    x = extend(window, j)
    return x[j:j + N]


# ---- glibc's log, FMA build -------------------------------------------------------------------------------------------
def _fma(a, b, c):
    """fma(a, b, c) exactly rounded (python 3.10 has no math.fma): rational arithmetic, one rounding."""
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def glibc_log_tables():
    """(ln2hi, ln2lo, A, B, T) parsed from the generated include file the kernels are compiled with
    (cora_amd/csrc/glibc_log_tab.inc, written by tools/gen_glibc_log_tab.py from the installed libm)."""
    import os
    import re

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cora_amd", "csrc", "glibc_log_tab.inc")
    txt = open(path).read()
    hx = r"-?0x[01]\.[0-9a-f]+p[+-]?\d+"
    ln2hi = float.fromhex(re.search(r"LN2HI (%s)" % hx, txt).group(1))
    ln2lo = float.fromhex(re.search(r"LN2LO (%s)" % hx, txt).group(1))
    A = [float.fromhex(t) for t in re.findall(hx, re.search(r"GLIBC_LOG_A \{(.*)\}", txt).group(1))]
    B = [float.fromhex(t) for t in re.findall(hx, re.search(r"GLIBC_LOG_B \{(.*)\}", txt).group(1))]
    T = [float.fromhex(t) for t in re.findall(hx, txt[txt.index("GLIBC_LOG_T"):])]
    assert len(A) == 5 and len(B) == 11 and len(T) == 256
    return ln2hi, ln2lo, A, B, T


_LOGTAB = None


def glibc_log_fma(x):
    """glibc's ``log`` (sysdeps/ieee754/dbl-64/e_log.c) for a positive normal double, in the evaluation order of its
    FMA build (``__log_fma``, what the loader selects on CPUs with FMA + AVX2 - read off the installed libm's code):
    every fused operation below is fused there, every separate one separate.  cora_amd/csrc/mtlegacy.hip runs the same
    sequence (``glibc_log_fma``); tests/test_oracle.py compares this with ``math.log`` on the host."""
    import struct

    global _LOGTAB
    if _LOGTAB is None:
        _LOGTAB = glibc_log_tables()
    ln2hi, ln2lo, A, B, T = _LOGTAB
    ix = struct.unpack("<Q", struct.pack("<d", x))[0]
    if (ix - 0x3FEE000000000000) & 0xFFFFFFFFFFFFFFFF <= 0x308FFFFFFFFFF:       # 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if x == 1.0:
            return 0.0
        r = x - 1.0
        p2 = _fma(r, B[2], B[1])
        p3 = _fma(r, B[5], B[4])
        r2 = r * r
        p5 = _fma(r, B[8], B[7])
        p2 = _fma(r2, B[3], p2)
        p3 = _fma(r2, B[6], p3)
        r3 = r * r2
        p1 = _fma(r2, B[9], p5)
        p1 = _fma(r3, B[10], p1)
        p1 = _fma(p1, r3, p3)
        p1 = _fma(p1, r3, p2)
        t = _fma(r, 134217728.0, r)
        rhi = _fma(-134217728.0, r, t)
        rhi2 = rhi * rhi
        rlo = r - rhi
        hi = _fma(rhi2, B[0], r)
        lo = _fma(rhi2, B[0], r - hi)
        lo = _fma(B[0] * rlo, r + rhi, lo)
        y = _fma(p1, r3, lo)
        return y + hi
    tmp = (ix - 0x3FE6000000000000) & 0xFFFFFFFFFFFFFFFF
    i = (tmp >> 45) & 127
    k = tmp >> 52
    if k >= 2048:
        k -= 4096
    iz = (ix - (tmp & 0xFFF0000000000000)) & 0xFFFFFFFFFFFFFFFF
    z = struct.unpack("<d", struct.pack("<Q", iz))[0]
    invc, logc = T[2 * i], T[2 * i + 1]
    kd = float(k)
    r = _fma(z, invc, -1.0)
    w = _fma(kd, ln2hi, logc)
    q = _fma(r, A[2], A[1])
    hi = r + w
    r2 = r * r
    lo = (w - hi) + r
    lo = _fma(kd, ln2lo, lo)
    r3 = r * r2
    p = _fma(r, A[4], A[3])
    lo = _fma(r2, A[0], lo)
    p = _fma(p, r2, q)
    y = _fma(r3, p, lo)
    return y + hi
