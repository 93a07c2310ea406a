"""numpy model of the compile-time ring-FFT pipeline of csrc/sht_ringfft_ct.hip (design check, CPU only).

It replays, on a flat "LDS" array, exactly the passes the kernels make - positions, twiddles, the Hermitian step
fused into the first pass of the direct class, the pruned first / last pass and the register-fused middle of the
Bluestein class, the digit-reversed store of the direct class - and compares the pixels with a plain DFT of the
folded ring spectrum.  Run:  python tools/k5_model.py
"""
import numpy as np


def sched(N):
    """DIF radices, largest stride first (3-smooth lengths start with the radix holding the factor 3)."""
    s = []
    n = N
    if n % 3 == 0:
        s.append(12 if n % 12 == 0 and n // 12 >= 16 else 3)
        n //= s[0]
    while n >= 16:
        s.append(16)
        n //= 16
    if n > 1:
        s.append(n)
    assert np.prod(s) == N
    return s


def dif_pass(buf, N, Ls, R, sign, tw=True):
    Q = Ls // R
    out = buf.copy()
    for b in range(N // Ls):
        for j in range(Q):
            idx = b * Ls + j + Q * np.arange(R)
            x = buf[idx]
            y = np.array([(x * np.exp(sign * 2j * np.pi * np.arange(R) * p / R)).sum() for p in range(R)])
            if tw:
                y = y * np.exp(sign * 2j * np.pi * j * np.arange(R) / Ls)
            out[idx] = y
    return out


def dit_pass(buf, N, Ls, R, sign):
    Q = Ls // R
    out = buf.copy()
    for b in range(N // Ls):
        for j in range(Q):
            idx = b * Ls + j + Q * np.arange(R)
            x = buf[idx] * np.exp(sign * 2j * np.pi * j * np.arange(R) / Ls)
            out[idx] = np.array([(x * np.exp(sign * 2j * np.pi * np.arange(R) * p / R)).sum() for p in range(R)])
    return out


def dif_pos(k, N):
    """storage position of frequency k after the DIF passes of sched(N)"""
    pos, length = 0, N
    for R in sched(N):
        length //= R
        pos += (k % R) * length
        k //= R
    return pos


def fft_dif(buf, N, sign):
    Ls = N
    for R in sched(N):
        buf = dif_pass(buf, N, Ls, R, sign)
        Ls //= R
    return buf


def fft_dit(buf, N, sign):
    rad = sched(N)[::-1]
    Ls = 1
    for R in rad:
        Ls *= R
        buf = dit_pass(buf, N, Ls, R, sign)
    return buf


def ring_reference(X, n):
    """T_j, j < n, from the folded Hermitian half spectrum X[0..h] (X_0, X_h real): the oracle's irfft."""
    return np.fft.irfft(X, n=n) * n


def direct_ring(X, h):
    n = 2 * h
    N = h
    rad = sched(N)
    # pass 1 with the Hermitian step fused: butterfly j reads X[j + r Q] and the partners X[h - (j + r Q)]
    Q = N // rad[0]
    buf = np.zeros(N, complex)
    for j in range(Q):
        k = j + Q * np.arange(rad[0])
        xa, xb = X[k], X[h - k]
        w = np.exp(2j * np.pi * k / n)        # = w^j * (32nd roots of unity) on the device
        Z = (xa + np.conj(xb)) + 1j * w * (xa - np.conj(xb))
        y = np.array([(Z * np.exp(2j * np.pi * np.arange(rad[0]) * p / rad[0])).sum() for p in range(rad[0])])
        buf[k] = y * np.exp(2j * np.pi * j * np.arange(rad[0]) / N)
    Ls = N // rad[0]
    for R in rad[1:-1]:
        buf = dif_pass(buf, N, Ls, R, +1)
        Ls //= R
    # last pass: butterfly t covers positions t RL + r; natural index = rev(t) + (N / RL) r, stored straight to HBM
    RL = rad[-1]
    assert Ls == RL
    out = np.zeros(N, complex)
    for t in range(N // RL):
        x = buf[t * RL + np.arange(RL)]
        y = np.array([(x * np.exp(2j * np.pi * np.arange(RL) * p / RL)).sum() for p in range(RL)])
        # t = k0 * R1 + k1 (two leading digits), natural k = k0 + R0 k1 + R0 R1 r
        if len(rad) == 3:
            k0, k1 = divmod(t, rad[1])
            nat = k0 + rad[0] * k1 + rad[0] * rad[1] * np.arange(RL)
        else:
            nat = t + rad[0] * np.arange(RL)
        out[nat] = y
        for r in range(RL):
            assert dif_pos(nat[r], N) == t * RL + r
    T = np.empty(n)
    T[0::2], T[1::2] = out.real, out.imag
    return T


def bluestein_ring(X, h, P):
    n = 2 * h
    k = np.arange(h)
    b = np.exp(1j * np.pi * ((k * k) % (2 * h)) / h)
    b2 = 1j * np.exp(1j * np.pi * ((k * k + k) % (2 * h)) / h)           # i w^k b_k
    # plan-time filter: DIF transform of the wrapped conjugate chirp, kept in DIF storage order
    f = np.zeros(P, complex)
    f[:h] = np.conj(b)
    f[P - h + 1:] = np.conj(b[1:][::-1])
    filt = fft_dif(f, P, -1)
    # Hermitian + chirp pre-pass (pairs k, h-k), zeros up to P/2
    y = np.zeros(P, complex)
    xa = X[:h]
    xb = X[h - k]
    y[:h] = b * (xa + np.conj(xb)) + b2 * (xa - np.conj(xb))
    rad = sched(P)
    # forward pass 1, pruned: inputs r >= R/2 are zero and not read
    R0 = rad[0]
    Q = P // R0
    assert h <= Q * (R0 // 2)
    buf = np.zeros(P, complex)
    for j in range(Q):
        idx = j + Q * np.arange(R0)
        x = np.where(np.arange(R0) < R0 // 2, y[idx], 0.0)
        yy = np.array([(x * np.exp(-2j * np.pi * np.arange(R0) * p / R0)).sum() for p in range(R0)])
        buf[idx] = yy * np.exp(-2j * np.pi * j * np.arange(R0) / P)
    Ls = P // R0
    for R in rad[1:-1]:
        buf = dif_pass(buf, P, Ls, R, -1)
        Ls //= R
    # middle, fused in registers: last forward pass (no twiddles), filter, first inverse pass
    RL = rad[-1]
    for t in range(P // RL):
        idx = t * RL + np.arange(RL)
        x = buf[idx]
        yy = np.array([(x * np.exp(-2j * np.pi * np.arange(RL) * p / RL)).sum() for p in range(RL)]) * filt[idx]
        buf[idx] = np.array([(yy * np.exp(2j * np.pi * np.arange(RL) * p / RL)).sum() for p in range(RL)])
    # inverse DIT passes after the first
    Ls = RL
    for R in rad[::-1][1:-1]:
        Ls *= R
        buf = dit_pass(buf, P, Ls, R, +1)
    # last inverse pass, pruned: only outputs j + r Q < h are formed, times b_j / P, stored
    out = np.zeros(h, complex)
    for j in range(Q):
        idx = j + Q * np.arange(R0)
        x = buf[idx] * np.exp(2j * np.pi * j * np.arange(R0) / P)
        for p in range(R0 // 2):
            jj = j + p * Q
            if jj < h:
                out[jj] = (x * np.exp(2j * np.pi * np.arange(R0) * p / R0)).sum() * b[jj] / P
    T = np.empty(n)
    T[0::2], T[1::2] = out.real, out.imag
    return T


def main():
    rng = np.random.default_rng(1)
    for N in (64, 1024, 2048, 4096):
        h = N
        X = rng.standard_normal(h + 1) + 1j * rng.standard_normal(h + 1)
        X[0] = X[0].real
        X[h] = X[h].real
        ref = ring_reference(X, 2 * h)
        got = direct_ring(X, h)
        print("direct  h=%5d sched=%s  err %.2e" % (h, sched(N), np.abs(got - ref).max() / np.abs(ref).max()))
    for h, P in ((2046, 4096), (1026, 4096), (1022, 2048), (514, 2048), (300, 1024), (1536, 3072), (1100, 3072), (700, 1536)):
        X = rng.standard_normal(h + 1) + 1j * rng.standard_normal(h + 1)
        X[0] = X[0].real
        X[h] = X[h].real
        ref = ring_reference(X, 2 * h)
        got = bluestein_ring(X, h, P)
        print("bluest. h=%5d P=%5d sched=%s  err %.2e" % (h, P, sched(P), np.abs(got - ref).max() / np.abs(ref).max()))


if __name__ == "__main__":
    main()
