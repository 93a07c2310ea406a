#!/usr/bin/env python3
"""Lane-group simulation of ds_read_b128 / ds_write_b128 bank conflicts (MI355X_MICROARCH.md, LDS table) for the
FFT pass patterns of sht_ringfft.hip (k5) and flatsky.hip (r4), and a hill-climb over XOR-linear slot swizzles.
    python tools/lds_bank_sim.py k5|r4 [restarts]
Cost units: read cycles + write cycles relative to conflict-free (2.0 = perfect for a read+write pass)."""
import numpy as np, itertools, sys
RG = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
RG += [[x+32 for x in g] for g in RG]
RG = np.array(RG)                      # [4,16]
WG = np.arange(64).reshape(8,8)        # [8,8]

def k5_passes(N, R0=16):
    """(name, idx[ninstr,64]) for DIF schedule of length N: radix 16 while fits then remainder; same index sets for DIT."""
    out = []
    Ls = N
    sched = []
    while Ls >= 16:
        sched.append((Ls, 16)); Ls >>= 4
    if Ls > 1: sched.append((Ls, Ls))
    for (Ls, R) in sched:
        q = Ls // R; nb = N // R
        t = np.arange(nb)
        b = t // q; j = t % q
        i0 = b * Ls + j
        rows = []
        for w in range(0, max(1, nb // 64)):
            tt = i0[64*w:64*w+64]
            if len(tt) < 64: tt = np.pad(tt, (0, 64-len(tt)), mode='edge')
            for r in range(R):
                rows.append(tt + r*q)
        out.append(("N%d_Ls%d_R%d" % (N, Ls, R), np.array(rows)))
    return out

def r4_passes(P):
    out = []
    logP = P.bit_length()-1
    lanes = np.arange(64)
    def brev(i): return int(format(i, '0%db'%logP)[::-1], 2)
    out.append(("P%d_commit" % P, np.array([[brev(64*w + j) for j in range(64)] for w in range(P//64)]), 'w'))
    h = 1
    if logP & 1:
        out.append(("P%d_r2" % P, np.array([[2*(64*w+l) + r for l in range(64)] for w in range(P//128) for r in range(2)])))
        h = 2
    q = P >> 2
    while h < P:
        rows = []
        for w in range(q // 64):
            j = 64*w + lanes
            pos = j & (h-1); grp = j // h
            for r in range(4):
                rows.append(grp*4*h + pos + r*h)
        out.append(("P%d_h%d" % (P, h), np.array(rows)))
        h <<= 2
    return out

def cost_of(pos, groups):
    # pos [ninstr, 64] element positions; returns total cycles / ideal
    quad = pos[:, groups] & 15                      # [ninstr, G, L]
    onehot = (quad[..., None] == np.arange(16)).sum(axis=2)   # [ninstr, G, 16]
    return onehot.max(axis=-1).sum() / (pos.shape[0] * groups.shape[0])

def make_swz(M):   # M: [4, nb] binary; out bit k ^= parity(M[k] & (i>>4))
    def f(i):
        x = i >> 4
        o = np.zeros_like(i)
        for k in range(4):
            m = int(sum(int(M[k][b]) << b for b in range(len(M[k]))))
            par = np.zeros_like(i)
            xm = x & m
            while True:
                par ^= xm & 1
                xm = xm >> 1
                if not xm.any(): break
            o |= par << k
        return i ^ o
    return f

def total(patterns, f, verbose=False):
    tot = 0
    for p in patterns:
        name, idx = p[0], p[1]
        kind = p[2] if len(p) > 2 else 'rw'
        pos = f(idx)
        c = 0
        if 'r' in kind: c += cost_of(pos, RG)
        if 'w' in kind: c += cost_of(pos, WG)
        if verbose: print("   ", name, round(c, 2))
        tot += c
    return tot

if __name__ == "__main__":
    which = sys.argv[1]
    if which == "k5":
        pats = []
        for N in (4096, 2048, 1024, 512, 256): pats += k5_passes(N)
        nb = 8
    else:
        pats = []
        for P in (256, 512, 1024, 2048, 4096): pats += r4_passes(P)
        nb = 8
    ident = lambda i: i
    fpad = lambda i: i + (i >> 3) + ((i >> 7) << 3)
    print("none", total(pats, ident, True))
    print("fpad", total(pats, fpad, True))
    rng = np.random.default_rng(0)
    best = None
    for restart in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        M = (rng.random((4, nb)) < 0.25).astype(int)
        cur = total(pats, make_swz(M))
        improved = True
        while improved:
            improved = False
            for k in range(4):
                for b in range(nb):
                    M[k][b] ^= 1
                    c = total(pats, make_swz(M))
                    if c < cur - 1e-9:
                        cur = c; improved = True
                    else:
                        M[k][b] ^= 1
        print("restart", restart, cur, M.tolist(), flush=True)
        if best is None or cur < best[0]: best = (cur, M.copy())
    print("BEST", best[0], best[1].tolist())
    total(pats, make_swz(best[1]), True)
