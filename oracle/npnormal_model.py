"""Oracle-side model (test infrastructure only) of the PARALLEL form of numpy's ziggurat stream that
cora_amd/csrc/npnormal.hip runs: the same decomposition restated in python so that its logic is checked on the CPU
against the sequential restatement (oracle/npnormal.py) and against numpy itself.

The sequential sampler consumes a data-dependent number of raw 64-bit draws per normal (1 on the fast path, 2 for a
wedge sample - accepted or not -, 1 + 2 i for a tail sample), so "which raw position starts a sample" is a prefix
problem.  The device uses R = 64 (one position per lane, the class masks are the compare results of a wave), T = 16 rows per
block = one wave, the rows of a block evaluated together in lanes 0..15 for both carries in and the carries resolved
with one add; this model keeps R and T free and resolves a block by fixed-point iteration - the same function.
Decomposition (R positions per chunk, T chunks per block):

  chunk    classification of every position AS IF it started a sample: nf (not fast), z (tail class: nf and
           idx == 0), wacc (wedge test of (p, p + 1) passes).  Position p of a chunk entered with `k` positions
           already consumed (by a sample that started earlier) starts a sample iff it is not the second draw of a
           wedge sample: inside a maximal run of nf positions the starts alternate from the run's first position - found
           for all positions at once with the carry trick of simdjson's odd-backslash scan.  A tail-class START is
           resolved by its owner reading on past the end of its chunk (2 draws per iteration), like the wedge draw of a
           chunk's last position; so the state at a chunk boundary is just k = positions of the next chunk(s) already
           consumed.
  block    k of thread t + 1 = k_out of thread t: fixed-point iteration from k = 0 (dependency chains are short: a
           fast position ends them).
  grid     pass 1 gives every block's (k_out, count) for block entry k in {0, 1}; a scan composes them; an entry k >= 2
           (a tail sample straddling a block boundary) re-evaluates that one block for its true k; pass 2 re-runs every
           block with its true (k, first ordinal) and writes the normals at their ordinals.
"""
import math

import numpy as np

from . import npnormal as seq

EVEN = 0x5555555555555555


class Stream:
    """Random access to the raw stream by position: raw(p) = output of the state after p + 1 steps."""

    def __init__(self, state, inc):
        self.s0, self.inc = state, inc
        self._cache = {}

    def raw(self, p):
        if p not in self._cache:
            blk = p - (p % 4096)
            s = seq.advance(self.s0, self.inc, blk)
            for q in range(blk, blk + 4096):
                s = seq.step(s, self.inc)
                self._cache[q] = seq.output(s)
        return self._cache[p]


def classify(r, ki):
    idx = r & 0xFF
    rabs = (r >> 9) & 0x000FFFFFFFFFFFFF
    return idx, rabs, (r >> 8) & 1, rabs < ki[idx]


def value_of(r, wi):
    idx = r & 0xFF
    rabs = (r >> 9) & 0x000FFFFFFFFFFFFF
    x = rabs * wi[idx]
    return -x if (r >> 8) & 1 else x


def wedge_accept(r0, r1, wi, fi):
    idx = r0 & 0xFF
    rabs = (r0 >> 9) & 0x000FFFFFFFFFFFFF
    x = rabs * wi[idx]
    u = (r1 >> 11) * (1.0 / 9007199254740992.0)
    return (fi[idx - 1] - fi[idx]) * u + fi[idx] < math.exp(-0.5 * x * x)


def tail_walk(stream, p):
    """Tail sample started at absolute position p: (value, positions consumed after p)."""
    r = stream.raw(p)
    rabs = (r >> 9) & 0x000FFFFFFFFFFFFF
    c = 0
    while True:
        u1 = (stream.raw(p + c + 1) >> 11) * (1.0 / 9007199254740992.0)
        u2 = (stream.raw(p + c + 2) >> 11) * (1.0 / 9007199254740992.0)
        c += 2
        xx = -seq.ZIG_INV_R * math.log1p(-u1)
        yy = -math.log1p(-u2)
        if yy + yy > xx * xx:
            break
    v = -(seq.ZIG_R + xx) if (rabs >> 8) & 1 else seq.ZIG_R + xx
    return v, c


class Chunk:
    def __init__(self, stream, a, R, tabs):
        ki, wi, fi = tabs
        self.a, self.R, self.stream, self.tabs = a, R, stream, tabs
        self.raw = [stream.raw(a + j) for j in range(R + 1)]
        self.nf = self.z = self.wacc = 0
        for p in range(R):
            idx, _rabs, _sg, fast = classify(self.raw[p], ki)
            if not fast:
                self.nf |= 1 << p
                if idx == 0:
                    self.z |= 1 << p
                elif wedge_accept(self.raw[p], self.raw[p + 1], wi, fi):
                    self.wacc |= 1 << p

    def eval(self, k, emit=None, base=0):
        """(k_out, count) for entry k; with ``emit`` (dict ordinal -> (value, end position)) also the values."""
        R = self.R
        full = (1 << R) - 1
        cnt = 0
        while True:
            if k >= R:
                return k - R, cnt
            low = (1 << k) - 1
            nf = self.nf & ~low
            starts = nf & ~(nf << 1)
            re = nf & ~(nf + (starts & EVEN))
            ro = nf & ~(nf + (starts & ~EVEN))
            sn = ((re & EVEN) | (ro & ~EVEN)) & full          # starts that are not fast
            S = ~(sn << 1)
            valid = full & ~low
            T = S & self.z & valid
            upto = valid if not T else valid & ((1 << (T & -T).bit_length() - 1) - 1)
            e_fast = S & ~nf & upto
            e_wedge = sn & ~self.z & self.wacc & upto
            if emit is not None:
                e = e_fast | e_wedge
                for p in range(R):
                    if (e >> p) & 1:
                        o = base + cnt + bin(e & ((1 << p) - 1)).count("1")
                        emit[o] = (value_of(self.raw[p], self.tabs[1]), self.a + p + (2 if (e_wedge >> p) & 1 else 1))
            cnt += bin(e_fast | e_wedge).count("1")
            if not T:
                return (sn >> (R - 1)) & 1, cnt
            p = (T & -T).bit_length() - 1
            v, c = tail_walk(self.stream, self.a + p)
            if emit is not None:
                emit[base + cnt] = (v, self.a + p + 1 + c)
            cnt += 1
            k = p + 1 + c


def block_resolve(chunks, k_in, emit=None, base=0):
    """Fixed point of the thread entries of one block; (k_out, count)."""
    T = len(chunks)
    kin = [0] * T
    kin[0] = k_in
    iters = 0
    while True:
        iters += 1
        res = [c.eval(k) for c, k in zip(chunks, kin)]
        new = [k_in] + [r[0] for r in res[:-1]]
        if new == kin:
            break
        kin = new
    if emit is not None:
        o = base
        for c, k, r in zip(chunks, kin, res):
            c.eval(k, emit, o)
            o += r[1]
    return res[-1][0], sum(r[1] for r in res), iters


def parallel_normals(state, inc, n, R=16, T=8, margin=1.0225, stats=None):
    """The first n normals of the stream and the raw draws they consume, by the block / scan / emit decomposition."""
    tabs = seq.tables()
    stream = Stream(state, inc)
    BLK = R * T
    out = np.empty(n)
    pos0, ord0 = 0, 0
    n_raw = None
    rounds = repairs = 0
    while ord0 < n:
        rounds += 1
        want = n - ord0
        nblk = (int(want * margin) + 64 + BLK - 1) // BLK
        blocks = [[Chunk(stream, pos0 + b * BLK + t * R, R, tabs) for t in range(T)] for b in range(nblk)]
        # pass 1: block functions on the domain {0, 1}
        fun = [[block_resolve(blk, e)[:2] for e in (0, 1)] for blk in blocks]
        # scan with repair
        entry = []
        k, o = 0, ord0
        for b in range(nblk):
            entry.append((k, o))
            if k >= 2:
                repairs += 1
                ko, c = block_resolve(blocks[b], k)[:2]
            else:
                ko, c = fun[b][k]
            k, o = ko, o + c
        # pass 2: emit
        emit = {}
        for b in range(nblk):
            block_resolve(blocks[b], entry[b][0], emit, entry[b][1])
        for o_, (v, end) in emit.items():
            if o_ < n:
                out[o_] = v
                if o_ == n - 1:
                    n_raw = end
        pos0, ord0 = pos0 + nblk * BLK + k, o
    if stats is not None:
        stats.update(rounds=rounds, repairs=repairs)
    return out, n_raw
