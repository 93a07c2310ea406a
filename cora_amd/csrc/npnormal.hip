// npnormal.hip - numpy's seeded normal stream, Generator(PCG64).standard_normal, generated on the device.
//
// The reference draws every normal of mkfullsky from the caller's numpy Generator (cora/util/nputil.py:121-125,
// called per l from cora/core/skysim.py:120; default_rng(seed) in cora/signal/lss.py:449-450): 1.07e9 values per
// cfg-3 realisation, 5-10 s of host time.  This file produces the SAME values on the GPU, bit for bit:
//
//   bit generator  PCG64 (pcg_setseq_128_xsl_rr_64): state <- state M + inc (mod 2^128), output rotr64(hi ^ lo, hi >> 58);
//                  a 128-bit LCG is position-addressable: f^n(s) = M^n s + inc (M^n - 1)/(M - 1)
//   sampler        numpy's 256-strip ziggurat (random_standard_normal of numpy/random/src/distributions/
//                  distributions.c, tables zig_tab.inc read out of numpy's own library by tools/gen_zig_tabs.py):
//                  one raw draw on the fast path (98.5 % of the positions), two for a wedge sample (accepted or
//                  not), 1 + 2 i for a tail sample.  Tail samples use glibc's log1p restated operation by operation
//                  (fdlibm's algorithm in glibc's evaluation order, no contraction), and the WEDGE test compares
//                  against glibc's exp restated the same way (glibc_exp_fma: the table-driven routine in the
//                  evaluation order of its FMA build, round 6; until then the device library's exp, within one ulp of
//                  it, could have flipped an accept at the threshold): bit-identical to numpy on glibc.
//
// "Which raw position starts a sample" is a prefix problem; oracle/npnormal_model.py states the decomposition in python
// (checked against numpy on the CPU), this is the same thing on the device:
//
//   row    64 consecutive positions, one per lane (lane p of row j of a block holds position 64 j + p; its LCG steps by
//          64: s <- M^64 s + inc G_64), classified AS IF each started a sample: nf (not fast), z (tail class), w (wedge
//          test of (p, p + 1) passes).  The compares that classify ARE the row's 64-bit masks (v_cmp writes a scalar
//          pair), so everything below is scalar code, once per wave.  With `pos` = the next position that starts a
//          sample, position p >= pos of the row starts one iff it is not the second draw of a wedge sample: inside a
//          maximal run of nf positions the starts alternate from the run's first position - found for all 64 at once
//          with an integer add (the carry runs through a run of ones; simdjson's odd-backslash scan).  A tail-class
//          start is walked where it is met (2 draws per iteration, wave-uniform), past the end of the block if need be,
//          like the wedge draw of a block's last position: the state at a block boundary is just k = positions of the
//          next block(s) already consumed.  The wedge tests (15 per block) are gathered in LDS and run 64 at a time.
//   block  16 rows = 1024 positions = one wave, rows in sequence (the scalar chain carries pos from row to row).
//   grid   zig_count_kernel: every block's (k_out, count) for block entry k in {0, 1} and the classes of its rows; a
//          block that hands k >= 2 to its successor (a tail sample across the boundary, 1.5e-4 of the blocks) evaluates
//          the successor for that k itself and appends the result to a patch list.  zig_tile / zig_top / zig_entry
//          compose the block functions (tiles of 512 blocks; inside a tile and across tiles by fixed-point iteration
//          from k = 0: almost every block maps 0 and 1 to the same k_out) and give every block its true
//          (k, first ordinal).  zig_emit_kernel re-runs the generator of every block with them and the stored classes:
//          the samples of a row leave as one store instruction at ordinal + mbcnt - consecutive addresses, no staging.
#include "common.h"

#include <algorithm>
#include <cstdlib>

#include "stream_internal.h"

#define ZIG_TAB_Q __device__ static const
#include "zig_tab.inc"

typedef unsigned __int128 u128;

#define ZIG_ROWS 16         // rows of 64 consecutive positions per block: lane p of row j holds position 64 j + p
#define ZIG_BLK (64 * ZIG_ROWS)
#define ZIG_WG 256          // threads per workgroup of the count / emit kernels (independent waves, one block each)
#ifndef ZIG_ABLATE
#define ZIG_ABLATE 0        // diagnostic builds (make zigablate; wrong results, timing only): 1 no wedge gather / tests,
#endif                      //   2 also no classification (generator only), 3 no chain in pass 1, 4 pass 2 without stores

namespace {

constexpr u128 mk128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }
constexpr u128 PCG_MULT = mk128(2549297995355413924ULL, 4865540595714422341ULL);

struct jump_t {
    uint64_t mhi, mlo, ghi, glo;   // f^n(s) = m s + inc g (mod 2^128)
};
template <int N>
struct jump_tab {
    jump_t v[N];
};
// n = 2^i
constexpr jump_tab<64> make_pow2() {
    jump_tab<64> t{};
    u128 m = PCG_MULT, g = 1;
    for (int i = 0; i < 64; i++) {
        t.v[i] = jump_t{(uint64_t)(m >> 64), (uint64_t)m, (uint64_t)(g >> 64), (uint64_t)g};
        g = g * (m + 1);
        m = m * m;
    }
    return t;
}
// n = l + 1: from the state before a block's first position to the state that outputs position l
constexpr jump_tab<64> make_lane() {
    jump_tab<64> t{};
    u128 m = PCG_MULT, g = 1;
    for (int i = 0; i < 64; i++) {
        t.v[i] = jump_t{(uint64_t)(m >> 64), (uint64_t)m, (uint64_t)(g >> 64), (uint64_t)g};
        g = g * PCG_MULT + 1;
        m = m * PCG_MULT;
    }
    return t;
}
constexpr jump_tab<64> H_POW2 = make_pow2();
__device__ const jump_tab<64> ZIG_POW2 = make_pow2();
__device__ const jump_tab<64> ZIG_LANE = make_lane();

__host__ __device__ inline u128 jump_apply(const jump_t &j, u128 s, u128 inc) {
    return mk128(j.mhi, j.mlo) * s + inc * mk128(j.ghi, j.glo);
}
__host__ __device__ inline u128 pcg_step(u128 s, u128 inc) { return s * PCG_MULT + inc; }
__host__ __device__ inline uint64_t pcg_out(u128 s) {
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
inline u128 host_advance(u128 s, u128 inc, uint64_t n) {
    for (int i = 0; i < 64 && (n >> i); i++)
        if ((n >> i) & 1) s = jump_apply(H_POW2.v[i], s, inc);
    return s;
}

constexpr double ZIG_NOR_R = 3.6541528853610087963519472518;
constexpr double ZIG_NOR_INV_R = 0.27366123732975827203338247596;
constexpr uint64_t M52 = 0x000fffffffffffffull;

// glibc's log1p (sysdeps/ieee754/dbl-64/s_log1p.c: fdlibm's algorithm, polynomial in glibc's split evaluation order),
// for -1 < x <= 0, operation by operation in IEEE double without contraction: bit-identical to the libm numpy calls
// (oracle/npnormal.py holds the same restatement; tests/test_oracle.py compares it with math.log1p on the host).
__device__ __attribute__((noinline)) double glibc_log1p_neg(double x) {
#pragma clang fp contract(off)
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    const int hx = __double2hiint(x);
    const int ax = hx & 0x7fffffff;
    int k = 1, hu = 0;
    double f = 0.0, c = 0.0;
    if (hx < 0x3FDA827A) {
        if (ax < 0x3e200000) {                       // |x| < 2^-29
            if (ax < 0x3c900000) return x;           // |x| < 2^-54
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= (int)0xbfd2bec3) {       // -0.2929 < x < 0.41422
            k = 0;
            f = x;
            hu = 1;
        }
    }
    if (k != 0) {
        double u = 1.0 + x;
        hu = __double2hiint(u);
        k = (hu >> 20) - 1023;
        c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
        c /= u;
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = __hiloint2double(hu | 0x3ff00000, __double2loint(u));
        } else {
            k += 1;
            u = __hiloint2double(hu | 0x3fe00000, __double2loint(u));
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    const double hfsq = 0.5 * f * f;
    if (hu == 0) {                                   // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        const double Rs = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - Rs;
        return k * ln2_hi - ((Rs - (k * ln2_lo + c)) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R1 = z * Lp1, z2 = z * z;
    const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
    const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
    const double R4 = Lp6 + z * Lp7;
    const double Rr = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + Rr));
    return k * ln2_hi - ((hfsq - (s * (hfsq + Rr) + (k * ln2_lo + c))) - f);
}

// glibc's exp (sysdeps/ieee754/dbl-64/e_exp.c: x = k ln2 / 128 + r, 2^(k/128) from a 128-entry table of (tail, scale
// bits), exp(r) - 1 by a degree-5 polynomial) in the evaluation order of its FMA build - the one the loader selects on
// every CPU with FMA + AVX2, read off the installed libm's code - operation by operation: what is fused there is an fma
// here, what is separate stays separate (contraction off).  Constants: glibc_exp_tab.inc (tools/gen_glibc_exp_tab.py
// reads them out of the installed libm).  For |x| < 512 (the wedge test's argument is in [-6.7, 0)); oracle/npnormal.py
// restates the same sequence and tests/test_oracle.py pins it to the host's exp bit for bit.
#include "glibc_exp_tab.inc"
__device__ const ulonglong2 g_gexp_T[128] = GLIBC_EXP_T;
__device__ inline double glibc_exp_fma(double x) {
#pragma clang fp contract(off)
    constexpr double C[4] = GLIBC_EXP_C;
    const unsigned abstop = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
    if (abstop < 0x3c9u) return 1.0 + x;                          // |x| < 2^-54
    double kd = fma(x, GLIBC_EXP_INVLN2N, GLIBC_EXP_SHIFT);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = kd - GLIBC_EXP_SHIFT;
    const double r = fma(kd, GLIBC_EXP_NEGLN2LON, fma(kd, GLIBC_EXP_NEGLN2HIN, x));
    const ulonglong2 t = g_gexp_T[ki & 127ull];
    const double tail = __longlong_as_double((long long)t.x);
    const unsigned long long sbits = t.y + (ki << 45);
    const double p23 = fma(r, C[1], C[0]);
    const double tr = r + tail;
    const double r2 = r * r;
    const double p45 = fma(r, C[3], C[2]);
    const double t1 = fma(p23, r2, tr);
    const double r4 = r2 * r2;
    const double tmp = fma(r4, p45, t1);
    const double scale = __longlong_as_double((long long)sbits);
    return fma(scale, tmp, scale);
}
__global__ void glibc_exp_kernel(const double *__restrict__ x, long n, double *__restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = glibc_exp_fma(x[i]);
}

__device__ inline double raw_to_double(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

// the value of a fast-path / wedge sample: rabs * wi[idx], negated by the sign bit
__device__ inline double zig_value(uint64_t r, const double *wi) {
    const double x = (double)((r >> 9) & M52) * wi[r & 0xff];
    return __longlong_as_double(__double_as_longlong(x) ^ (long long)(((r >> 8) & 1) << 63));
}

// wedge test of a sample started by r0 (idx != 0, not fast) with the uniform of the next draw r1
__device__ __attribute__((noinline)) bool zig_wedge_accept(uint64_t r0, uint64_t r1, const double *wi, const double *fi) {
#pragma clang fp contract(off)
    const unsigned idx = (unsigned)(r0 & 0xff);
    const double x = (double)((r0 >> 9) & M52) * wi[idx];
    const double u = raw_to_double(r1);
    const double lhs = (fi[idx - 1] - fi[idx]) * u + fi[idx];
    return lhs < glibc_exp_fma(-0.5 * x * x);
}

struct zig_status {
    unsigned long long total;      // ordinal after the last block of the round
    unsigned long long n_raw;      // raw draws consumed by the first n normals (absolute position count)
    unsigned k_last, npatch, error, pad;
};

#define ZIG_TAIL_CAP 4096   // iterations of one tail loop before the status word is flagged (rejection rate 8 %)

// ---- wave-level primitives ---------------------------------------------------------------------------------------
// (the builtins return int: without the casts the low word sign-extends over the high one)
__device__ inline unsigned uni32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane(v); }
__device__ inline uint64_t uni64(uint64_t v) { return ((uint64_t)uni32((unsigned)(v >> 32)) << 32) | uni32((unsigned)v); }
__device__ inline unsigned readlane32(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane(v, l); }
// number of set bits of m below this lane
// lane i <- lane i + 1 (lane 63 keeps its own value): a DPP wave shift, two VALU moves - __shfl_down compiles to two
// ds_bpermute round trips through the LDS crossbar, which the classifying loop took in more than half of its rows
__device__ inline uint64_t wave_shl1(uint64_t v) {
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    const int slo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);   // wave_shl:1
    const int shi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
    return ((uint64_t)(unsigned)shi << 32) | (unsigned)slo;
}
__device__ inline unsigned mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ inline uint64_t readlane64(uint64_t v, int l) {
    return ((uint64_t)readlane32((unsigned)(v >> 32), l) << 32) | readlane32((unsigned)v, l);
}
__device__ inline void wave_lds_fence() {      // LDS serves a wave's instructions in order; this stops the compiler
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// the 64-bit class masks of the 16 rows of a block, row j in lane j of each word ("columns"): wave-uniform values
// without spending 6 x 16 scalar registers
struct cols_t {
    unsigned nf_lo = 0, nf_hi = 0, z_lo = 0, z_hi = 0, w_lo = 0, w_hi = 0;
};
__device__ inline void col_set(unsigned &lo, unsigned &hi, int j, uint64_t m) {
    const bool mine = (int)(threadIdx.x & 63) == j;           // (m is wave-uniform: two selects against scalars)
    lo = mine ? (unsigned)m : lo;
    hi = mine ? (unsigned)(m >> 32) : hi;
}
__device__ inline uint64_t col_get(unsigned lo, unsigned hi, int j) {
    return ((uint64_t)readlane32(hi, j) << 32) | readlane32(lo, j);
}

// tail sample whose first draw came out of state `s` (wave-uniform; every lane computes the same): the value with its
// sign, and the draws consumed after that first one
__device__ __attribute__((noinline)) void zig_tail_from(u128 s, u128 inc, double &val, unsigned &consumed, unsigned *err) {
#pragma clang fp contract(off)
    const uint64_t r0 = pcg_out(s);
    unsigned c = 0;
    double xx = 0.0;
    for (int it = 0;; it++) {
        s = pcg_step(s, inc);
        const double u1 = raw_to_double(pcg_out(s));
        s = pcg_step(s, inc);
        const double u2 = raw_to_double(pcg_out(s));
        c += 2;
        xx = -ZIG_NOR_INV_R * glibc_log1p_neg(-u1);
        const double yy = -glibc_log1p_neg(-u2);
        if (yy + yy > xx * xx) break;
        if (it >= ZIG_TAIL_CAP) {
            atomicOr(err, 1u);
            break;
        }
    }
    const bool neg = (((r0 >> 9) & M52) >> 8) & 1;
    val = neg ? -(ZIG_NOR_R + xx) : ZIG_NOR_R + xx;
    consumed = c;
}
// the same for position q of a block, from the state before the block's first position (a jump of q + 1 steps)
__device__ inline void zig_tail_walk(u128 s_blk, u128 inc, unsigned q, double &val, unsigned &consumed, unsigned *err) {
    u128 s = s_blk;
    const unsigned n = q + 1;
    for (int i = 0; i < 32 && (n >> i); i++)
        if ((n >> i) & 1) s = jump_apply(ZIG_POW2.v[i], s, inc);
    zig_tail_from(s, inc, val, consumed, err);
}
__device__ inline u128 bcast128(u128 s, int l) {
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    return mk128(readlane64(hi, l), readlane64(lo, l));
}

// The tail-class positions of a block with what a sample started there consumes: up to four 16-bit entries
// q | (consumed / 2) << 10 (never 0: a tail sample consumes at least two draws); bit 63 = more than fits - the block
// then takes the sequential scalar chain.  Pass 1 makes the table where the row's generator state is live, pass 2
// reads it back.
#define ZIG_TT_OVER (1ull << 63)
__device__ inline uint64_t ttab_add(uint64_t tt, unsigned ntail, unsigned q, unsigned consumed) {
    if (ntail >= 4 || (consumed >> 1) >= 32) return tt | ZIG_TT_OVER;
    return tt | ((uint64_t)(q | ((consumed >> 1) << 10)) << (16 * ntail));
}
// draws consumed by the tail sample started at position q (0: not in the table)
__device__ inline unsigned ttab_lookup(uint64_t tt, unsigned q) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned en = (unsigned)(tt >> (16 * i)) & 0x7fffu;
        if (en != 0 && (en & 0x3ffu) == q) c = (en >> 10) << 1;
    }
    return c;
}

// ---- the run scan ------------------------------------------------------------------------------------------------
// One row = 64 consecutive positions, bit p of a mask = position p.  `pos` = the next position of the BLOCK that starts a
// sample (everything below it is consumed).  Returns for row j: e_fast / e_wedge = the fast-path / accepted-wedge
// samples of the row up to its first tail-class start, tl_bit = that start (64 = none), and moves pos on past the row
// or past what the row's samples up to there consume.  All values are wave-uniform: this is scalar code.
struct row_eval_t {
    uint64_t e_fast, e_wedge;
    int tl_bit;
};
__device__ inline row_eval_t row_eval(uint64_t nf_j, uint64_t z_j, uint64_t w_j, int j, unsigned &pos) {
    constexpr uint64_t EVEN = 0x5555555555555555ull;
    const unsigned lowk = pos - 64u * j;                                 // < 64 (the caller skips consumed rows)
    const uint64_t low = (1ull << lowk) - 1ull;
    const uint64_t nf = nf_j & ~low;
    const uint64_t starts = nf & ~(nf << 1);
    const uint64_t re = nf & ~(nf + (starts & EVEN));                    // runs whose first position is even
    const uint64_t ro = nf & ~(nf + (starts & ~EVEN));                   // ... odd
    const uint64_t sn = (re & EVEN) | (ro & ~EVEN);                      // sample starts that are not fast
    const uint64_t S = ~(sn << 1);                                       // sample starts
    const uint64_t valid = ~low;
    const uint64_t tl = S & z_j & valid;
    row_eval_t r;
    r.tl_bit = tl ? __builtin_ctzll(tl) : 64;
    const uint64_t upto = tl ? (valid & ((tl & (0ull - tl)) - 1ull)) : valid;
    r.e_fast = S & ~nf & upto;
    r.e_wedge = sn & ~z_j & w_j & upto;
    if (!tl) pos = 64u * (j + 1) + (unsigned)(sn >> 63);                 // a wedge sample at bit 63 takes the next row's first draw
    return r;
}

// (k_out, count) of a block entered with k positions consumed (pass 1: no values)
__device__ inline void block_chain(const cols_t &c, u128 s_blk, u128 inc, unsigned k, unsigned &kout, unsigned &cnt,
                                   unsigned &pos_row0, unsigned &cnt_row0, int rows, unsigned *err) {
    unsigned pos = k, n = 0;
    for (int j = 0; j < ZIG_ROWS; j++) {
        if (j == 1) {
            pos_row0 = pos;
            cnt_row0 = n;
            if (rows == 1) break;
        }
        if (pos >= 64u * (j + 1)) continue;
        const uint64_t nf_j = col_get(c.nf_lo, c.nf_hi, j), z_j = col_get(c.z_lo, c.z_hi, j), w_j = col_get(c.w_lo, c.w_hi, j);
        while (pos < 64u * (j + 1)) {
            const row_eval_t r = row_eval(nf_j, z_j, w_j, j, pos);
            n += __builtin_popcountll(r.e_fast | r.e_wedge);
            if (r.tl_bit == 64) break;
            double v;
            unsigned consumed;
            zig_tail_walk(s_blk, inc, 64u * j + r.tl_bit, v, consumed, err);
            n += 1;
            pos = 64u * j + r.tl_bit + 1 + consumed;
        }
    }
    kout = pos > ZIG_BLK ? pos - ZIG_BLK : 0u;
    cnt = n;
}

// ---- the run scan of all 16 rows at once (lanes 0..15 = rows, 64-bit VALU) -------------------------------------------
// Row j evaluated for both possible carries in (0: its first position starts a sample, 1: it is the wedge draw of the
// previous row's last position); the 16 carries follow from the rows' (carry out | carry in) pairs with one integer add
// (generate = out for carry 0, propagate = out only for carry 1).  A tail-class START anywhere on the chosen path
// (a quarter of the blocks) sends the block to the sequential scalar chain instead.
struct vrow_t {
    uint64_t e, ew, t;      // samples of the row: fast | accepted wedge, the accepted wedges among them, tail samples
    unsigned cout;          // positions of the next row the row's samples consume (0 / 1; more: `big`)
    bool big;
};
__device__ inline vrow_t row_eval_vec(uint64_t nf_j, uint64_t z_j, uint64_t w_j, unsigned cin, uint64_t tt, unsigned q0) {
    constexpr uint64_t EVEN = 0x5555555555555555ull;
    vrow_t r;
    r.e = r.ew = r.t = 0;
    r.cout = 0;
    r.big = false;
    uint64_t low = cin;                                   // positions of the row already consumed
    for (int it = 0; it < 6; it++) {
        const uint64_t nf = nf_j & ~low;
        const uint64_t starts = nf & ~(nf << 1);
        const uint64_t re = nf & ~(nf + (starts & EVEN));
        const uint64_t ro = nf & ~(nf + (starts & ~EVEN));
        const uint64_t sn = (re & EVEN) | (ro & ~EVEN);
        const uint64_t S = ~(sn << 1);
        const uint64_t valid = ~low;
        const uint64_t tl = S & z_j & valid;
        const uint64_t upto = tl ? (valid & ((tl & (0ull - tl)) - 1ull)) : valid;
        const uint64_t ew = sn & ~z_j & w_j & upto;
        r.e |= (S & ~nf & upto) | ew;
        r.ew |= ew;
        if (!tl) {
            r.cout = (unsigned)(sn >> 63);
            return r;
        }
        const unsigned p = (unsigned)__builtin_ctzll(tl);
        r.t |= 1ull << p;
        const unsigned nxt = p + 1 + ttab_lookup(tt, q0 + p);
        if (nxt >= 64) {
            r.cout = nxt - 64;
            r.big = r.cout > 1;
            return r;
        }
        low = (1ull << nxt) - 1ull;
    }
    r.big = true;                                          // (more tail samples in one row than the loop allows)
    return r;
}
// carries into rows 0..16 (bit j = row j's carry in; bit 16 = the block's carry out) for block entry k in {0, 1}
__device__ inline unsigned row_carries(unsigned G, unsigned C1, unsigned k) {
    if ((G & ~C1) == 0) {                       // no row inverts its carry: generate / propagate, one add
        const unsigned sum = C1 + G + k;
        return (sum ^ C1 ^ G) & 0x1ffffu;
    }
    unsigned c = k, cins = 0;
    for (int j = 0; j < ZIG_ROWS; j++) {
        cins |= c << j;
        c = c ? (C1 >> j) & 1u : (G >> j) & 1u;
    }
    return cins | (c << 16);
}
// sum over lanes 0..15 (DPP row shifts; lanes 16.. must hold 0 or are ignored): the total, wave-uniform
__device__ inline unsigned row16_sum(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    return readlane32(v, 15);
}
// inclusive prefix over lanes 0..15
__device__ inline unsigned row16_scan(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); v += t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); v += t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); v += t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); v += t;
    return v;
}

struct zig_lds {
    uint64_t ki[256];
    double wi[256], fi[256];
};
__device__ inline void zig_lds_fill(zig_lds &L) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        L.ki[i] = ZIG_KI[i];
        L.wi[i] = ZIG_WI[i];
        L.fi[i] = ZIG_FI[i];
    }
    __syncthreads();
}

// state before the first position of block b: f^(pos0 + b BLK)(s0)
__global__ void zig_seek_kernel(uint64_t s_hi, uint64_t s_lo, uint64_t i_hi, uint64_t i_lo, unsigned long long pos0,
                                long nblk, ulonglong2 *__restrict__ blk_state) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const u128 inc = mk128(i_hi, i_lo);
    u128 s = mk128(s_hi, s_lo);
    const unsigned long long n = pos0 + (unsigned long long)b * ZIG_BLK;
    for (int i = 0; i < 64 && (n >> i); i++)
        if ((n >> i) & 1) s = jump_apply(ZIG_POW2.v[i], s, inc);
    blk_state[b] = make_ulonglong2((uint64_t)(s >> 64), (uint64_t)s);
}

// wave-private LDS of the wedge tests: 1.45 % of the positions are wedge class (15 of a block's 1024); tested where
// they stand they cost a full instruction stream each, so the (draw, next draw) pairs of a block are gathered and
// tested 64 at a time, the verdicts coming back as bits of the rows' masks
#define ZIG_WCAP 128
struct zig_wlds {
    ulonglong2 cand[ZIG_WCAP];
    unsigned short where[ZIG_WCAP];     // position in the block
    unsigned wmask[2 * ZIG_ROWS];       // accepted-wedge masks, row j in words 2 j, 2 j + 1
};

__device__ inline void wedge_flush(zig_wlds *W, unsigned ncand, const double *wi, const double *fi) {
    const int lane = threadIdx.x & 63;
    wave_lds_fence();
    for (unsigned i = lane; i < ncand; i += 64) {
        const ulonglong2 pr = W->cand[i];
        const unsigned q = W->where[i];
        if (zig_wedge_accept(pr.x, pr.y, wi, fi)) atomicOr(&W->wmask[q >> 5], 1u << (q & 31));
    }
    wave_lds_fence();
}

// rows 0 .. 15 of block b: the class masks of every row (columns) - pass 1's vector work.  KEEP (the single-pass
// kernel): the raw draws of the 16 rows stay in `raw` (row j of lane p = position 64 j + p) and the values of the
// tail samples of the table in `tv` (entry i of the table <-> tv[i]; wave-uniform).
template <bool KEEP = false>
__device__ inline uint64_t block_classify(cols_t &c, u128 s_blk, u128 inc, u128 lane_m, u128 lane_c, u128 c64,
                                          const zig_lds &L, zig_wlds *W, unsigned *err, uint64_t *raw = nullptr,
                                          double *tv = nullptr) {
    const int lane = threadIdx.x & 63;
    constexpr u128 M64 = mk128(H_POW2.v[6].mhi, H_POW2.v[6].mlo);
    u128 s = lane_m * s_blk + lane_c;                            // state after position `lane` of row 0
    uint64_t raw_cur = pcg_out(s);
    if (lane < 2 * ZIG_ROWS) W->wmask[lane] = 0;
    unsigned ncand = 0, ntail = 0;
    uint64_t tt = 0;
#pragma unroll
    for (int j = 0; j < ZIG_ROWS; j++) {
        const u128 s_row = s;                                    // the states that put out row j
        if (KEEP) raw[64 * j + lane] = raw_cur;                 // (wave-private LDS: a lane reads back what it wrote)
        s = s * M64 + c64;
        const uint64_t raw_next = pcg_out(s);                    // row j + 1 (row 16: the next block's first row)
        const unsigned idx = (unsigned)(raw_cur & 0xff);
#if ZIG_ABLATE == 2
        const bool slow = (raw_cur >> 9) == 12345;
#else
        const bool slow = ((raw_cur >> 9) & M52) >= L.ki[idx];
#endif
        const uint64_t NF = __ballot(slow), Z = __ballot(slow && idx == 0);
#if ZIG_ABLATE == 1 || ZIG_ABLATE == 2
        const uint64_t Wm = 0;
#else
        const uint64_t Wm = NF & ~Z;
#endif
        if (Wm) {
            // the wedge draw of position p is the draw of position p + 1: the next lane, or lane 0 of the next row
            uint64_t r1 = wave_shl1(raw_cur);
            const uint64_t first_next = readlane64(raw_next, 0);
            if (lane == 63) r1 = first_next;
            const unsigned add = __builtin_popcountll(Wm);
            if (ncand + add > ZIG_WCAP) {
                wedge_flush(W, ncand, L.wi, L.fi);
                ncand = 0;
            }
            if ((Wm >> lane) & 1) {
                const unsigned slot = ncand + mbcnt64(Wm);
                W->cand[slot] = make_ulonglong2(raw_cur, r1);
                W->where[slot] = (unsigned short)(64 * j + lane);
            }
            ncand += add;
        }
        if (Z) {                                                 // 1.6 % of the rows: what a sample started there would consume
            for (uint64_t zz = Z; zz; zz &= zz - 1) {
                const int pz = __builtin_ctzll(zz);
                double v;
                unsigned consumed;
                zig_tail_from(bcast128(s_row, pz), inc, v, consumed, err);
                if (KEEP) {
#pragma unroll
                    for (int i = 0; i < 4; i++) tv[i] = ntail == (unsigned)i ? v : tv[i];
                }
                tt = ttab_add(tt, ntail++, 64u * j + pz, consumed);
            }
        }
        col_set(c.nf_lo, c.nf_hi, j, NF);
        col_set(c.z_lo, c.z_hi, j, Z);
        raw_cur = raw_next;
    }
    wedge_flush(W, ncand, L.wi, L.fi);
    c.w_lo = lane < ZIG_ROWS ? W->wmask[2 * lane] : 0u;
    c.w_hi = lane < ZIG_ROWS ? W->wmask[2 * lane + 1] : 0u;
    wave_lds_fence();
    return tt;
}

// pass 1: fun[2 b + e] = k_out << 16 | count of block b entered with e in {0, 1} positions consumed; classes[b][j] =
// (A, B) of row j with A = wedge class, B = accepted wedge | tail class; successors that will be entered with k >= 2
// are evaluated here and appended to the patch list (key = b << 16 | k, value as fun).
// One wave per block; the waves of a workgroup share nothing but the tables.
__global__ void __launch_bounds__(ZIG_WG)
zig_count_kernel(const ulonglong2 *__restrict__ blk_state, uint64_t i_hi, uint64_t i_lo, long nblk,
                 unsigned *__restrict__ fun, ulonglong2 *__restrict__ classes, unsigned long long *__restrict__ tails,
                 ulonglong2 *__restrict__ patch, unsigned patch_cap, zig_status *st) {
    __shared__ zig_lds L;
    __shared__ zig_wlds Wall[ZIG_WG / 64];
    zig_wlds *W = &Wall[threadIdx.x >> 6];
    zig_lds_fill(L);
    const u128 inc = mk128(i_hi, i_lo);
    const u128 c64 = inc * mk128(H_POW2.v[6].ghi, H_POW2.v[6].glo);
    const int lane = threadIdx.x & 63;
    // from the state before a block's first position to the one that outputs position `lane`: s -> lane_m s + lane_c
    const jump_t lj = ZIG_LANE.v[lane];
    const u128 lane_m = mk128(lj.mhi, lj.mlo), lane_c = inc * mk128(lj.ghi, lj.glo);
    const long wave0 = (long)blockIdx.x * (ZIG_WG / 64) + (threadIdx.x >> 6), nwave = (long)gridDim.x * (ZIG_WG / 64);
    for (long b = wave0; b < nblk; b += nwave) {
        const ulonglong2 bs = blk_state[b];
        const u128 s_blk = mk128(uni64(bs.x), uni64(bs.y));
        cols_t c;
        const uint64_t tt = block_classify(c, s_blk, inc, lane_m, lane_c, c64, L, W, &st->error);
        if (lane < ZIG_ROWS) {
            const uint64_t nf = ((uint64_t)c.nf_hi << 32) | c.nf_lo, z = ((uint64_t)c.z_hi << 32) | c.z_lo,
                           w = ((uint64_t)c.w_hi << 32) | c.w_lo;
            classes[b * ZIG_ROWS + lane] = make_ulonglong2(nf & ~z, w | z);
        }
        if (lane == 0) tails[b] = tt;
        unsigned kb0, total0, kb1, total1;
#if ZIG_ABLATE == 3
        kb0 = kb1 = 0;
        total0 = total1 = 1000 + (c.nf_lo & 1);
#else
        {
            const bool row = lane < ZIG_ROWS;
            const uint64_t nf = ((uint64_t)c.nf_hi << 32) | c.nf_lo, z = ((uint64_t)c.z_hi << 32) | c.z_lo,
                           w = ((uint64_t)c.w_hi << 32) | c.w_lo;
            const vrow_t v0 = row_eval_vec(nf, z, w, 0u, tt, 64u * lane), v1 = row_eval_vec(nf, z, w, 1u, tt, 64u * lane);
            const unsigned G = (unsigned)__ballot(row && v0.cout), C1 = (unsigned)__ballot(row && v1.cout);
            const unsigned p0 = row ? __builtin_popcountll(v0.e | v0.t) : 0u;
            const unsigned p1 = row ? __builtin_popcountll(v1.e | v1.t) : 0u;
            const unsigned sum0 = row16_sum(p0);
            const bool over = (tt & ZIG_TT_OVER) != 0;
            unsigned kk[2], tt_[2];
            for (unsigned e = 0; e < 2; e++) {
                const unsigned cins = row_carries(G, C1, e);
                const bool c1 = (cins >> lane) & 1u;
                const bool slow = over || __ballot(row && (c1 ? v1.big : v0.big)) != 0;
                if (!slow) {
                    kk[e] = cins >> 16;
                    tt_[e] = sum0 + row16_sum(c1 && row ? p1 - p0 : 0u);
                } else {
                    unsigned pp, cc;
                    block_chain(c, s_blk, inc, e, kk[e], tt_[e], pp, cc, ZIG_ROWS, &st->error);
                }
            }
            kb0 = kk[0];
            total0 = tt_[0];
            kb1 = kk[1];
            total1 = tt_[1];
        }
#endif
        if (lane == 0) *reinterpret_cast<uint2 *>(fun + 2 * b) = make_uint2((kb0 << 16) | total0, (kb1 << 16) | total1);
        // rare: a tail sample (or several) reaches across the end of the block
        for (int e = 0; e < 2; e++) {
            unsigned k = e ? kb1 : kb0;
            if (e == 1 && kb1 == kb0) break;
            for (long bb = b + 1; k >= 2 && bb < nblk; bb++) {
                const ulonglong2 bs2 = blk_state[bb];
                const u128 s2 = mk128(uni64(bs2.x), uni64(bs2.y));
                cols_t c2;
                block_classify(c2, s2, inc, lane_m, lane_c, c64, L, W, &st->error);
                unsigned kb, total, pp, cc;
                block_chain(c2, s2, inc, k, kb, total, pp, cc, ZIG_ROWS, &st->error);
                if (lane == 0) {
                    const unsigned slot = atomicAdd(&st->npatch, 1u);
                    if (slot < patch_cap) patch[slot] = make_ulonglong2(((unsigned long long)bb << 16) | k, (kb << 16) | total);
                    else atomicOr(&st->error, 2u);
                }
                k = kb;
            }
        }
    }
}

// ---- the scan over the block functions ---------------------------------------------------------------------------
// Three small kernels: (1) every tile of 64 x 8 consecutive blocks (one wave; lane = 8 blocks) reduced to its function
// on {0, 1}; (2) one workgroup composes the tile functions and gives every tile its (k, ordinal); (3) every tile
// writes the (k, ordinal) of its blocks.  Inside a tile the lane entries come from the same fixed-point iteration
// from k = 0: almost every block maps 0 and 1 to the same k_out, which ends the dependency.
__device__ inline unsigned patch_lookup(const ulonglong2 *patch, unsigned npatch, long b, unsigned k, unsigned *err) {
    const unsigned long long key = ((unsigned long long)b << 16) | k;
    for (unsigned i = 0; i < npatch; i++)
        if (patch[i].x == key) return (unsigned)patch[i].y;
    atomicOr(err, 4u);
    return 0u;
}
__device__ inline void fun_step(long b, uint2 w2, unsigned &k, unsigned &cnt, const ulonglong2 *patch, unsigned npatch,
                                unsigned *err) {
    const unsigned w = k < 2 ? (k ? w2.y : w2.x) : patch_lookup(patch, npatch, b, k, err);
    k = w >> 16;
    cnt += w & 0xffffu;
}
#define ZIG_SEG 8                       // blocks per lane
#define ZIG_TILE (64 * ZIG_SEG)         // blocks per tile
struct tile_t {
    uint2 w2[ZIG_SEG];
    long b0;                            // first block of the lane
    int nv;                             // blocks of the lane that exist
};
__device__ inline void tile_load(tile_t &t, const unsigned *fun, long tile, long nblk) {
    const int lane = threadIdx.x & 63;
    t.b0 = tile * ZIG_TILE + (long)lane * ZIG_SEG;
    t.nv = (int)std::max<long>(0, std::min<long>(ZIG_SEG, nblk - t.b0));
    const uint2 *f2 = reinterpret_cast<const uint2 *>(fun);
#pragma unroll
    for (int j = 0; j < ZIG_SEG; j++) t.w2[j] = j < t.nv ? f2[t.b0 + j] : make_uint2(0u, 0u);
}
// The function value of block b entered with k >= 2 (a tail sample reached across the boundary: 1.5e-4 of the blocks)
// is in the patch list, which pass 1 appends to in no order.  A lane that walked the list by itself held its whole wave
// for npatch dependent loads (~150 at cfg 3, ~2500 at cfg 5: the scan kernels took 0.2-0.26 ms each for trivial work);
// here the WAVE searches for it: every lane looks at npatch / 64 entries.  `need` lanes get their value in `w`.
__device__ inline void patch_lookup_wave(bool need, long b, unsigned k, unsigned &w, const ulonglong2 *patch, unsigned npatch,
                                         unsigned *err) {
    const int lane = threadIdx.x & 63;
    unsigned long long nm = __ballot(need);
    while (nm) {
        const int src = __builtin_ctzll(nm);
        const unsigned long long key = (readlane64((uint64_t)b, src) << 16) | readlane32(k, src);
        unsigned found = 0;
        bool hit = false;
        for (unsigned i = lane; i < npatch; i += 64)
            if (patch[i].x == key) {
                found = (unsigned)patch[i].y;
                hit = true;
            }
        const unsigned long long hm = __ballot(hit);
        unsigned val = 0;
        if (hm) val = readlane32(found, __builtin_ctzll(hm));
        else if (lane == src) atomicOr(err, 4u);
        if (lane == src) w = val;
        nm &= nm - 1;
    }
}
// one lane = ZIG_SEG consecutive blocks; all lanes of the wave call this together (uniform control flow)
__device__ inline void seg_eval(const tile_t &t, unsigned k, unsigned &kout, unsigned &cnt, const ulonglong2 *patch,
                                unsigned npatch, unsigned *err) {
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < ZIG_SEG; j++) {
        const bool act = j < t.nv;
        unsigned w = k ? t.w2[j].y : t.w2[j].x;
        patch_lookup_wave(act && k >= 2, t.b0 + j, k, w, patch, npatch, err);
        if (act) {
            k = w >> 16;
            c += w & 0xffffu;
        }
    }
    kout = k;
    cnt = c;
}
// lane entries of a tile entered with k_tile; returns the tile's k_out, every lane keeps its (kin, cnt)
__device__ inline unsigned tile_resolve(const tile_t &t, unsigned k_tile, unsigned &kin, unsigned &cnt,
                                        const ulonglong2 *patch, unsigned npatch, unsigned *err) {
    const int lane = threadIdx.x & 63;
    kin = lane == 0 ? k_tile : 0u;
    unsigned assumed = 0, kout = 0;
    for (int it = 0; it <= 64; it++) {
        seg_eval(t, kin, kout, cnt, patch, npatch, err);
        const bool changed = lane != 63 && kout != assumed;
        assumed = kout;
        if (!__any(changed)) break;
        const unsigned up = __shfl_up(kout, 1);
        kin = lane == 0 ? k_tile : up;
    }
    return __shfl(kout, 63);
}
__device__ inline unsigned wave_sum(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256)
zig_tile_kernel(long nblk, long ntile, const unsigned *__restrict__ fun, const ulonglong2 *__restrict__ patch,
                unsigned patch_cap, uint4 *__restrict__ tile_fun, zig_status *st) {
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntile) return;
    const unsigned npatch = min(st->npatch, patch_cap);
    tile_t t;
    tile_load(t, fun, tile, nblk);
    unsigned kin, cnt;
    const unsigned k0 = tile_resolve(t, 0u, kin, cnt, patch, npatch, &st->error);
    const unsigned c0 = wave_sum(cnt);
    const unsigned k1 = tile_resolve(t, 1u, kin, cnt, patch, npatch, &st->error);
    const unsigned c1 = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) tile_fun[tile] = make_uint4(k0, c0, k1, c1);
}

// one workgroup: tile_entry[t] = (k, ordinal) of every tile; a tile entered with k >= 2 (a tail sample across a tile
// boundary) is walked block by block here
#define ZIG_TOP_T 1024
__global__ void __launch_bounds__(ZIG_TOP_T)
zig_top_kernel(long nblk, long ntile, const unsigned *__restrict__ fun, const ulonglong2 *__restrict__ patch,
               unsigned patch_cap, const uint4 *__restrict__ tile_fun, unsigned long long ord0,
               ulonglong2 *__restrict__ tile_entry, zig_status *st, int force_serial) {
    constexpr int T = ZIG_TOP_T;
    __shared__ unsigned g_k[T][2];
    __shared__ unsigned long long g_c[T][2];
    __shared__ unsigned s_k[T];
    __shared__ unsigned long long s_o[T];
    const int t = threadIdx.x;
    const unsigned npatch = min(st->npatch, patch_cap);
    const long seg = (ntile + T - 1) / T;
    const long t0 = std::min<long>(ntile, t * seg), t1 = std::min<long>(ntile, t0 + seg);
    auto tile_step = [&](long tile, unsigned &k, unsigned long long &o) {
        if (k < 2) {
            const uint4 f = tile_fun[tile];
            o += k ? f.w : f.y;
            k = k ? f.z : f.x;
        } else {
            const uint2 *f2 = reinterpret_cast<const uint2 *>(fun);
            const long c0 = tile * ZIG_TILE, c1 = std::min<long>(nblk, c0 + ZIG_TILE);
            unsigned c = 0;
            for (long b = c0; b < c1; b++) fun_step(b, f2[b], k, c, patch, npatch, &st->error);
            o += c;
        }
    };
    for (int e = 0; e < 2; e++) {
        unsigned k = e;
        unsigned long long o = 0;
        for (long tile = t0; tile < t1; tile++) tile_step(tile, k, o);
        g_k[t][e] = k;
        g_c[t][e] = o;
    }
    __syncthreads();
    // The T thread functions are composed by ONE wave in three short steps (round 6; a single thread walking all T entries -
    // 1024 dependent LDS reads - was 0.25 ms of the 0.35 ms scan): lane L composes its T / 64 consecutive entries for both
    // entries k, the 64 lane functions are chained through readlane, every lane walks its entries again with its true
    // (k, ordinal).  An exit k >= 2 anywhere (a tail sample across a thread's boundary: ~15 % of the cfg-3 streams have
    // one) has no table entry: the serial walk below, which evaluates such tiles block by block, takes over.
    __shared__ int s_serial;
    if (t == 0) s_serial = 0;
    __syncthreads();
    if (t < 64) {
        constexpr int PER = T / 64;
        unsigned fk[2];
        unsigned long long fc[2];
        bool bad = false;
        for (int e = 0; e < 2; e++) {
            unsigned k = e;
            unsigned long long c = 0;
            for (int i = t * PER; i < (t + 1) * PER; i++) {
                if (k >= 2) {
                    bad = true;
                    break;
                }
                c += g_c[i][k];
                k = g_k[i][k];
            }
            fk[e] = k;
            fc[e] = c;
            bad = bad || k >= 2;
        }
        if (__any(bad) || force_serial) {
            if (t == 0) s_serial = 1;
        } else {
            unsigned k = 0, my_k = 0;
            unsigned long long o = ord0, my_o = 0;
            for (int L = 0; L < 64; L++) {
                if (t == L) {
                    my_k = k;
                    my_o = o;
                }
                const unsigned k0 = readlane32(fk[0], L), k1 = readlane32(fk[1], L);
                const unsigned long long c0 = readlane64(fc[0], L), c1 = readlane64(fc[1], L);
                o += k ? c1 : c0;
                k = k ? k1 : k0;
            }
            if (t == 0) {
                st->total = o;
                st->k_last = k;
            }
            for (int i = t * PER; i < (t + 1) * PER; i++) {
                s_k[i] = my_k;
                s_o[i] = my_o;
                my_o += g_c[i][my_k];
                my_k = g_k[i][my_k];
            }
        }
    }
    __syncthreads();
    if (s_serial && t == 0) {
        unsigned k = 0;
        unsigned long long o = ord0;
        for (int i = 0; i < T; i++) {
            s_k[i] = k;
            s_o[i] = o;
            if (k < 2) {
                o += g_c[i][k];
                k = g_k[i][k];
            } else {
                const long a0 = std::min<long>(ntile, i * seg), a1 = std::min<long>(ntile, a0 + seg);
                for (long tile = a0; tile < a1; tile++) tile_step(tile, k, o);
            }
        }
        st->total = o;
        st->k_last = k;
    }
    __syncthreads();
    unsigned k = s_k[t];
    unsigned long long o = s_o[t];
    for (long tile = t0; tile < t1; tile++) {
        tile_entry[tile] = make_ulonglong2(k, o);
        tile_step(tile, k, o);
    }
}

__global__ void __launch_bounds__(256)
zig_entry_kernel(long nblk, long ntile, const unsigned *__restrict__ fun, const ulonglong2 *__restrict__ patch,
                 unsigned patch_cap, const ulonglong2 *__restrict__ tile_entry, ulonglong2 *__restrict__ entry,
                 zig_status *st) {
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntile) return;
    const int lane = threadIdx.x & 63;
    const unsigned npatch = min(st->npatch, patch_cap);
    tile_t t;
    tile_load(t, fun, tile, nblk);
    const ulonglong2 te = tile_entry[tile];
    unsigned kin, cnt;
    tile_resolve(t, (unsigned)te.x, kin, cnt, patch, npatch, &st->error);
    unsigned incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    unsigned k = kin;
    unsigned long long o = te.y + (incl - cnt);
#pragma unroll
    for (int j = 0; j < ZIG_SEG; j++) {
        const bool act = j < t.nv;
        if (act) entry[t.b0 + j] = make_ulonglong2(k, o);
        unsigned w = k ? t.w2[j].y : t.w2[j].x;
        patch_lookup_wave(act && k >= 2, t.b0 + j, k, w, patch, npatch, &st->error);
        if (act) {
            k = w >> 16;
            o += w & 0xffffu;
        }
    }
}

// pass 2: every block with its true entry; the normals with ordinals in [o_lo, o_hi) go to g[ordinal - o_lo] (the whole
// stream: o_lo = 0, o_hi = n; one range of the l-range pipeline: its ordinals, g = the ring slot).  blk_first: device
// word holding the first block to look at - the one that contains ordinal o_lo (zig_range_kernel) -, NULL = block 0.
// Lane p of row j holds position 64 j + p: the samples of a row leave as one store instruction to consecutive addresses.
__global__ void __launch_bounds__(ZIG_WG)
zig_emit_kernel(const ulonglong2 *__restrict__ blk_state, uint64_t i_hi, uint64_t i_lo, long nblk,
                const ulonglong2 *__restrict__ entry, const ulonglong2 *__restrict__ classes,
                const unsigned long long *__restrict__ tails, unsigned long long pos0, unsigned long long n,
                unsigned long long o_lo, unsigned long long o_hi, const long *__restrict__ blk_first,
                double *__restrict__ g, zig_status *st) {
    __shared__ zig_lds L;
    zig_lds_fill(L);
    const u128 inc = mk128(i_hi, i_lo);
    const u128 c64 = inc * mk128(H_POW2.v[6].ghi, H_POW2.v[6].glo);
    constexpr u128 M64 = mk128(H_POW2.v[6].mhi, H_POW2.v[6].mlo);
    const int lane = threadIdx.x & 63;
    const jump_t lj = ZIG_LANE.v[lane];
    const u128 lane_m = mk128(lj.mhi, lj.mlo), lane_c = inc * mk128(lj.ghi, lj.glo);
    const long wave0 = (long)blockIdx.x * (ZIG_WG / 64) + (threadIdx.x >> 6), nwave = (long)gridDim.x * (ZIG_WG / 64);
    const long b_first = blk_first ? *blk_first : 0L;
    const unsigned long long o_n = o_hi - o_lo;            // ordinal o is written iff o - o_lo < o_n (unsigned: o < o_lo wraps)
    for (long b = b_first + wave0; b < nblk; b += nwave) {
        const ulonglong2 en = entry[b];
        const unsigned long long ord_blk = uni64(en.y);
        if (ord_blk >= o_hi) break;                         // (blocks are in ordinal order: nothing left to write)
        const ulonglong2 bs = blk_state[b];
        const u128 s_blk = mk128(uni64(bs.x), uni64(bs.y));
        // classes of the 16 rows, row j in lane j
        ulonglong2 cl = make_ulonglong2(0ull, 0ull);
        if (lane < ZIG_ROWS) cl = classes[b * ZIG_ROWS + lane];
        const uint64_t z_col = cl.y & ~cl.x, w_col = cl.y & cl.x, nf_col = cl.x | z_col;
        u128 s = lane_m * s_blk + lane_c;                    // state after position `lane` of row 0
        unsigned pos = uni32((unsigned)en.x);
        const unsigned long long blk_pos = pos0 + (unsigned long long)b * ZIG_BLK;
        // fast path (entry 0 or 1): all rows evaluated at once in lanes 0..15
        const uint64_t tt = uni64(tails[b]);
        if (pos < 2 && !(tt & ZIG_TT_OVER)) {
            const bool row = lane < ZIG_ROWS;
            const vrow_t v0 = row_eval_vec(nf_col, z_col, w_col, 0u, tt, 64u * lane),
                         v1 = row_eval_vec(nf_col, z_col, w_col, 1u, tt, 64u * lane);
            const unsigned G = (unsigned)__ballot(row && v0.cout), C1 = (unsigned)__ballot(row && v1.cout);
            const unsigned cins = row_carries(G, C1, pos);
            const bool c1 = (cins >> lane) & 1u;
            if (__ballot(row && (c1 ? v1.big : v0.big)) == 0) {
                const uint64_t ew_col = row ? (c1 ? v1.ew : v0.ew) : 0ull;
                const uint64_t t_col = row ? (c1 ? v1.t : v0.t) : 0ull;
                const uint64_t e_col = row ? (c1 ? v1.e : v0.e) : 0ull;
                const unsigned pc = __builtin_popcountll(e_col | t_col);
                const unsigned base_col = row16_scan(pc) - pc;          // samples of the block before row `lane`
                const unsigned trows = (unsigned)__ballot(t_col != 0);  // rows with a tail sample
#pragma unroll
                for (int j = 0; j < ZIG_ROWS; j++) {
                    const uint64_t raw = pcg_out(s);
                    const uint64_t e = readlane64(e_col, j);
                    if ((trows >> j) & 1) {
                        const uint64_t tj = readlane64(t_col, j), all = e | tj;
                        for (uint64_t zz = tj; zz; zz &= zz - 1) {
                            const int pz = __builtin_ctzll(zz);
                            double v;
                            unsigned consumed;
                            zig_tail_from(bcast128(s, pz), inc, v, consumed, &st->error);
                            const unsigned long long o = ord_blk + readlane32(base_col, j) +
                                                         __builtin_popcountll(all & ((1ull << pz) - 1ull));
                            if (lane == 0 && o - o_lo < o_n) {
                                g[o - o_lo] = v;
                                if (o + 1 == n) st->n_raw = blk_pos + 64u * j + pz + 1 + consumed;
                            }
                        }
                        if ((e >> lane) & 1) {
                            const unsigned long long o = ord_blk + readlane32(base_col, j) + mbcnt64(all);
                            if (o - o_lo < o_n) {
                                g[o - o_lo] = zig_value(raw, L.wi);
                                if (o + 1 == n) st->n_raw = blk_pos + 64u * j + lane + 1 + ((readlane64(ew_col, j) >> lane) & 1);
                            }
                        }
                    } else if ((e >> lane) & 1) {
                        const unsigned long long o = ord_blk + readlane32(base_col, j) + mbcnt64(e);
#if ZIG_ABLATE == 4
                        if (o - o_lo < o_n && raw == 12345) {
#else
                        if (o - o_lo < o_n) {
#endif
                            g[o - o_lo] = zig_value(raw, L.wi);
                            if (o + 1 == n) st->n_raw = blk_pos + 64u * j + lane + 1 + ((readlane64(ew_col, j) >> lane) & 1);
                        }
                    }
                    s = s * M64 + c64;
                }
                continue;
            }
        }
        unsigned long long ord = ord_blk;                    // ordinal of the next sample
#pragma unroll
        for (int j = 0; j < ZIG_ROWS; j++) {
            const uint64_t raw = pcg_out(s);
            s = s * M64 + c64;
            if (pos >= 64u * (j + 1)) continue;
            const uint64_t nf_j = readlane64(nf_col, j), z_j = readlane64(z_col, j), w_j = readlane64(w_col, j);
            while (pos < 64u * (j + 1)) {
                const row_eval_t r = row_eval(nf_j, z_j, w_j, j, pos);
                const uint64_t e = r.e_fast | r.e_wedge;
                if ((e >> lane) & 1) {
                    const unsigned long long o = ord + mbcnt64(e);
                    if (o - o_lo < o_n) {
                        g[o - o_lo] = zig_value(raw, L.wi);
                        if (o + 1 == n) st->n_raw = blk_pos + 64u * j + lane + 1 + ((r.e_wedge >> lane) & 1);
                    }
                }
                ord += __builtin_popcountll(e);
                if (r.tl_bit == 64) break;
                double v;
                unsigned consumed;
                zig_tail_walk(s_blk, inc, 64u * j + r.tl_bit, v, consumed, &st->error);
                if (lane == 0 && ord - o_lo < o_n) {
                    g[ord - o_lo] = v;
                    if (ord + 1 == n) st->n_raw = blk_pos + 64u * j + r.tl_bit + 1 + consumed;
                }
                ord += 1;
                pos = 64u * j + r.tl_bit + 1 + consumed;
            }
        }
    }
}

// first block of every range of the l-range pipeline: blk_first[r] = the block that contains ordinal bounds[r] (the
// largest b with entry[b].ordinal <= bounds[r]; entry ordinals increase with b)
__global__ void zig_range_kernel(const ulonglong2 *__restrict__ entry, long nblk, const unsigned long long *__restrict__ bounds,
                                 int nr, long *__restrict__ blk_first) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nr) return;
    const unsigned long long o = bounds[r];
    long lo = 0, hi = nblk - 1;                              // invariant: entry[lo].y <= o (or lo = 0), answer in [lo, hi]
    while (lo < hi) {
        const long mid = (lo + hi + 1) >> 1;
        if (entry[mid].y <= o) lo = mid;
        else hi = mid - 1;
    }
    blk_first[r] = lo;
}


// ---- ONE pass: classify, chained scan with decoupled look-back, emit (round 6) -----------------------------------------
// The two-pass form above steps the 128-bit generator twice: once to find every block's (entry, first ordinal), once to
// write the samples.  Here a workgroup of ZC_WAVES waves takes a CHUNK of ZC_WAVES consecutive blocks (a ticket: chunks
// are handed out in order, so every predecessor of a running chunk is running or done - no residency assumption), each
// wave classifies its block and KEEPS its 1024 raw draws in wave-private LDS (in registers the kernel needed 242 VGPRs:
// one workgroup per CU), the workgroup composes its blocks'
// functions on {0, 1} into the chunk's aggregate and publishes it; wave 0 then looks back over the predecessors'
// words (64 chunks per poll; an aggregate can be folded, an inclusive word ends the walk - Merrill & Garland's
// decoupled look-back on a function composition instead of a sum), publishes the chunk's inclusive word (exit k,
// ordinal after) BEFORE anything is written, and every wave emits from the kept draws.  Words are single naturally
// aligned 8-byte values with their valid bit inside, stored and polled with agent-scope relaxed atomics (L1 bypass):
// no fences, nothing else is handed over.
// An entry k >= 2 (a tail sample across a boundary, 1.5e-4 of the blocks) has no table entry: inside a chunk the wave
// that owns the block evaluates it for that k on request (block_chain), across chunks the look-back simply waits for
// the inclusive word of the chunk that was entered that way.
// One launch = one RANGE of ordinals: it starts at the raw position where the previous range ended (*carry_in, left by
// the lane that wrote the previous range's last sample) and leaves its own end in *carry_out.
#ifndef ZC_WAVES
#define ZC_WAVES 4
#endif
#ifndef ZC_SLEEP
#define ZC_SLEEP 4
#endif
#define ZC_WG (64 * ZC_WAVES)
#define ZC_PATCH 16
#define ZC_NONE 0xffffffffu
#define ZC_SPIN_CAP (1u << 22)          // polls of one look-back before the status word is flagged (seconds; never met)
#define ZC_VALID (1ull << 63)
#define ZC_K(w, e) ((unsigned)((w) >> (28 * (e))) & 0x3fffu)
#define ZC_C(w, e) ((unsigned)((w) >> (28 * (e) + 14)) & 0x3fffu)
#define ZC_ORD_MASK ((1ull << 48) - 1ull)

struct zc_head {
    unsigned ticket, pad0;
    unsigned long long pad1;
#ifdef ZC_STATS
    unsigned long long polls, folds, cyc_classify, cyc_lookback, cyc_emit, cyc_total, chunks, dist;
#endif
};

__device__ inline unsigned long long zc_ld(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void zc_st(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// state before the first position of block b of a range that starts at raw position *pos_base
__global__ void zig_seek_rel_kernel(uint64_t s_hi, uint64_t s_lo, uint64_t i_hi, uint64_t i_lo,
                                    const unsigned long long *__restrict__ pos_base, long nblk, ulonglong2 *__restrict__ blk_state) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const u128 inc = mk128(i_hi, i_lo);
    u128 s = mk128(s_hi, s_lo);
    const unsigned long long n = *pos_base + (unsigned long long)b * ZIG_BLK;
    for (int i = 0; i < 64 && (n >> i); i++)
        if ((n >> i) & 1) s = jump_apply(ZIG_POW2.v[i], s, inc);
    blk_state[b] = make_ulonglong2((uint64_t)(s >> 64), (uint64_t)s);
}

// the chunk's blocks entered with k: entry and sample offset of every block (when ent != NULL), exit k and sample
// count; false = the function value of block req >> 16 for k = req & 0xffff is not known yet
__device__ inline bool zc_walk(unsigned k, const unsigned (*fun)[2], const uint2 *patch, unsigned np, unsigned *ent, unsigned *off,
                               unsigned &kout, unsigned &tot, unsigned &req) {
    unsigned c = 0;
    for (int w = 0; w < ZC_WAVES; w++) {
        if (ent) {
            ent[w] = k;
            off[w] = c;
        }
        unsigned v = 0;
        if (k < 2) v = fun[w][k];
        else {
            const unsigned key = ((unsigned)w << 16) | k;
            bool hit = false;
            for (unsigned i = 0; i < np && i < ZC_PATCH; i++)
                if (patch[i].x == key) {
                    v = patch[i].y;
                    hit = true;
                }
            if (!hit) {
                req = key;
                return false;
            }
        }
        k = v >> 16;
        c += v & 0xffffu;
    }
    kout = k;
    tot = c;
    return true;
}

// A chunk's function on {0, 1} in one 64-bit word for the look-back: entry e in bits 32 e .. 32 e + 31 as count << 8 | k,
// k = 0xff ("poison") when the exit is >= 2 - such a chunk's successor is not in anybody's table, the walk waits for an
// inclusive word behind it instead.  (Counts of one round of ZC_NWIN windows: < 2^24.)
#define ZC_NWIN 4
#define ZC_IDENT ((1ull << 32) | 0ull)
__device__ inline unsigned long long zc_pack(unsigned long long a) {
    const unsigned k0 = ZC_K(a, 0), k1 = ZC_K(a, 1);
    const unsigned e0 = (ZC_C(a, 0) << 8) | (k0 < 2 ? k0 : 0xffu), e1 = (ZC_C(a, 1) << 8) | (k1 < 2 ? k1 : 0xffu);
    return ((unsigned long long)e1 << 32) | e0;
}
// h = g o f: f (the farther chunks) first
__device__ inline unsigned long long zc_compose(unsigned long long g, unsigned long long f) {
    unsigned h[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const unsigned fe = (unsigned)(f >> (32 * e)), kf = fe & 0xffu;
        const unsigned ge = kf ? (unsigned)(g >> 32) : (unsigned)g;
        h[e] = kf >= 2 ? 0xffu : (((fe >> 8) + (ge >> 8)) << 8) | (ge & 0xffu);
    }
    return ((unsigned long long)h[1] << 32) | h[0];
}
__device__ inline unsigned long long shfl_down64(unsigned long long v, int d) {
    const unsigned lo = __shfl_down((unsigned)v, d), hi = __shfl_down((unsigned)(v >> 32), d);
    return ((unsigned long long)hi << 32) | lo;
}

// wave 0: (entry k, first ordinal) of chunk c from the words of its predecessors.  Every round has the words of
// ZC_NWIN x 64 predecessors in flight together (a poll is a round trip to the memory side of the L2s: the words are
// stored write-through), the windows are evaluated nearest first: an inclusive word ends the walk, 64 aggregates
// without one are composed by a log-step reduction across the lanes and the walk goes on.  false: gave up (status flagged).
__device__ inline bool zc_lookback(long c, const unsigned long long *agg, const unsigned long long *incw, unsigned &k_out,
                                   unsigned long long &ord_out, unsigned *err, zc_head *head = nullptr) {
    const int lane = threadIdx.x & 63;
    for (unsigned spin = 0;; spin++) {
        // pending = the composition of the windows already folded (the chunks between `base` and c), on {0, 1}
        unsigned pk[2] = {0u, 1u};
        unsigned long long pc[2] = {0ull, 0ull};
        bool any_pending = false, failed = false;
        for (long base = c - 1; !failed; base -= 64 * ZC_NWIN) {
#ifdef ZC_STATS
            if (lane == 0) atomicAdd(&head->polls, 1ull);
#endif
            unsigned long long iw[ZC_NWIN], aw[ZC_NWIN];
#pragma unroll
            for (int w = 0; w < ZC_NWIN; w++) {
                const long j = base - 64 * w - lane;
                iw[w] = aw[w] = 0;
                if (j >= 0) {
                    iw[w] = zc_ld(incw + j);
                    aw[w] = zc_ld(agg + j);
                } else if (j == -1) iw[w] = ZC_VALID;             // in front of chunk 0: k = 0, ordinal 0
            }
#pragma unroll
            for (int w = 0; w < ZC_NWIN && !failed; w++) {
                const unsigned long long im = __ballot((iw[w] >> 63) != 0), am = __ballot((aw[w] >> 63) != 0);
                const int li = im ? __builtin_ctzll(im) : 64;       // the nearest inclusive word of the window
                const unsigned long long need = li == 64 ? ~0ull : (1ull << li) - 1ull;
                if ((am & need) != need) {
                    failed = true;                                  // an aggregate on the way is not there yet
                    break;
                }
                // the aggregates of lanes 0 .. li - 1, the nearest (lane 0) applied last
                unsigned long long f = lane < li ? zc_pack(aw[w]) : ZC_IDENT;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned long long far = shfl_down64(f, d);
                    f = zc_compose(f, lane + d < 64 ? far : ZC_IDENT);
                }
                f = readlane64(f, 0);
                if (li < 64) {
                    const unsigned long long w0 = readlane64(iw[w], li);
                    unsigned k = (unsigned)(w0 >> 48) & 0x3fffu;
                    unsigned long long ord = w0 & ZC_ORD_MASK;
                    // (li = 0 and nothing pending: the word of c - 1 itself, whose k may be anything)
                    if (li > 0) {
                        const unsigned fe = k < 2 ? (unsigned)(f >> (32 * k)) : 0xffu;
                        if ((fe & 0xffu) >= 2) {                    // a chunk on the way was entered with k >= 2, or hands one to c:
                            failed = true;                          // its own inclusive word will say which k
                            break;
                        }
                        ord += fe >> 8;
                        k = fe & 0xffu;
                    }
                    if (any_pending) {
                        if (k >= 2) {
                            failed = true;
                            break;
                        }
                        ord += pc[k];
                        k = pk[k];
                    }
                    k_out = k;
                    ord_out = ord;
#ifdef ZC_STATS
                    if (lane == 0) atomicAdd(&head->dist, (unsigned long long)(c - 1 - base + 64 * w + li));
#endif
                    return true;
                }
                // 64 aggregates, no inclusive word: pending <- pending o f
                unsigned nk[2];
                unsigned long long nc[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const unsigned fe = (unsigned)(f >> (32 * e)), kf = fe & 0xffu;
                    if (kf >= 2) failed = true;
                    nk[e] = any_pending ? pk[kf & 1u] : kf;
                    nc[e] = (fe >> 8) + (any_pending ? pc[kf & 1u] : 0ull);
                    if (nk[e] >= 2) failed = true;
                }
                pk[0] = nk[0], pk[1] = nk[1], pc[0] = nc[0], pc[1] = nc[1];
                any_pending = true;
#ifdef ZC_STATS
                if (lane == 0) atomicAdd(&head->folds, 1ull);
#endif
            }
        }
        if (spin >= ZC_SPIN_CAP) {
            if (lane == 0) atomicOr(err, 8u);
            return false;
        }
        // a word is missing, or a chunk on the way was entered with k >= 2 (it publishes its own inclusive word): start
        // again from the nearest window - by then the inclusive words have come closer
        __builtin_amdgcn_s_sleep(ZC_SLEEP);
    }
}

// the samples of one block, from the raw draws kept in LDS: entry `pos`, first ordinal ord_blk (relative to the
// range); ordinals below o_n are written, and whoever writes the range's last one leaves the raw position behind it
__device__ inline void block_emit(const uint64_t *raw, const cols_t &c, uint64_t tt, const double (&tv)[4], unsigned G,
                                  unsigned C1, u128 s_blk, u128 inc, unsigned pos, unsigned long long ord_blk,
                                  unsigned long long o_n, double *__restrict__ g, unsigned long long blk_pos,
                                  unsigned long long *carry_out, const double *wi, unsigned *err) {
    const int lane = threadIdx.x & 63;
    const bool row = lane < ZIG_ROWS;
    const uint64_t nf_col = ((uint64_t)c.nf_hi << 32) | c.nf_lo, z_col = ((uint64_t)c.z_hi << 32) | c.z_lo,
                   w_col = ((uint64_t)c.w_hi << 32) | c.w_lo;
    if (pos < 2 && !(tt & ZIG_TT_OVER)) {
        const unsigned cins = row_carries(G, C1, pos);
        const bool c1 = (cins >> lane) & 1u;
        const vrow_t v = row_eval_vec(nf_col, z_col, w_col, c1 ? 1u : 0u, tt, 64u * lane);
        if (__ballot(row && v.big) == 0) {
            const uint64_t ew_col = row ? v.ew : 0ull, t_col = row ? v.t : 0ull, e_col = row ? v.e : 0ull;
            const unsigned pc = __builtin_popcountll(e_col | t_col);
            const unsigned base_col = row16_scan(pc) - pc;          // samples of the block before row `lane`
            const unsigned trows = (unsigned)__ballot(t_col != 0);  // rows with a tail sample
#pragma unroll
            for (int j = 0; j < ZIG_ROWS; j++) {
                const uint64_t e = readlane64(e_col, j);
                const unsigned long long ord_row = ord_blk + readlane32(base_col, j);
                if ((trows >> j) & 1) {
                    const uint64_t tj = readlane64(t_col, j), all = e | tj;
                    for (uint64_t zz = tj; zz; zz &= zz - 1) {
                        const int pz = __builtin_ctzll(zz);
                        const unsigned q = 64u * j + pz;
                        double v2 = 0.0;
                        unsigned consumed = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const unsigned en = (unsigned)(tt >> (16 * i)) & 0x7fffu;
                            if (en != 0 && (en & 0x3ffu) == q) {
                                v2 = tv[i];
                                consumed = (en >> 10) << 1;
                            }
                        }
                        const unsigned long long o = ord_row + __builtin_popcountll(all & ((1ull << pz) - 1ull));
                        if (lane == 0 && o < o_n) {
                            g[o] = v2;
                            if (o + 1 == o_n) *carry_out = blk_pos + q + 1 + consumed;
                        }
                    }
                    if ((e >> lane) & 1) {
                        const unsigned long long o = ord_row + mbcnt64(all);
                        if (o < o_n) {
                            g[o] = zig_value(raw[64 * j + lane], wi);
                            if (o + 1 == o_n) *carry_out = blk_pos + 64u * j + lane + 1 + ((readlane64(ew_col, j) >> lane) & 1);
                        }
                    }
                } else if ((e >> lane) & 1) {
                    const unsigned long long o = ord_row + mbcnt64(e);
                    if (o < o_n) {
                        g[o] = zig_value(raw[64 * j + lane], wi);
                        if (o + 1 == o_n) *carry_out = blk_pos + 64u * j + lane + 1 + ((readlane64(ew_col, j) >> lane) & 1);
                    }
                }
            }
            return;
        }
    }
    unsigned long long ord = ord_blk;                        // ordinal of the next sample
#pragma unroll
    for (int j = 0; j < ZIG_ROWS; j++) {
        if (pos >= 64u * (j + 1)) continue;
        const uint64_t nf_j = readlane64(nf_col, j), z_j = readlane64(z_col, j), w_j = readlane64(w_col, j);
        while (pos < 64u * (j + 1)) {
            const row_eval_t r = row_eval(nf_j, z_j, w_j, j, pos);
            const uint64_t e = r.e_fast | r.e_wedge;
            if ((e >> lane) & 1) {
                const unsigned long long o = ord + mbcnt64(e);
                if (o < o_n) {
                    g[o] = zig_value(raw[64 * j + lane], wi);
                    if (o + 1 == o_n) *carry_out = blk_pos + 64u * j + lane + 1 + ((r.e_wedge >> lane) & 1);
                }
            }
            ord += __builtin_popcountll(e);
            if (r.tl_bit == 64) break;
            double v;
            unsigned consumed;
            zig_tail_walk(s_blk, inc, 64u * j + r.tl_bit, v, consumed, err);
            if (lane == 0 && ord < o_n) {
                g[ord] = v;
                if (ord + 1 == o_n) *carry_out = blk_pos + 64u * j + r.tl_bit + 1 + consumed;
            }
            ord += 1;
            pos = 64u * j + r.tl_bit + 1 + consumed;
        }
    }
}

__global__ void __launch_bounds__(ZC_WG, 3)
zig_chain_kernel(const ulonglong2 *__restrict__ blk_state, uint64_t i_hi, uint64_t i_lo, long nchunk, zc_head *head,
                 unsigned long long *agg, unsigned long long *incw, const unsigned long long *__restrict__ carry_in,
                 unsigned long long *carry_out, unsigned long long o_n, double *__restrict__ g, zig_status *st) {
    __shared__ zig_lds L;
    __shared__ zig_wlds Wall[ZC_WAVES];
    __shared__ uint64_t s_raw[ZC_WAVES][ZIG_BLK];       // the raw draws of the chunk: position q of block w at [w][q]
    __shared__ unsigned s_fun[ZC_WAVES][2], s_ent[ZC_WAVES], s_off[ZC_WAVES];
    __shared__ uint2 s_patch[ZC_PATCH];
    __shared__ unsigned s_req1, s_req2, s_np, s_skip;   // (one request word per phase: a late reader of phase 1 must not see phase 2's)
    __shared__ unsigned long long s_ord;
    __shared__ long s_chunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    zig_wlds *W = &Wall[wave];
    zig_lds_fill(L);
    const u128 inc = mk128(i_hi, i_lo);
    const u128 c64 = inc * mk128(H_POW2.v[6].ghi, H_POW2.v[6].glo);
    const jump_t lj = ZIG_LANE.v[lane];
    const u128 lane_m = mk128(lj.mhi, lj.mlo), lane_c = inc * mk128(lj.ghi, lj.glo);
    const unsigned long long pos_base = *carry_in;
    long next = 0;
    if (threadIdx.x == 0) next = (long)atomicAdd(&head->ticket, 1u);
    for (;;) {
        if (threadIdx.x == 0) {
            s_chunk = next;
            s_np = 0;
        }
        __syncthreads();
        const long chunk = s_chunk;
        if (chunk >= nchunk) break;
        const long b = chunk * ZC_WAVES + wave;
#ifdef ZC_STATS
        const unsigned long long tA = __builtin_readcyclecounter();
#endif
        const ulonglong2 bs = blk_state[b];
        const u128 s_blk = mk128(uni64(bs.x), uni64(bs.y));
        cols_t c;
        uint64_t *raw = s_raw[wave];
        double tv[4] = {0.0, 0.0, 0.0, 0.0};
        const uint64_t tt = block_classify<true>(c, s_blk, inc, lane_m, lane_c, c64, L, W, &st->error, raw, tv);
        // the block's function on {0, 1} (as pass 1 of the two-pass form)
        unsigned G, C1;
        {
            const bool row = lane < ZIG_ROWS;
            const uint64_t nf = ((uint64_t)c.nf_hi << 32) | c.nf_lo, z = ((uint64_t)c.z_hi << 32) | c.z_lo,
                           w = ((uint64_t)c.w_hi << 32) | c.w_lo;
            const vrow_t v0 = row_eval_vec(nf, z, w, 0u, tt, 64u * lane), v1 = row_eval_vec(nf, z, w, 1u, tt, 64u * lane);
            G = (unsigned)__ballot(row && v0.cout);
            C1 = (unsigned)__ballot(row && v1.cout);
            const unsigned p0 = row ? __builtin_popcountll(v0.e | v0.t) : 0u;
            const unsigned p1 = row ? __builtin_popcountll(v1.e | v1.t) : 0u;
            const unsigned sum0 = row16_sum(p0);
            const bool over = (tt & ZIG_TT_OVER) != 0;
            for (unsigned e = 0; e < 2; e++) {
                const unsigned cins = row_carries(G, C1, e);
                const bool c1 = (cins >> lane) & 1u;
                const bool slow = over || __ballot(row && (c1 ? v1.big : v0.big)) != 0;
                unsigned kk, cnt;
                if (!slow) {
                    kk = cins >> 16;
                    cnt = sum0 + row16_sum(c1 && row ? p1 - p0 : 0u);
                } else {
                    unsigned pp, cc;
                    block_chain(c, s_blk, inc, e, kk, cnt, pp, cc, ZIG_ROWS, &st->error);
                }
                if (lane == 0) s_fun[wave][e] = (kk << 16) | cnt;
            }
        }
        __syncthreads();
        // the chunk's aggregate; a block entered with k >= 2 is evaluated by its wave on request
        unsigned ak[2] = {0, 0}, ac[2] = {0, 0};
        for (;;) {
            if (threadIdx.x == 0) {
                unsigned req = ZC_NONE;
                const bool ok = zc_walk(0u, s_fun, s_patch, s_np, nullptr, nullptr, ak[0], ac[0], req) &&
                                zc_walk(1u, s_fun, s_patch, s_np, nullptr, nullptr, ak[1], ac[1], req);
                if (ok) {
                    zc_st(agg + chunk, ZC_VALID | (unsigned long long)(ak[0] & 0x3fffu) | ((unsigned long long)ac[0] << 14) |
                                           ((unsigned long long)(ak[1] & 0x3fffu) << 28) | ((unsigned long long)ac[1] << 42));
                    if ((ak[0] | ak[1]) > 0x3fffu) atomicOr(&st->error, 1u);
                } else if (s_np >= ZC_PATCH) {
                    atomicOr(&st->error, 2u);
                    req = ZC_NONE;
                    zc_st(agg + chunk, ZC_VALID);
                }
                s_req1 = ok ? ZC_NONE : req;
            }
            __syncthreads();
            const unsigned req = s_req1;
            if (req == ZC_NONE) break;
            if ((unsigned)wave == (req >> 16)) {
                unsigned kb, total, pp, cc;
                block_chain(c, s_blk, inc, req & 0xffffu, kb, total, pp, cc, ZIG_ROWS, &st->error);
                if (lane == 0) {
                    s_patch[s_np] = make_uint2(req, (kb << 16) | total);
                    s_np = s_np + 1;
                }
            }
            __syncthreads();
        }
        // look-back (wave 0), then the entries of the chunk's blocks; the inclusive word goes out before any sample
        unsigned k_c = 0;
        unsigned long long ord_c = 0;
        bool lb_ok = true;
#ifdef ZC_STATS
        const unsigned long long tB = __builtin_readcyclecounter();
#endif
#if defined(ZC_ABLATE) && ZC_ABLATE == 1      // (timing only, wrong ordinals: no look-back at all - the work of the kernel by itself)
        ord_c = (unsigned long long)chunk * 4000ull;
#else
        if (wave == 0) lb_ok = zc_lookback(chunk, agg, incw, k_c, ord_c, &st->error, head);
#endif
#ifdef ZC_STATS
        const unsigned long long tC = __builtin_readcyclecounter();
#endif
        for (;;) {
            if (threadIdx.x == 0) {
                unsigned req = ZC_NONE, kx = 0, cx = 0;
                bool ok = true;
                if (!lb_ok) {
                    zc_st(incw + chunk, ZC_VALID);               // (status flagged: let the successors finish)
                    s_skip = 1;
                } else {
                    ok = zc_walk(k_c, s_fun, s_patch, s_np, s_ent, s_off, kx, cx, req);
                    if (!ok && s_np >= ZC_PATCH) {
                        atomicOr(&st->error, 2u);
                        ok = true;
                        lb_ok = false;
                        zc_st(incw + chunk, ZC_VALID);
                        s_skip = 1;
                    } else if (ok) {
                        zc_st(incw + chunk, ZC_VALID | ((unsigned long long)(kx & 0x3fffu) << 48) | ((ord_c + cx) & ZC_ORD_MASK));
                        s_ord = ord_c;
                        s_skip = ord_c >= o_n ? 1u : 0u;           // (the range ended in front of this chunk)
                        if (chunk == nchunk - 1) {
                            st->total = ord_c + cx;
                            if (ord_c + cx < o_n) atomicOr(&st->error, 16u);   // the range's blocks did not hold its normals
                        }
                    }
                }
                s_req2 = ok ? ZC_NONE : req;
            }
            __syncthreads();
            const unsigned req = s_req2;
            if (req == ZC_NONE) break;
            if ((unsigned)wave == (req >> 16)) {
                unsigned kb, total, pp, cc;
                block_chain(c, s_blk, inc, req & 0xffffu, kb, total, pp, cc, ZIG_ROWS, &st->error);
                if (lane == 0) {
                    s_patch[s_np] = make_uint2(req, (kb << 16) | total);
                    s_np = s_np + 1;
                }
            }
            __syncthreads();
        }
        // the next ticket is taken only now, behind this chunk's inclusive word: a ticket held by a workgroup that still
        // waits in its look-back is a chunk nobody classifies, and every later chunk waits for its aggregate (taken one
        // chunk ahead, the scan ran at 100 polls per chunk); here the atomic's latency hides behind the emission
        if (threadIdx.x == 0) next = (long)atomicAdd(&head->ticket, 1u);
        if (!s_skip)
            block_emit(raw, c, tt, tv, G, C1, s_blk, inc, s_ent[wave], s_ord + s_off[wave], o_n, g,
                       pos_base + (unsigned long long)b * ZIG_BLK, carry_out, L.wi, &st->error);
#ifdef ZC_STATS
        if (threadIdx.x == 0) {
            const unsigned long long tD = __builtin_readcyclecounter();
            atomicAdd(&head->cyc_classify, tB - tA);
            atomicAdd(&head->cyc_lookback, tC - tB);
            atomicAdd(&head->cyc_emit, tD - tC);
            atomicAdd(&head->chunks, 1ull);
        }
#endif
    }
}

__global__ void zig_debug_kernel(const ulonglong2 *blk_state, uint64_t i_hi, uint64_t i_lo, long b, uint64_t *out) {
    const u128 inc = mk128(i_hi, i_lo);
    const int lane = threadIdx.x & 63;
    const jump_t lj = ZIG_LANE.v[lane];
    const u128 lane_m = mk128(lj.mhi, lj.mlo), lane_c = inc * mk128(lj.ghi, lj.glo);
    const ulonglong2 bs = blk_state[b];
    const u128 s_blk = mk128(uni64(bs.x), uni64(bs.y));
    const u128 s = lane_m * s_blk + lane_c;
    out[lane] = pcg_out(s);
    const u128 s2 = jump_apply(ZIG_LANE.v[lane], mk128(bs.x, bs.y), inc);
    out[64 + lane] = pcg_out(s2);
    constexpr u128 M64 = mk128(H_POW2.v[6].mhi, H_POW2.v[6].mlo);
    const u128 c64 = inc * mk128(H_POW2.v[6].ghi, H_POW2.v[6].glo);
    out[128 + lane] = pcg_out(s2 * M64 + c64);
}

}  // namespace

// ---- host side: one round = the tables of nblk blocks from raw position pos0 on ------------------------------------------
struct zig_round {
    long nblk = 0, ntile = 0;
    unsigned patch_cap = 0, grid = 0;
    ulonglong2 *blk_state = nullptr, *entry = nullptr, *patch = nullptr, *classes = nullptr, *tile_entry = nullptr;
    unsigned *fun = nullptr;
    unsigned long long *tails = nullptr;
    uint4 *tile_fun = nullptr;
    zig_status *st = nullptr;
    char *extra = nullptr;          // `extra_bytes` more, 16-byte aligned, behind the status word (range tables of the l-range pipeline)
};
// the block tables of a round that has to yield `want` normals, cut from scratch slot 6
static int zig_round_tables(corahip_ctx *ctx, unsigned long long want, size_t extra_bytes, zig_round &rd) {
    // 1.02145 raw draws per normal on average; the margin covers 200 sigma, and a short round is followed by another
    rd.nblk = (long)((want + want / 44 + 2 * ZIG_BLK) / ZIG_BLK) + 1;
    rd.patch_cap = (unsigned)(rd.nblk / 64 + 1024);
    rd.ntile = (rd.nblk + ZIG_TILE - 1) / ZIG_TILE;
    const size_t nblk = (size_t)rd.nblk, ntile = (size_t)rd.ntile;
    const size_t off_fun = sizeof(ulonglong2) * nblk;
    const size_t off_entry = off_fun + sizeof(unsigned) * 2 * nblk;
    const size_t off_patch = off_entry + sizeof(ulonglong2) * nblk;
    const size_t off_cls = off_patch + sizeof(ulonglong2) * rd.patch_cap;
    const size_t off_tt = off_cls + sizeof(ulonglong2) * ZIG_ROWS * nblk;
    const size_t off_tf = off_tt + sizeof(unsigned long long) * nblk;
    const size_t off_te = off_tf + sizeof(uint4) * ntile;
    const size_t off_st = off_te + sizeof(ulonglong2) * ntile;
    const size_t off_ex = (off_st + sizeof(zig_status) + 15) & ~(size_t)15;
    char *ws = nullptr;
    int rc = corahip_ctx_scratch(ctx, 6, off_ex + extra_bytes, (void **)&ws);
    if (rc) return rc;
    rd.blk_state = (ulonglong2 *)ws;
    rd.fun = (unsigned *)(ws + off_fun);
    rd.entry = (ulonglong2 *)(ws + off_entry);
    rd.patch = (ulonglong2 *)(ws + off_patch);
    rd.classes = (ulonglong2 *)(ws + off_cls);
    rd.tails = (unsigned long long *)(ws + off_tt);
    rd.tile_fun = (uint4 *)(ws + off_tf);
    rd.tile_entry = (ulonglong2 *)(ws + off_te);
    rd.st = (zig_status *)(ws + off_st);
    rd.extra = ws + off_ex;
    rd.grid = (unsigned)std::min<long>((rd.nblk + ZIG_WG / 64 - 1) / (ZIG_WG / 64), (long)ctx->num_cu * 32);
    return 0;
}
// seek + pass 1 + the scan over the block functions, on `stream`: afterwards every block has its (k, first ordinal)
static int zig_round_count_scan(corahip_ctx *ctx, hipStream_t stream, const uint64_t state[2], const uint64_t inc[2],
                                unsigned long long pos0, unsigned long long ord0, const zig_round &rd) {
    const bool other = stream != ctx->stream;
    HIP_TRY(hipMemsetAsync(rd.st, 0, sizeof(zig_status), stream));
    {
        StageTimer t0(ctx, "zig_seek", stream, other);
        zig_seek_kernel<<<(unsigned)((rd.nblk + 255) / 256), 256, 0, stream>>>(state[0], state[1], inc[0], inc[1], pos0, rd.nblk,
                                                                              rd.blk_state);
        LAUNCH_CHECK();
    }
    {
        StageTimer t1(ctx, "zig_count", stream, other);
        zig_count_kernel<<<rd.grid, ZIG_WG, 0, stream>>>(rd.blk_state, inc[0], inc[1], rd.nblk, rd.fun, rd.classes, rd.tails, rd.patch,
                                                        rd.patch_cap, rd.st);
        LAUNCH_CHECK();
    }
    {
        StageTimer t2(ctx, "zig_scan", stream, other);
        const unsigned tgrid = (unsigned)((rd.ntile + 3) / 4);
        zig_tile_kernel<<<tgrid, 256, 0, stream>>>(rd.nblk, rd.ntile, rd.fun, rd.patch, rd.patch_cap, rd.tile_fun, rd.st);
        LAUNCH_CHECK();
        const char *fs = getenv("CORAHIP_ZIG_TOP_SERIAL");       // (tests: the serial walk that takes over when a thread hands on k >= 2)
        zig_top_kernel<<<1, ZIG_TOP_T, 0, stream>>>(rd.nblk, rd.ntile, rd.fun, rd.patch, rd.patch_cap, rd.tile_fun, ord0, rd.tile_entry,
                                                   rd.st, fs && atoi(fs) ? 1 : 0);
        LAUNCH_CHECK();
        zig_entry_kernel<<<tgrid, 256, 0, stream>>>(rd.nblk, rd.ntile, rd.fun, rd.patch, rd.patch_cap, rd.tile_entry, rd.entry, rd.st);
        LAUNCH_CHECK();
    }
    return 0;
}
static void zig_debug_dump(corahip_ctx *ctx, const zig_round &rd, const uint64_t inc[2], const zig_status &hs) {
    const long nb = std::min<long>(rd.nblk, 8);
    std::vector<unsigned> hf(2 * nb);
    std::vector<ulonglong2> he(nb), hb(nb);
    uint4 tf;
    ulonglong2 te;
    (void)hipMemcpy(hf.data(), rd.fun, sizeof(unsigned) * 2 * nb, hipMemcpyDeviceToHost);
    (void)hipMemcpy(he.data(), rd.entry, sizeof(ulonglong2) * nb, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), rd.blk_state, sizeof(ulonglong2) * nb, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&tf, rd.tile_fun, sizeof(tf), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&te, rd.tile_entry, sizeof(te), hipMemcpyDeviceToHost);
    fprintf(stderr, "zig debug: nblk %ld ntile %ld total %llu k_last %u npatch %u err %u n_raw %llu | tile0 fun (%u %u %u %u) entry (%llu %llu)\n",
            rd.nblk, rd.ntile, hs.total, hs.k_last, hs.npatch, hs.error, hs.n_raw, tf.x, tf.y, tf.z, tf.w, te.x, te.y);
    {
        uint64_t *dbg = nullptr, h[192];
        (void)hipMalloc(&dbg, sizeof(h));
        zig_debug_kernel<<<1, 64, 0, ctx->stream>>>(rd.blk_state, inc[0], inc[1], 1, dbg);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        fprintf(stderr, "  blk 1 raws: hoisted %016llx %016llx %016llx | table %016llx %016llx %016llx | row 1 %016llx %016llx\n",
                (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[63], (unsigned long long)h[64],
                (unsigned long long)h[65], (unsigned long long)h[127], (unsigned long long)h[128], (unsigned long long)h[129]);
    }
    for (long i = 0; i < nb; i++)
        fprintf(stderr, "  blk %ld: state %016llx%016llx fun0 (k %u cnt %u) fun1 (k %u cnt %u) entry (k %llu ord %llu)\n", i,
                hb[i].x, hb[i].y, hf[2 * i] >> 16, hf[2 * i] & 0xffff, hf[2 * i + 1] >> 16, hf[2 * i + 1] & 0xffff,
                he[i].x, he[i].y);
}

// ---- the stream in ranges, two-pass form (the default; CORAHIP_ZIG_ONEPASS=1 takes the single pass below) ---------------
struct zig2_session {
    zig_round rd;
    uint64_t state[2], inc[2];
    unsigned long long n = 0;
    std::vector<unsigned long long> bounds;      // [nr + 1] ordinals; kept until finish (the upload is asynchronous)
    unsigned long long *d_bounds = nullptr;
    long *d_first = nullptr;
    int nr = 0;
};

static int zig2_stream_prepare(corahip_ctx *ctx, hipStream_t stream, const uint64_t state[2], const uint64_t inc[2], int64_t n,
                       const std::vector<unsigned long long> &bounds, zig2_session **out) {
    ARG_CHECK(n > 0 && bounds.size() >= 2 && bounds.front() == 0 && bounds.back() == (unsigned long long)n);
    zig2_session *s = new zig2_session();
    s->n = (unsigned long long)n;
    s->bounds = bounds;
    s->nr = (int)bounds.size() - 1;
    for (int i = 0; i < 2; i++) {
        s->state[i] = state[i];
        s->inc[i] = inc[i];
    }
    const size_t nb = bounds.size();
    int rc = zig_round_tables(ctx, s->n, (sizeof(unsigned long long) + sizeof(long)) * nb, s->rd);
    if (!rc) {
        s->d_bounds = (unsigned long long *)s->rd.extra;
        s->d_first = (long *)(s->rd.extra + sizeof(unsigned long long) * nb);
        rc = zig_round_count_scan(ctx, stream, state, inc, 0ull, 0ull, s->rd);
    }
    if (!rc) {
        hipError_t e = hipMemcpyAsync(s->d_bounds, s->bounds.data(), sizeof(unsigned long long) * nb, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) rc = (int)e;
    }
    if (rc) {
        delete s;
        return rc;
    }
    zig_range_kernel<<<(s->nr + 63) / 64, 64, 0, stream>>>(s->rd.entry, s->rd.nblk, s->d_bounds, s->nr, s->d_first);
    if (hipGetLastError() != hipSuccess) {
        delete s;
        corahip_set_error("zig_range_kernel launch failed");
        return CORAHIP_ESTATE;
    }
    *out = s;
    return 0;
}

static int zig2_stream_emit_range(corahip_ctx *ctx, hipStream_t stream, zig2_session *s, int r, double *slot) {
    ARG_CHECK(s != nullptr && r >= 0 && r < s->nr && slot != nullptr);
    const unsigned long long o_lo = s->bounds[r], o_hi = s->bounds[r + 1];
    // one wave per block of ~1000 normals; the grid covers the range's blocks (+ slack: the loop strides, nothing is missed)
    const long est = (long)((o_hi - o_lo) / 960) + 8;
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((est + ZIG_WG / 64 - 1) / (ZIG_WG / 64), (long)ctx->num_cu * 32));
    StageTimer t3(ctx, "zig_emit", stream, stream != ctx->stream);
    zig_emit_kernel<<<grid, ZIG_WG, 0, stream>>>(s->rd.blk_state, s->inc[0], s->inc[1], s->rd.nblk, s->rd.entry, s->rd.classes,
                                                s->rd.tails, 0ull, s->n, o_lo, o_hi, s->d_first + r, slot, s->rd.st);
    LAUNCH_CHECK();
    return 0;
}

static int zig2_stream_finish(corahip_ctx *ctx, hipStream_t stream, zig2_session *s, uint64_t *n_raw) {
    ARG_CHECK(s != nullptr && n_raw != nullptr);
    zig_status hs;
    HIP_TRY(hipMemcpyAsync(&hs, s->rd.st, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (hs.error) {
        corahip_set_error("normals_pcg64 (ranges): device status %u (1 tail loop cap, 2 patch list full, 4 patch missing)", hs.error);
        return CORAHIP_ESTATE;
    }
    if (hs.total < s->n) {      // (the margin of the block count is 200 sigma: never met)
        corahip_set_error("normals_pcg64 (ranges): the raw range held %llu normals, %llu needed", hs.total, s->n);
        return CORAHIP_ESTATE;
    }
    *n_raw = hs.n_raw;
    return 0;
}

static void zig2_stream_free(zig2_session *s) { delete s; }


// ---- the stream in ranges (stream_internal.h), single pass: every range is one memset + seek + chain launch ---------------
struct zig_session {
    zig2_session *two = nullptr;                 // the default (two-pass) form; NULL: CORAHIP_ZIG_ONEPASS=1
    uint64_t state[2], inc[2];
    unsigned long long n = 0;
    std::vector<unsigned long long> bounds;      // [nr + 1] ordinals
    int nr = 0;
    long nblk_cap = 0;                           // blocks the tables hold (the largest range)
    ulonglong2 *blk_state = nullptr;
    zc_head *head = nullptr;                     // head | agg[nchunk] | inc[nchunk]: zeroed in front of every range
    unsigned long long *agg = nullptr, *incw = nullptr, *carry = nullptr;   // carry[r] = raw position where range r starts
    zig_status *st = nullptr;
    size_t zero_bytes = 0;
};
// blocks of a range of `want` normals: 1.02145 raw draws per normal on average; the margin (0.13 % + two blocks) is
// 85 sigma for a 1 GiB slot, more for smaller ranges; whole chunks
static long zc_blocks(unsigned long long want) {
    const long nblk = (long)((want + want / 44 + 2 * ZIG_BLK) / ZIG_BLK) + 1;
    return (nblk + ZC_WAVES - 1) / ZC_WAVES * ZC_WAVES;
}
// The two-pass form is the default: measured at cfg 3 (HISTORY.md, round 6) the single pass takes 14.5 ms against 6.3 -
// its work alone (look-back ablated) is 5.5 ms, and the chunks wait ~19 polls each for the aggregates of stragglers that
// the LDS-bound run-ahead (768 chunks) cannot absorb.  CORAHIP_ZIG_ONEPASS=1 selects it (tests/test_gpu_npnormal.py runs both).
static bool zig_two_pass() {
    const char *e = getenv("CORAHIP_ZIG_ONEPASS");
    return !(e && atoi(e) != 0);
}

int zig_stream_prepare(corahip_ctx *ctx, hipStream_t stream, const uint64_t state[2], const uint64_t inc[2], int64_t n,
                       const std::vector<unsigned long long> &bounds, zig_session **out) {
    ARG_CHECK(n > 0 && bounds.size() >= 2 && bounds.front() == 0 && bounds.back() == (unsigned long long)n);
    zig_session *s = new zig_session();
    if (zig_two_pass()) {
        const int rc = zig2_stream_prepare(ctx, stream, state, inc, n, bounds, &s->two);
        if (rc) {
            delete s;
            return rc;
        }
        *out = s;
        return 0;
    }
    s->n = (unsigned long long)n;
    s->bounds = bounds;
    s->nr = (int)bounds.size() - 1;
    for (int i = 0; i < 2; i++) {
        s->state[i] = state[i];
        s->inc[i] = inc[i];
    }
    unsigned long long widest = 0;
    for (int r = 0; r < s->nr; r++) {
        if (bounds[r + 1] <= bounds[r]) {
            delete s;
            corahip_set_error("normals_pcg64 (ranges): empty range %d", r);
            return CORAHIP_EINVAL;
        }
        widest = std::max(widest, bounds[r + 1] - bounds[r]);
    }
    if (widest >= ZC_ORD_MASK / 2) {
        delete s;
        corahip_set_error("normals_pcg64 (ranges): a range of %llu normals is beyond the ordinal field of the scan", widest);
        return CORAHIP_EINVAL;
    }
    s->nblk_cap = zc_blocks(widest);
    const size_t nblk = (size_t)s->nblk_cap, nchunk = nblk / ZC_WAVES;
    const size_t off_head = sizeof(ulonglong2) * nblk;
    const size_t off_agg = off_head + sizeof(zc_head);
    const size_t off_inc = off_agg + sizeof(unsigned long long) * nchunk;
    const size_t off_carry = off_inc + sizeof(unsigned long long) * nchunk;
    const size_t off_st = (off_carry + sizeof(unsigned long long) * (s->nr + 1) + 15) & ~(size_t)15;
    char *ws = nullptr;
    int rc = corahip_ctx_scratch(ctx, 6, off_st + sizeof(zig_status), (void **)&ws);
    if (rc) {
        delete s;
        return rc;
    }
    s->blk_state = (ulonglong2 *)ws;
    s->head = (zc_head *)(ws + off_head);
    s->agg = (unsigned long long *)(ws + off_agg);
    s->incw = (unsigned long long *)(ws + off_inc);
    s->carry = (unsigned long long *)(ws + off_carry);
    s->st = (zig_status *)(ws + off_st);
    s->zero_bytes = off_carry - off_head;
    // carry[0] = 0: the first range starts at the generator's position; the status word
    hipError_t e = hipMemsetAsync(s->carry, 0, off_st + sizeof(zig_status) - off_carry, stream);
    if (e != hipSuccess) {
        delete s;
        corahip_set_error("normals_pcg64 (ranges): memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    *out = s;
    return 0;
}

int zig_stream_emit_range(corahip_ctx *ctx, hipStream_t stream, zig_session *s, int r, double *slot) {
    ARG_CHECK(s != nullptr && slot != nullptr);
    if (s->two) return zig2_stream_emit_range(ctx, stream, s->two, r, slot);
    ARG_CHECK(r >= 0 && r < s->nr);
    const unsigned long long o_n = s->bounds[r + 1] - s->bounds[r];
    const long nblk = zc_blocks(o_n), nchunk = nblk / ZC_WAVES;
    ARG_CHECK(nblk <= s->nblk_cap);
    StageTimer t3(ctx, "zig_chain", stream, stream != ctx->stream);
    HIP_TRY(hipMemsetAsync(s->head, 0, sizeof(zc_head) + 2 * sizeof(unsigned long long) * (size_t)(s->nblk_cap / ZC_WAVES), stream));
    zig_seek_rel_kernel<<<(unsigned)((nblk + 255) / 256), 256, 0, stream>>>(s->state[0], s->state[1], s->inc[0], s->inc[1], s->carry + r,
                                                                          nblk, s->blk_state);
    LAUNCH_CHECK();
    // tickets: a workgroup that is not resident yet has no chunk, so the grid may exceed what fits
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>(nchunk, (long)ctx->num_cu * 4));
    zig_chain_kernel<<<grid, ZC_WG, 0, stream>>>(s->blk_state, s->inc[0], s->inc[1], nchunk, s->head, s->agg, s->incw, s->carry + r,
                                                s->carry + r + 1, o_n, slot, s->st);
    LAUNCH_CHECK();
    return 0;
}

int zig_stream_finish(corahip_ctx *ctx, hipStream_t stream, zig_session *s, uint64_t *n_raw) {
    ARG_CHECK(s != nullptr && n_raw != nullptr);
    if (s->two) return zig2_stream_finish(ctx, stream, s->two, n_raw);
    zig_status hs;
    unsigned long long last = 0;
    HIP_TRY(hipMemcpyAsync(&hs, s->st, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&last, s->carry + s->nr, sizeof(last), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
#ifdef ZC_STATS
    {
        zc_head hh;
        (void)hipMemcpy(&hh, s->head, sizeof(hh), hipMemcpyDeviceToHost);
        const double nc = (double)std::max<unsigned long long>(1, hh.chunks);
        fprintf(stderr, "zc stats (last range): chunks %llu polls/chunk %.2f folds/chunk %.3f distance %.1f | cycles per chunk (100 MHz clock): classify %.0f "
                "lookback %.0f emit %.0f\n", hh.chunks, hh.polls / nc, hh.folds / nc, hh.dist / nc, hh.cyc_classify / nc, hh.cyc_lookback / nc,
                hh.cyc_emit / nc);
    }
#endif
#ifdef ZC_ABLATE
    *n_raw = 1;
    return 0;
#endif
    if (hs.error) {
        corahip_set_error("normals_pcg64: device status %u (1 tail loop cap, 2 patch list full, 8 look-back gave up, 16 a range's raw blocks "
                          "did not hold its normals)", hs.error);
        return CORAHIP_ESTATE;
    }
    if (last == 0) {            // (nobody wrote the last sample: a range's blocks did not hold its normals - 85 sigma)
        corahip_set_error("normals_pcg64: the raw range of a slot did not hold its normals (%llu counted in the last one)", hs.total);
        return CORAHIP_ESTATE;
    }
    *n_raw = last;
    return 0;
}

void zig_stream_free(zig_session *s) {
    if (s && s->two) zig2_stream_free(s->two);
    delete s;
}

extern "C" {

int corahip_glibc_exp(corahip_ctx *ctx, const double *x, int64_t n, double *y) {
    ARG_CHECK(ctx != nullptr && n >= 0 && (n == 0 || (x != nullptr && y != nullptr)));
    if (n == 0) return 0;
    glibc_exp_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(x, (long)n, y);
    LAUNCH_CHECK();
    return 0;
}

int corahip_pcg64_advance(const uint64_t state[2], const uint64_t inc[2], uint64_t delta, uint64_t out_state[2]) {
    ARG_CHECK(state != nullptr && inc != nullptr && out_state != nullptr);
    const u128 s = host_advance(mk128(state[0], state[1]), mk128(inc[0], inc[1]), delta);
    out_state[0] = (uint64_t)(s >> 64);
    out_state[1] = (uint64_t)s;
    return 0;
}

int corahip_normals_pcg64(corahip_ctx *ctx, const uint64_t state[2], const uint64_t inc[2], int64_t n, double *g,
                          uint64_t *n_raw) {
    ARG_CHECK(ctx != nullptr && state != nullptr && inc != nullptr && n >= 0 && n_raw != nullptr);
    ARG_CHECK(n == 0 || g != nullptr);
    *n_raw = 0;
    if (n == 0) return 0;
    if (ctx->draw_pending) {
        corahip_set_error("normals_pcg64: a corahip_draw_alm_numpy_begin session is pending (its tables share this call's scratch)");
        return CORAHIP_ESTATE;
    }
    StageTimer timer(ctx, "normals_pcg64");
    if (!zig_two_pass()) {                      // the whole stream as one range of the single-pass form
        zig_session *s = nullptr;
        int rc = zig_stream_prepare(ctx, ctx->stream, state, inc, n, std::vector<unsigned long long>{0ull, (unsigned long long)n}, &s);
        if (rc) return rc;
        rc = zig_stream_emit_range(ctx, ctx->stream, s, 0, g);
        uint64_t raw = 0;
        if (!rc) rc = zig_stream_finish(ctx, ctx->stream, s, &raw);
        else (void)hipStreamSynchronize(ctx->stream);
        zig_stream_free(s);
        if (!rc) *n_raw = raw;
        return rc;
    }
    unsigned long long pos0 = 0, ord0 = 0;
    for (int round = 0; round < 64; round++) {
        zig_round rd;
        int rc = zig_round_tables(ctx, (unsigned long long)n - ord0, 0, rd);
        if (rc) return rc;
        if ((rc = zig_round_count_scan(ctx, ctx->stream, state, inc, pos0, ord0, rd))) return rc;
        {
            StageTimer t3(ctx, "zig_emit");
            zig_emit_kernel<<<rd.grid, ZIG_WG, 0, ctx->stream>>>(rd.blk_state, inc[0], inc[1], rd.nblk, rd.entry, rd.classes, rd.tails,
                                                                pos0, (unsigned long long)n, 0ull, (unsigned long long)n, nullptr,
                                                                g, rd.st);
            LAUNCH_CHECK();
        }
        zig_status hs;
        HIP_TRY(hipMemcpyAsync(&hs, rd.st, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (getenv("CORAHIP_ZIG_DEBUG")) zig_debug_dump(ctx, rd, inc, hs);
        if (hs.error) {
            corahip_set_error("normals_pcg64: device status %u (1 tail loop cap, 2 patch list full, 4 patch missing)", hs.error);
            return CORAHIP_ESTATE;
        }
        if (hs.total >= (unsigned long long)n) {
            *n_raw = hs.n_raw;
            return 0;
        }
        pos0 += (unsigned long long)rd.nblk * ZIG_BLK + hs.k_last;
        ord0 = hs.total;
    }
    corahip_set_error("normals_pcg64: no convergence");
    return CORAHIP_ESTATE;
}

}  // extern "C"
