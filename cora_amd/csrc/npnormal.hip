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
//                  (fdlibm's algorithm in glibc's evaluation order, no contraction): bit-identical to numpy on glibc.
//
// "Which raw position starts a sample" is a prefix problem; oracle/npnormal_model.py states the decomposition in python
// (checked against numpy on the CPU), this is the same thing on the device:
//
//   chunk  R consecutive positions per thread, classified AS IF each started a sample: nf (not fast), z (tail class),
//          wacc (wedge test of (p, p + 1) passes).  Entered with k positions already consumed by an earlier sample,
//          position p >= k starts a sample iff it is not the second draw of a wedge sample: inside a maximal run of nf
//          positions the starts alternate from the run's first position - found for all positions at once with an
//          integer add (the carry runs through a run of ones; simdjson's odd-backslash scan).  A tail-class start is
//          resolved by its owner reading on past its chunk, like the wedge draw of a chunk's last position: the state
//          at a chunk boundary is just k = positions of the next chunk(s) already consumed.
//   block  256 chunks; k of thread t + 1 = k_out of thread t, by fixed-point iteration from k = 0 (a fast position ends
//          every dependency chain: 2-3 iterations).
//   grid   zig_count_kernel: every block's (k_out, count) for block entry k in {0, 1}; a block that hands k >= 2 to its
//          successor (a tail sample across the boundary, 6e-4 of the blocks) evaluates the successor for that k itself
//          and appends the result to a patch list.  zig_scan_kernel (one workgroup) composes the block functions and
//          gives every block its true (k, first ordinal).  zig_emit_kernel re-runs every block with them and writes the
//          normals at their ordinals through an LDS staging buffer (coalesced stores).
#include "common.h"

#include <algorithm>

#define ZIG_TAB_Q __device__ static const
#include "zig_tab.inc"

typedef unsigned __int128 u128;

#define ZIG_R_CHUNK 16      // positions per thread
#define ZIG_T 256           // threads per block
#define ZIG_BLK (ZIG_R_CHUNK * ZIG_T)

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
// n = t R: the offset of thread t's chunk inside its block
template <int R, int T>
constexpr jump_tab<T> make_thr() {
    jump_tab<T> t{};
    u128 m = 1, g = 0;
    for (int i = 0; i < T; i++) {
        t.v[i] = jump_t{(uint64_t)(m >> 64), (uint64_t)m, (uint64_t)(g >> 64), (uint64_t)g};
        for (int r = 0; r < R; r++) {
            g = g * PCG_MULT + 1;
            m = m * PCG_MULT;
        }
    }
    return t;
}
constexpr jump_tab<64> H_POW2 = make_pow2();
__device__ const jump_tab<64> ZIG_POW2 = make_pow2();
__device__ const jump_tab<ZIG_T> ZIG_THR = make_thr<ZIG_R_CHUNK, ZIG_T>();

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
// (tools/log1p_probe.py compares the same restatement with math.log1p on the host).
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
    return lhs < exp(-0.5 * x * x);
}

struct zig_status {
    unsigned long long total;      // ordinal after the last block of the round
    unsigned long long n_raw;      // raw draws consumed by the first n normals (absolute position count)
    unsigned k_last, npatch, error, pad;
};

#define ZIG_TAIL_CAP 4096   // iterations of one tail loop before the status word is flagged (rejection rate 8 %)

template <int R>
struct chunk_t {
    uint64_t raw[R + 1];
    u128 s0;               // state before the chunk's first position
    unsigned nf, z, wacc;
};

template <int R>
__device__ inline void chunk_build(chunk_t<R> &c, u128 s, u128 inc, const uint64_t *ki, const double *wi,
                                   const double *fi) {
    c.s0 = s;
#pragma unroll
    for (int j = 0; j <= R; j++) {
        s = pcg_step(s, inc);
        c.raw[j] = pcg_out(s);
    }
    unsigned nf = 0, z = 0;
#pragma unroll
    for (int p = 0; p < R; p++) {
        const uint64_t r = c.raw[p];
        const unsigned idx = (unsigned)(r & 0xff);
        const bool slow = ((r >> 9) & M52) >= ki[idx];
        nf |= (slow ? 1u : 0u) << p;
        z |= ((slow && idx == 0) ? 1u : 0u) << p;
    }
    unsigned wacc = 0;
    const unsigned w = nf & ~z;
    if (w) {
#pragma unroll
        for (int p = 0; p < R; p++)
            if ((w >> p) & 1) wacc |= (zig_wedge_accept(c.raw[p], c.raw[p + 1], wi, fi) ? 1u : 0u) << p;
    }
    c.nf = nf;
    c.z = z;
    c.wacc = wacc;
}

// tail sample started at position p of the chunk (r_p its draw): value, and the positions consumed after p
__device__ __attribute__((noinline)) void zig_tail_walk(u128 s, u128 inc, int p, uint64_t r_p, double &val, unsigned &consumed,
                                                        unsigned *err) {
#pragma clang fp contract(off)
    for (int i = 0; i <= p; i++) s = pcg_step(s, inc);
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
    const bool neg = (((r_p >> 9) & M52) >> 8) & 1;
    val = neg ? -(ZIG_NOR_R + xx) : ZIG_NOR_R + xx;
    consumed = c;
}

struct emit_ctx {
    double *stage;                 // LDS staging buffer of the block (padded index)
    unsigned obase;                // ordinal of the thread's first sample relative to the block
    unsigned long long blk_ord;    // ordinal of the block's first sample
    unsigned long long n;          // samples wanted in all
    unsigned long long pos;        // absolute position of the chunk's first draw
    unsigned long long *n_raw;
};
__device__ inline unsigned stage_pad(unsigned i) { return i + (i >> 4); }

// (k_out, count) of a chunk entered with k positions consumed; EMIT: also writes the values at their ordinals
template <int R, bool EMIT>
__device__ inline void chain_eval(const chunk_t<R> &c, u128 inc, unsigned k, unsigned &kout, unsigned &cnt,
                                  const double *wi, unsigned *err, const emit_ctx *ec) {
    constexpr unsigned FULL = (R == 32) ? 0xffffffffu : ((1u << R) - 1u);
    constexpr unsigned EVEN = 0x55555555u;
    static_assert(R <= 31, "the run scan needs a spare bit for the carry");
    unsigned n = 0;
    for (;;) {
        if (k >= (unsigned)R) {
            kout = k - R;
            cnt = n;
            return;
        }
        const unsigned low = (1u << k) - 1u;
        const unsigned nf = c.nf & ~low;
        const unsigned starts = nf & ~(nf << 1);
        const unsigned re = nf & ~(nf + (starts & EVEN));       // runs whose first position is even
        const unsigned ro = nf & ~(nf + (starts & ~EVEN));      // ... odd
        const unsigned sn = ((re & EVEN) | (ro & ~EVEN)) & FULL; // sample starts that are not fast
        const unsigned S = ~(sn << 1);                           // sample starts
        const unsigned valid = FULL & ~low;
        const unsigned tl = S & c.z & valid;                     // tail-class starts
        const unsigned upto = tl ? (valid & ((tl & (0u - tl)) - 1u)) : valid;
        const unsigned e_fast = S & ~nf & upto;
        const unsigned e_wedge = sn & ~c.z & c.wacc & upto;
        const unsigned e = e_fast | e_wedge;
        if constexpr (EMIT) {
#pragma unroll
            for (int p = 0; p < R; p++) {
                if ((e >> p) & 1) {
                    const unsigned o = ec->obase + n + __builtin_popcount(e & ((1u << p) - 1u));
                    ec->stage[stage_pad(o)] = zig_value(c.raw[p], wi);
                    if (ec->blk_ord + o + 1 == ec->n) *ec->n_raw = ec->pos + p + 1 + ((e_wedge >> p) & 1);
                }
            }
        }
        n += __builtin_popcount(e);
        if (!tl) {
            kout = (sn >> (R - 1)) & 1u;
            cnt = n;
            return;
        }
        const int p = __builtin_ctz(tl);
        // (the draw of position p by a select chain: raw[] lives in registers)
        uint64_t rp = 0;
#pragma unroll
        for (int q = 0; q < R; q++) rp = (q == p) ? c.raw[q] : rp;
        double v;
        unsigned consumed;
        zig_tail_walk(c.s0, inc, p, rp, v, consumed, err);
        if constexpr (EMIT) {
            const unsigned o = ec->obase + n;
            ec->stage[stage_pad(o)] = v;
            if (ec->blk_ord + o + 1 == ec->n) *ec->n_raw = ec->pos + p + 1 + consumed;
        }
        n += 1;
        k = p + 1 + consumed;
    }
}

// fixed point of the thread entries of one block: every thread leaves with its own (kin, kout, cnt)
template <int R, int T>
__device__ inline void block_resolve(const chunk_t<R> &c, u128 inc, unsigned k_block, unsigned *lds_k /* [T + 1] */,
                                     const double *wi, unsigned *err, unsigned &kin, unsigned &kout, unsigned &cnt) {
    const int t = threadIdx.x;
    kin = t == 0 ? k_block : 0u;
    for (int it = 0; it <= T; it++) {
        chain_eval<R, false>(c, inc, kin, kout, cnt, wi, err, nullptr);
        lds_k[t + 1] = kout;
        __syncthreads();
        const unsigned nk = t == 0 ? k_block : lds_k[t];
        const int changed = nk != kin;
        kin = nk;
        if (!__syncthreads_or(changed)) break;
    }
}

// sum over the block (every thread gets it); red: [T / 64 + 1] words of LDS
template <int T>
__device__ inline unsigned block_sum(unsigned v, unsigned *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int t = threadIdx.x;
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    unsigned s = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) s += red[w];
    return s;
}

struct zig_lds {
    uint64_t ki[256];
    double wi[256], fi[256];
    unsigned k[ZIG_T + 1];
    unsigned red[ZIG_T / 64 + 4];
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

__device__ inline u128 thread_state(const ulonglong2 *blk_state, long b, u128 inc) {
    const ulonglong2 bs = blk_state[b];
    return jump_apply(ZIG_THR.v[threadIdx.x], mk128(bs.x, bs.y), inc);
}

// pass 1: fun[2 b + e] = k_out << 16 | count of block b entered with e in {0, 1} positions consumed; successors that
// will be entered with k >= 2 are evaluated here and appended to the patch list (key = b << 16 | k, value as fun)
__global__ void __launch_bounds__(ZIG_T)
zig_count_kernel(const ulonglong2 *__restrict__ blk_state, uint64_t i_hi, uint64_t i_lo, long nblk,
                 unsigned *__restrict__ fun, ulonglong2 *__restrict__ patch, unsigned patch_cap, zig_status *st) {
    constexpr int R = ZIG_R_CHUNK, T = ZIG_T;
    __shared__ zig_lds L;
    zig_lds_fill(L);
    const u128 inc = mk128(i_hi, i_lo);
    const int t = threadIdx.x;
    for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
        chunk_t<R> c;
        chunk_build<R>(c, thread_state(blk_state, b, inc), inc, L.ki, L.wi, L.fi);
        unsigned kfwd[2];
        for (int e = 0; e < 2; e++) {
            unsigned kin, kout, cnt;
            block_resolve<R, T>(c, inc, (unsigned)e, L.k, L.wi, &st->error, kin, kout, cnt);
            const unsigned total = block_sum<T>(cnt, L.red);
            const unsigned kb = L.k[T];                  // k_out of the last thread
            if (t == 0) fun[2 * b + e] = (kb << 16) | total;
            kfwd[e] = kb;
            __syncthreads();
        }
        // rare: a tail sample (or several) reaches across the end of the block
        for (int e = 0; e < 2; e++) {
            unsigned k = kfwd[e];
            if (e == 1 && k == kfwd[0]) break;
            for (long bb = b + 1; k >= 2 && bb < nblk; bb++) {
                chunk_t<R> c2;
                chunk_build<R>(c2, thread_state(blk_state, bb, inc), inc, L.ki, L.wi, L.fi);
                unsigned kin, kout, cnt;
                block_resolve<R, T>(c2, inc, k, L.k, L.wi, &st->error, kin, kout, cnt);
                const unsigned total = block_sum<T>(cnt, L.red);
                const unsigned kb = L.k[T];
                if (t == 0) {
                    const unsigned slot = atomicAdd(&st->npatch, 1u);
                    if (slot < patch_cap) patch[slot] = make_ulonglong2(((unsigned long long)bb << 16) | k, (kb << 16) | total);
                    else atomicOr(&st->error, 2u);
                }
                k = kb;
                __syncthreads();
            }
        }
    }
}

__device__ inline unsigned patch_lookup(const ulonglong2 *patch, unsigned npatch, long b, unsigned k, unsigned *err) {
    const unsigned long long key = ((unsigned long long)b << 16) | k;
    for (unsigned i = 0; i < npatch; i++)
        if (patch[i].x == key) return (unsigned)patch[i].y;
    atomicOr(err, 4u);
    return 0u;
}

// one workgroup: composes the block functions, entry[b] = (k, first ordinal) of every block
__global__ void __launch_bounds__(ZIG_T)
zig_scan_kernel(long nblk, const unsigned *__restrict__ fun, const ulonglong2 *__restrict__ patch, unsigned patch_cap,
                unsigned long long ord0, ulonglong2 *__restrict__ entry, zig_status *st) {
    constexpr int T = ZIG_T;
    __shared__ unsigned g_k[T][2];
    __shared__ unsigned long long g_c[T][2];
    __shared__ unsigned s_k[T + 1];
    __shared__ unsigned long long s_o[T + 1];
    const int t = threadIdx.x;
    const unsigned npatch = min(st->npatch, patch_cap);
    const long seg = (nblk + T - 1) / T;
    const long b0 = std::min<long>(nblk, t * seg), b1 = std::min<long>(nblk, b0 + seg);
    auto walk = [&](unsigned k, unsigned long long o, bool write, unsigned &k_end, unsigned long long &o_end) {
        for (long b = b0; b < b1; b++) {
            if (write) entry[b] = make_ulonglong2(k, o);
            const unsigned w = k < 2 ? fun[2 * b + k] : patch_lookup(patch, npatch, b, k, &st->error);
            k = w >> 16;
            o += w & 0xffffu;
        }
        k_end = k;
        o_end = o;
    };
    for (int e = 0; e < 2; e++) walk((unsigned)e, 0ull, false, g_k[t][e], g_c[t][e]);
    __syncthreads();
    if (t == 0) {
        unsigned k = 0;
        unsigned long long o = ord0;
        for (int i = 0; i < T; i++) {
            s_k[i] = k;
            s_o[i] = o;
            if (k < 2) {
                o += g_c[i][k];
                k = g_k[i][k];
            } else {
                // a tail sample across a segment boundary: this segment is walked here (its owner starts from k < 2)
                const long c0 = std::min<long>(nblk, i * seg), c1 = std::min<long>(nblk, c0 + seg);
                for (long b = c0; b < c1; b++) {
                    const unsigned w = k < 2 ? fun[2 * b + k] : patch_lookup(patch, npatch, b, k, &st->error);
                    k = w >> 16;
                    o += w & 0xffffu;
                }
            }
        }
        s_k[T] = k;
        s_o[T] = o;
        st->total = o;
        st->k_last = k;
    }
    __syncthreads();
    unsigned ke;
    unsigned long long oe;
    walk(s_k[t], s_o[t], true, ke, oe);
}

// pass 2: every block with its true entry; the normals go to g[ordinal] for ordinal < n
__global__ void __launch_bounds__(ZIG_T)
zig_emit_kernel(const ulonglong2 *__restrict__ blk_state, uint64_t i_hi, uint64_t i_lo, long nblk,
                const ulonglong2 *__restrict__ entry, unsigned long long pos0, unsigned long long n,
                double *__restrict__ g, zig_status *st) {
    constexpr int R = ZIG_R_CHUNK, T = ZIG_T;
    __shared__ zig_lds L;
    __shared__ double stage[ZIG_BLK + ZIG_BLK / 16 + 2];
    __shared__ unsigned wsum[T / 64];
    zig_lds_fill(L);
    const u128 inc = mk128(i_hi, i_lo);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
        const ulonglong2 en = entry[b];
        if (en.y >= n) break;                               // (blocks are in ordinal order: nothing left to write)
        chunk_t<R> c;
        chunk_build<R>(c, thread_state(blk_state, b, inc), inc, L.ki, L.wi, L.fi);
        unsigned kin, kout, cnt;
        block_resolve<R, T>(c, inc, (unsigned)en.x, L.k, L.wi, &st->error, kin, kout, cnt);
        // exclusive prefix of the counts over the block
        unsigned incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < T / 64; w++) {
            if (w < wave) wbase += wsum[w];
            total += wsum[w];
        }
        emit_ctx ec;
        ec.stage = stage;
        ec.obase = wbase + incl - cnt;
        ec.blk_ord = en.y;
        ec.n = n;
        ec.pos = pos0 + (unsigned long long)b * ZIG_BLK + (unsigned long long)t * R;
        ec.n_raw = &st->n_raw;
        unsigned k2, c2;
        chain_eval<R, true>(c, inc, kin, k2, c2, L.wi, &st->error, &ec);
        __syncthreads();
        const unsigned long long room = n - en.y;
        const unsigned m = (unsigned)std::min<unsigned long long>(total, room);
        for (unsigned i = t; i < m; i += T) g[en.y + i] = stage[stage_pad(i)];
        __syncthreads();
    }
}

}  // namespace

extern "C" {

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
    StageTimer timer(ctx, "normals_pcg64");
    unsigned long long pos0 = 0, ord0 = 0;
    for (int round = 0; round < 64; round++) {
        const unsigned long long want = (unsigned long long)n - ord0;
        // 1.02145 raw draws per normal on average; the margin covers 200 sigma, and a short round is followed by another
        const long nblk = (long)((want + want / 44 + 2 * ZIG_BLK) / ZIG_BLK) + 1;
        const unsigned patch_cap = (unsigned)(nblk / 64 + 1024);
        const size_t off_fun = sizeof(ulonglong2) * (size_t)nblk;
        const size_t off_entry = off_fun + sizeof(unsigned) * 2 * (size_t)nblk;
        const size_t off_patch = off_entry + sizeof(ulonglong2) * (size_t)nblk;
        const size_t off_st = off_patch + sizeof(ulonglong2) * patch_cap;
        char *ws = nullptr;
        int rc = corahip_ctx_scratch(ctx, 6, off_st + sizeof(zig_status), (void **)&ws);
        if (rc) return rc;
        ulonglong2 *blk_state = (ulonglong2 *)ws;
        unsigned *fun = (unsigned *)(ws + off_fun);
        ulonglong2 *entry = (ulonglong2 *)(ws + off_entry);
        ulonglong2 *patch = (ulonglong2 *)(ws + off_patch);
        zig_status *st = (zig_status *)(ws + off_st);
        HIP_TRY(hipMemsetAsync(st, 0, sizeof(zig_status), ctx->stream));
        zig_seek_kernel<<<(unsigned)((nblk + 255) / 256), 256, 0, ctx->stream>>>(state[0], state[1], inc[0], inc[1], pos0, nblk,
                                                                                blk_state);
        LAUNCH_CHECK();
        const unsigned grid = (unsigned)std::min<long>(nblk, (long)ctx->num_cu * 8);
        zig_count_kernel<<<grid, ZIG_T, 0, ctx->stream>>>(blk_state, inc[0], inc[1], nblk, fun, patch, patch_cap, st);
        LAUNCH_CHECK();
        zig_scan_kernel<<<1, ZIG_T, 0, ctx->stream>>>(nblk, fun, patch, patch_cap, ord0, entry, st);
        LAUNCH_CHECK();
        zig_emit_kernel<<<grid, ZIG_T, 0, ctx->stream>>>(blk_state, inc[0], inc[1], nblk, entry, pos0, (unsigned long long)n, g,
                                                         st);
        LAUNCH_CHECK();
        zig_status hs;
        HIP_TRY(hipMemcpyAsync(&hs, st, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (hs.error) {
            corahip_set_error("normals_pcg64: device status %u (1 tail loop cap, 2 patch list full, 4 patch missing)", hs.error);
            return CORAHIP_ESTATE;
        }
        if (hs.total >= (unsigned long long)n) {
            *n_raw = hs.n_raw;
            return 0;
        }
        pos0 += (unsigned long long)nblk * ZIG_BLK + hs.k_last;
        ord0 = hs.total;
    }
    corahip_set_error("normals_pcg64: no convergence");
    return CORAHIP_ESTATE;
}

}  // extern "C"
