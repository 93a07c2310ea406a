// mtlegacy.hip - numpy's LEGACY normal stream (np.random.standard_normal, the global MT19937 RandomState) on the device.
//
// The reference called without a generator - `Sky3d.getsky()` -> `skysim.mkfullsky(cla, nside)`, cora/core/maps.py:235-237;
// `rng=None` in cora/util/nputil.py:121-123 - draws its 2 F nalm normals from numpy's global legacy state: MT19937 +
// Marsaglia's polar method with its one cached value (numpy/random/src/legacy/legacy-distributions.c `legacy_gauss`,
// numpy/random/src/mt19937/mt19937.c).  ~10 s per cfg-3 realisation on the host.  The same stream here:
//
//   uniform   a = next32 >> 5, b = next32 >> 6, (a 2^26 + b) / 2^53                               (exact in fp64)
//   attempt   x1 = 2 u1 - 1, x2 = 2 u2 - 1, r2 = x1^2 + x2^2; accepted iff 0 < r2 < 1 (pi / 4 of them);
//             f = sqrt(-2 log(r2) / r2); the normals f x2, then f x1
// An attempt ALWAYS consumes four 32-bit words: attempt a of a stream sits at words 4 a .. 4 a + 3, whatever happened
// before - the accept decisions and every x are exact arithmetic on those words, so which attempts are accepted, and
// with it the generator state afterwards, is numpy's bit for bit; the values go through log / sqrt / two divisions
// (sqrt and division are correctly rounded here as in libm; log is glibc's routine restated operation by operation -
// glibc_log_fma below -: the values are numpy's bit for bit on a host whose libm is glibc's FMA build, within 4 ulp on any
// other; tests/test_gpu_npnormal.py asserts whichever applies).
//
// MT19937 itself is the sequential part: x[k + 624] = x[k + 397] ^ twist(x[k], x[k + 1]).  It is linear over GF(2), so
// the window J words on is g(T) applied to the window, g = x^J mod phi (phi the minimal polynomial, degree 19937):
// mt_jump.inc (tools/gen_mt_jump.py: Berlekamp-Massey on numpy's own output, square-and-multiply mod phi, checked
// against plain stepping; restated in oracle/mtlegacy.py) holds x^(q 8^j S - 1), q = 1 .. 7, for the segment length S = 2^20 words.
// The stream is cut into segments of S words = 262144 attempts; the segment start windows come from a radix-8 tree
// (level j: the starts q 8^j + a, q = 1 .. 7, from the starts a < 8^j, one launch; round 6 - rounds 4-5 doubled: twelve
// sequential launches at cfg 3 where there are four now), each application = extend the source window by 19938 words in
// LDS and XOR the windows at the polynomial's ~10^4 set bits (6e6 word operations: with S = 2^18 the tree was 10.9 of
// 19.6 ms, with 2^20 it is 4.0 of 15.3 - fewer, longer segments cost the two passes 2.6 ms of occupancy).
// Then one wave per segment: pass 1 counts the accepted attempts, a one-workgroup scan gives every segment its first
// output position, pass 2 regenerates and writes the normals (an accepted lane writes its pair: consecutive lanes,
// consecutive addresses).  The wave that meets the last needed attempt writes out the state numpy would be left in:
// the 624-word block it is in, the position inside it, and the cached second value when an odd count was asked for.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "mt_jump.inc"
#include "stream_internal.h"

namespace {

constexpr int MTN = 624, MTM = 397;
constexpr int MT_DEG = 19937;
constexpr int MT_ATT_BLOCK = MTN / 4;                       // attempts per 624-word block
constexpr long MT_SEG_WORDS = 1L << MT_SEG_LOG2;
constexpr long MT_SEG_ATT = MT_SEG_WORDS / 4;               // attempts per segment

__host__ __device__ inline unsigned mt_next(unsigned u, unsigned v, unsigned w) {
    const unsigned y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return w ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ inline unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// the next 624 words in place (mt19937_gen), by ONE wave, in three phases of <= 227 words: within a phase every word
// needs only OLD words (k, k + 1, and k + 397 for k < 227) or words of an EARLIER phase (k - 227), so all reads of a
// phase go out together, then all its writes - three LDS round trips per block (64 words at a time took ten).  Word 623
// takes the NEW word 0 as its k + 1.
__device__ inline void mt_block_next(unsigned *mt, int lane) {
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {
        const int k0 = 227 * ph, k1 = ph == 2 ? MTN : 227 * (ph + 1);
        unsigned nw[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int k = k0 + 64 * c + lane;
            if (k < k1) {
                const unsigned u = mt[k], v = mt[k + 1 < MTN ? k + 1 : 0], w = mt[k + MTM < MTN ? k + MTM : k + MTM - MTN];
                nw[c] = mt_next(u, v, w);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int k = k0 + 64 * c + lane;
            if (k < k1) mt[k] = nw[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// the next 624 words OUT OF PLACE: dst = the block behind src.  Same three phases (word k >= 227 needs the NEW word
// k - 227, word 623 the new word 0), but nothing is overwritten, so a phase needs no separation of its reads from its
// writes: one LDS round trip less per phase.  Used to keep TWO consecutive blocks resident (round 6): a block holds
// 156 attempts = 2.44 rounds of 64 lanes, a pair 312 = 4.875 rounds - the third round of every block ran at 44 % lane
// occupancy, the fifth round of a pair runs at 88 %.
__device__ inline void mt_block_next_to(const unsigned *src, unsigned *dst, int lane) {
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {
        const int k0 = 227 * ph, k1 = ph == 2 ? MTN : 227 * (ph + 1);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int k = k0 + 64 * c + lane;
            if (k < k1) {
                const unsigned u = src[k], v = k + 1 < MTN ? src[k + 1] : dst[0];
                const unsigned w = k + MTM < MTN ? src[k + MTM] : dst[k + MTM - MTN];
                dst[k] = mt_next(u, v, w);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

struct mt_status {
    unsigned key[MTN];          // the block the generator is left in
    int pos, has_gauss;
    double gauss;
    unsigned long long total_accepted;
    int done, pad;
};

// ---- jump: dst windows from src windows, one workgroup per application ---------------------------------------------
// xs = the source window and the 19938 words that follow it (wave 0 steps the recurrence, 64 words at a time: word
// 624 + k needs words k, k + 1 and k + 397, the last one at least 227 words back); meanwhile the other waves expand the
// polynomial's bit mask into the list of its set positions (LDS).  Then out[w] = xor over the list of xs[i + 1 + w]:
// 16 list entries per round, all their reads in flight together (one LDS round trip per BIT was 64 ms at cfg 3).
#define MT_JUMP_T 256
#define MT_JUMP_XS (MTN + MT_DEG + 1 + 256 + MTM + 8)
#define MT_PLIST_STRIDE (MT_DEG + 32)      // positions per polynomial in the device-wide list (a polynomial has <= 19937 terms)
// one-off per context: the set positions (+ 1) of every jump polynomial as a list, npos[k] entries at plist[k][...] -
// the kernels below walk the list with wave-uniform (scalar) loads instead of rebuilding it in LDS per application
__global__ void __launch_bounds__(MT_JUMP_T)
mt_plist_kernel(unsigned *__restrict__ plist_all, unsigned *__restrict__ npos_all) {
    __shared__ unsigned wcnt[MTN + 1];
    const int k = blockIdx.x, tid = threadIdx.x;
    const unsigned *g = MT_JUMP[k];
    for (int i = tid; i < MTN; i += MT_JUMP_T) wcnt[i] = (unsigned)__builtin_popcount(g[i]);
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;                                     // exclusive prefix of the per-word bit counts
        for (int i = 0; i < MTN; i++) {
            const unsigned c = wcnt[i];
            wcnt[i] = run;
            run += c;
        }
        npos_all[k] = run;
    }
    __syncthreads();
    unsigned *pl = plist_all + (size_t)k * MT_PLIST_STRIDE;
    for (int i = tid; i < MTN; i += MT_JUMP_T) {
        unsigned m = g[i];
        unsigned o = wcnt[i];
        while (m) {
            pl[o++] = (unsigned)(32 * i + __builtin_ctz(m) + 1);
            m &= m - 1;
        }
    }
}

// `split` workgroups share one application: workgroup (a, part) makes the output words [part, part + 1) * 624 / split (split
// divides 624).  An application is bound by the LDS of its CU (10^4 windows x 624 words of 4-byte reads) and the tree's
// first levels have 1, 2, 4, ... applications: split over up to 24 CUs (each extends the window for itself: 20 us) they
// take a fraction of one application's latency (tree 3.93 -> 3.54 ms at cfg 3).  Inside a workgroup the threads are
// (g, wi): word wi of the slice, positions g, g + G, ... of the list (G = 256 / words per slice when the slice is
// narrower than the workgroup); the G partial words are combined through LDS.  The positions are copied into LDS from
// the device-wide list (walking the list in global memory by scalar loads instead was measured SLOWER, tree 3.5 -> 4.7
// ms: a load's latency per 16 positions is not hidden at one wave per SIMD).
#define MT_JUMP_WG 1024                     // threads of a jump workgroup: MT_JUMP_WG / 256 position groups on the full window
// One launch = one LEVEL of the radix-8 tree (round 6): from the segment starts 0 .. have - 1 the starts q have + a, q = 1 .. 7 -
// application (q, a) applies polynomial kbase + q - 1 (x^(q have S - 1)) to start a.  (Rounds 4-5: a doubling tree, one
// polynomial and `have` applications per level - twelve sequential launches at cfg 3, eight of them with fewer applications
// than CUs, each bound by one application's latency; now four launches, two of them small.)
__global__ void __launch_bounds__(MT_JUMP_WG)
mt_jump_kernel(unsigned *__restrict__ seg_state, long have, long nseg, int kbase, int split,
               const unsigned *__restrict__ plist_all, const unsigned *__restrict__ npos_all) {
    extern __shared__ unsigned xs[];                          // [MT_JUMP_XS] words, then the position list (u16)
    constexpr int NQ = (MTN + MT_JUMP_T - 1) / MT_JUMP_T;     // words per thread on the full window: 3
    __shared__ unsigned red[NQ][MT_JUMP_WG];
    unsigned short *plist = (unsigned short *)(xs + MT_JUMP_XS);
    const long app = blockIdx.x / split;
    const int part = blockIdx.x - (int)app * split;
    const long q = app / have + 1, a = app - (q - 1) * have;
    const long dst_seg = q * have + a;
    if (q > 7 || dst_seg >= nseg) return;
    const int k = kbase + (int)q - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned *src = seg_state + a * MTN;
    const int npos = (int)npos_all[k];
    {
        const unsigned *__restrict__ pl = plist_all + (size_t)k * MT_PLIST_STRIDE;
        for (int i = tid; i < npos; i += MT_JUMP_WG) plist[i] = (unsigned short)pl[i];
    }
    for (int i = tid; i < MTN; i += MT_JUMP_WG) xs[i] = src[i];
    __syncthreads();
    if (tid < 64) {
        // 227 words per step (four reads-then-writes of 64): word 624 + k needs words k, k + 1 and k + 397, the last at
        // least 227 words back, i.e. of an earlier step
        for (int k0 = 0; k0 < MT_DEG + 1; k0 += 227) {
            unsigned nw[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int kk = k0 + 64 * c + lane;
                nw[c] = mt_next(xs[kk], xs[kk + 1], xs[kk + MTM]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int kk = k0 + 64 * c + lane;
                if (64 * c + lane < 227 && kk < MT_DEG + 1) xs[MTN + kk] = nw[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // threads as (gi, wi): words wi, wi + 256, .. of the slice, positions gi, gi + G, .. of the list.  A wave per SIMD
    // cannot hide the LDS latency (the counter of outstanding LDS reads has four bits): sixteen waves take the reads of
    // four position groups at once (jump tree of cfg 3: 3.5 -> 2.3 ms)
    const int nw = MTN / split, w0 = part * nw;               // this workgroup's slice of the output window
    const int nwt = nw < MT_JUMP_T ? nw : MT_JUMP_T;          // threads along the words
    const int G = MT_JUMP_WG / nwt;                           // position groups
    const int wi = tid % nwt, gi = tid / nwt;
    unsigned acc[NQ] = {};
    auto xor_windows = [&](auto nqe_c) {
        constexpr int NQE = decltype(nqe_c)::value;
        int p = gi;
        for (; p + 15 * G < npos; p += 16 * G) {
            unsigned v[16][NQE];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = plist[p + e * G];
#pragma unroll
                for (int q = 0; q < NQE; q++) v[e][q] = xs[i + w0 + wi + q * MT_JUMP_T];   // (past the slice: inside xs, discarded)
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
#pragma unroll
                for (int q = 0; q < NQE; q++) acc[q] ^= v[e][q];
        }
        for (; p < npos; p += G) {
            const int i = plist[p];
#pragma unroll
            for (int q = 0; q < NQE; q++) acc[q] ^= xs[i + w0 + wi + q * MT_JUMP_T];
        }
    };
    if (gi < G) {
        if (nw <= MT_JUMP_T) xor_windows(std::integral_constant<int, 1>{});
        else if (nw <= 2 * MT_JUMP_T) xor_windows(std::integral_constant<int, 2>{});
        else xor_windows(std::integral_constant<int, NQ>{});
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) red[q][tid] = gi < G ? acc[q] : 0u;
    __syncthreads();
    unsigned *dst = seg_state + dst_seg * MTN + w0;
    if (gi == 0) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int w = wi + q * MT_JUMP_T;
            if (w < nw) {
                unsigned r = 0;
                for (int g2 = 0; g2 < G; g2++) r ^= red[q][g2 * nwt + wi];
                dst[w] = r;
            }
        }
    }
}

// ---- glibc's log as numpy calls it -------------------------------------------------------------------------------------
// legacy_gauss takes log(r2) from libm.  glibc's routine (sysdeps/ieee754/dbl-64/e_log.c: 128-entry table of (1 / c,
// log c), degree-5 polynomial in r = z / c - 1; a degree-11 polynomial for 1 - 2^-4 <= x < 1 + 0x1.09p-4) in the
// evaluation order of its FMA build - the one the loader selects on every CPU with FMA + AVX2 - operation by operation:
// what is fused there is an fma here, what is separate stays separate (contraction off).  Constants: glibc_log_tab.inc
// (tools/gen_glibc_log_tab.py reads them out of the installed libm).  For positive normal x (r2 >= 2^-104 here);
// oracle/mtlegacy.py restates the same sequence and tests/test_oracle.py pins it to the host's log bit for bit.
#include "glibc_log_tab.inc"
__device__ const double2 g_glog_T[128] = GLIBC_LOG_T;
__device__ inline double glibc_log_fma(double x) {
#pragma clang fp contract(off)
    constexpr double A[5] = GLIBC_LOG_A;
    constexpr double B[11] = GLIBC_LOG_B;
    const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    if (ix - 0x3FEE000000000000ull <= 0x308FFFFFFFFFFull) {
        if (x == 1.0) return 0.0;
        const double r = x - 1.0;
        double p2 = fma(r, B[2], B[1]);
        double p3 = fma(r, B[5], B[4]);
        const double r2 = r * r;
        const double p5 = fma(r, B[8], B[7]);
        p2 = fma(r2, B[3], p2);
        p3 = fma(r2, B[6], p3);
        const double r3 = r * r2;
        double p1 = fma(r2, B[9], p5);
        p1 = fma(r3, B[10], p1);
        p1 = fma(p1, r3, p3);
        p1 = fma(p1, r3, p2);
        const double t = fma(r, 134217728.0, r);
        const double rhi = fma(-134217728.0, r, t);
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = fma(rhi2, B[0], r);
        double lo = fma(rhi2, B[0], r - hi);
        lo = fma(B[0] * rlo, r + rhi, lo);
        return fma(p1, r3, lo) + hi;
    }
    const unsigned long long tmp = ix - 0x3FE6000000000000ull;
    const int i = (int)(tmp >> 45) & 127;
    const int k = (int)((long long)tmp >> 52);
    const double z = __longlong_as_double((long long)(ix - (tmp & 0xFFF0000000000000ull)));
    const double2 tc = g_glog_T[i];
    const double kd = (double)k;
    const double r = fma(z, tc.x, -1.0);
    const double w = fma(kd, GLIBC_LOG_LN2HI, tc.y);
    const double q = fma(r, A[2], A[1]);
    const double hi = r + w;
    const double r2 = r * r;
    double lo = (w - hi) + r;
    lo = fma(kd, GLIBC_LOG_LN2LO, lo);
    const double r3 = r * r2;
    double p = fma(r, A[4], A[3]);
    lo = fma(r2, A[0], lo);
    p = fma(p, r2, q);
    return fma(r3, p, lo) + hi;
}

// ---- one attempt of the polar method from four words of the block --------------------------------------------------
struct mt_attempt {
    double x1, x2, r2;
    bool ok;
};
__device__ inline mt_attempt mt_try(const unsigned *mt, int a) {
#pragma clang fp contract(off)
    const uint4 w = *reinterpret_cast<const uint4 *>(mt + 4 * a);
    const double u1 = ((double)(mt_temper(w.x) >> 5) * 67108864.0 + (double)(mt_temper(w.y) >> 6)) / 9007199254740992.0;
    const double u2 = ((double)(mt_temper(w.z) >> 5) * 67108864.0 + (double)(mt_temper(w.w) >> 6)) / 9007199254740992.0;
    mt_attempt t;
    t.x1 = 2.0 * u1 - 1.0;
    t.x2 = 2.0 * u2 - 1.0;
    t.r2 = t.x1 * t.x1 + t.x2 * t.x2;
    t.ok = !(t.r2 >= 1.0 || t.r2 == 0.0);
    return t;
}

#define MT_WG 256
// The count pass runs one wave per SEGMENT (a segment start costs a jump application), the emit pass one wave per
// SUB-SEGMENT of MT_SUB_BLOCKS blocks: the count pass leaves a snapshot of the generator (624 words) and the number of
// accepted attempts for every sub-segment, so the emit pass can start anywhere with ~3400 accepted attempts of
// granularity - which is what lets it run on a RANGE of the stream (one slot of the l-range ring, drawstream.hip) with
// a thousand waves in flight instead of the forty segments a range spans - and runs at full occupancy on the whole
// stream as well.
constexpr int MT_SUB_BLOCKS = 28;
constexpr int MT_SEG_BLOCKS = (int)((MT_SEG_ATT + MT_ATT_BLOCK - 1) / MT_ATT_BLOCK);   // 1681, the last one partial (64 attempts)
constexpr int MT_NSUB = MT_SEG_BLOCKS / MT_SUB_BLOCKS;                                 // 60; the last sub-segment takes the remainder

// pass 1, one wave per segment: sub_state[(j NSUB + k)] = the block the generator is in at block k SUB_BLOCKS of segment
// j, sub_cnt[..] = accepted attempts of that sub-segment
__global__ void __launch_bounds__(MT_WG)
mt_count_kernel(const unsigned *__restrict__ seg_state, long nseg, unsigned *__restrict__ sub_state, unsigned *__restrict__ sub_cnt) {
#pragma clang fp contract(off)
    // two consecutive blocks per wave, [A | B] contiguous: attempt a < 156 of a pair reads words 4 a .. 4 a + 3 of A, attempt
    // 156 + a' those of B - one index into one array
    __shared__ __attribute__((aligned(16))) unsigned blk[MT_WG / 64][2 * MTN + 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * (MT_WG / 64) + wv;
    if (j >= nseg) return;
    unsigned *mt = blk[wv], *mtb = blk[wv] + MTN;
    for (int i = lane; i < MTN; i += 64) mt[i] = seg_state[j * MTN + i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // (sub-segment by sub-segment: the snapshot and the count are written at the loop level they belong to, no test
    //  per block)
    for (int k = 0; k < MT_NSUB; k++) {
        const long sidx = j * MT_NSUB + k;
        for (int i = lane; i < MTN; i += 64) sub_state[sidx * MTN + i] = mt[i];
        const int b0 = k * MT_SUB_BLOCKS, b1 = k == MT_NSUB - 1 ? MT_SEG_BLOCKS : b0 + MT_SUB_BLOCKS;
        unsigned cnt = 0;
        int bi = b0;
        for (; bi + 1 < b1; bi += 2) {
            mt_block_next_to(mt, mtb, lane);
            // attempts of the two blocks that belong to the segment (only a segment's LAST block is partial)
            const int nbb = (int)std::min<long>(MT_ATT_BLOCK, MT_SEG_ATT - (long)(bi + 1) * MT_ATT_BLOCK);
#pragma unroll
            for (int r = 0; r < (2 * MT_ATT_BLOCK + 63) / 64; r++) {
                const int a = 64 * r + lane;
                const bool in = a < MT_ATT_BLOCK + nbb;
                mt_attempt t;
                t.ok = false;
                if (in) t = mt_try(mt, a);
                cnt += (unsigned)__builtin_popcountll(__ballot(in && t.ok));
            }
            mt_block_next_to(mtb, mt, lane);
        }
        if (bi < b1) {                                     // an odd block count: the sub-segment's last block by itself
            const int nb = (int)std::min<long>(MT_ATT_BLOCK, MT_SEG_ATT - (long)bi * MT_ATT_BLOCK);
#pragma unroll
            for (int r = 0; r < (MT_ATT_BLOCK + 63) / 64; r++) {
                const int a = 64 * r + lane;
                const bool in = a < nb;
                mt_attempt t;
                t.ok = false;
                if (in) t = mt_try(mt, a);
                cnt += (unsigned)__builtin_popcountll(__ballot(in && t.ok));
            }
            mt_block_next(mt, lane);
        }
        if (lane == 0) sub_cnt[sidx] = cnt;
    }
}

// pass 2, one wave per sub-segment (grid-stride from *sub_first, NULL = 0): the normals of accepted attempt o are
// elements e = off0 + 2 o and e + 1 of the stream (off0 = 1 when numpy's cached second value came first); an element is
// written to g[e - w_lo] iff w_lo <= e < w_hi and e < n.  The wave that meets the last needed attempt (number
// `pairs`) leaves the state numpy would be left in.
__global__ void __launch_bounds__(MT_WG)
mt_emit_kernel(const unsigned *__restrict__ sub_state, long nsub, const unsigned long long *__restrict__ sub_base,
               const long *__restrict__ sub_first, unsigned long long w_lo, unsigned long long w_hi, unsigned off0,
               unsigned long long n, double *__restrict__ g, mt_status *st) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) unsigned blk[MT_WG / 64][2 * MTN + 8];    // two consecutive blocks [A | B]: see mt_count_kernel
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned *mt = blk[wv], *mtb = blk[wv] + MTN;
    const unsigned long long need = n - off0;
    const unsigned long long pairs = (need + 1) / 2;          // accepted attempts that are needed
    // accepted attempts o with an element below w_hi: off0 + 2 o < w_hi
    const unsigned long long p_end = std::min<unsigned long long>(pairs, w_hi > off0 ? (w_hi - off0 + 1) / 2 : 0ull);
    const unsigned long long w_n = w_hi - w_lo;
    const bool al16 = ((reinterpret_cast<size_t>(g) >> 3) + (size_t)off0 - (size_t)(w_lo & 1ull)) % 2 == 0;   // element off0 + 2 o at a 16-byte boundary
    const long s0 = sub_first ? *sub_first : 0L;
    for (long s = s0 + (long)blockIdx.x * (MT_WG / 64) + wv; s < nsub; s += (long)gridDim.x * (MT_WG / 64)) {
        unsigned long long ord = sub_base[s];                  // accepted attempts before the next one
        if (ord >= p_end) break;
        for (int i = lane; i < MTN; i += 64) mt[i] = sub_state[s * MTN + i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int k = (int)(s % MT_NSUB);
        const int b0 = k * MT_SUB_BLOCKS, b1 = k == MT_NSUB - 1 ? MT_SEG_BLOCKS : b0 + MT_SUB_BLOCKS;
        // one round of 64 attempts of the resident window [A | B]: attempt a (< 156: block A, else block B), `in` = it exists
        auto round = [&](const int a, const bool in) {
            mt_attempt t;
            t.ok = false;
            if (in) t = mt_try(mt, a);
            const unsigned long long acc = __ballot(in && t.ok);
            bool last = false;
            if (in && t.ok) {
                const unsigned long long o = ord + __builtin_amdgcn_mbcnt_hi((unsigned)(acc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)acc, 0u));
                if (o < pairs) {
                    const unsigned long long e = (unsigned long long)off0 + 2 * o;
                    const bool in0 = e - w_lo < w_n, in1 = e + 1 - w_lo < w_n && 2 * o + 1 < need;
                    last = o + 1 == pairs;
                    if (in0 || in1 || last) {
                        const double f = sqrt(-2.0 * glibc_log_fma(t.r2) / t.r2);     // (sqrt and the division are correctly rounded on both sides)
                        const double first = f * t.x2, second = f * t.x1;
                        if (in0 && in1 && al16) *reinterpret_cast<double2 *>(g + (e - w_lo)) = make_double2(first, second);
                        else {             // (a pair behind an odd offset, or one cut by the window's edge)
                            if (in0) g[e - w_lo] = first;
                            if (in1) g[e + 1 - w_lo] = second;
                        }
                        if (last) {
                            // the generator after this attempt: inside its block, behind the attempt's four words; the
                            // second value stays cached when an odd number of normals was asked for
                            st->pos = 4 * ((a < MT_ATT_BLOCK ? a : a - MT_ATT_BLOCK) + 1);
                            st->has_gauss = (need & 1ull) ? 1 : 0;
                            st->gauss = (need & 1ull) ? second : 0.0;
                            st->done = 1;
                        }
                    }
                }
            }
            const unsigned long long hit = __ballot(last);
            if (hit) {
                // (wave-uniform) the block that holds the last needed attempt is the key numpy is left with
                const int a_hit = a - lane + (int)__builtin_ctzll(hit);
                const unsigned *kb = a_hit < MT_ATT_BLOCK ? mt : mtb;
                for (int i = lane; i < MTN; i += 64) st->key[i] = kb[i];
            }
            ord += __builtin_popcountll(acc);
        };
        int bi = b0;
        for (; bi + 1 < b1; bi += 2) {
            mt_block_next_to(mt, mtb, lane);
            const int nbb = (int)std::min<long>(MT_ATT_BLOCK, MT_SEG_ATT - (long)(bi + 1) * MT_ATT_BLOCK);
#pragma unroll
            for (int r = 0; r < (2 * MT_ATT_BLOCK + 63) / 64; r++) round(64 * r + lane, 64 * r + lane < MT_ATT_BLOCK + nbb);
            if (ord >= p_end) break;
            mt_block_next_to(mtb, mt, lane);
        }
        if (bi < b1 && bi + 1 >= b1 && ord < p_end) {          // an odd block count: the sub-segment's last block by itself
            const int nb = (int)std::min<long>(MT_ATT_BLOCK, MT_SEG_ATT - (long)bi * MT_ATT_BLOCK);
#pragma unroll
            for (int r = 0; r < (MT_ATT_BLOCK + 63) / 64; r++) round(64 * r + lane, 64 * r + lane < nb);
        }
    }
}

// the cached second value of numpy's last pair is the first element of the stream
__global__ void mt_put_kernel(double *g, double v) { g[0] = v; }

// first sub-segment of every range [bounds[r], bounds[r + 1]) of stream elements: the one that holds accepted attempt
// (bounds[r] - off0) / 2 (the largest s with sub_base[s] <= that)
__global__ void mt_range_kernel(const unsigned long long *__restrict__ sub_base, long nsub, const unsigned long long *__restrict__ bounds,
                                int nr, unsigned off0, long *__restrict__ sub_first) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nr) return;
    const unsigned long long o = bounds[r] > off0 ? (bounds[r] - off0) / 2 : 0ull;
    long lo = 0, hi = nsub - 1;
    while (lo < hi) {
        const long mid = (lo + hi + 1) >> 1;
        if (sub_base[mid] <= o) lo = mid;
        else hi = mid - 1;
    }
    sub_first[r] = lo;
}

// one workgroup: first accepted-attempt ordinal of every sub-segment
__global__ void __launch_bounds__(1024)
mt_scan_kernel(long nseg, const unsigned *__restrict__ seg_cnt, unsigned long long *__restrict__ seg_base, mt_status *st) {
    // exclusive prefix of the accepted attempts per sub-segment (157 k entries at cfg 3): every wave takes a contiguous
    // chunk with coalesced loads and a shuffle scan per 64 entries (round 5: one thread per 153 strided entries and a
    // serial pass over 1024 partial sums took 0.34 ms)
    __shared__ unsigned long long wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long per = ((nseg + 15) / 16 + 63) / 64 * 64;
    const long a = std::min<long>(nseg, wave * per), b = std::min<long>(nseg, a + per);
    unsigned long long s = 0;
    for (long i = a + lane; i < b; i += 64) s += seg_cnt[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) wtot[wave] = s;
    __syncthreads();
    unsigned long long run = 0;
    for (int w = 0; w < wave; w++) run += wtot[w];
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < 16; w++) tot += wtot[w];
        st->total_accepted = tot;
    }
    for (long i0 = a; i0 < b; i0 += 64) {
        const long i = i0 + lane;
        const unsigned long long v = i < b ? seg_cnt[i] : 0ull;
        unsigned long long incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (i < b) seg_base[i] = run + incl - v;
        run += __shfl(incl, 63);
    }
}

}  // namespace

// ---- host side (stream_internal.h): prepare = jump tree + count + scan; emit = one window of the stream -------------------
struct mt_session {
    std::vector<unsigned> x;                     // the window at the generator's position (kept until finish: asynchronous upload)
    std::vector<unsigned long long> bounds;      // [nr + 1] stream elements
    long nseg = 0, nsub = 0;
    unsigned off0 = 0;
    double gauss0 = 0.0;
    unsigned long long n = 0, pairs = 0;
    unsigned *seg_state = nullptr, *sub_state = nullptr, *sub_cnt = nullptr;
    unsigned long long *sub_base = nullptr, *d_bounds = nullptr;
    long *d_first = nullptr;
    mt_status *st = nullptr;
    int nr = 0;
};

int mt_stream_prepare(corahip_ctx *ctx, hipStream_t stream, corahip_mt_state *state, int64_t n,
                      const std::vector<unsigned long long> &bounds, mt_session **out) {
    ARG_CHECK(state != nullptr && n > 0 && state->pos >= 0 && state->pos <= MTN);
    ARG_CHECK(bounds.size() >= 2 && bounds.front() == 0 && bounds.back() == (unsigned long long)n);
    const bool other = stream != ctx->stream;
    mt_session *s = new mt_session();
    s->n = (unsigned long long)n;
    s->bounds = bounds;
    s->nr = (int)bounds.size() - 1;
    s->off0 = state->has_gauss ? 1u : 0u;          // the value legacy_gauss kept from its last pair comes first
    s->gauss0 = state->gauss;
    const unsigned long long need = s->n - s->off0;
    // attempts to look at: need / 2 accepted ones at an acceptance of pi / 4, + 2 % and a floor (the count of accepted
    // attempts among N has a relative sigma of 0.5 / sqrt(N))
    s->pairs = (need + 1) / 2;
    const long double want = (long double)s->pairs / 0.78539816339744830962L;
    const long natt = (long)(want * 1.02L) + 8192;
    s->nseg = std::max<long>(1, (natt + MT_SEG_ATT - 1) / MT_SEG_ATT);
    s->nsub = s->nseg * MT_NSUB;
    int levels = 0;
    for (long have = 1; have < s->nseg; have *= 8) levels++;
    if (levels > MT_NLEV) {
        corahip_set_error("normals_mt19937_legacy: %lld normals need %ld segments, the jump table holds %d radix-8 levels", (long long)n,
                          s->nseg, MT_NLEV);
        delete s;
        return CORAHIP_EINVAL;
    }
    // the window at the generator's position: x[pos .. pos + 623] (host: at most 624 steps)
    s->x.assign(state->key, state->key + MTN);
    s->x.resize(MTN + state->pos);
    for (int k = 0; k < state->pos; k++) s->x[MTN + k] = mt_next(s->x[k], s->x[k + 1], s->x[k + MTM]);
    const size_t nb = bounds.size();
    const size_t off_sub = sizeof(unsigned) * MTN * (size_t)s->nseg;
    const size_t off_cnt = off_sub + sizeof(unsigned) * MTN * (size_t)s->nsub;
    const size_t off_base = (off_cnt + sizeof(unsigned) * (size_t)s->nsub + 15) & ~(size_t)15;
    const size_t off_bnd = off_base + sizeof(unsigned long long) * (size_t)s->nsub;
    const size_t off_first = off_bnd + sizeof(unsigned long long) * nb;
    const size_t off_st = (off_first + sizeof(long) * nb + 15) & ~(size_t)15;
    char *ws = nullptr;
    int rc = corahip_ctx_scratch(ctx, 6, off_st + sizeof(mt_status), (void **)&ws);
    if (rc) {
        delete s;
        return rc;
    }
    s->seg_state = (unsigned *)ws;
    s->sub_state = (unsigned *)(ws + off_sub);
    s->sub_cnt = (unsigned *)(ws + off_cnt);
    s->sub_base = (unsigned long long *)(ws + off_base);
    s->d_bounds = (unsigned long long *)(ws + off_bnd);
    s->d_first = (long *)(ws + off_first);
    s->st = (mt_status *)(ws + off_st);
    auto fail = [&](hipError_t e, const char *what) {
        corahip_set_error("normals_mt19937_legacy: %s failed: %s", what, hipGetErrorString(e));
        delete s;
        return (int)e;
    };
    hipError_t e;
    if ((e = hipMemsetAsync(s->st, 0, sizeof(mt_status), stream)) != hipSuccess) return fail(e, "memset");
    if ((e = hipMemcpyAsync(s->seg_state, s->x.data() + state->pos, sizeof(unsigned) * MTN, hipMemcpyHostToDevice, stream)) != hipSuccess)
        return fail(e, "upload of the generator window");
    if ((e = hipMemcpyAsync(s->d_bounds, s->bounds.data(), sizeof(unsigned long long) * nb, hipMemcpyHostToDevice, stream)) != hipSuccess)
        return fail(e, "upload of the range bounds");
    // the position lists of the jump polynomials: made once per context (scratch slot 9), on this stream
    unsigned *plist_all = nullptr;
    {
        const size_t pb = sizeof(unsigned) * ((size_t)MT_NPOLY * MT_PLIST_STRIDE + MT_NPOLY);
        const bool fresh = ctx->scratch_bytes[9] < pb;
        if (corahip_ctx_scratch(ctx, 9, pb, (void **)&plist_all)) {
            delete s;
            return CORAHIP_ENOMEM;
        }
        if (fresh || !ctx->mt_plist_ready) {
            mt_plist_kernel<<<MT_NPOLY, MT_JUMP_T, 0, stream>>>(plist_all, plist_all + (size_t)MT_NPOLY * MT_PLIST_STRIDE);
            // (later calls may run on another stream: the list must be complete before any of them can start)
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e, "position lists of the jump polynomials");
            ctx->mt_plist_ready = true;
        }
    }
    const unsigned *npos_all = plist_all + (size_t)MT_NPOLY * MT_PLIST_STRIDE;
    {
        StageTimer t0(ctx, "mt_jump", stream, other);
        const size_t shm = sizeof(unsigned) * MT_JUMP_XS + sizeof(unsigned short) * (MT_DEG + 8);
        if ((e = hipFuncSetAttribute((const void *)mt_jump_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)) != hipSuccess)
            return fail(e, "hipFuncSetAttribute");
        int lev = 0;
        for (long have = 1; have < s->nseg; have *= 8, lev++) {
            const long count = std::min<long>(7 * have, s->nseg - have);       // applications of this level
            // levels with fewer applications than CUs: several workgroups per application (divisors of 624).  (Splitting
            // the last partial round of the larger levels as well was measured: no gain, 2.32 vs 2.36 ms.)
            int split = 1;
            for (int cand : {24, 12, 8, 6, 4, 3, 2})
                if (count * cand <= ctx->num_cu + ctx->num_cu / 2) {
                    split = cand;
                    break;
                }
            // (the grid covers q = 1 .. 7 for every start a < have: applications past nseg return at once)
            const long napp = std::min<long>(7 * have, ((s->nseg - 1) / have) * have);
            mt_jump_kernel<<<(unsigned)(napp * split), MT_JUMP_WG, shm, stream>>>(s->seg_state, have, s->nseg, 7 * lev, split, plist_all,
                                                                                 npos_all);
        }
    }
    {
        StageTimer t1(ctx, "mt_count", stream, other);
        const unsigned grid = (unsigned)((s->nseg + MT_WG / 64 - 1) / (MT_WG / 64));
        mt_count_kernel<<<grid, MT_WG, 0, stream>>>(s->seg_state, s->nseg, s->sub_state, s->sub_cnt);
        mt_scan_kernel<<<1, 1024, 0, stream>>>(s->nsub, s->sub_cnt, s->sub_base, s->st);
        mt_range_kernel<<<(s->nr + 63) / 64, 64, 0, stream>>>(s->sub_base, s->nsub, s->d_bounds, s->nr, s->off0, s->d_first);
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "a launch of the count pass");
    // numpy hands the cached value back and clears it; the state is rewritten by finish
    *out = s;
    return 0;
}

int mt_stream_emit_range(corahip_ctx *ctx, hipStream_t stream, mt_session *s, int r, double *slot) {
    ARG_CHECK(s != nullptr && r >= 0 && r < s->nr && slot != nullptr);
    const unsigned long long w_lo = s->bounds[r], w_hi = s->bounds[r + 1];
    StageTimer t2(ctx, "mt_emit", stream, stream != ctx->stream);
    if (w_lo == 0 && s->off0) {
        mt_put_kernel<<<1, 1, 0, stream>>>(slot, s->gauss0);
        LAUNCH_CHECK();
    }
    // one wave per sub-segment of ~3400 accepted attempts (MT_SUB_BLOCKS x 156 attempts x pi / 4); the loop strides
    const long est = (long)((w_hi - w_lo) / 2 / 3300) + 8;
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((est + MT_WG / 64 - 1) / (MT_WG / 64), (long)ctx->num_cu * 64));
    mt_emit_kernel<<<grid, MT_WG, 0, stream>>>(s->sub_state, s->nsub, s->sub_base, s->d_first + r, w_lo, w_hi, s->off0, s->n, slot, s->st);
    LAUNCH_CHECK();
    return 0;
}

int mt_stream_finish(corahip_ctx *ctx, hipStream_t stream, mt_session *s, corahip_mt_state *state) {
    ARG_CHECK(s != nullptr && state != nullptr);
    mt_status hs;
    HIP_TRY(hipMemcpyAsync(&hs, s->st, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (s->pairs == 0) {                           // n = 1 with a cached value: no attempt was needed, the key stays
        state->has_gauss = 0;
        state->gauss = 0.0;
        return 0;
    }
    if (!hs.done || hs.total_accepted < s->pairs) {
        corahip_set_error("normals_mt19937_legacy: %llu accepted attempts in %ld segments, %llu needed", hs.total_accepted, s->nseg, s->pairs);
        return CORAHIP_ESTATE;
    }
    std::copy(hs.key, hs.key + MTN, state->key);
    state->pos = hs.pos;
    state->has_gauss = hs.has_gauss;
    state->gauss = hs.gauss;
    return 0;
}

void mt_stream_free(mt_session *s) { delete s; }

extern "C" {

int corahip_normals_mt19937_legacy(corahip_ctx *ctx, corahip_mt_state *state, int64_t n, double *g) {
    ARG_CHECK(ctx != nullptr && state != nullptr && n >= 0 && (n == 0 || g != nullptr));
    ARG_CHECK(state->pos >= 0 && state->pos <= MTN);
    if (n == 0) return 0;
    if (ctx->draw_pending) {
        corahip_set_error("normals_mt19937_legacy: a corahip_draw_alm_numpy_begin session is pending (its tables share this call's scratch)");
        return CORAHIP_ESTATE;
    }
    StageTimer timer(ctx, "normals_legacy");
    mt_session *s = nullptr;
    int rc = mt_stream_prepare(ctx, ctx->stream, state, n, std::vector<unsigned long long>{0ull, (unsigned long long)n}, &s);
    if (rc) return rc;
    rc = mt_stream_emit_range(ctx, ctx->stream, s, 0, g);
    if (!rc) rc = mt_stream_finish(ctx, ctx->stream, s, state);
    mt_stream_free(s);
    return rc;
}

}  // extern "C"
