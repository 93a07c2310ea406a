// draw.hip - correlated Gaussian a_lm draw:  a_lm(nu) = sum_nu' T_l[nu,nu'] g_lm(nu')
//
// Replaces nputil.complex_std_normal (cora/util/nputil.py:104-125) in its device
// (counter-based) form and the np.dot of cora/core/skysim.py:121, writing a_lm in
// the device layout consumed by the synthesis ([nalm][g][c][v], see sht_internal.h).
//
// K3 is one FP64 MFMA GEMM per l: C[(c,m)][nu] = sum_nu' G[(c,m)][nu'] * T_l[nu][nu'],
// rows = the 2(l+1) real/imag normal vectors of that l, cols = channels.  T_l is staged
// k-chunk by k-chunk through LDS (transposed on the fly); the normals are read straight
// from the stream-ordered buffer (each element is used by exactly one wave).
#include "common.h"

#include <cstdlib>
#include <type_traits>

#include "rng_dev.h"
#include "stream_internal.h"

// The device stream: the real and the imaginary normal of (l, nu', m) are the two Box-Muller outputs of Philox
// counter {lo = m, hi = l F + nu'} under key = seed, built from the four output words as rng_dev.h describes
// (u1 from 52 bits of (r0, r1), the angle from 60 bits of (r2, r3)).
// A value depends only on (seed, l, nu', m, re/im): the same for any number of GPUs, and the same whether it is
// materialised in HBM (normals_kernel, stream-order layout) or generated inside K3.  oracle/philox.py
// restates the stream in numpy.
__device__ static inline double2 philox_normal_pair(uint64_t seed, int l, int F, int nup, int m,
                                                    const double2 *lg = RNG_LOG_TAB, const double2 *sc = RNG_SC_TAB) {
    const uint64_t ctr = ((uint64_t)((uint32_t)l * (uint32_t)F + (uint32_t)nup) << 32) | (uint32_t)m;
    return philox_boxmuller(ctr, seed, lg, sc);
}

// one thread per (l, nu', m): writes the stream-order buffer  g[F l(l+1) + c F(l+1) + nu'(l+1) + m], c = 0 (re), 1 (im)
__global__ void normals_kernel(uint64_t seed, int lmax, int F, double *__restrict__ g) {
    const int l = blockIdx.y;
    const int lp1 = l + 1;
    const long n = (long)F * lp1;
    double *gl = g + (size_t)F * l * lp1;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int m = (int)(q % lp1);
        const int nup = (int)(q / lp1);
        const double2 v = philox_normal_pair(seed, l, F, nup, m);
        gl[q] = v.x;
        gl[n + q] = v.y;
    }
}

// ------------------------------------------------------------------------------------
// K3: per-l GEMM on FP64 MFMA
// ------------------------------------------------------------------------------------
#ifndef DRAW_ABLATE
#define DRAW_ABLATE 0  // diagnostic builds of the fused-RNG kernel: 1 no RNG, 2 no MFMA, 3 no a_lm stores, 4 no staging of T
#endif
#ifndef DRAW_KK_UNROLL
#define DRAW_KK_UNROLL 2   // unroll factor of the k-step loop of a chunk in the fused-RNG kernel (two generator chains interleaved: 7.58 -> 7.45 ms; 4: the same)
#endif
#ifndef DRAW_BEARLY
#define DRAW_BEARLY 0  // 1: B-operand LDS reads of a k-step issued before its generator chain
#endif
#define DRAW_KC 32   // nu' per LDS stage
#define DRAW_ROWS 64 // (c,m) rows per block (4 waves x 16)

// NCT = 16-column tiles per block (block covers 16*NCT channels starting at col0)
template <int NCT>
__global__ void __launch_bounds__(256)
draw_kernel(const double *__restrict__ T, size_t t_ldl, int t_row0, const int32_t *__restrict__ info,
            const double *__restrict__ g, size_t g_off, int l_lo, int lmax, int F, int nu0, int nnu, int Gout,
            double *__restrict__ alm) {
    constexpr int NC = 16 * NCT;
    constexpr int STRIDE = DRAW_KC + 2;  // doubles per channel row: 272 B, so 16 consecutive rows hit 16 distinct 16-B slots
    extern __shared__ __attribute__((aligned(16))) double lds[];  // Bs[n][k] = T_l[nu0+col0+n][k0+k], [NC][STRIDE]

    const int l = l_lo + blockIdx.x;
    const int nrow = 2 * (l + 1);
    const int row0 = blockIdx.y * DRAW_ROWS;
    if (row0 >= nrow) return;
    const int col0 = blockIdx.z * NC;  // local channel index of first column
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    const int lp1 = l + 1;

    // stream offset of this l: sum_{l'<l} 2 F (l'+1) = F l (l+1); g[0] is element g_off of the stream
    const double *gl = g + ((size_t)F * l * (l + 1) - g_off);
    const double *Tl = T + (size_t)l * t_ldl - (size_t)t_row0 * F;  // row nu of T_l at Tl + nu F (rows < t_row0 never read)
    const bool dense = (info == nullptr) || (info[l] != 0);

    // A operand row of this lane
    const int rr = row0 + wave * 16 + ri;
    const bool row_ok = rr < nrow;
    const int c_of = rr >= lp1 ? 1 : 0;
    const int m_of = rr - c_of * lp1;
    const double *grow = gl + (size_t)c_of * F * lp1 + m_of;  // + nu' * lp1

    d4_t acc[NCT];
#pragma unroll
    for (int t = 0; t < NCT; t++) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // lower-triangular T: columns nu only need nu' <= nu
    const int kmax = dense ? F : min(F, nu0 + col0 + NC);
    for (int k0 = 0; k0 < kmax; k0 += DRAW_KC) {
        __syncthreads();
        // stage the 256-byte k-run of every channel row: 16 lanes x 16 B per row, coalesced in HBM/L2
        // and conflict-free in LDS (row stride 272 B)
        for (int it = tid; it < NC * (DRAW_KC / 2); it += 256) {
            const int n = it / (DRAW_KC / 2), q = it % (DRAW_KC / 2);
            const int nu = nu0 + col0 + n;
            double2 v = make_double2(0.0, 0.0);
            const int k = k0 + 2 * q;
            if (col0 + n < nnu && nu < F) {
                if (k + 1 < F) v = *reinterpret_cast<const double2 *>(Tl + (size_t)nu * F + k);
                else if (k < F) v.x = Tl[(size_t)nu * F + k];
            }
            *reinterpret_cast<double2 *>(lds + n * STRIDE + 2 * q) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DRAW_KC / 4; kk++) {
            const int kbase = k0 + 4 * kk;
            if (kbase >= kmax) break;
            const int kp = kbase + kq;
            double a = 0.0;
            if (row_ok && kp < F) a = grow[(size_t)kp * lp1];
            const double *bs = lds + ri * STRIDE + 4 * kk + kq;
#pragma unroll
            for (int t = 0; t < NCT; t++) {
                // triangular skip: tile t holds channels nu0+col0+16t .. +15
                if (!dense && kbase > nu0 + col0 + 16 * t + 15) continue;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bs[16 * t * STRIDE], acc[t], 0, 0, 0);
            }
        }
    }

    // epilogue: 1/sqrt(2) of complex_std_normal, store into [idx][g][c][v]
    const double sc = 0.70710678118654752440;
    const long base = alm_idx(l, 0, lmax);  // idx(l,m) = m(2 lmax+1-m)/2 + l
#pragma unroll
    for (int t = 0; t < NCT; t++) {
        const int col = col0 + 16 * t + ri;
        if (col >= 4 * Gout) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int ro = row0 + wave * 16 + kq + 4 * r;
            if (ro < nrow) {
                const int c = ro >= lp1 ? 1 : 0;
                const int m = ro - c * lp1;
                const long idx = (long)m * (2 * lmax + 1 - m) / 2 + l;
                (void)base;
                alm[((size_t)idx * Gout + (col >> 2)) * 8 + c * 4 + (col & 3)] = acc[t][r] * sc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// K3 with the normals generated in registers (device-RNG mode): no 8.6 GB normal buffer is written or read.
//
// Work item = (l, block of 128 m, column group of 16 NCT channels); persistent workgroups of 8 waves pull items
// from an atomic queue (heavy column groups first).  Wave tile = 16 m x (re, im) x 16 NCT channels: the two
// Box-Muller outputs of one Philox block are the real and the imaginary normal of the SAME (l, m, nu'), so a wave
// owns whole 64-byte a_lm cells ([re x4 | im x4]) and stores them as full lines (16 bytes per lane after a quad
// exchange).  T_l is staged by LDS-DMA through a ring of three 32-nu' stages that runs on across items: the first
// two stages of the next item are requested before the epilogue stores of the current one.
// Round-2 form, for the record (stamps, `make k3stamps`): 4-wave workgroups of one (l, 64 m, column group) each
// spent 13 % of their wave cycles in the prologue (tables, first stage), 30 % issuing / waiting for the LDS-DMA of
// T_l - every 64 m re-staged the same 256 KB - and 28 % in an epilogue of half-line (32-byte) stores.
// ------------------------------------------------------------------------------------
// LDS-DMA of 16 bytes per lane from inline asm (see sht_internal.h: hipcc would drain a builtin DMA with
// vmcnt(0) before every later ds_read); lane i's bytes land at lds_byte_addr + 16 i.
__device__ static inline void draw_glds16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

#ifndef DRAW_STAMPS
#define DRAW_STAMPS 0   // diagnostic build (make k3stamps): s_memtime per phase, summed over all waves
#endif
#if DRAW_STAMPS
__device__ unsigned long long g_draw_stamps[8];
#define DSTAMP(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); d_acc[k] += _t - d_last; d_last = _t; }
#else
#define DSTAMP(k)
#endif

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), in order
template <int N, int J = 0, class Fn>
__device__ static inline void draw_static_for(Fn &&f) {
    if constexpr (J < N) {
        f(std::integral_constant<int, J>{});
        draw_static_for<N, J + 1>(f);
    }
}

#define DRAW_WAVES 8                 // waves per workgroup of the fused-RNG kernel
#define DRAW_MB (16 * DRAW_WAVES)    // m per work item
#define DRAW_NBUF 3                  // stages in the LDS ring

// number of (l, m-block) slots of the multipoles l_lo .. l_hi: l in band j = l >> 7 has j + 1 blocks of 128 m
static inline long draw_slots(int l_lo, int l_hi) {
    long n = 0;
    for (int l = l_lo; l <= l_hi; l++) n += (l / DRAW_MB) + 1;
    return n;
}

// FROMG: the normals come from a stream-order buffer `gsrc` (the caller's numpy stream: uploaded, or generated on the
// device by corahip_normals_pcg64) instead of the Philox chain.  The A operands of a chunk (8 k-steps x (re, im) = 16
// doubles per lane) are requested one chunk ahead by asm loads that take their place in the counted vmcnt scheme of the
// staging by HALF chunks: behind the barrier of chunk c a wave requests the second half of chunk c, THEN the DMA pieces
// of stage c + 2 - the wait in front of the second half leaves those QPW pieces in flight and covers the loads; the
// first half of chunk c + 1 is requested at that point and waited for, with everything else, at the next chunk begin.
template <int NCT, bool FROMG = false, int NB = DRAW_NBUF>
__global__ void __launch_bounds__(64 * DRAW_WAVES, 2)
draw_rng_kernel(const double *__restrict__ T, size_t t_ldl, int rows, const int32_t *__restrict__ info,
                const double *__restrict__ zeros, uint64_t seed, const double *__restrict__ gsrc, size_t g_off, int l_lo,
                int l_hi, int lmax, int F, int nu0, int nu1, int cw, int nnu, int Gout, int nslots, int ncg0, int ncg,
                double *__restrict__ alm, unsigned *__restrict__ queue, const unsigned *__restrict__ slot_tab) {
    constexpr int NC = 16 * NCT;
    constexpr int ROWD = DRAW_KC;            // doubles per channel row in LDS: 256 B, unpadded (DMA is lane-linear)
    constexpr int BUF = NC * ROWD;           // doubles per stage
    constexpr int QPW = (NC / 4 + DRAW_WAVES - 1) / DRAW_WAVES;   // row quads (= DMA instructions) a staging wave issues per stage
    static_assert(QPW * DRAW_WAVES == NC / 4 || NC / 4 < DRAW_WAVES, "every staging wave issues the same number of pieces");
    // Bs[n][slot' = slot ^ (n & 15)][2]: the 16-byte slots of a row are XOR-swizzled with the row number
    // (applied on the DMA source address), so that 16 rows read at the same k hit 16 distinct slots
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [NB][NC][ROWD]
    static_assert(NB == 3 || (NB == 2 && !FROMG), "two stages: the Philox instantiation only (no operand loads to count)");
    __shared__ double2 lg_s[257], sc_s[256];   // LDS copies of the Box-Muller tables (rng_dev.h): 8 KB
    __shared__ int s_next[2];                  // next work item, double-buffered (written one item ahead)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ri = lane & 15, kq = lane >> 4;
    const int nitems = nslots * ncg;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;
#if DRAW_STAMPS
    unsigned long long d_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d_last;
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); d_last = _t; }
#endif

    // The launch's channels: local channel c < cw is global channel nu0 + c, local channel c >= cw is nu1 + (c - cw) (the
    // second chunk of a FOLDED frequency shard, parallel.py; cw >= nnu: one contiguous block).  Column groups never
    // straddle the two chunks: groups 0 .. ncg0 - 1 tile chunk 0, the others chunk 1.
    struct item_t {
        int l, mb, base0, lbase, lend, kmax, nchunk, c_full;
        bool tri_tail;
        const double *Tl;        // row of the item's column n (global channel base0 + n, local channel lbase + n) at Tl + n F
    };
    auto decode = [&](int it) {
        item_t w;
        const int cg = ncg - 1 - it / nslots;          // the column groups with the longest nu' range first
        // slot -> (l, m block): the multipoles of the launch from l_hi down, the blocks of an l in order - a table of the
        // whole range 0 .. lmax (draw_slot_table: the same order; the launch's slots are the run that starts at its l_hi),
        // read by one scalar load (round 5: the walk over the bands of l with its integer divisions was 4-6 % of the
        // kernel's wave cycles, per item)
        const int s = it - (it / nslots) * nslots;
        const unsigned e = slot_tab[s];
        const int l = (int)(e >> 8), mb = (int)(e & 255u);
        w.l = l;
        w.mb = mb;
        const bool second = cg >= ncg0;
        w.lbase = second ? cw + (cg - ncg0) * NC : cg * NC;        // first local channel of the item
        w.lend = second ? nnu : min(cw, nnu);                      // end of its chunk (local)
        w.base0 = second ? nu1 + (cg - ncg0) * NC : nu0 + cg * NC; // first (global) channel of the item
        const bool dense = (info == nullptr) || (info[l] != 0);
        // lower-triangular T: channel nu only needs nu' <= nu
        w.kmax = dense ? F : min(F, w.base0 + NC);
        w.nchunk = (w.kmax + DRAW_KC - 1) / DRAW_KC;
        // Tile t (channels base0 + 16 t .. + 15) of a triangular factor is zero for nu' > base0 + 16 t + 15.  With base0 a
        // multiple of the chunk length the chunks below the item's own channels take every tile and the chunks across
        // them drop one tile per half: that tail is unrolled so that every half knows its tiles at compile time (per-tile
        // tests inside the k-steps were 45 scalar instructions per k-step, issue time next to the MFMAs; a run-time
        // dispatch per half made the register allocator copy the accumulators between the cases).  A tile that is kept
        // is multiplied as a whole: the entries above the diagonal are stored zeros.  Any other base0 (uneven channel
        // shards) takes every tile up to kmax - correct for the same reason, just not minimal.
        w.tri_tail = !dense && (w.base0 % DRAW_KC) == 0;
        w.c_full = w.tri_tail ? min(w.nchunk, w.base0 / DRAW_KC) : w.nchunk;
        w.Tl = T + (size_t)l * t_ldl + (size_t)(rows ? w.lbase : w.base0) * F;    // (a row block holds the local channels in order)
        return w;
    };
    // stage chunk c of the item (rows base0 .. base0+NC-1 of T_l, nu' in [c KC, c KC + KC)) into ring slot `slot`:
    // each wave-instruction moves 4 rows x 16 slots; every row is staged (zeros past nnu / F), so that every
    // staging wave issues exactly QPW pieces per stage and the waits below can count them
    auto stage = [&](const item_t &w, int c, int slot) {
        const int k0 = c * DRAW_KC;
#pragma unroll
        for (int it = 0; it < QPW; it++) {
            const int rq = wave + DRAW_WAVES * it;  // row quad index
            if (rq >= NC / 4) break;
            const int n = 4 * rq + (lane >> 4);     // row of this lane
            const int slot_dst = lane & 15;
            const int slot_src = slot_dst ^ (n & 15);
            const int nu = w.base0 + n;
            const int k = k0 + 2 * slot_src;
            const double *src = zeros;  // F is even on this path (host wrapper), so k + 1 < F whenever k < F
            if (w.lbase + n < w.lend && nu < F && k + 1 < F) src = w.Tl + (size_t)n * F + k;
#if DRAW_ABLATE != 4   // diagnostic 4: no staging of T
            draw_glds16(src, lds_base + (unsigned)((slot * BUF + 4 * rq * ROWD) * sizeof(double)));
#else
            (void)src;
#endif
        }
    };

    if constexpr (!FROMG) {
        for (int i = tid; i < 257; i += 64 * DRAW_WAVES) lg_s[i] = RNG_LOG_TAB[i];   // (visible after the first barrier)
        for (int i = tid; i < 256; i += 64 * DRAW_WAVES) sc_s[i] = RNG_SC_TAB[i];
    }

    // FROMG: A operands by HALF chunks (4 k-steps x (re, im) = 8 doubles per lane): a_x holds the first half of a chunk,
    // a_y the second; each is requested half a chunk before its use
    double a_x[8], a_y[8];
    // requests half `half` of chunk c of item `w` for this wave's rows: element (l, c, nu', m) of the stream-order buffer
    // sits at F l (l + 1) + c F (l + 1) + nu' (l + 1) + m - g_off (g_off = F l_lo (l_lo + 1) when gsrc holds the stream
    // from multipole l_lo on: one slot of the l-range ring); rows past l and nu' past F - 1 are read from the clamped
    // address (finite values: their products meet staged zeros or are never stored)
    auto issue_a = [&](const item_t &w, int c, int half, double (&dst)[8]) {
        const int lp1 = w.l + 1;
        const int m = min(w.mb * DRAW_MB + 16 * wave + ri, w.l);
        const double *gre = gsrc + ((size_t)F * w.l * lp1 - g_off) + m;
        const size_t im_off = (size_t)F * lp1;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int kp = min(c * DRAW_KC + 16 * half + 4 * kk + kq, F - 1);
            const double *pr = gre + (size_t)kp * lp1;
            const double *pi = pr + im_off;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst[2 * kk]) : "v"(pr) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst[2 * kk + 1]) : "v"(pi) : "memory");
        }
    };
    // behind the wait that covers them: ties the later uses of the values to this point
    auto pin_a = [&](double (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("" : "+v"(dst[i]));
    };

    int item = blockIdx.x;      // the first gridDim.x items are pre-assigned; the queue starts behind them
    if (item >= nitems) return;
    item_t w = decode(item);
    int ring = 0, par = 0;
    stage(w, 0, 0);
    if (NB > 2 && w.nchunk > 1) stage(w, 1, 1);
    if constexpr (FROMG) {
        if (w.mb * DRAW_MB + 16 * wave < w.l + 1) issue_a(w, 0, 0, a_x);
    }
    DSTAMP(0);                           // prologue + first stages issued

    for (;;) {
        if (tid == 0) s_next[par] = (int)(gridDim.x + atomicAdd(queue, 1u));   // latency hidden behind this item
        const int l = w.l, lp1 = w.l + 1;
        const int mw = w.mb * DRAW_MB + 16 * wave;     // first m of this wave
        const bool wave_has_rows = mw < lp1;
        const int m_lane = mw + ri;                    // the A-operand row of this lane
        const int kmax = w.kmax, nchunk = w.nchunk;
        d4_t acc0[NCT], acc1[NCT];                     // real / imaginary part of a_lm: rows m = mw + kq + 4 r, channel 16 t + ri
#pragma unroll
        for (int t = 0; t < NCT; t++) {
            acc0[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            acc1[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
        }
        // k-steps of one half (16 nu' = 4 k-steps) of the chunk in buffer `sb`, multiplying the tiles TMIN .. NCT-1
        auto half_steps = [&](auto tmin_c, const double *sb, int k0, auto half_c) {
            constexpr int TMIN = decltype(tmin_c)::value;
            constexpr int half = decltype(half_c)::value;
            constexpr int UNR = FROMG ? 4 : (NCT > 8 ? 1 : DRAW_KK_UNROLL);     // (FROMG indexes its operand registers by kk; the 256-column shape has no registers for a second chain)
#pragma unroll UNR
            for (int kk = 4 * half; kk < 4 * half + 4; kk++) {
                const int kp = k0 + 4 * kk + kq;
                const int kl = 4 * kk + kq;          // k within the chunk
                // all B operands of the k-step are read up front (one address + immediate offsets; the 256-column shape:
                // the first eight tiles' - the others behind the first MFMAs)
                double bv[NCT > 8 ? 8 : NCT];
                const double *brow = sb + ri * ROWD + 2 * ((kl >> 1) ^ ri) + (kl & 1);
#pragma unroll
                for (int t = TMIN; t < (NCT > 8 ? 8 : NCT); t++) bv[t] = brow[16 * t * ROWD];
#if DRAW_ABLATE == 1   // diagnostic: no RNG
                double2 a = make_double2(1.0 + kp, 0.5 * m_lane);
#else
                // (rows past l and nu' >= F are generated like any other: their products meet staged zeros or are
                //  never stored - no exec masking around the chain)
                double2 a;
                if constexpr (FROMG) a = half ? make_double2(a_y[2 * (kk - 4 * half)], a_y[2 * (kk - 4 * half) + 1])
                                              : make_double2(a_x[2 * (kk - 4 * half)], a_x[2 * (kk - 4 * half) + 1]);
                else a = philox_normal_pair(seed, l, F, kp, m_lane, lg_s, sc_s);
#endif
#pragma unroll
                for (int t = TMIN; t < (NCT > 8 ? 8 : NCT); t++) {
#if DRAW_ABLATE == 2   // diagnostic: no MFMA
                    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(bv[t]));
#else
                    acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bv[t], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bv[t], acc1[t], 0, 0, 0);
#endif
                }
                if constexpr (NCT > 8) {
#pragma unroll
                    for (int t = (TMIN > 8 ? TMIN : 8); t < NCT; t++) bv[t - 8] = brow[16 * t * ROWD];
#pragma unroll
                    for (int t = (TMIN > 8 ? TMIN : 8); t < NCT; t++) {
                        acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bv[t - 8], acc0[t], 0, 0, 0);
                        acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bv[t - 8], acc1[t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (the next k-step's generator chain must not be scheduled across: no registers for it)
                }
            }
        };
        // Stage c of the item sits in ring slot (ring + c) % NBUF.  On entry stages 0 and 1 have been requested (before
        // the previous item's stores): the first wait drains everything; later waits leave the QPW pieces of the
        // younger stage in flight.  After the barrier of chunk c every wave has finished chunk c - 1, whose slot
        // takes stage c + 2.
        // FROMG, in front of the second half of chunk c: its operands are older than the stage requested at the chunk's
        // start; then the first half of the next chunk is requested (a_x is free: the first half is done)
        auto mid_chunk = [&](int c) {
            if constexpr (FROMG) {
                // (a wave that stages nothing - the 16-column shape has four row quads for eight waves - has only its own
                //  operand loads in flight: it waits for all of them; counting QPW pieces it never issued would let the
                //  youngest operand load through unfinished)
                constexpr bool ALL_STAGE = QPW * DRAW_WAVES == NC / 4;
                if (c + 2 < nchunk && (ALL_STAGE || wave < NC / 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pin_a(a_y);
                if (c + 1 < nchunk) issue_a(w, c + 1, 0, a_x);
            }
        };
        auto chunk_begin = [&](int c) {
            // FROMG: the first-half operands of this chunk were requested at the middle of the previous one, BEHIND the
            // DMA pieces of stage c + 1 - they are the youngest requests in flight, so everything is waited for (stage
            // c + 1 has then had one chunk, not two, to land; the counted wait is the one in front of the second half)
            // (NB = 2 - the 256-column shape, whose three stages would not fit the LDS -: stage c was requested behind the
            //  barrier of chunk c - 1 and nothing younger is in flight)
            if (NB == 2 || FROMG || c == 0 || c + 1 >= nchunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPW) : "memory");
            if constexpr (FROMG) {
                if (wave_has_rows) pin_a(a_x);
            }
            __syncthreads();
#if DRAW_STAMPS
            if (c == 0) { DSTAMP(6); } else
#endif
            DSTAMP(1);                       // wait for the stage + barrier (stamp 6: the first chunk of an item, which also waits for the previous item's stores)
            if constexpr (FROMG) {
                // second half of this chunk: ahead of the DMA pieces, so that the wait in front of the half leaves them
                if (wave_has_rows && c * DRAW_KC + 16 < kmax) issue_a(w, c, 1, a_y);
            }
            if (c + NB - 1 < nchunk) stage(w, c + NB - 1, (ring + c + NB - 1) % NB);
            DSTAMP(2);                       // issue of the stage after next
        };
        int c = 0;
        for (; c < w.c_full; c++) {
            chunk_begin(c);
            if (!wave_has_rows) continue;
            const double *sb = lds + ((ring + c) % NB) * BUF;
            half_steps(std::integral_constant<int, 0>{}, sb, c * DRAW_KC, std::integral_constant<int, 0>{});
            if (c * DRAW_KC + 16 < kmax) {
                mid_chunk(c);
                half_steps(std::integral_constant<int, 0>{}, sb, c * DRAW_KC, std::integral_constant<int, 1>{});
            }
            DSTAMP(3);                       // generator + MFMAs of the chunk
        }
        draw_static_for<(NC + DRAW_KC - 1) / DRAW_KC>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if (c >= nchunk) return;         // (uniform; also the dense / unaligned case, where c == nchunk here)
            chunk_begin(c);
            if (wave_has_rows) {
                const double *sb = lds + ((ring + c) % NB) * BUF;
                half_steps(std::integral_constant<int, (2 * J < NCT ? 2 * J : NCT)>{}, sb, c * DRAW_KC, std::integral_constant<int, 0>{});
                if (c * DRAW_KC + 16 < kmax) {
                    mid_chunk(c);
                    if constexpr (2 * J + 1 < NCT)
                        half_steps(std::integral_constant<int, (2 * J + 1 < NCT ? 2 * J + 1 : NCT)>{}, sb, c * DRAW_KC,
                                   std::integral_constant<int, 1>{});
                }
                DSTAMP(3);
            }
            c++;
        });

        // ---- next item: its first two stages are requested now, ahead of this item's stores.  Only the slot of the
        //      last chunk can still be in use by a slower wave, and the ring moves on past it.
        ring = (ring + nchunk) % NB;
        const int cur_lbase = w.lbase, cur_lend4 = (w.lend + 3) & ~3;
        item = __builtin_amdgcn_readfirstlane(s_next[par]);   // (written before this item's first barrier)
        par ^= 1;
        const bool have_next = item < nitems;
        if (have_next) {
            w = decode(item);
            stage(w, 0, ring);
            if (NB > 2 && w.nchunk > 1) stage(w, 1, (ring + 1) % NB);      // (NB = 2: that slot is the last chunk's, maybe still read)
            if constexpr (FROMG) {
                if (w.mb * DRAW_MB + 16 * wave < w.l + 1) issue_a(w, 0, 0, a_x);
            }
        }
        DSTAMP(4);                           // next item decoded, its first stages issued

        // ---- epilogue: 1/sqrt(2) of complex_std_normal; lane (ri, kq) holds re and im of rows m = mw + kq + 4 r, channel
        //      16 t + ri.  The four lanes of a quad hold one 64-byte cell [re x4 | im x4]: lane q of the quad stores its
        //      16-byte quarter - (re0, re1), (re2, re3), (im0, im1), (im2, im3) - after a quad exchange (DPP), so every
        //      store instruction writes 16 full lines.
        if (wave_has_rows) {
            const double sc = 0.70710678118654752440;
            const int q = ri & 3;
            const bool upper = q >= 2;                 // this lane stores imaginary parts
            auto quad = [](double v, auto ctrl_c) {    // v of the quad lane selected by the quad_perm pattern
                constexpr int CTRL = decltype(ctrl_c)::value;
                int lo = __double2loint(v), hi = __double2hiint(v);
                lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
                hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
                return __hiloint2double(hi, lo);
            };
#pragma unroll
            for (int t = 0; t < NCT; t++) {
                const int col = cur_lbase + 16 * t + (ri & ~3);          // first (local) channel of the quad's cell
                const bool col_ok = col < cur_lend4;                      // (the chunk's end rounded up to a cell: the whole cell or none)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double re = acc0[t][r] * sc, im = acc1[t][r] * sc;
                    // quarter 0 / 2 takes the values of quad lanes (0, 1), quarter 1 / 3 those of lanes (2, 3)
                    const double re_a = quad(re, std::integral_constant<int, 0x88>{});   // from lanes [0, 2, 0, 2]
                    const double im_a = quad(im, std::integral_constant<int, 0x88>{});
                    const double re_b = quad(re, std::integral_constant<int, 0xDD>{});   // from lanes [1, 3, 1, 3]
                    const double im_b = quad(im, std::integral_constant<int, 0xDD>{});
                    const int m = mw + kq + 4 * r;
                    if (col_ok && m < lp1) {
                        const long idx = (long)m * (2 * lmax + 1 - m) / 2 + l;
#if DRAW_ABLATE == 3   // diagnostic: no a_lm stores
                        if (re == 1.2345e300)
#endif
                        *reinterpret_cast<double2 *>(alm + ((size_t)idx * Gout + (col >> 2)) * 8 + 2 * q) =
                            upper ? make_double2(im_a, im_b) : make_double2(re_a, re_b);
                    }
                }
            }
        }
        DSTAMP(5);                           // epilogue (scale, exchange, a_lm stores issued)
        if (!have_next) break;
    }
#if DRAW_STAMPS
    if (lane == 0) {
        for (int k = 0; k < 7; k++) atomicAdd(&g_draw_stamps[k], d_acc[k]);
        atomicAdd(&g_draw_stamps[7], 1ull);
    }
#endif
}

// l_lo .. l_hi: the multipoles of this launch (0 .. lmax: all); FROMG: gsrc[0] is element g_off of the stream-order
// buffer; `stream`: where the launch goes (the context's stream, or the draw stream of the l-range pipeline)
// chan: the launch's channels {nu0, nu1, cw, nnu} (see the kernel); rows: T holds the row block of the local channels
struct draw_chan {
    int nu0, nu1, cw, nnu;
};
// (l, m block) of every slot of the range 0 .. lmax in the order the kernel takes them: l from lmax down, the l / 128 + 1
// blocks of an l in order; entry = l << 8 | block.  Built once per lmax, kept in the context.
static int draw_slot_table(corahip_ctx *ctx, int lmax, const unsigned **tab) {
    if (ctx->draw_slot_tab && ctx->draw_slot_lmax == lmax) {
        *tab = ctx->draw_slot_tab;
        return 0;
    }
    std::vector<unsigned> h;
    h.reserve((size_t)draw_slots(0, lmax));
    for (int l = lmax; l >= 0; l--)
        for (int mb = 0; mb <= l / DRAW_MB; mb++) h.push_back(((unsigned)l << 8) | (unsigned)mb);
    if (ctx->draw_slot_tab) (void)hipFree(ctx->draw_slot_tab);
    ctx->draw_slot_tab = nullptr;
    ctx->draw_slot_lmax = -1;
    unsigned *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(unsigned) * h.size()));
    HIP_TRY(hipMemcpy(d, h.data(), sizeof(unsigned) * h.size(), hipMemcpyHostToDevice));   // (synchronous: visible to every stream)
    ctx->draw_slot_tab = d;
    ctx->draw_slot_lmax = lmax;
    *tab = d;
    return 0;
}

template <int NCT, bool FROMG = false, int NB = DRAW_NBUF>
static int launch_draw_rng(corahip_ctx *ctx, hipStream_t stream, const double *T, size_t t_ldl, int rows, const int32_t *info,
                           uint64_t seed, const double *gsrc, size_t g_off, int l_lo, int l_hi, int lmax, int F,
                           draw_chan ch, int Gout, double *alm) {
    constexpr int NC = 16 * NCT;
    const size_t shm = sizeof(double) * NB * NC * DRAW_KC;
    HIP_TRY(hipFuncSetAttribute((const void *)draw_rng_kernel<NCT, FROMG, NB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    // scratch slot 3: 4096 bytes of zeros (DMA source of padded rows) + the work queue counter behind them
    char *zq = nullptr;
    int rc = corahip_ctx_scratch(ctx, 3, 8192, (void **)&zq);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(zq, 0, 8192, stream));
    const int n0 = std::min(ch.cw, ch.nnu), n1 = ch.nnu - n0;
    const int ncg0 = (((n0 + 3) & ~3) + NC - 1) / NC, ncg = ncg0 + (((n1 + 3) & ~3) + NC - 1) / NC;
    const long nslots = draw_slots(l_lo, l_hi);
    const long nitems = nslots * ncg;
    ARG_CHECK(nitems < (1L << 30) && lmax < (1 << 24) && lmax / DRAW_MB < 256);
    const unsigned *slot_tab = nullptr;
    if ((rc = draw_slot_table(ctx, lmax, &slot_tab))) return rc;
    slot_tab += l_hi < lmax ? draw_slots(l_hi + 1, lmax) : 0;      // the launch's run of the table starts at its l_hi
    // persistent: one workgroup per CU for the 128-channel shape (106 KB of LDS), two for the narrower ones
    const int per_cu = (shm + 8300 > 80 * 1024) ? 1 : 2;
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    draw_rng_kernel<NCT, FROMG, NB><<<grid, 64 * DRAW_WAVES, shm, stream>>>(T, t_ldl, rows, info, (const double *)zq, seed, gsrc,
                                                                        g_off, l_lo, l_hi, lmax, F, ch.nu0, ch.nu1, ch.cw,
                                                                        ch.nnu, Gout, (int)nslots, ncg0, ncg, alm,
                                                                        (unsigned *)(zq + 4096), slot_tab);
    LAUNCH_CHECK();
#if DRAW_STAMPS
    {
        unsigned long long hs[8], z[8] = {0};
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_draw_stamps), sizeof(hs)));
        const double per = 1.0 / (double)std::max<unsigned long long>(hs[7], 1);
        fprintf(stderr, "K3 NCT=%d waves=%llu: cycles/wave  prologue %.0f wait+barrier %.0f (first chunk of an item %.0f) stage-issue %.0f rng+mfma %.0f next-item %.0f epilogue %.0f\n",
                NCT, hs[7], hs[0] * per, hs[1] * per, hs[6] * per, hs[2] * per, hs[3] * per, hs[4] * per, hs[5] * per);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_draw_stamps), z, sizeof(z)));
    }
#endif
    return 0;
}

// ------------------------------------------------------------------------------------
// layout converters
// ------------------------------------------------------------------------------------
// alm_dev [nalm][G][2][4] -> square [nnu][1][L][L] complex128 (m > l entries zero)
__global__ void dev_to_square_kernel(const double *__restrict__ alm, int lmax, int nnu, int G,
                                     double *__restrict__ sq) {
    const int L = lmax + 1;
    const long n = (long)nnu * L * L;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int m = (int)(q % L);
        const int l = (int)((q / L) % L);
        const int nu = (int)(q / ((long)L * L));
        double2 v = make_double2(0.0, 0.0);
        if (m <= l) {
            const long idx = (long)m * (2 * lmax + 1 - m) / 2 + l;
            const double *cell = alm + ((size_t)idx * G + (nu >> 2)) * 8 + (nu & 3);
            v = make_double2(cell[0], cell[4]);
        }
        *reinterpret_cast<double2 *>(sq + 2 * q) = v;
    }
}

// packed [nnu][nalm] complex128 (healpy order) -> alm_dev [nalm][G][2][4]; padding channels zero
__global__ void packed_to_dev_kernel(const double *__restrict__ packed, long nalm, int nnu, int G,
                                     double *__restrict__ alm) {
    const long n = nalm * G * 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int v = (int)(q & 3);
        const int gg = (int)((q >> 2) % G);
        const long idx = q / (4L * G);
        const int nu = 4 * gg + v;
        double2 val = make_double2(0.0, 0.0);
        if (nu < nnu) val = *reinterpret_cast<const double2 *>(packed + 2 * ((size_t)nu * nalm + idx));
        double *cell = alm + ((size_t)idx * G + gg) * 8 + v;
        cell[0] = val.x;
        cell[4] = val.y;
    }
}

template <int NCT>
static int launch_draw(corahip_ctx *ctx, hipStream_t stream, const double *T, size_t t_ldl, int t_row0, const int32_t *info,
                       const double *g, size_t g_off, int l_lo, int l_hi, int lmax, int F, int nu0, int nnu, int Gout,
                       double *alm) {
    constexpr int NC = 16 * NCT;
    const size_t shm = sizeof(double) * NC * (DRAW_KC + 2);
    HIP_TRY(hipFuncSetAttribute((const void *)draw_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    dim3 grid(l_hi - l_lo + 1, (2 * (l_hi + 1) + DRAW_ROWS - 1) / DRAW_ROWS, (4 * Gout + NC - 1) / NC);
    draw_kernel<NCT><<<grid, 256, shm, stream>>>(T, t_ldl, t_row0, info, g, g_off, l_lo, lmax, F, nu0, nnu, Gout, alm);
    LAUNCH_CHECK();
    return 0;
}

extern "C" {

int corahip_normals_philox(corahip_ctx *ctx, uint64_t seed, int lmax, int F, double *g) {
    ARG_CHECK(ctx != nullptr && g != nullptr && lmax >= 0 && F >= 1);
    StageTimer t(ctx, "normals");
    dim3 grid(std::max(1, std::min(64, (F * (lmax + 1) + 255) / 256)), lmax + 1);
    normals_kernel<<<grid, 256, 0, ctx->stream>>>(seed, lmax, F, g);
    LAUNCH_CHECK();
    return 0;
}

// widest chunk of the launch's channels, in columns of whole cells: decides the tile shape
static inline int chan_ncol(const draw_chan &ch) {
    const int n0 = std::min(ch.cw, ch.nnu), n1 = ch.nnu - n0;
    return (std::max(n0, n1) + 3) & ~3;
}
static inline bool chan_folded(const draw_chan &ch) { return ch.nnu > ch.cw; }

// K3 of the multipoles l_lo .. l_hi from a stream-order buffer whose first element is element g_off of the stream
static int draw_stream_range(corahip_ctx *ctx, hipStream_t stream, const double *T, int rows, const int32_t *info,
                             const double *g, size_t g_off, int l_lo, int l_hi, int lmax, int F, draw_chan ch,
                             double *alm_dev) {
    const int Gout = (ch.nnu + 3) / 4;
    const size_t t_ldl = rows ? (size_t)ch.nnu * F : (size_t)F * F;
    if (!(F & 1) && !getenv("CORAHIP_DRAW_GENERIC")) {
        // the persistent MFMA kernel of the device-RNG mode with its A operands read from the stream buffer
        const int ncol = chan_ncol(ch);
        if (ncol <= 16) return launch_draw_rng<1, true>(ctx, stream, T, t_ldl, rows, info, 0, g, g_off, l_lo, l_hi, lmax, F, ch, Gout, alm_dev);
        if (ncol <= 32) return launch_draw_rng<2, true>(ctx, stream, T, t_ldl, rows, info, 0, g, g_off, l_lo, l_hi, lmax, F, ch, Gout, alm_dev);
        if (ncol <= 64) return launch_draw_rng<4, true>(ctx, stream, T, t_ldl, rows, info, 0, g, g_off, l_lo, l_hi, lmax, F, ch, Gout, alm_dev);
        return launch_draw_rng<8, true>(ctx, stream, T, t_ldl, rows, info, 0, g, g_off, l_lo, l_hi, lmax, F, ch, Gout, alm_dev);
    }
    ARG_CHECK(!chan_folded(ch));          // (the generic kernel - odd F - takes one contiguous block of channels)
    const int ncol = 4 * Gout, nu0 = ch.nu0, nnu = ch.nnu, t_row0 = rows ? nu0 : 0;
    if (ncol <= 16) return launch_draw<1>(ctx, stream, T, t_ldl, t_row0, info, g, g_off, l_lo, l_hi, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 32) return launch_draw<2>(ctx, stream, T, t_ldl, t_row0, info, g, g_off, l_lo, l_hi, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 64) return launch_draw<4>(ctx, stream, T, t_ldl, t_row0, info, g, g_off, l_lo, l_hi, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 128) return launch_draw<8>(ctx, stream, T, t_ldl, t_row0, info, g, g_off, l_lo, l_hi, lmax, F, nu0, nnu, Gout, alm_dev);
    return launch_draw<16>(ctx, stream, T, t_ldl, t_row0, info, g, g_off, l_lo, l_hi, lmax, F, nu0, nnu, Gout, alm_dev);
}

static int draw_host_stream(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, const double *g, int lmax, int F,
                            draw_chan ch, double *alm_dev) {
    StageTimer t(ctx, "draw");
    return draw_stream_range(ctx, ctx->stream, T, rows, info, g, 0, 0, lmax, lmax, F, ch, alm_dev);
}

static int draw_philox(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, uint64_t seed, int lmax, int F,
                       draw_chan ch, double *alm_dev) {
    if (F & 1) {
        // odd F: a 16-byte LDS-DMA piece would straddle the end of a T row; materialise the (identical)
        // device stream and use the generic kernel instead
        double *g = nullptr;
        int rc = corahip_ctx_scratch(ctx, 1, sizeof(double) * 2 * (size_t)F * nalm_of(lmax), (void **)&g);
        if (rc) return rc;
        if ((rc = corahip_normals_philox(ctx, seed, lmax, F, g))) return rc;
        return draw_host_stream(ctx, T, rows, info, g, lmax, F, ch, alm_dev);
    }
    StageTimer t(ctx, "draw");
    const int Gout = (ch.nnu + 3) / 4;
    const size_t t_ldl = rows ? (size_t)ch.nnu * F : (size_t)F * F;
    const int ncol = chan_ncol(ch);
    if (ncol <= 16) return launch_draw_rng<1>(ctx, ctx->stream, T, t_ldl, rows, info, seed, nullptr, 0, 0, lmax, lmax, F, ch, Gout, alm_dev);
    if (ncol <= 32) return launch_draw_rng<2>(ctx, ctx->stream, T, t_ldl, rows, info, seed, nullptr, 0, 0, lmax, lmax, F, ch, Gout, alm_dev);
    if (ncol <= 64) return launch_draw_rng<4>(ctx, ctx->stream, T, t_ldl, rows, info, seed, nullptr, 0, 0, lmax, lmax, F, ch, Gout, alm_dev);
    // Round-6 experiment, OFF by default (CORAHIP_K3_WIDE=1 selects it; tools/k3_wide_probe.py): ONE column group of 256 per
    // (l, m block) where there are two of 128 - every normal pair generated once per 256 columns instead of once per 128
    // (the generator is half of this kernel's issue time), two 64 KB stages instead of three.  Same a_lm - and 23.1 ms
    // against 7.32: 128 accumulator registers + the generator chain do not fit the 256 of two waves per SIMD (1228 bytes
    // of scratch per lane, ~25 scratch operations per k-step in the MFMA loop, each a vmcnt-ordered memory instruction).
    static const bool wide = getenv("CORAHIP_K3_WIDE") && atoi(getenv("CORAHIP_K3_WIDE")) != 0;
    if (wide && ncol >= 256 && ch.nu1 == 0 && ch.cw >= ch.nnu)
        return launch_draw_rng<16, false, 2>(ctx, ctx->stream, T, t_ldl, rows, info, seed, nullptr, 0, 0, lmax, lmax, F, ch, Gout, alm_dev);
    return launch_draw_rng<8>(ctx, ctx->stream, T, t_ldl, rows, info, seed, nullptr, 0, 0, lmax, lmax, F, ch, Gout, alm_dev);
}

// a rank's channels as the C ABI hands them over: nchunks x chunk_nnu channels from nu0[0] (and nu0[1])
static int chan_of(const corahip_chanset *set, int F, draw_chan &ch) {
    ARG_CHECK(set != nullptr && (set->nchunks == 1 || set->nchunks == 2) && set->chunk_nnu >= 1);
    ARG_CHECK(set->nu0[0] >= 0 && set->nu0[0] + set->chunk_nnu <= F);
    if (set->nchunks == 2) {
        ARG_CHECK(set->chunk_nnu % 4 == 0);                                   // whole a_lm cells per chunk
        ARG_CHECK(set->nu0[1] >= set->nu0[0] + set->chunk_nnu && set->nu0[1] + set->chunk_nnu <= F);
        ch = draw_chan{set->nu0[0], set->nu0[1], set->chunk_nnu, 2 * set->chunk_nnu};
    } else {
        ch = draw_chan{set->nu0[0], 0, set->chunk_nnu, set->chunk_nnu};
    }
    return 0;
}

int corahip_draw_alm_philox(corahip_ctx *ctx, const double *T, const int32_t *info, uint64_t seed, int lmax, int F,
                            int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_philox(ctx, T, 0, info, seed, lmax, F, draw_chan{nu0, 0, nnu, nnu}, alm_dev);
}

int corahip_draw_alm_philox_rows(corahip_ctx *ctx, const double *T_rows, const int32_t *info, uint64_t seed, int lmax,
                                 int F, int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T_rows != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_philox(ctx, T_rows, 1, info, seed, lmax, F, draw_chan{nu0, 0, nnu, nnu}, alm_dev);
}

int corahip_draw_alm_philox_rows_set(corahip_ctx *ctx, const double *T_rows, const int32_t *info, uint64_t seed, int lmax,
                                     int F, const corahip_chanset *set, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T_rows != nullptr && alm_dev != nullptr && lmax >= 0 && F >= 1);
    draw_chan ch;
    int rc = chan_of(set, F, ch);
    if (rc) return rc;
    ARG_CHECK(!(chan_folded(ch) && (F & 1)));
    return draw_philox(ctx, T_rows, 1, info, seed, lmax, F, ch, alm_dev);
}

int corahip_draw_alm(corahip_ctx *ctx, const double *T, const int32_t *info, const double *g, int lmax, int F,
                     int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T != nullptr && g != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_host_stream(ctx, T, 0, info, g, lmax, F, draw_chan{nu0, 0, nnu, nnu}, alm_dev);
}

int corahip_draw_alm_rows(corahip_ctx *ctx, const double *T_rows, const int32_t *info, const double *g, int lmax, int F,
                          int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T_rows != nullptr && g != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_host_stream(ctx, T_rows, 1, info, g, lmax, F, draw_chan{nu0, 0, nnu, nnu}, alm_dev);
}

int corahip_alm_dev_to_square(corahip_ctx *ctx, const double *alm_dev, int lmax, int nnu, double *square) {
    ARG_CHECK(ctx != nullptr && alm_dev != nullptr && square != nullptr && lmax >= 0 && nnu >= 1);
    const long n = (long)nnu * (lmax + 1) * (lmax + 1);
    const int blocks = (int)std::min<long>((n + 255) / 256, 256L * 16);
    dev_to_square_kernel<<<blocks, 256, 0, ctx->stream>>>(alm_dev, lmax, nnu, (nnu + 3) / 4, square);
    LAUNCH_CHECK();
    return 0;
}

int corahip_alm_packed_to_dev(corahip_ctx *ctx, const double *packed, int lmax, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && alm_dev != nullptr && packed != nullptr && lmax >= 0 && nnu >= 1);
    const long nalm = nalm_of(lmax);
    const int G = (nnu + 3) / 4;
    const long n = nalm * G * 4;
    const int blocks = (int)std::min<long>((n + 255) / 256, 256L * 16);
    packed_to_dev_kernel<<<blocks, 256, 0, ctx->stream>>>(packed, nalm, nnu, G, alm_dev);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// (stream_internal.h) K3 of one l range of the numpy-stream pipeline (drawstream.hip)
int corahip_draw_range(corahip_ctx *ctx, hipStream_t stream, const double *T, int rows, const int32_t *info, const double *gslot,
                       size_t g_off, int l_lo, int l_hi, int lmax, int F, const corahip_chanset *set, double *alm_dev) {
    draw_chan ch;
    int rc = chan_of(set, F, ch);
    if (rc) return rc;
    return draw_stream_range(ctx, stream, T, rows, info, gslot, g_off, l_lo, l_hi, lmax, F, ch, alm_dev);
}
