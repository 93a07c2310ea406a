// sht.hip - HEALPix spherical-harmonic synthesis (alm -> RING maps) for many channels.
//
// Replaces hputil.sphtrans_inv_sky -> healpy.alm2map (cora/util/hputil.py:369-391,500-531).
// Definition implemented: SURVEY.md Appendix A (HEALPix software conventions).
//
// Two kernels per pass over a chunk of channels:
//   K4 legendre_kernel : F_m(ring) = sum_l a_lm lambda_lm(cos theta_ring) for every ring pair,
//        as FP64 MFMA (v_mfma_f64_16x16x4_f64): A = lambda (rows = rings, generated in
//        registers by the three-term recurrence), B = a_lm (LDS-staged rows of the
//        [nalm][cols] device layout), even/odd (l-m) accumulated separately so the
//        north ring gets e+o and its southern mirror e-o.
//   K5 ringfft_kernel  : per ring and channel: phase e^{i m phi0}, alias fold onto nphi
//        bins, complex-to-real FFT of length nphi (radix-2 in LDS; Bluestein for the
//        cap rings whose length 4i is not a power of two), pixel store.
// Intermediate F_m layout: inter[ring][g][m][c*4+v] (g = channel/4, v = channel%4,
// c = re/im): 64-byte cells, contiguous in m for K5, 64-byte segments for K4's stores.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

// ------------------------------------------------------------------------------------
struct corahip_sht_plan {
    int nside = 0, lmax = 0, L = 0, npair = 0, nring = 0;
    long npix = 0, nalm = 0;
    std::vector<int64_t> h_start;
    std::vector<int32_t> h_nphi;
    std::vector<double> h_z, h_sth, h_phi0;
    // device
    double *d_z = nullptr, *d_sth = nullptr;              // [npair] (north rings + equator)
    int32_t *d_nphi = nullptr;                            // [nring]
    int64_t *d_start = nullptr;                           // [nring]
    double *d_phi0 = nullptr;                             // [nring]
    double2 *d_coef = nullptr;                            // [nalm]: (A_l, B_l) at alm_idx(l,m)
    int32_t *d_lstart = nullptr;                          // [L][npair]
    double2 *d_seed = nullptr;                            // [L][npair]: (lambda_{lstart-1}, lambda_{lstart})
    int32_t *d_lmin = nullptr;                            // [L][ntile] first l per (m, ring tile)
    unsigned *d_queue = nullptr;                          // K4 work-queue head
    int32_t *d_mcut = nullptr;                            // [nring] number of m with any non-negligible lambda_lm
    double *d_polc = nullptr;                             // [nalm + 64][4] spin-2 coefficients (g1..g4), built on first use
    double *d_zeros = nullptr;                            // 4 KiB of zeros (source of padding rows for LDS-DMA)
    double2 *d_tw = nullptr;                              // e^{+2 pi i k/pmax}, k < pmax/2
    int pmax = 0, log_pmax = 0;
    // Bluestein tables, indexed by north-cap ring number i-1 (i = 1..nside-1)
    int32_t *d_blu_P = nullptr;                           // [nside]: 0 = power-of-two ring
    int64_t *d_blu_boff = nullptr, *d_blu_foff = nullptr; // offsets into chirp / filter arrays
    double2 *d_bchirp = nullptr, *d_bfilt = nullptr;
    int max_fft_len = 0;                                  // largest LDS FFT buffer (complex elems)
    // K5 launch classes: rings grouped by transform kind/length so each launch sizes its LDS
    struct ring_class {
        int P = 0;        // Bluestein length, 0 = direct power-of-two transform
        int nch = 4;      // channels transformed together per workgroup
        int threads = 0;  // workgroup size (0: K5_THREADS)
        int bstride = 0;  // complex elements per channel buffer in LDS
        int count = 0;
        int32_t *d_list = nullptr;
    };
    std::vector<ring_class> classes;
};

static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ------------------------------------------------------------------------------------
// plan-time kernels
// ------------------------------------------------------------------------------------
__device__ static inline void scaled_pow(double s, int n, double &mant, int &ex) {
    // s^n = mant * 2^ex with mant in [0.5, 1); exponentiation by squaring, renormalised
    int e;
    double f = frexp(s, &e);
    double rm = 1.0;
    int re = 0;
    double bm = f;
    int be = e;
    while (n) {
        if (n & 1) {
            rm *= bm;
            re += be;
            int t;
            rm = frexp(rm, &t);
            re += t;
        }
        bm *= bm;
        be *= 2;
        int t;
        bm = frexp(bm, &t);
        be += t;
        n >>= 1;
    }
    mant = rm;
    ex = re;
}

#define SEED_MIN_EXP (-900)

// lstart[m][r]: first l at which |lambda_lm(ring r)| >= 2^SEED_MIN_EXP, with the two
// recurrence values there; terms below are < 1e-270 and are dropped (libsharp does the same).
__global__ void seed_kernel(int lmax, int npair, const double *__restrict__ z, const double *__restrict__ sth,
                            const double *__restrict__ pref, const double2 *__restrict__ coef,
                            int32_t *__restrict__ lstart, double2 *__restrict__ seed) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    int m = blockIdx.y;
    if (r >= npair) return;
    double x = z[r];
    double pm;
    int pe;
    scaled_pow(sth[r], m, pm, pe);
    int t;
    double mant = frexp(pm * pref[m], &t);
    int sc = pe + t;
    if (m & 1) mant = -mant;
    long o = (long)m * npair + r;
    if (sc >= SEED_MIN_EXP) {
        lstart[o] = m;
        seed[o] = make_double2(0.0, ldexp(mant, sc));
        return;
    }
    const double2 *cf = coef + alm_idx(0, m, lmax);
    double p0 = 0.0, p1 = mant;  // scaled by 2^sc
    int found = lmax + 1;
    double s0 = 0.0, s1 = 0.0;
    for (int l = m + 1; l <= lmax; l++) {
        double2 c = cf[l];
        double v = fma(c.x * x, p1, -(c.y * p0));
        p0 = p1;
        p1 = v;
        if (fabs(p1) > 0x1p100) {
            p0 *= 0x1p-100;
            p1 *= 0x1p-100;
            sc += 100;
        }
        if (p1 != 0.0 && sc + ilogb(p1) >= SEED_MIN_EXP) {
            found = l;
            s0 = ldexp(p0, sc);
            s1 = ldexp(p1, sc);
            break;
        }
    }
    lstart[o] = found;
    seed[o] = make_double2(s0, s1);
}

// test hook: lambda_lm for one (m, ring pair), l = m..lmax
__global__ void lambda_kernel(int lmax, int npair, int m, int r, const double *__restrict__ z,
                              const double2 *__restrict__ coef, const int32_t *__restrict__ lstart,
                              const double2 *__restrict__ seed, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double x = z[r];
    long o = (long)m * npair + r;
    int ls = lstart[o];
    double2 sd = seed[o];
    const double2 *cf = coef + alm_idx(0, m, lmax);
    double p0 = 0.0, p1 = 0.0;
    for (int l = m; l <= lmax; l++) {
        double2 c = cf[l];
        double v = fma(c.x * x, p1, -(c.y * p0));
        bool inj = (l == ls);
        v = inj ? sd.y : v;
        p0 = inj ? sd.x : p1;
        p1 = v;
        out[l - m] = v;
    }
}

// ------------------------------------------------------------------------------------
// K4: Legendre contraction on FP64 MFMA
// ------------------------------------------------------------------------------------
#ifndef LEG_ABLATE
#define LEG_ABLATE 0  // diagnostic builds only (make ablate): 1 no MFMA, 2 no recurrence, 3 no B reads, 4 no epilogue stores
#endif
#ifndef LEG_KT
#define LEG_KT 48      // l rows per LDS stage (32 rows x 4 buffers: 75.6 ms; 48 x 3: 74.1 ms; 48 x 2 and 56 x 2: 74.2 ms; 64 needs > 64 coefficient lanes)
#endif
#ifndef LEG_WAVES
#define LEG_WAVES 8
#endif
#define LEG_RINGS (16 * LEG_WAVES)  // ring pairs per workgroup (x RT)
#define LMIN_RINGS 128              // granularity of the plan's per-(m, ring block) first-l table
#define ADJ_WAVES 8                 // waves per workgroup of the analysis kernel
#ifndef LEG_NBUF
#define LEG_NBUF 3     // LDS stage ring: one being read + two in flight
#endif

// LDS-DMA issued from inline asm: hipcc does not count it, so it does not drain the DMA with a
// vmcnt(0) in front of every later ds_read (which it does for the builtin: the DMA is a pending LDS
// write it cannot disambiguate).  The kernel waits itself: s_waitcnt vmcnt(0) before the stage barrier.
// lds_byte_addr must be wave-uniform; lane i's 16 bytes land at lds_byte_addr + 16 i.
__device__ static inline void glds16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// Lane roles (wave = 16 rings x 4 k-slots, the A operand of v_mfma_f64_16x16x4_f64):
// lane (ri = lane&15, kq = lane>>4) runs the recurrence of ring ri STAGGERED by 2 kq steps, so that
// at every macro-step (8 consecutive l, base l0) the first two values it produces are exactly the
// ones its k-slot must feed: lambda at l0+2kq (even l-m -> north+south accumulator) and l0+2kq+1
// (odd).  No cross-lane movement, no selects; the price is that the recurrence coefficients are
// no longer wave-uniform (4 distinct rows per step) - they are staged through LDS with the a_lm
// rows and read with one broadcast ds_read_b128 per step.
// NT = 16-column tiles per wave, RT = 16-ring row tiles per wave (RT x NT x 2 parities = 16 accumulator
// tiles = 128 VGPRs in both shipped shapes: <8,1> for >= 128 columns, <4,2> for 64-column shards, where a
// second, independent recurrence per lane keeps the recurrence : MFMA ratio of the wide shape).
template <int NT, int RT>
__global__ void __launch_bounds__(64 * LEG_WAVES, LEG_WAVES <= 4 ? 2 : 1)
legendre_kernel(int lmax, int npair, int nring, int ncols, const double *__restrict__ z,
                const double2 *__restrict__ coef, const int32_t *__restrict__ lstart,
                const double2 *__restrict__ seed, const int32_t *__restrict__ lmin_tab,
                const double *__restrict__ alm, const double *__restrict__ zeros, double *__restrict__ inter,
                unsigned *__restrict__ queue) {
    constexpr int TCOLS = 16 * NT;          // columns of this block
    constexpr int STRIDE = TCOLS + 8;       // LDS row stride (doubles): 2 rows apart = 128 B mod 256
    constexpr int CROWS = LEG_KT + 8;       // coefficient rows per stage (staggered lanes look 6 ahead)
    constexpr int STAGE = LEG_KT * STRIDE + 2 * CROWS;  // doubles per stage: a_lm rows + (A,B) pairs
    constexpr int RPW = LEG_KT / LEG_WAVES;             // a_lm rows each wave moves per stage
    constexpr int PIECES = RPW + 1;                     // LDS-DMA pieces per wave per stage (+ coefficients)
    constexpr int TRINGS = LEG_RINGS * RT;              // ring pairs per workgroup
    constexpr int RPM = RPW / (LEG_KT / 8);             // a_lm pieces each wave issues per macro-step
    static_assert(RPM * (LEG_KT / 8) == RPW && RPM >= 1, "whole a_lm pieces per macro-step");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int &s_next = *reinterpret_cast<int *>(lds + LEG_NBUF * STAGE);  // next work item (carved after the ring)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ri = lane & 15, kq = lane >> 4;
    const int d = 2 * kq;
    const int L = lmax + 1;
    const int G = ncols >> 3;
    const int ntile128 = (npair + LMIN_RINGS - 1) / LMIN_RINGS;   // granularity of lmin_tab
    const int ntile = (npair + TRINGS - 1) / TRINGS;
    const int ncg = ncols / TCOLS;
    const int nitems = L * ncg * ntile;
    const long last_row = nalm_of(lmax) - 1;
    const unsigned lds_base_bytes = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;
    const bool odd_lane = lane & 1;

    // Persistent workgroups.  Work item = (m, column group, ring tile); consecutive items are the ring
    // tiles of one a_lm slice, so the workgroups running at the same time share slices in L2 (they are
    // dealt over all 8 XCDs; packing a slice group onto ONE XCD was measured 24 % slower: every resident
    // workgroup of the XCD then hits the same 1-2 L2 channels in lock step).  Small m (long K) first.
    struct item_t {
        int m, cg, rtile, l_begin, nstage;
        long base_m;
    };
    auto decode = [&](int it) {
        item_t w;
        const int gidx = it / ntile;
        w.rtile = it - gidx * ntile;
        w.m = gidx / ncg;
        w.cg = gidx - w.m * ncg;
        int lmin = lmax + 1;
        const int t_first = (w.rtile * TRINGS) / LMIN_RINGS;
        const int t_last = min((w.rtile * TRINGS + TRINGS - 1) / LMIN_RINGS, ntile128 - 1);
        for (int t128 = t_first; t128 <= t_last; t128++) lmin = min(lmin, lmin_tab[w.m * ntile128 + t128]);
        w.l_begin = w.m + ((lmin - w.m) & ~7);
        w.nstage = lmin <= lmax ? (lmax - w.l_begin) / LEG_KT + 1 : 0;
        w.base_m = alm_idx(0, w.m, lmax);
        return w;
    };
    // LDS-DMA pieces: every wave issues exactly PIECES per stage (counted vmcnt): RPW a_lm rows (rows past
    // lmax are never used - their lambda is 0 - so any valid row is read, keeping the address scalar) and
    // the CROWS coefficient pairs (all waves write the same bytes).
    auto issue_row = [&](const item_t &w, int st, int rr) {
        const int row = wv + LEG_WAVES * rr;
        long rowidx = w.base_m + w.l_begin + st * LEG_KT + row;
        rowidx = rowidx < last_row ? rowidx : last_row;
        const double *src = alm + (size_t)w.cg * TCOLS + (size_t)rowidx * ncols + 2 * lane;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % LEG_NBUF) * STAGE + row * STRIDE) * sizeof(double));
        if (lane < 8 * NT) glds16(src, dst);
    };
    auto issue_coef = [&](const item_t &w, int st) {
        const int l = w.l_begin + st * LEG_KT + lane;
        const double *src = (l <= lmax) ? reinterpret_cast<const double *>(coef + w.base_m + l) : zeros;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % LEG_NBUF) * STAGE + LEG_KT * STRIDE) * sizeof(double));
        if (lane < CROWS) glds16(src, dst);
    };
    auto issue_stage = [&](const item_t &w, int st) {
        issue_coef(w, st);
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) issue_row(w, st, rr);
    };

    // dynamic work queue (one atomic per item, fetched one item ahead): items differ a lot in length
    // (polar ring tiles start late, large m is short), a static assignment left ~10 % on the table
    int item = blockIdx.x;  // the first gridDim.x items are pre-assigned; the queue starts behind them
    if (item >= nitems) return;
    item_t w = decode(item);
#pragma unroll
    for (int st = 0; st < LEG_NBUF - 1; st++)
        if (st < w.nstage) issue_stage(w, st);

    for (;;) {
        const int m = w.m;
        if (tid == 0) s_next = (int)(gridDim.x + atomicAdd(queue, 1u));  // latency hidden behind this item
        d4_t acce[RT][NT], acco[RT][NT];
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acce[q][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
                acco[q][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            }
        if (w.nstage > 0) {
            // rings are dealt to the waves interleaved (ring = tile base + 8 (ri + 16 q) + wave) so that every
            // wave of the workgroup has the same mix of first-contributing l and reaches the barriers together
            double x[RT], p0[RT], p1[RT];
            double2 sd[RT];
            int my_ls[RT], inj_l[RT];
            const double2 *cf = coef + w.base_m;
            int ls_min = lmax + 1;
#pragma unroll
            for (int q = 0; q < RT; q++) {
                const int ring = w.rtile * TRINGS + (ri + 16 * q) * LEG_WAVES + wave;
                x[q] = 0.0;
                my_ls[q] = lmax + 1;
                sd[q] = make_double2(0.0, 0.0);
                if (ring < npair) {
                    x[q] = z[ring];
                    const long o = (long)m * npair + ring;
                    my_ls[q] = lstart[o];
                    sd[q] = seed[o];
                }
                ls_min = min(ls_min, my_ls[q]);
                // per-lane start state: (lambda_{lf-2}, lambda_{lf-1}) with lf = l_begin + d the first l of this
                // lane.  If the ring's first contributing l lies before lf, advance from the seeds.
                p0[q] = 0.0;
                p1[q] = 0.0;
                inj_l[q] = my_ls[q];  // l at which the seeds are injected
                const int lf = w.l_begin + d;
                if (my_ls[q] < lf) {
                    p0[q] = sd[q].x;
                    p1[q] = sd[q].y;
                    for (int l = my_ls[q] + 1; l < lf; l++) {
                        const double2 c = (l <= lmax) ? cf[l] : make_double2(0.0, 0.0);
                        const double vv = fma(c.x * x[q], p1[q], -(c.y * p0[q]));
                        p0[q] = p1[q];
                        p1[q] = vv;
                    }
                    inj_l[q] = 0x7fffffff;
                }
            }

            for (int st = 0; st < w.nstage; st++) {
                // own pieces of stage st have landed when at most the pieces of the (up to LEG_NBUF-2) younger
                // stages are still in flight (anything younger than those only makes the wait stricter)
                if (LEG_NBUF >= 4 && st + 2 < w.nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
                else if (LEG_NBUF >= 3 && st + 1 < w.nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // everyone's pieces of stage st landed; everyone is done reading stage st-1
                const bool refill = st + LEG_NBUF - 1 < w.nstage;
                if (refill) issue_coef(w, st + LEG_NBUF - 1);
                const int ls = w.l_begin + st * LEG_KT;
                const double *sb = lds + (st % LEG_NBUF) * STAGE;
                const double2 *sc = reinterpret_cast<const double2 *>(sb + LEG_KT * STRIDE) + d;
#pragma unroll 1
                for (int ms = 0; ms < LEG_KT / 8; ms++) {
                    // one a_lm piece of the stage being refilled per macro-step: spreads the LDS-DMA issue
                    // over the MFMA work instead of an 8-wave burst behind the barrier (measured 7 % of time)
                    if (refill) {
#pragma unroll
                        for (int r = 0; r < RPM; r++) issue_row(w, st + LEG_NBUF - 1, ms * RPM + r);
                    }
                    const int l0 = ls + 8 * ms;
                    if (l0 > lmax) continue;
                    // nothing of this wave starts before l0+14: skip the macro-step entirely
                    if (__all(ls_min > l0 + 13)) continue;
                    double ae[RT], ao[RT];
#if LEG_ABLATE == 2  // diagnostic: no recurrence
#pragma unroll
                    for (int q = 0; q < RT; q++) {
                        ae[q] = x[q];
                        ao[q] = x[q] + 1.0;
                        asm volatile("" : "+v"(ae[q]), "+v"(ao[q]));
                    }
#else
                    __builtin_amdgcn_s_setprio(2);  // the short recurrence outranks the partner wave's MFMAs
                    double2 c[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) c[j] = sc[8 * ms + j];
                    const int lf = l0 + d;
                    bool any_inj = false;
#pragma unroll
                    for (int q = 0; q < RT; q++) any_inj |= (inj_l[q] >= lf && inj_l[q] < lf + 8);
                    if (__any(any_inj)) {
#pragma unroll
                        for (int q = 0; q < RT; q++) {
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                double vv = fma(c[j].x * x[q], p1[q], -(c[j].y * p0[q]));
                                const bool inj = (lf + j == inj_l[q]);
                                vv = inj ? sd[q].y : vv;
                                p0[q] = inj ? sd[q].x : p1[q];
                                p1[q] = vv;
                                if (j == 0) ae[q] = vv;
                                if (j == 1) ao[q] = vv;
                            }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
#pragma unroll
                            for (int q = 0; q < RT; q++) {
                                const double vv = fma(c[j].x * x[q], p1[q], -(c[j].y * p0[q]));
                                p0[q] = p1[q];
                                p1[q] = vv;
                                if (j == 0) ae[q] = vv;
                                if (j == 1) ao[q] = vv;
                            }
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
#endif
                    if (__all(ls_min > l0 + 7)) continue;  // all A operands of this macro-step are zero
                    const double *be = sb + (8 * ms + d) * STRIDE + ri;
                    const double *bo = be + STRIDE;
#if LEG_ABLATE == 1  // diagnostic: no MFMA (keep the operands alive)
                    asm volatile("" ::"v"(ae[0]), "v"(ao[0]), "v"(be), "v"(bo));
#else
#pragma unroll
                    for (int t = 0; t < NT; t++) {
#if LEG_ABLATE == 3  // diagnostic: no B operand reads from LDS
                        const double bev = ae[0] + t, bov = ao[0] - t;
                        asm volatile("" ::"v"(be), "v"(bo));
#else
                        const double bev = be[16 * t], bov = bo[16 * t];
#endif
#pragma unroll
                        for (int q = 0; q < RT; q++) {
                            acce[q][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[q], bev, acce[q][t], 0, 0, 0);
                            acco[q][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[q], bov, acco[q][t], 0, 0, 0);
                        }
                    }
#endif
                }
            }
        }

        // ---- next item: start its first stages now, so they land behind this item's epilogue stores
        const int cur_rtile = w.rtile, cur_cg = w.cg, cur_m = w.m, cur_nstage = w.nstage;
        __syncthreads();  // all waves are done reading the stage ring; s_next is visible
        item = __builtin_amdgcn_readfirstlane(s_next);
        const bool have_next = item < nitems;
        if (have_next) {
            w = decode(item);
#pragma unroll
            for (int st = 0; st < LEG_NBUF - 1; st++)
                if (st < w.nstage) issue_stage(w, st);
        }

        // ---- epilogue: north = even + odd, south mirror = even - odd.  Adjacent lanes (columns n, n+1 of the
        //      same rows) swap one value each so that every lane stores 16 bytes: half the store instructions.
        //      A tile with no contributing l at all is not written: K5 never reads cells with m >= mcut(ring).
        if (cur_nstage > 0) {
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int col = cur_cg * TCOLS + 16 * t + (ri & ~1);  // even column of the lane pair
                    const int g = col >> 3, cv = col & 7;
#pragma unroll
                    for (int rp = 0; rp < 2; rp++) {
                        const int r0 = 2 * rp, r1 = 2 * rp + 1;
                        const double n0 = acce[q][t][r0] + acco[q][t][r0], n1 = acce[q][t][r1] + acco[q][t][r1];
                        const double s0 = acce[q][t][r0] - acco[q][t][r0], s1 = acce[q][t][r1] - acco[q][t][r1];
                        // even lane keeps row r0 and sends its r1 value; odd lane keeps row r1 and sends its r0 value
                        const double nrecv = __shfl_xor(odd_lane ? n0 : n1, 1);
                        const double srecv = __shfl_xor(odd_lane ? s0 : s1, 1);
                        const int rr = odd_lane ? r1 : r0;
                        const int ro = cur_rtile * TRINGS + (kq + 4 * rr + 16 * q) * LEG_WAVES + wave;
#if LEG_ABLATE == 4  // diagnostic: no epilogue stores (unless a value is absurd: keeps the arithmetic alive)
                        if (ro < npair && n0 == 1.2345e300) {
#else
                        if (ro < npair) {
#endif
                            const double2 nv = odd_lane ? make_double2(nrecv, n1) : make_double2(n0, nrecv);
                            *reinterpret_cast<double2 *>(inter + (((size_t)ro * G + g) * L + cur_m) * 8 + cv) = nv;
                            const int rs = nring - 1 - ro;
                            if (rs != ro) {
                                const double2 sv = odd_lane ? make_double2(srecv, s1) : make_double2(s0, srecv);
                                *reinterpret_cast<double2 *>(inter + (((size_t)rs * G + g) * L + cur_m) * 8 + cv) = sv;
                            }
                        }
                    }
                }
        }
        if (!have_next) break;
        __syncthreads();  // everyone has read s_next before thread 0 overwrites it
    }
}

// K4 for polarisation (spin 2): (E, B) -> (Q, U), what healpy.alm2map([T, E, B]) does for Q and U behind
// hputil.sphtrans_inv_real_pol (cora/util/hputil.py:394-432).  Same structure as legendre_kernel (persistent
// workgroups + queue, staggered-lane recurrence, LDS-DMA stage ring, even/odd parity accumulators), with the two
// A operands per parity derived from the scalar recurrence:
//     W_lm = (g1 r1 + g2) lambda_l + g3 r2 lambda_{l-1},     X_lm = g4 r2 lambda_l - m g3 r1 lambda_{l-1},
//     r1 = 1/sin^2, r2 = cos/sin^2 per ring (lane),  (g1..g4)(l, m) from the plan's table (staged through LDS):
//     g1 = -N2 (l - m^2), g2 = -N2 l(l-1)/2, g3 = N2 (2l+1)/A_l, g4 = N2 m (l-1), N2 = 2/sqrt((l+2)(l+1)l(l-1)).
// The kernel accumulates S_W = sum_l W a and S_X = sum_l X a for the natural columns; channels are interleaved
// (E_f, B_f) so that one a_lm cell [re x4 | im x4] holds (Re E, Re B, .., Im E, Im B, ..) and
//     Re Q = -(S_W[ReE] + S_X[ImB]),  Im Q = -(S_W[ImE] - S_X[ReB]),  Re U = -(S_W[ReB] - S_X[ImE]),  Im U = -(S_W[ImB] + S_X[ReE])
// is a lane-xor-5 exchange in the epilogue; Q_f, U_f leave in the cell positions of E_f, B_f.  W has the parity
// (-1)^{l+m} of lambda under theta -> pi - theta, X the opposite: north = (W_e + W_o, X_e + X_o), south = (W_e - W_o, X_o - X_e).
template <int NT>
__global__ void __launch_bounds__(64 * LEG_WAVES)
legendre_pol_kernel(int lmax, int npair, int nring, int ncols, const double *__restrict__ z,
                    const double *__restrict__ sth, const double2 *__restrict__ coef,
                    const double *__restrict__ polc, const int32_t *__restrict__ lstart,
                    const double2 *__restrict__ seed, const int32_t *__restrict__ lmin_tab,
                    const double *__restrict__ alm, const double *__restrict__ zeros, double *__restrict__ inter,
                    unsigned *__restrict__ queue) {
    constexpr int KT = 32;                  // l rows per stage (the 4 accumulator sets leave room for NT = 4 only)
    constexpr int NBUF = 3;
    constexpr int TCOLS = 16 * NT;
    constexpr int STRIDE = TCOLS + 8;
    constexpr int CROWS = KT + 8;
    constexpr int STAGE = KT * STRIDE + 2 * CROWS + 4 * CROWS;   // a_lm rows + (A, B) pairs + (g1..g4) rows
    constexpr int RPW = KT / LEG_WAVES;
    constexpr int PIECES = RPW + 3;         // a_lm rows + the (A, B) piece + two pieces of the g table (see issue_coef)
    constexpr int TRINGS = LEG_RINGS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int &s_next = *reinterpret_cast<int *>(lds + NBUF * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ri = lane & 15, kq = lane >> 4;
    const int d = 2 * kq;
    const int L = lmax + 1;
    const int G = ncols >> 3;
    const int ntile128 = (npair + LMIN_RINGS - 1) / LMIN_RINGS;
    const int ntile = (npair + TRINGS - 1) / TRINGS;
    const int ncg = ncols / TCOLS;
    const int nitems = L * ncg * ntile;
    const long last_row = nalm_of(lmax) - 1;
    const unsigned lds_base_bytes = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;

    struct item_t {
        int m, cg, rtile, l_begin, nstage;
        long base_m;
    };
    auto decode = [&](int it) {
        item_t w;
        const int gidx = it / ntile;
        w.rtile = it - gidx * ntile;
        w.m = gidx / ncg;
        w.cg = gidx - w.m * ncg;
        int lmin = lmax + 1;
        const int t_first = (w.rtile * TRINGS) / LMIN_RINGS;
        const int t_last = min((w.rtile * TRINGS + TRINGS - 1) / LMIN_RINGS, ntile128 - 1);
        for (int t128 = t_first; t128 <= t_last; t128++) lmin = min(lmin, lmin_tab[w.m * ntile128 + t128]);
        w.l_begin = w.m + ((lmin - w.m) & ~7);
        w.nstage = lmin <= lmax ? (lmax - w.l_begin) / KT + 1 : 0;
        w.base_m = alm_idx(0, w.m, lmax);
        return w;
    };
    auto issue_row = [&](const item_t &w, int st, int rr) {
        const int row = wv + LEG_WAVES * rr;
        long rowidx = w.base_m + w.l_begin + st * KT + row;
        rowidx = rowidx < last_row ? rowidx : last_row;
        const double *src = alm + (size_t)w.cg * TCOLS + (size_t)rowidx * ncols + 2 * lane;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % NBUF) * STAGE + row * STRIDE) * sizeof(double));
        if (lane < 8 * NT) glds16(src, dst);
    };
    // coefficient pieces of a stage: CROWS (A, B) pairs (16 B each) and CROWS (g1..g4) rows (32 B each = 2 CROWS
    // 16-byte chunks, contiguous in the table): 1 + 2 wave-instructions (CROWS = 40: 40 + 80 lanes)
    auto issue_coef = [&](const item_t &w, int st) {
        const int l0 = w.l_begin + st * KT;
        const int l = l0 + lane;
        const double *src = (l <= lmax) ? reinterpret_cast<const double *>(coef + w.base_m + l) : zeros;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % NBUF) * STAGE + KT * STRIDE) * sizeof(double));
        if (lane < CROWS) glds16(src, dst);
        const double *gsrc = polc + 4 * (size_t)(w.base_m + l0);     // the table is padded: rows past the end exist
        const unsigned gdst = dst + (unsigned)(2 * CROWS * sizeof(double));
        glds16(gsrc + 2 * lane, gdst);
        if (lane < 2 * CROWS - 64) glds16(gsrc + 2 * (64 + lane), gdst + 64 * 16);
    };
    auto issue_stage = [&](const item_t &w, int st) {
        issue_coef(w, st);
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) issue_row(w, st, rr);
    };

    int item = blockIdx.x;
    if (item >= nitems) return;
    item_t w = decode(item);
#pragma unroll
    for (int st = 0; st < NBUF - 1; st++)
        if (st < w.nstage) issue_stage(w, st);

    for (;;) {
        const int m = w.m;
        const double mval = (double)m;
        if (tid == 0) s_next = (int)(gridDim.x + atomicAdd(queue, 1u));
        d4_t awe[NT], awo[NT], axe[NT], axo[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            awe[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            awo[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            axe[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            axo[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
        }
        if (w.nstage > 0) {
            const int ring = w.rtile * TRINGS + ri * LEG_WAVES + wave;
            double x = 0.0, r1 = 0.0, r2 = 0.0, p0 = 0.0, p1 = 0.0;
            double2 sd = make_double2(0.0, 0.0);
            int my_ls = lmax + 1;
            if (ring < npair) {
                x = z[ring];
                const double s = sth[ring];
                r1 = 1.0 / (s * s);
                r2 = x * r1;
                const long o = (long)m * npair + ring;
                my_ls = lstart[o];
                sd = seed[o];
            }
            const int ls_min = my_ls;
            int inj_l = my_ls;
            const double2 *cf = coef + w.base_m;
            {
                const int lf = w.l_begin + d;
                if (my_ls < lf) {
                    p0 = sd.x;
                    p1 = sd.y;
                    for (int l = my_ls + 1; l < lf; l++) {
                        const double2 c = (l <= lmax) ? cf[l] : make_double2(0.0, 0.0);
                        const double vv = fma(c.x * x, p1, -(c.y * p0));
                        p0 = p1;
                        p1 = vv;
                    }
                    inj_l = 0x7fffffff;
                }
            }
            for (int st = 0; st < w.nstage; st++) {
                if (st + 1 < w.nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const bool refill = st + NBUF - 1 < w.nstage;
                if (refill) issue_stage(w, st + NBUF - 1);
                const int ls = w.l_begin + st * KT;
                const double *sb = lds + (st % NBUF) * STAGE;
                const double2 *sc = reinterpret_cast<const double2 *>(sb + KT * STRIDE) + d;
                const double *sg = sb + KT * STRIDE + 2 * CROWS + 4 * d;
#pragma unroll 1
                for (int ms = 0; ms < KT / 8; ms++) {
                    const int l0 = ls + 8 * ms;
                    if (l0 > lmax) continue;
                    if (__all(ls_min > l0 + 13)) continue;
                    double2 c[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) c[j] = sc[8 * ms + j];
                    const int lf = l0 + d;
                    // lambda at lf-1, lf (even l-m slot of this lane), lf+1 (odd slot); then 6 more steps
                    double lm1 = p1, le = 0.0, lo = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        double vv = fma(c[j].x * x, p1, -(c[j].y * p0));
                        const bool inj = (lf + j == inj_l);
                        vv = inj ? sd.y : vv;
                        p0 = inj ? sd.x : p1;
                        p1 = vv;
                        if (j == 0) {
                            le = vv;
                            lm1 = p0;      // lambda_{lf-1} as the recurrence sees it (the seed if injected here)
                        }
                        if (j == 1) lo = vv;
                    }
                    double lem1 = lm1, lom1 = le;
                    if (lf + 1 == inj_l) lom1 = sd.x;
                    if (__all(ls_min > l0 + 7)) continue;
                    // W, X at the two l of this lane
                    const double4 ge = *reinterpret_cast<const double4 *>(sg + 4 * (8 * ms));
                    const double4 go = *reinterpret_cast<const double4 *>(sg + 4 * (8 * ms + 1));
                    const double We = fma(fma(ge.x, r1, ge.y), le, (ge.z * r2) * lem1);
                    const double Xe = fma(ge.w * r2, le, -((mval * ge.z) * r1) * lem1);
                    const double Wo = fma(fma(go.x, r1, go.y), lo, (go.z * r2) * lom1);
                    const double Xo = fma(go.w * r2, lo, -((mval * go.z) * r1) * lom1);
                    const double *be = sb + (8 * ms + d) * STRIDE + ri;
                    const double *bo = be + STRIDE;
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const double bev = be[16 * t], bov = bo[16 * t];
                        awe[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(We, bev, awe[t], 0, 0, 0);
                        awo[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wo, bov, awo[t], 0, 0, 0);
                        axe[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xe, bev, axe[t], 0, 0, 0);
                        axo[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xo, bov, axo[t], 0, 0, 0);
                    }
                }
            }
        }

        const int cur_rtile = w.rtile, cur_cg = w.cg, cur_m = w.m, cur_nstage = w.nstage;
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_next);
        const bool have_next = item < nitems;
        if (have_next) {
            w = decode(item);
#pragma unroll
            for (int st = 0; st < NBUF - 1; st++)
                if (st < w.nstage) issue_stage(w, st);
        }
        if (cur_nstage > 0) {
            // sigma of the column inside its 8-wide cell: +1 for (Re B, Im E) columns, -1 for (Re E, Im B)
            const int c8 = ri & 7;
            const double sigma = (((c8 & 1) ^ ((c8 >> 2) & 1)) != 0) ? 1.0 : -1.0;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int col = cur_cg * TCOLS + 16 * t + ri;
                const int g = col >> 3, cv = col & 7;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double swn = awe[t][r] + awo[t][r], sws = awe[t][r] - awo[t][r];
                    const double sxn = axe[t][r] + axo[t][r], sxs = axo[t][r] - axe[t][r];
                    const double pxn = __shfl_xor(sxn, 5), pxs = __shfl_xor(sxs, 5);
                    const int ro = cur_rtile * TRINGS + (kq + 4 * r) * LEG_WAVES + wave;
                    if (ro < npair) {
                        inter[(((size_t)ro * G + g) * L + cur_m) * 8 + cv] = -(swn - sigma * pxn);
                        const int rs = nring - 1 - ro;
                        if (rs != ro) inter[(((size_t)rs * G + g) * L + cur_m) * 8 + cv] = -(sws - sigma * pxs);
                    }
                }
            }
        }
        if (!have_next) break;
        __syncthreads();
    }
}

// per ring: mcut = number of m (from 0) whose lambda_lm reach 2^-900 for some l <= lmax; F_m of the ring
// is exactly zero beyond (lstart is monotone in m), so K4 need not write and K5 need not read those cells
__global__ void mcut_kernel(int lmax, int npair, int nring, const int32_t *__restrict__ lstart,
                            int32_t *__restrict__ mcut) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npair) return;
    int c = 0;
    for (int m = 0; m <= lmax; m++)
        if (lstart[(long)m * npair + r] <= lmax) c = m + 1;
    mcut[r] = c;
    mcut[nring - 1 - r] = c;
}

// per (m, ring tile) minimum of lstart: the first l the tile's workgroup has to visit
__global__ void lmin_kernel(int lmax, int npair, int ntile, const int32_t *__restrict__ lstart,
                            int32_t *__restrict__ lmin_tab) {
    const int m = blockIdx.x, t = threadIdx.x;
    if (t >= ntile) return;
    int v = lmax + 1;
    for (int r = t * LMIN_RINGS; r < min((t + 1) * LMIN_RINGS, npair); r++) v = min(v, lstart[(long)m * npair + r]);
    lmin_tab[m * ntile + t] = v;
}

// ------------------------------------------------------------------------------------
// K5: per-ring phase / fold / FFT
// ------------------------------------------------------------------------------------
__device__ static inline double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ static inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ static inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by SIGN * i
template <int SIGN>
__device__ static inline double2 cmuli(double2 a) {
    return SIGN > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

// small DFTs in registers, kernel e^{SIGN 2 pi i r p / R}, natural order in and out
template <int SIGN>
__device__ static inline void dft2(double2 &a, double2 &b) {
    const double2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}
template <int SIGN>
__device__ static inline void dft4(double2 &x0, double2 &x1, double2 &x2, double2 &x3) {
    const double2 s02 = cadd(x0, x2), d02 = csub(x0, x2);
    const double2 s13 = cadd(x1, x3), d13 = cmuli<SIGN>(csub(x1, x3));
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = cadd(d02, d13);
    x3 = csub(d02, d13);
}
template <int R, int SIGN>
struct DftR;
template <int SIGN>
struct DftR<2, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[2]) { dft2<SIGN>(x[0], x[1]); }
};
template <int SIGN>
struct DftR<4, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[4]) { dft4<SIGN>(x[0], x[1], x[2], x[3]); }
};
template <int SIGN>
struct DftR<8, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[8]) {
        // even / odd halves, then radix-2 combine with eighth roots
        dft4<SIGN>(x[0], x[2], x[4], x[6]);
        dft4<SIGN>(x[1], x[3], x[5], x[7]);
        const double h = 0.70710678118654752440;
        const double2 w1 = make_double2(h, SIGN * h), w3 = make_double2(-h, SIGN * h);
        const double2 o0 = x[1], o1 = cmul(x[3], w1), o2 = cmuli<SIGN>(x[5]), o3 = cmul(x[7], w3);
        const double2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
        x[0] = cadd(e0, o0);
        x[4] = csub(e0, o0);
        x[1] = cadd(e1, o1);
        x[5] = csub(e1, o1);
        x[2] = cadd(e2, o2);
        x[6] = csub(e2, o2);
        x[3] = cadd(e3, o3);
        x[7] = csub(e3, o3);
    }
};
template <int SIGN>
struct DftR<16, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[16]) {
        // n = 4a + c, k = k1 + 4 k2: DFT4 over a, twiddle w16^{c k1}, DFT4 over c
        const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;  // cos, sin(pi/8)
        const double h = 0.70710678118654752440;
#pragma unroll
        for (int c = 0; c < 4; c++) dft4<SIGN>(x[c], x[4 + c], x[8 + c], x[12 + c]);
        // after this x[4 k1 + c] holds t_c[k1]
        // twiddles w16^{c k1}, c,k1 in 1..3: exponents 1,2,3,2,4,6,3,6,9
        const double2 w1 = make_double2(c1, SIGN * s1), w2 = make_double2(h, SIGN * h), w3 = make_double2(s1, SIGN * c1);
        const double2 w6 = make_double2(-h, SIGN * h), w9 = make_double2(-c1, -SIGN * s1);
        x[4 + 1] = cmul(x[4 + 1], w1);
        x[4 + 2] = cmul(x[4 + 2], w2);
        x[4 + 3] = cmul(x[4 + 3], w3);
        x[8 + 1] = cmul(x[8 + 1], w2);
        x[8 + 2] = cmuli<SIGN>(x[8 + 2]);
        x[8 + 3] = cmul(x[8 + 3], w6);
        x[12 + 1] = cmul(x[12 + 1], w3);
        x[12 + 2] = cmul(x[12 + 2], w6);
        x[12 + 3] = cmul(x[12 + 3], w9);
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) dft4<SIGN>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
        // now x[4 k1 + k2] = X[k1 + 4 k2]: transpose to natural order
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
            for (int k2 = k1 + 1; k2 < 4; k2++) {
                const double2 t = x[4 * k1 + k2];
                x[4 * k1 + k2] = x[4 * k2 + k1];
                x[4 * k2 + k1] = t;
            }
    }
};

// LDS index padding of the FFT buffers: one spare slot per 8 elements plus 8 per 128.  Makes the
// unit-stride last pass (lane t owns elements R t .. R t + R-1) and the stride-8/16 middle pass at
// most 2-way bank conflicted for ds_read/write_b128 instead of 8..16-way.
__host__ __device__ static inline int fpad(int i) { return i + (i >> 3) + ((i >> 7) << 3); }

// e^{+2 pi i idx/pmax} from the half-circle table in HBM, tw[k] = e^{+2 pi i k/pmax}, k < pmax/2
__device__ static inline double2 tw_global(const double2 *__restrict__ tw, int pmax, int idx) {
    const int hp = pmax >> 1;
    double2 w = tw[idx >= hp ? idx - hp : idx];
    if (idx >= hp) w = make_double2(-w.x, -w.y);
    return w;
}
// The kernels look twiddles up in a two-level LDS table instead: tl[lo] = e^{2 pi i lo/pmax}, lo < 64, and
// tl[64 + hi] = e^{2 pi i 64 hi/pmax}; e^{2 pi i idx/pmax} = tl[idx & 63] * tl[64 + (idx >> 6)].  A twiddle
// fetched from HBM inside an FFT pass made every pass wait (vmcnt is in-order) for the register prefetch of
// the NEXT ring's cells issued just before it - the whole HBM latency was exposed once per ring.
// Measured at cfg 3: the extra LDS reads + complex multiply and the higher register pressure cost more
// (belt class 13.2 -> 15.8 ms) than the exposed latency they remove, so the switch is OFF; kept for the record.
#ifndef K5_LDS_TW
#define K5_LDS_TW 0
#endif
#if K5_LDS_TW
#define TWL_ENTRIES(pmax) (64 + ((pmax) >= 64 ? (pmax) / 64 : 1))
#else
#define TWL_ENTRIES(pmax) 0
#endif
__device__ static inline void twl_fill(double2 *tl, const double2 *__restrict__ tw, int pmax) {
    const int nhi = pmax >= 64 ? pmax / 64 : 1;
    for (int i = threadIdx.x; i < 64 + nhi; i += blockDim.x) {
        const int idx = i < 64 ? (i < pmax ? i : 0) : 64 * (i - 64);
        tl[i] = tw_global(tw, pmax, idx);
    }
    __syncthreads();
}
template <int SIGN>
__device__ static inline double2 tw_get(const double2 *tl, int pmax, int idx) {
#if K5_LDS_TW
    double2 w = cmul(tl[idx & 63], tl[64 + (idx >> 6)]);
#else
    double2 w = tw_global(tl, pmax, idx);
#endif
    if (SIGN < 0) w.y = -w.y;
    return w;
}

// One radix-R pass over `nch` channel buffers (channel c at buf + c*bstride), transform
// length N, current sub-length Ls.  DIT = false: decimation in frequency (DFT then twiddle),
// true: its transpose (twiddle then DFT).  Ends with a workgroup barrier.
template <int R, int SIGN, bool DIT>
__device__ __forceinline__ static void fft_pass(double2 *buf, int bstride, int nch, int N, int Ls, const double2 *__restrict__ tw,
                                int pmax, const double2 *__restrict__ postmul = nullptr) {
    const int q = Ls / R;
    const int nb = N / R;
    const int total = nch * nb;
    const int twstep = pmax / Ls;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int ch = idx / nb, t = idx - ch * nb;
        const int b = t / q, j = t - b * q;
        double2 *cbuf = buf + (size_t)ch * bstride;
        const int i0 = b * Ls + j;
        double2 x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = cbuf[fpad(i0 + r * q)];
        // twiddles w_Ls^{SIGN j r}: only the binary powers w, w^2, w^4, w^8 are kept in registers and
        // x[r] is multiplied by the ones its index selects (16 VGPRs instead of a 64-VGPR power table)
        constexpr int NB = R == 16 ? 4 : (R == 8 ? 3 : (R == 4 ? 2 : 1));
        double2 wp[NB];
        wp[0] = tw_get<SIGN>(tw, pmax, j * twstep);
#pragma unroll
        for (int b = 1; b < NB; b++) wp[b] = cmul(wp[b - 1], wp[b - 1]);
        auto twiddle_all = [&]() {
#pragma unroll
            for (int r = 1; r < R; r++) {
#pragma unroll
                for (int b = 0; b < NB; b++)
                    if (r & (1 << b)) x[r] = cmul(x[r], wp[b]);
            }
        };
        if (DIT) twiddle_all();
        DftR<R, SIGN>::run(x);
        if (!DIT) twiddle_all();
        if (postmul) {  // pointwise factor indexed by storage position (Bluestein filter, digit-reversed order)
#pragma unroll
            for (int r = 0; r < R; r++) x[r] = cmul(x[r], postmul[i0 + r * q]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) cbuf[fpad(i0 + r * q)] = x[r];
    }
    __syncthreads();
}

#ifndef K5_RADIX
#define K5_RADIX 16    // largest butterfly: 16 -> 512-thread workgroups; 8 -> 1024 threads (<= 128 VGPRs: measured 30 % slower, spills)
#endif
#define K5_LOGR (K5_RADIX == 16 ? 4 : 3)
#define K5_THREADS (K5_RADIX == 16 ? 512 : 1024)
// pass schedule for N = 2^k: radix K5_RADIX while it fits, then the remainder
// `postmul` (optional) is multiplied into the output of the LAST pass, indexed by storage position
template <int SIGN>
__device__ __forceinline__ static void fft_dif(double2 *buf, int bstride, int nch, int N, const double2 *__restrict__ tw, int pmax,
                                               const double2 *__restrict__ postmul = nullptr) {
    int Ls = N;
    while (Ls >= K5_RADIX) {
        fft_pass<K5_RADIX, SIGN, false>(buf, bstride, nch, N, Ls, tw, pmax, Ls == K5_RADIX ? postmul : nullptr);
        Ls >>= K5_LOGR;
    }
    if (K5_RADIX == 16 && Ls == 8) fft_pass<8, SIGN, false>(buf, bstride, nch, N, Ls, tw, pmax, postmul);
    else if (Ls == 4) fft_pass<4, SIGN, false>(buf, bstride, nch, N, Ls, tw, pmax, postmul);
    else if (Ls == 2) fft_pass<2, SIGN, false>(buf, bstride, nch, N, Ls, tw, pmax, postmul);
}
// transpose of fft_dif: digit-reversed order in -> natural order out
template <int SIGN>
__device__ __forceinline__ static void fft_dit(double2 *buf, int bstride, int nch, int N, const double2 *__restrict__ tw, int pmax) {
    int rem = N;
    while (rem >= K5_RADIX) rem >>= K5_LOGR;  // remainder radix handled first (it was last in DIF)
    int Ls = rem;
    if (K5_RADIX == 16 && rem == 8) fft_pass<8, SIGN, true>(buf, bstride, nch, N, Ls, tw, pmax);
    else if (rem == 4) fft_pass<4, SIGN, true>(buf, bstride, nch, N, Ls, tw, pmax);
    else if (rem == 2) fft_pass<2, SIGN, true>(buf, bstride, nch, N, Ls, tw, pmax);
    if (rem == 1) Ls = 1;
    while (Ls < N) {
        Ls <<= K5_LOGR;
        fft_pass<K5_RADIX, SIGN, true>(buf, bstride, nch, N, Ls, tw, pmax);
    }
}
// position of frequency index k in the digit-reversed output of fft_dif
__device__ static inline int fft_dif_pos(int k, int N) {
    int pos = 0, len = N, Ls = N;
    while (Ls >= K5_RADIX) {
        len >>= K5_LOGR;
        pos += (k & (K5_RADIX - 1)) * len;
        k >>= K5_LOGR;
        Ls >>= K5_LOGR;
    }
    if (Ls > 1) {
        len /= Ls;
        pos += (k & (Ls - 1)) * len;
    }
    return pos;
}

// ---- Bluestein convolution with the middle and the end kept in registers ----------------------
// forward DIF: every pass but the last
template <int SIGN>
__device__ __forceinline__ static int fft_dif_head(double2 *buf, int bstride, int nch, int N, const double2 *__restrict__ tw, int pmax) {
    int Ls = N;
    while (Ls > K5_RADIX) {   // stop with the final sub-length (<= K5_RADIX) left
        fft_pass<K5_RADIX, SIGN, false>(buf, bstride, nch, N, Ls, tw, pmax);
        Ls >>= K5_LOGR;
    }
    return Ls;  // radix of the last forward pass == radix of the first inverse pass (stride 1)
}
// last forward pass (DFT_R, sign -), pointwise filter, first inverse pass (DFT_R, sign +): both act on the
// same R consecutive storage positions and their twiddles are 1, so the data never leaves the registers
template <int R>
__device__ __forceinline__ static void fft_mid_fused(double2 *buf, int bstride, int nch, int N,
                                                     const double2 *__restrict__ filt) {
    const int nb = N / R;
    const int total = nch * nb;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int ch = idx / nb, b = idx - ch * nb;
        double2 *cbuf = buf + (size_t)ch * bstride;
        const int i0 = b * R;
        double2 x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = cbuf[fpad(i0 + r)];
        DftR<R, -1>::run(x);
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = cmul(x[r], filt[i0 + r]);
        DftR<R, 1>::run(x);
#pragma unroll
        for (int r = 0; r < R; r++) cbuf[fpad(i0 + r)] = x[r];
    }
    __syncthreads();
}
// inverse DIT passes after the first one, except the last (Ls == N), which is fft_dit_last_out
template <int SIGN>
__device__ __forceinline__ static void fft_dit_middle(double2 *buf, int bstride, int nch, int N, int Ls_first,
                                                      const double2 *__restrict__ tw, int pmax) {
    int Ls = Ls_first;
    while ((Ls << K5_LOGR) < N) {
        Ls <<= K5_LOGR;
        fft_pass<K5_RADIX, SIGN, true>(buf, bstride, nch, N, Ls, tw, pmax);
    }
}
// last inverse pass (radix K5_RADIX, Ls = N): results are natural-order j = j0 + r N/R; multiply by the
// chirp b_j / P and store the pixel pair (2j, 2j+1) of every channel straight to HBM (consecutive lanes ->
// consecutive j: coalesced 16-byte stores), j < h only.
template <int SIGN>
__device__ __forceinline__ static void fft_dit_last_out(const double2 *buf, int bstride, int nch, int N,
                                                        const double2 *__restrict__ tw, int pmax,
                                                        const double2 *__restrict__ chirpb, double invP, int h,
                                                        double *__restrict__ maps, long npix, long start, int ch0,
                                                        int nnu) {
    constexpr int R = K5_RADIX;
    const int q = N / R;
    const int total = nch * q;
    const int twstep = pmax / N;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int ch = idx / q, j = idx - ch * q;
        const double2 *cbuf = buf + (size_t)ch * bstride;
        double2 x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = cbuf[fpad(j + r * q)];
        constexpr int NB = R == 16 ? 4 : 3;
        double2 wp[NB];
        wp[0] = tw_get<SIGN>(tw, pmax, j * twstep);
#pragma unroll
        for (int b = 1; b < NB; b++) wp[b] = cmul(wp[b - 1], wp[b - 1]);
#pragma unroll
        for (int r = 1; r < R; r++) {
#pragma unroll
            for (int b = 0; b < NB; b++)
                if (r & (1 << b)) x[r] = cmul(x[r], wp[b]);
        }
        DftR<R, SIGN>::run(x);
#if K5_ABLATE == 2
        if (x[0].x == 1.2345e300)
#endif
        if (ch0 + ch < nnu) {
            double *out = maps + (size_t)(ch0 + ch) * npix + start;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int jj = j + r * q;
                if (jj < h) {
                    double2 zv = cmul(x[r], chirpb[jj]);
                    zv.x *= invP;
                    zv.y *= invP;
                    *reinterpret_cast<double2 *>(out + 2 * jj) = zv;
                }
            }
        }
    }
}

// Bluestein tables for cap ring i (h = 2i not a power of two): chirp b_j = e^{i pi j^2/h}, j < h,
// and filt = FFT_P(conj chirp wrapped), stored in the digit-reversed order fft_dif produces.
__global__ void __launch_bounds__(256)
bluestein_table_kernel(const int32_t *__restrict__ blu_P, const int64_t *__restrict__ boff,
                       const int64_t *__restrict__ foff, double2 *__restrict__ chirp, double2 *__restrict__ filt,
                       const double2 *__restrict__ tw, int pmax, int tl_off) {
    extern __shared__ __attribute__((aligned(16))) double2 fbuf[];   // [fpad(maxlen) + 1] then the twiddle table
    const int i = blockIdx.x + 1;
    const int P = blu_P[i - 1];
    if (P == 0) return;
#if K5_LDS_TW
    double2 *tl = fbuf + tl_off;
    twl_fill(tl, tw, pmax);
#else
    const double2 *tl = tw;
#endif
    const int h = 2 * i;
    double2 *b = chirp + boff[i - 1];
    for (int j = threadIdx.x; j < fpad(P); j += blockDim.x) fbuf[j] = make_double2(0.0, 0.0);
    __syncthreads();
    for (int j = threadIdx.x; j < h; j += blockDim.x) {
        const long q = ((long)j * j) % (2 * h);
        double s, c;
        sincospi((double)q / (double)h, &s, &c);
        b[j] = make_double2(c, s);
        fbuf[fpad(j)] = make_double2(c, -s);
        if (j > 0) fbuf[fpad(P - j)] = make_double2(c, -s);
    }
    __syncthreads();
    fft_dif<-1>(fbuf, 0, 1, P, tl, pmax);
    double2 *f = filt + foff[i - 1];
    for (int j = threadIdx.x; j < P; j += blockDim.x) f[j] = fbuf[fpad(j)];
}

// Persistent workgroups: each loops over work items (ring of the class, NCH consecutive channels),
// all NCH channels transformed together in LDS.  The F_m cells of the NEXT item are fetched into
// registers while the current item is in its FFT passes, so HBM reads overlap the FP64 work and the
// pixel stores of one item drain during the next.  P > 0: Bluestein of length P; P == 0: h = nphi/2
// is a power of two.
#define K5_MC (K5_RADIX == 16 ? 4 : 2)  // cells per thread held in registers for the next item (rest read in place)
#ifndef K5_STAMPS
#define K5_STAMPS 0  // diagnostic build: s_memtime phase breakdown
#endif
#ifndef K5_ABLATE
#define K5_ABLATE 0  // diagnostic builds (make k5ablate; wrong results, timing only): 1 no FFT, 2 no pixel stores, 3 no cell loads
#endif
#if K5_STAMPS
__device__ unsigned long long g_k5_stamps[8];
#define K5STAMP(acc) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); acc += _t - k5_last; k5_last = _t; }
#else
#define K5STAMP(acc)
#endif

// BLU = false: class of power-of-two rings only (the belt); the Bluestein code and its registers are compiled out
template <int NCH, bool BLU>
__global__ void __launch_bounds__(K5_THREADS)
ringfft_kernel(const int32_t *__restrict__ ring_list, int nlist, int nside, int lmax, int G, int nnu, long npix,
               const int32_t *__restrict__ nphi_a, const int64_t *__restrict__ start_a,
               const double *__restrict__ phi0_a, const double *__restrict__ inter, double *__restrict__ maps,
               const double2 *__restrict__ tw_hbm, int pmax, const int32_t *__restrict__ blu_P,
               const int64_t *__restrict__ boff, const int64_t *__restrict__ foff,
               const double2 *__restrict__ chirp, const double2 *__restrict__ filt, int bstride,
               const int32_t *__restrict__ mcut) {
    // cells per thread prefetched into registers for the next item (the rest are read in place): the Bluestein
    // instantiations need the registers for the fused filter pass (4 cells made them spill 66 VGPRs)
    constexpr int MC = BLU ? (K5_MC > 2 ? 2 : K5_MC) : K5_MC;
    extern __shared__ __attribute__((aligned(16))) double2 sm[];  // [NCH][bstride], then the twiddle table
    const int tid = threadIdx.x, nt = blockDim.x;
    const int L = lmax + 1;
    const int ngrp = (nnu + NCH - 1) / NCH;
    const int nitems = nlist * ngrp;
    double *smd = reinterpret_cast<double *>(sm);
#if K5_LDS_TW
    double2 *tl = sm + (size_t)NCH * bstride;
    twl_fill(tl, tw_hbm, pmax);
    const double2 *tw = tl;   // every twiddle below comes from LDS
#else
    const double2 *tw = tw_hbm;
#endif

    // register prefetch of the cells m = tid + k nt, k < MC, of one item
    struct cell_t {
        double re[NCH], im[NCH];
    };
    cell_t pf0, pf1, pf2, pf3;
    auto cell_ptr = [&](int item) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        return inter + ((size_t)ring * G + (ch0 >> 2)) * L * 8 + (ch0 & 3);
    };
    auto load_cell = [&](const double *cell, int m) {
        cell_t c;
#if K5_ABLATE == 3
        for (int q = 0; q < NCH; q++) { c.re[q] = 1.0 + m; c.im[q] = 0.5; }
        return c;
#endif
        if (NCH == 4) {
            const double4 a = *reinterpret_cast<const double4 *>(cell + (unsigned)m * 8u);
            const double4 b = *reinterpret_cast<const double4 *>(cell + (unsigned)m * 8u + 4u);
            c.re[0] = a.x; c.re[1 % NCH] = a.y; c.re[2 % NCH] = a.z; c.re[3 % NCH] = a.w;
            c.im[0] = b.x; c.im[1 % NCH] = b.y; c.im[2 % NCH] = b.z; c.im[3 % NCH] = b.w;
        } else if (NCH == 2) {
            const double2 a = *reinterpret_cast<const double2 *>(cell + (unsigned)m * 8u);
            const double2 b = *reinterpret_cast<const double2 *>(cell + (unsigned)m * 8u + 4u);
            c.re[0] = a.x; c.re[1 % NCH] = a.y;
            c.im[0] = b.x; c.im[1 % NCH] = b.y;
        } else {
            c.re[0] = cell[(unsigned)m * 8u];
            c.im[0] = cell[(unsigned)m * 8u + 4u];
        }
        return c;
    };
    // Branch-free: every thread loads its MC cells (index clamped to the last cell of the row; cells at or
    // beyond mcut(ring) hold stale data that the fold never consumes).  Conditional loads made hipcc drain
    // vmcnt at every join, serialising the prefetch and exposing the whole HBM latency to the next fold.
    auto prefetch = [&](int item) {
        // uniform base (+ k nt cells) and ONE per-thread 32-bit offset: saddr-form loads, no 64-bit address
        // arithmetic or spilled per-cell offsets between them.  Cells past the row end belong to the next
        // row / the workspace tail pad (alm2map_workspace_bytes adds it) and are never consumed.
        const double *cell = cell_ptr(item);
        pf0 = load_cell(cell, tid);
        pf1 = load_cell(cell + (size_t)nt * 8, tid);
        if (MC > 2) {
            pf2 = load_cell(cell + (size_t)2 * nt * 8, tid);
            pf3 = load_cell(cell + (size_t)3 * nt * 8, tid);
        }
    };

#if K5_STAMPS
    unsigned long long k5_last, t_zero = 0, t_fold = 0, t_z = 0, t_fft = 0, t_out = 0, t_pre = 0, t_mid = 0, t_dit = 0;
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); k5_last = _t; }
#endif
    int item = blockIdx.x;
    if (item < nitems) prefetch(item);
    for (; item < nitems; item += gridDim.x) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const int n = nphi_a[ring];
        const int h = n >> 1;
        const long start = start_a[ring];
        const double phi0_over_pi = phi0_a[ring] / M_PI;
        int icap = 0;
        if (ring + 1 < nside) icap = ring + 1;
        else if (ring + 1 > 3 * nside) icap = 4 * nside - (ring + 1);
        const int P = (BLU && icap) ? blu_P[icap - 1] : 0;
        const int flen = P ? P : h + 1;
        const int Lr = mcut[ring];

        // every m present is <= h (polar rings: mcut(ring) << lmax): each bin gets at most one contribution, so the
        // fold is a plain store, and only the bins behind the last cell need zeroing
        const bool noalias = Lr - 1 <= h;
        const bool need_zero = !(noalias && Lr == h + 1 && P == 0);  // direct ring whose bins 0..h are all written
        K5STAMP(t_pre);
        __syncthreads();  // previous item's LDS reads are done
        if (need_zero) {
            for (int j = (noalias ? fpad(Lr) : 0) + tid; j < fpad(flen); j += nt)
#pragma unroll
                for (int c = 0; c < NCH; c++) sm[(size_t)c * bstride + j] = make_double2(0.0, 0.0);
            if (!noalias) __syncthreads();   // (the stores of the fold and the zeroed tail are disjoint otherwise)
        }

        // ---- phase + alias fold onto bins 0..h of the Hermitian length-n spectrum X
        K5STAMP(t_zero);
        const double *cell = cell_ptr(item);
        // e^{i m phi0}: one sincospi for m = tid, then the fixed rotation e^{i nt phi0} per further cell
        double2 ph, phstep;
        {
            double s, c;
            sincospi(fmod((double)tid * phi0_over_pi, 2.0), &s, &c);
            ph = make_double2(c, s);
            sincospi(fmod((double)nt * phi0_over_pi, 2.0), &s, &c);
            phstep = make_double2(c, s);
        }
        auto fold_one = [&](int m, const cell_t cv) {
            const int k = m % n;
            const int kc = (n - k) % n;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const double2 val = cmul(make_double2(cv.re[c], cv.im[c]), ph);
                double *bd = smd + (size_t)c * bstride * 2;
                if (m == 0) {
                    if (noalias) *reinterpret_cast<double2 *>(bd) = make_double2(val.x, 0.0);  // Re(c_0) only
                    else atomicAdd(&bd[0], val.x);
                } else if (noalias) {
                    if (m < h) *reinterpret_cast<double2 *>(bd + 2 * fpad(m)) = val;
                    else *reinterpret_cast<double2 *>(bd + 2 * fpad(h)) = make_double2(2.0 * val.x, 0.0);  // m == h: c + conj(c)
                } else {
                    if (k <= h) {
                        atomicAdd(&bd[2 * fpad(k)], val.x);
                        atomicAdd(&bd[2 * fpad(k) + 1], val.y);
                    }
                    if (kc <= h) {
                        atomicAdd(&bd[2 * fpad(kc)], val.x);
                        atomicAdd(&bd[2 * fpad(kc) + 1], -val.y);
                    }
                }
            }
        };
        if (tid < Lr) fold_one(tid, pf0);
        ph = cmul(ph, phstep);
        if (tid + nt < Lr) fold_one(tid + nt, pf1);
        if (MC > 2) {
            ph = cmul(ph, phstep);
            if (tid + 2 * nt < Lr) fold_one(tid + 2 * nt, pf2);
            ph = cmul(ph, phstep);
            if (tid + 3 * nt < Lr) fold_one(tid + 3 * nt, pf3);
        }
        for (int m = tid + MC * nt; m < Lr; m += nt) {
            ph = cmul(ph, phstep);
            fold_one(m, load_cell(cell, m));
        }
        // the registers are free again: fetch the next item's cells behind the FFT passes
        if (item + (int)gridDim.x < nitems) prefetch(item + gridDim.x);
        __syncthreads();
        K5STAMP(t_fold);
        // ---- Hermitian -> half-length complex: Z_k = (X_k + conj X_{h-k}) + i w^k (X_k - conj X_{h-k}),
        //      w = e^{2 pi i/n}; pairs (k, h-k) updated together.  Bluestein: times chirp b_k.
        const double2 *bch = P ? chirp + boff[icap - 1] : nullptr;
        const bool n_in_table = (pmax % n) == 0;
        for (int k = tid; k <= h / 2; k += nt) {
            const int k2 = h - k;
            double2 w;
            if (n_in_table) w = tw_get<1>(tw, pmax, k * (pmax / n));
            else {
                double s, c;
                sincospi(2.0 * (double)k / (double)n, &s, &c);
                w = make_double2(c, s);
            }
            double2 bk = make_double2(1.0, 0.0), bk2 = make_double2(1.0, 0.0);
            if (P) {
                if (k < h) bk = bch[k];
                if (k2 < h) bk2 = bch[k2];
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                double2 *bc = sm + (size_t)c * bstride;
                const double2 xa = bc[fpad(k)], xb = bc[fpad(k2)];
                double2 sum = make_double2(xa.x + xb.x, xa.y - xb.y);
                double2 dif = make_double2(xa.x - xb.x, xa.y + xb.y);
                double2 t = cmul(dif, w);
                double2 zk = make_double2(sum.x - t.y, sum.y + t.x);
                sum = make_double2(xb.x + xa.x, xb.y - xa.y);
                dif = make_double2(xb.x - xa.x, xb.y + xa.y);
                t = cmul(dif, make_double2(-w.x, w.y));  // w^{h-k} = -conj(w^k)
                double2 zk2 = make_double2(sum.x - t.y, sum.y + t.x);
                if (P) {
                    zk = cmul(zk, bk);
                    zk2 = cmul(zk2, bk2);
                }
                if (k2 < h) bc[fpad(k2)] = zk2;
                else if (P) bc[fpad(k2)] = make_double2(0.0, 0.0);  // slot h is padding for the length-P transform
                if (k < h) bc[fpad(k)] = zk;
            }
        }
        __syncthreads();
        K5STAMP(t_z);

        if (!BLU || P == 0) {
#if K5_ABLATE != 1
            fft_dif<1>(sm, bstride, NCH, h, tw, pmax);
#endif
            K5STAMP(t_fft);
            for (int j = tid; j < h; j += nt) {
                const int pos = fpad(fft_dif_pos(j, h));
#pragma unroll
                for (int c = 0; c < NCH; c++) {
#if K5_ABLATE == 2
                    if (sm[(size_t)c * bstride + pos].x == 1.2345e300)
#endif
                    if (ch0 + c < nnu)
                        *reinterpret_cast<double2 *>(maps + (size_t)(ch0 + c) * npix + start + 2 * j) =
                            sm[(size_t)c * bstride + pos];
                }
            }
        } else {
            const double2 *f = filt + foff[icap - 1];
            const double invP = 1.0 / (double)P;
#if K5_ABLATE == 1
            if (false) {
#else
            if (P >= K5_RADIX * K5_RADIX) {
#endif
                // >= 3 passes each way: the filter step and the final chirp/store are fused into the passes
                const int rl = fft_dif_head<-1>(sm, bstride, NCH, P, tw, pmax);
                K5STAMP(t_fft);      // stamps build: forward passes but the last
                if (rl == 16) fft_mid_fused<16>(sm, bstride, NCH, P, f);
                else if (rl == 8) fft_mid_fused<8>(sm, bstride, NCH, P, f);
                else if (rl == 4) fft_mid_fused<4>(sm, bstride, NCH, P, f);
                else fft_mid_fused<2>(sm, bstride, NCH, P, f);
                K5STAMP(t_mid);      // last forward pass + filter + first inverse pass (registers)
                fft_dit_middle<1>(sm, bstride, NCH, P, rl, tw, pmax);
                K5STAMP(t_dit);      // middle inverse passes
                fft_dit_last_out<1>(sm, bstride, NCH, P, tw, pmax, bch, invP, h, maps, npix, start, ch0, nnu);
                K5STAMP(t_out);      // last inverse pass + chirp + pixel stores
                continue;
            }
#if K5_ABLATE != 1
            fft_dif<-1>(sm, bstride, NCH, P, tw, pmax, f);  // filter multiplied in by the last pass
            fft_dit<1>(sm, bstride, NCH, P, tw, pmax);
#endif
            K5STAMP(t_fft);
            for (int j = tid; j < h; j += nt) {
                const double2 bj = bch[j];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
#if K5_ABLATE == 2
                    if (bj.x == 1.2345e300)
#endif
                    if (ch0 + c < nnu) {
                        double2 zv = cmul(sm[(size_t)c * bstride + fpad(j)], bj);
                        zv.x *= invP;
                        zv.y *= invP;
                        *reinterpret_cast<double2 *>(maps + (size_t)(ch0 + c) * npix + start + 2 * j) = zv;
                    }
                }
            }
        }
        K5STAMP(t_out);
    }
#if K5_STAMPS
    if ((tid & 63) == 0) {
        atomicAdd(&g_k5_stamps[0], t_pre);
        atomicAdd(&g_k5_stamps[1], t_zero);
        atomicAdd(&g_k5_stamps[2], t_fold);
        atomicAdd(&g_k5_stamps[3], t_z);
        atomicAdd(&g_k5_stamps[4], t_fft);
        atomicAdd(&g_k5_stamps[5], t_out);
        atomicAdd(&g_k5_stamps[6], t_mid);
        atomicAdd(&g_k5_stamps[7], t_dit);
    }
#endif
}

// ------------------------------------------------------------------------------------
// Analysis (adjoint of K5 and K4): healpy.map2alm as the reference reaches it through
// hputil.sphtrans_real / sphtrans_sky / sph_ps (cora/util/hputil.py:195-234,460-497,607-619)
// ------------------------------------------------------------------------------------
// K5^T  ringana_kernel: per ring, G_m = w_ring (4 pi / npix) e^{-i m phi0} sum_j x_j e^{-2 pi i j m / n},
//       m < mcut(ring), for NCH channels at once, written in the `inter` cell layout K4 writes and K5 reads.
//       The n real pixels are packed as h = n/2 complex numbers z_j = x_2j + i x_2j+1; a complex transform of
//       length h (radix-16 LDS passes; Bluestein with the synthesis' chirp/filter tables for the cap rings,
//       run on conj(z) so that the same e^{+...} machinery serves) and the split
//           X_k = 1/2 [(Z_k + conj Z_{h-k}) - i e^{-2 pi i k/n} (Z_k - conj Z_{h-k})]
//       give bins 0..h; m >= n aliases back (k = m mod n, conjugate above h).
template <int NCH>
__global__ void __launch_bounds__(K5_THREADS)
ringana_kernel(const int32_t *__restrict__ ring_list, int nlist, int nside, int lmax, int G, int nnu, long npix,
               const int32_t *__restrict__ nphi_a, const int64_t *__restrict__ start_a,
               const double *__restrict__ phi0_a, const double *__restrict__ maps, double *__restrict__ inter,
               const double2 *__restrict__ tw_hbm, int pmax, const int32_t *__restrict__ blu_P,
               const int64_t *__restrict__ boff, const int64_t *__restrict__ foff,
               const double2 *__restrict__ chirp, const double2 *__restrict__ filt, int bstride,
               const int32_t *__restrict__ mcut, const double *__restrict__ ring_w, int nvalid) {
    // nnu: channels incl. padding (every cell K4^T reads gets written); nvalid: channels present in `maps`
    extern __shared__ __attribute__((aligned(16))) double2 sm[];  // [NCH][bstride], then the twiddle table
    const int tid = threadIdx.x, nt = blockDim.x;
#if K5_LDS_TW
    double2 *tl = sm + (size_t)NCH * bstride;
    twl_fill(tl, tw_hbm, pmax);
    const double2 *tw = tl;
#else
    const double2 *tw = tw_hbm;
#endif
    const int L = lmax + 1;
    const int ngrp = (nnu + NCH - 1) / NCH;
    const int nitems = nlist * ngrp;
    const int nring = 4 * nside - 1;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const int n = nphi_a[ring];
        const int h = n >> 1;
        const long start = start_a[ring];
        const double phi0_over_pi = phi0_a[ring] / M_PI;
        int icap = 0;
        if (ring + 1 < nside) icap = ring + 1;
        else if (ring + 1 > 3 * nside) icap = 4 * nside - (ring + 1);
        const int P = icap ? blu_P[icap - 1] : 0;
        const int Lr = mcut[ring];
        const double wr = (ring_w ? ring_w[min(ring, nring - 1 - ring)] : 1.0) * (4.0 * M_PI / (double)npix);
        const double2 *bch = P ? chirp + boff[icap - 1] : nullptr;
        __syncthreads();  // previous item's LDS reads are done
        if (P) {
            for (int j = h + tid; j < P; j += nt)
#pragma unroll
                for (int c = 0; c < NCH; c++) sm[(size_t)c * bstride + fpad(j)] = make_double2(0.0, 0.0);
        }
        // ---- pixels -> packed complex (conjugated and chirped for the Bluestein path)
        for (int j = tid; j < h; j += nt) {
            const double2 bj = P ? bch[j] : make_double2(1.0, 0.0);
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                double2 zv = make_double2(0.0, 0.0);
                if (ch0 + c < nvalid) zv = *reinterpret_cast<const double2 *>(maps + (size_t)(ch0 + c) * npix + start + 2 * j);
                if (P) zv = cmul(make_double2(zv.x, -zv.y), bj);
                sm[(size_t)c * bstride + fpad(j)] = zv;
            }
        }
        __syncthreads();
        double scaleZ = 1.0;
        if (P == 0) {
            fft_dif<-1>(sm, bstride, NCH, h, tw, pmax);          // Z_k at fft_dif_pos(k)
        } else {
            const double2 *f = filt + foff[icap - 1];
            fft_dif<-1>(sm, bstride, NCH, P, tw, pmax, f);
            fft_dit<1>(sm, bstride, NCH, P, tw, pmax);           // W'_k natural order; Z_k = conj(W'_k b_k / P)
            scaleZ = 1.0 / (double)P;
        }
        auto getZ = [&](const double2 *bc, int k) {  // k in [0, h)
            if (P == 0) return bc[fpad(fft_dif_pos(k, h))];
            const double2 v = cmul(bc[fpad(k)], bch[k]);
            return make_double2(v.x * scaleZ, -v.y * scaleZ);
        };
        // ---- bins -> G_m cells
        const bool n_in_table = (pmax % n) == 0;
        double *cell0 = inter + ((size_t)ring * G + (ch0 >> 2)) * L * 8 + (ch0 & 3);
        for (int m = tid; m < Lr; m += nt) {
            const int k = m % n;
            const bool cj = k > h;
            const int kk = cj ? n - k : k;            // 0..h
            const int ka = kk == h ? 0 : kk;          // Z_h := Z_0
            const int kb = kk == 0 ? 0 : h - kk;      // partner h - kk (kk = 0 -> Z_h = Z_0)
            double2 w;                                // e^{-2 pi i kk / n}
            if (n_in_table) w = tw_get<-1>(tw, pmax, kk * (pmax / n));
            else {
                double sv, cv;
                sincospi(2.0 * (double)kk / (double)n, &sv, &cv);
                w = make_double2(cv, -sv);
            }
            double sp, cp;
            sincospi(fmod((double)m * phi0_over_pi, 2.0), &sp, &cp);
            const double2 ph = make_double2(cp * wr, -sp * wr);   // w_ring area e^{-i m phi0}
            double re[NCH], im[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const double2 *bc = sm + (size_t)c * bstride;
                const double2 za = getZ(bc, ka), zb = getZ(bc, kb);
                const double2 sum = make_double2(za.x + zb.x, za.y - zb.y);   // Z_k + conj Z_{h-k}
                const double2 dif = make_double2(za.x - zb.x, za.y + zb.y);   // Z_k - conj Z_{h-k}
                const double2 t = cmul(dif, w);
                // X = 1/2 [sum - i t]
                double2 X = make_double2(0.5 * (sum.x + t.y), 0.5 * (sum.y - t.x));
                if (cj) X.y = -X.y;
                const double2 g = cmul(X, ph);
                re[c] = g.x;
                im[c] = g.y;
            }
            double *cell = cell0 + (size_t)m * 8;
            if (NCH == 4) {
                *reinterpret_cast<double4 *>(cell) = make_double4(re[0], re[1 % NCH], re[2 % NCH], re[3 % NCH]);
                *reinterpret_cast<double4 *>(cell + 4) = make_double4(im[0], im[1 % NCH], im[2 % NCH], im[3 % NCH]);
            } else if (NCH == 2) {
                *reinterpret_cast<double2 *>(cell) = make_double2(re[0], re[1 % NCH]);
                *reinterpret_cast<double2 *>(cell + 4) = make_double2(im[0], im[1 % NCH]);
            } else {
                cell[0] = re[0];
                cell[4] = im[0];
            }
        }
    }
}

// K4^T  legendre_adj_kernel: a_lm(col) = sum_rings lambda_lm(ring) [G_m(north) + (-1)^{l+m} G_m(south)](col)
// on FP64 MFMA with M = l, K = ring pairs, N = columns (channel re/im).  Work item = (m, 16 NCT columns,
// tile of 512 ring pairs).  A wave owns 64 ring pairs: lane = ring steps the recurrence once per l (no
// redundancy), the 32 lambda values of an l-block go through a wave-private LDS transpose into the A-operand
// layout (16 same-parity l x 4 rings), and the wave's G tile (64 rings x 16 NCT columns, even = N+S and
// odd = N-S combinations) stays in REGISTERS as the B operand for the whole item.  The eight waves hold
// different rings, so their [32 l x 16 NCT] partial sums are added through LDS once per l-block; the four
// ring tiles of an (m, column group) go to separate partial buffers summed by alm_reduce_kernel
// (deterministic - no atomics).
template <int NCT>
__global__ void __launch_bounds__(512)
legendre_adj_kernel(int lmax, int npair, int nring, int ncols, const double *__restrict__ z,
                    const double2 *__restrict__ coef, const int32_t *__restrict__ lstart,
                    const double2 *__restrict__ seed, const int32_t *__restrict__ lmin_tab,
                    const int32_t *__restrict__ mcut, const double *__restrict__ inter, double *__restrict__ part,
                    unsigned *__restrict__ queue) {
    constexpr int TCOLS = 16 * NCT;
    constexpr int TRINGS = 64 * ADJ_WAVES;     // 512 ring pairs per workgroup
    constexpr int LB = 32;                     // l per block: 16 even + 16 odd (l - m)
    constexpr int LSTR = LB + 1;               // LDS row stride of the transpose: conflict-free both ways
    constexpr int WREG = 64 * LSTR;            // doubles per wave
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int &s_next = *reinterpret_cast<int *>(lds + ADJ_WAVES * WREG);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    const int L = lmax + 1;
    const int G = ncols >> 3;
    const long nalm = nalm_of(lmax);
    const int ntile128 = (npair + LMIN_RINGS - 1) / LMIN_RINGS;
    const int ntile = (npair + TRINGS - 1) / TRINGS;
    const int ncg = ncols / TCOLS;
    const int nitems = L * ncg * ntile;
    double *lamw = lds + wave * WREG;

    int item = blockIdx.x;
    while (item < nitems) {
        if (tid == 0) s_next = (int)(gridDim.x + atomicAdd(queue, 1u));
        const int gidx = item / ntile;
        const int rtile = item - gidx * ntile;
        const int m = gidx / ncg;
        const int cg = gidx - m * ncg;
        int lmin = lmax + 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int t128 = rtile * 4 + q;
            if (t128 < ntile128) lmin = min(lmin, lmin_tab[m * ntile128 + t128]);
        }
        const long base_m = alm_idx(0, m, lmax);
        const int lb0 = lmin <= lmax ? m + ((lmin - m) & ~(LB - 1)) : lmax + 1;
        double *pout = part + ((size_t)rtile * nalm + base_m) * ncols + (size_t)cg * TCOLS;
        // multipoles this tile cannot reach contribute zero
        for (int e = tid; e < (lb0 - m) * TCOLS; e += 512) pout[(size_t)(m + e / TCOLS) * ncols + e % TCOLS] = 0.0;

        if (lb0 <= lmax) {
            // ---- this wave's G tile -> registers (B operand): k-step s covers rings 4s..4s+3 of the wave
            double ge[16][NCT], go[16][NCT];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int ro = rtile * TRINGS + (4 * s + kq) * ADJ_WAVES + wave;
                const bool ok = ro < npair && m < mcut[min(ro, npair - 1)];
                const int rs = nring - 1 - ro;
#pragma unroll
                for (int t = 0; t < NCT; t++) {
                    const int col = cg * TCOLS + 16 * t + ri;
                    const size_t off = ((size_t)(col >> 3) * L + m) * 8 + (col & 7);
                    double gn = 0.0, gsv = 0.0;
                    if (ok) {
                        gn = inter[(size_t)ro * G * L * 8 + off];
                        if (rs != ro) gsv = inter[(size_t)rs * G * L * 8 + off];
                    }
                    ge[s][t] = gn + gsv;
                    go[s][t] = gn - gsv;
                }
            }
            // ---- recurrence state of this lane's ring
            const int ring = rtile * TRINGS + lane * ADJ_WAVES + wave;
            double x = 0.0, p0 = 0.0, p1 = 0.0;
            double2 sd = make_double2(0.0, 0.0);
            int my_ls = lmax + 1;
            if (ring < npair) {
                x = z[ring];
                const long o = (long)m * npair + ring;
                my_ls = lstart[o];
                sd = seed[o];
            }
            // first l of each group of 4 lanes (= one MFMA k-step): lets whole k-steps be skipped
            int ls4 = min(my_ls, __shfl_xor(my_ls, 1));
            ls4 = min(ls4, __shfl_xor(ls4, 2));
            const double2 *cf = coef + base_m;

            for (int lb = lb0; lb <= lmax; lb += LB) {
                const unsigned long long act = __ballot(ls4 <= lb + LB - 1);   // bit 4s: k-step s has a started ring
                // lambda_{lb .. lb+31} of this lane's ring -> transpose buffer [ring][l - lb]
                // (coefficients are read unconditionally - the table is padded by 32 entries - so that the scalar
                // loads of a whole unrolled group are issued together; rows past lmax are discarded below)
#pragma unroll 8
                for (int j = 0; j < LB; j++) {
                    const int l = lb + j;
                    const double2 c = cf[l];
                    double vv = fma(c.x * x, p1, -(c.y * p0));
                    const bool inj = (l == my_ls);
                    vv = inj ? sd.y : vv;
                    p0 = inj ? sd.x : p1;
                    p1 = vv;
                    lamw[lane * LSTR + j] = vv;
                }
                d4_t acc[2][NCT];
#pragma unroll
                for (int par = 0; par < 2; par++)
#pragma unroll
                    for (int t = 0; t < NCT; t++) acc[par][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 16; s++) {
                    if (!((act >> (4 * s)) & 1ull)) continue;
                    const double ae = lamw[(4 * s + kq) * LSTR + 2 * ri];
                    const double ao = lamw[(4 * s + kq) * LSTR + 2 * ri + 1];
#pragma unroll
                    for (int t = 0; t < NCT; t++) {
                        acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, ge[s][t], acc[0][t], 0, 0, 0);
                        acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, go[s][t], acc[1][t], 0, 0, 0);
                    }
                }
                // ---- add the eight waves' partial tiles through LDS (the wave's transpose buffer is free now:
                //      LDS operations of one wave are ordered)
#pragma unroll
                for (int par = 0; par < 2; par++)
#pragma unroll
                    for (int t = 0; t < NCT; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) lamw[((par * NCT + t) * 4 + r) * 64 + lane] = acc[par][t][r];
                __syncthreads();
#pragma unroll
                for (int u = 0; u < NCT; u++) {
                    const int e = tid + 512 * u;          // element (par, t, r, lane) of the reduced tile
                    double sum = 0.0;
#pragma unroll
                    for (int w = 0; w < ADJ_WAVES; w++) sum += lds[w * WREG + e];
                    const int el = e & 63, r = (e >> 6) & 3, t = (e >> 8) % NCT, par = (e >> 8) / NCT;
                    const int l = lb + 2 * ((el >> 4) + 4 * r) + par;
                    if (l <= lmax) pout[(size_t)l * ncols + 16 * t + (el & 15)] = sum;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();
    }
}

// alm_dev[idx][col] = sum over ring tiles of part[rt][idx][col]
__global__ void alm_reduce_kernel(const double *__restrict__ part, long n, int ntile, double *__restrict__ alm) {
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < ntile; t++) s += part[(size_t)t * n + q];
        alm[q] = s;
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
template <typename T>
static int dev_upload(T **dptr, const std::vector<T> &h, hipStream_t s) {
    HIP_TRY(hipMalloc((void **)dptr, std::max<size_t>(1, h.size()) * sizeof(T)));
    if (!h.empty()) {
        HIP_TRY(hipMemcpyAsync(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return 0;
}

extern "C" {

int corahip_sht_plan_destroy(corahip_ctx *ctx, corahip_sht_plan *p) {
    if (!p) return 0;
    (void)hipFree(p->d_z);
    (void)hipFree(p->d_sth);
    (void)hipFree(p->d_nphi);
    (void)hipFree(p->d_start);
    (void)hipFree(p->d_phi0);
    (void)hipFree(p->d_coef);
    (void)hipFree(p->d_lstart);
    (void)hipFree(p->d_seed);
    (void)hipFree(p->d_tw);
    (void)hipFree(p->d_zeros);
    (void)hipFree(p->d_lmin);
    (void)hipFree(p->d_queue);
    (void)hipFree(p->d_mcut);
    (void)hipFree(p->d_polc);
    (void)hipFree(p->d_blu_P);
    (void)hipFree(p->d_blu_boff);
    (void)hipFree(p->d_blu_foff);
    (void)hipFree(p->d_bchirp);
    (void)hipFree(p->d_bfilt);
    for (auto &c : p->classes) (void)hipFree(c.d_list);
    delete p;
    return 0;
}

int corahip_sht_plan_create(corahip_ctx *ctx, int nside, int lmax, corahip_sht_plan **out) {
    ARG_CHECK(ctx != nullptr && out != nullptr);
    ARG_CHECK(nside >= 1 && is_pow2(nside) && nside <= 8192);
    ARG_CHECK(lmax >= 0 && lmax <= 16384);
    HIP_TRY(hipSetDevice(ctx->device));
    corahip_sht_plan *p = new corahip_sht_plan();
    // every early return below (HIP_TRY / rc checks) releases what was allocated so far
    struct plan_guard {
        corahip_sht_plan *p;
        ~plan_guard() {
            if (p) corahip_sht_plan_destroy(nullptr, p);
        }
    } guard{p};
    p->nside = nside;
    p->lmax = lmax;
    p->L = lmax + 1;
    p->npair = 2 * nside;
    p->nring = 4 * nside - 1;
    p->npix = 12L * nside * nside;
    p->nalm = nalm_of(lmax);
    const int nring = p->nring;
    p->h_start.resize(nring);
    p->h_nphi.resize(nring);
    p->h_z.resize(nring);
    p->h_sth.resize(nring);
    p->h_phi0.resize(nring);
    // HEALPix RING geometry (pix2ang_ring conventions)
    const double fact2 = 4.0 / (double)p->npix;       // 1/(3 nside^2)
    const double fact1 = 2.0 * nside * fact2;         // 2/(3 nside)
    for (int r = 0; r < nring; r++) {
        const int i = r + 1;
        if (i < nside) {
            const double tmp = (double)i * i * fact2;
            p->h_z[r] = 1.0 - tmp;
            p->h_sth[r] = sqrt(tmp * (2.0 - tmp));
            p->h_nphi[r] = 4 * i;
            p->h_phi0[r] = M_PI / (4.0 * i);
            p->h_start[r] = 2L * i * (i - 1);
        } else if (i <= 3 * nside) {
            const double zz = (2 * nside - i) * fact1;
            p->h_z[r] = zz;
            p->h_sth[r] = sqrt((1.0 - zz) * (1.0 + zz));
            p->h_nphi[r] = 4 * nside;
            p->h_phi0[r] = (((i - nside) & 1) == 0) ? M_PI / (4.0 * nside) : 0.0;
            p->h_start[r] = 2L * nside * (nside - 1) + (long)(i - nside) * 4 * nside;
        } else {
            const int ip = 4 * nside - i;
            const double tmp = (double)ip * ip * fact2;
            p->h_z[r] = -(1.0 - tmp);
            p->h_sth[r] = sqrt(tmp * (2.0 - tmp));
            p->h_nphi[r] = 4 * ip;
            p->h_phi0[r] = M_PI / (4.0 * ip);
            p->h_start[r] = p->npix - 2L * ip * (ip + 1);
        }
    }
    hipStream_t s = ctx->stream;
    int rc;
    {
        std::vector<double> zz(p->h_z.begin(), p->h_z.begin() + p->npair);
        std::vector<double> ss(p->h_sth.begin(), p->h_sth.begin() + p->npair);
        if ((rc = dev_upload(&p->d_z, zz, s))) return rc;
        if ((rc = dev_upload(&p->d_sth, ss, s))) return rc;
    }
    if ((rc = dev_upload(&p->d_nphi, p->h_nphi, s))) return rc;
    if ((rc = dev_upload(&p->d_start, p->h_start, s))) return rc;
    if ((rc = dev_upload(&p->d_phi0, p->h_phi0, s))) return rc;

    // recurrence coefficients: lambda_l = A_l x lambda_{l-1} - B_l lambda_{l-2},
    // A_l = alpha_lm, B_l = alpha_lm/alpha_{l-1,m}, alpha_lm = sqrt((4l^2-1)/(l^2-m^2))
    {
        std::vector<double2> coef(p->nalm + 32, make_double2(0.0, 0.0));  // +32: K4 prefetches past the end
        for (int m = 0; m <= lmax; m++) {
            long double aprev = 0.0L;
            for (int l = m; l <= lmax; l++) {
                const long o = alm_idx(l, m, lmax);
                if (l == m) {
                    coef[o] = make_double2(0.0, 0.0);
                    continue;
                }
                const long double ll = l, mm = m;
                const long double al = sqrtl((4.0L * ll * ll - 1.0L) / (ll * ll - mm * mm));
                coef[o] = make_double2((double)al, l == m + 1 ? 0.0 : (double)(al / aprev));
                aprev = al;
            }
        }
        if ((rc = dev_upload(&p->d_coef, coef, s))) return rc;
    }
    // |lambda_mm| prefactor sqrt((2m+1)!!/(4 pi (2m)!!))
    double *d_pref = nullptr;
    {
        std::vector<double> pref(p->L);
        long double pr = 1.0L / sqrtl(4.0L * acosl(-1.0L));
        pref[0] = (double)pr;
        for (int m = 1; m <= lmax; m++) {
            pr *= sqrtl((2.0L * m + 1.0L) / (2.0L * m));
            pref[m] = (double)pr;
        }
        if ((rc = dev_upload(&d_pref, pref, s))) return rc;
    }
    HIP_TRY(hipMalloc((void **)&p->d_lstart, sizeof(int32_t) * (size_t)p->L * p->npair));
    HIP_TRY(hipMalloc((void **)&p->d_seed, sizeof(double2) * (size_t)p->L * p->npair));
    {
        dim3 grid((p->npair + 63) / 64, p->L);
        seed_kernel<<<grid, 64, 0, s>>>(lmax, p->npair, p->d_z, p->d_sth, d_pref, p->d_coef, p->d_lstart, p->d_seed);
        LAUNCH_CHECK();
    }
    {
        const int ntile = (p->npair + LMIN_RINGS - 1) / LMIN_RINGS;
        HIP_TRY(hipMalloc((void **)&p->d_queue, 64));
        HIP_TRY(hipMalloc((void **)&p->d_mcut, sizeof(int32_t) * (size_t)p->nring));
        mcut_kernel<<<(p->npair + 63) / 64, 64, 0, s>>>(lmax, p->npair, p->nring, p->d_lstart, p->d_mcut);
        LAUNCH_CHECK();
        HIP_TRY(hipMalloc((void **)&p->d_lmin, sizeof(int32_t) * (size_t)p->L * ntile));
        lmin_kernel<<<p->L, 64 * ((ntile + 63) / 64), 0, s>>>(lmax, p->npair, ntile, p->d_lstart, p->d_lmin);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(d_pref);

    HIP_TRY(hipMalloc((void **)&p->d_zeros, 4096));
    HIP_TRY(hipMemsetAsync(p->d_zeros, 0, 4096, s));
    // FFT twiddles and Bluestein tables
    p->pmax = std::max(4 * nside, 4);
    p->log_pmax = ilog2(p->pmax);
    {
        std::vector<double2> tw(p->pmax / 2);
        for (int k = 0; k < p->pmax / 2; k++) {
            const long double a = 2.0L * acosl(-1.0L) * k / p->pmax;
            tw[k] = make_double2((double)cosl(a), (double)sinl(a));
        }
        if ((rc = dev_upload(&p->d_tw, tw, s))) return rc;
    }
    {
        std::vector<int32_t> bp(nside, 0);
        std::vector<int64_t> bo(nside, 0), fo(nside, 0);
        int64_t nb = 0, nf = 0;
        int maxlen = 2 * nside + 1;  // belt: h + 1
        for (int i = 1; i < nside; i++) {
            const int h = 2 * i;
            if (is_pow2(h)) continue;
            int P = 1;
            while (P < 2 * h - 1) P <<= 1;
            bp[i - 1] = P;
            bo[i - 1] = nb;
            fo[i - 1] = nf;
            nb += h;
            nf += P;
            maxlen = std::max(maxlen, P);
        }
        p->max_fft_len = maxlen;
        if ((rc = dev_upload(&p->d_blu_P, bp, s))) return rc;
        if ((rc = dev_upload(&p->d_blu_boff, bo, s))) return rc;
        if ((rc = dev_upload(&p->d_blu_foff, fo, s))) return rc;
        HIP_TRY(hipMalloc((void **)&p->d_bchirp, sizeof(double2) * std::max<int64_t>(1, nb)));
        HIP_TRY(hipMalloc((void **)&p->d_bfilt, sizeof(double2) * std::max<int64_t>(1, nf)));
        if (nside > 1) {
            const int tl_off = fpad(maxlen) + 1;
            const size_t shm = sizeof(double2) * (size_t)(tl_off + TWL_ENTRIES(p->pmax));
            HIP_TRY(hipFuncSetAttribute((const void *)bluestein_table_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            bluestein_table_kernel<<<nside - 1, 256, shm, s>>>(p->d_blu_P, p->d_blu_boff, p->d_blu_foff,
                                                                p->d_bchirp, p->d_bfilt, p->d_tw, p->pmax, tl_off);
            LAUNCH_CHECK();
        }
    }
    // K5 ring classes
    {
        std::map<int, std::vector<int32_t>> by_len;
        for (int r = 0; r < nring; r++) {
            const int i = r + 1;
            int icap = 0;
            if (i < nside) icap = i;
            else if (i > 3 * nside) icap = 4 * nside - i;
            int P = 0;
            if (icap) {
                const int h = 2 * icap;
                if (!is_pow2(h)) {
                    P = 1;
                    while (P < 2 * h - 1) P <<= 1;
                }
            }
            by_len[P].push_back(r);
        }
        size_t lds_budget = 160 * 1024;
        if (getenv("CORAHIP_K5_LDS_KB")) lds_budget = (size_t)atoi(getenv("CORAHIP_K5_LDS_KB")) * 1024;
        for (auto &kv : by_len) {
            corahip_sht_plan::ring_class c;
            c.P = kv.first;
            c.bstride = fpad(c.P ? c.P : 2 * nside + 1) + 1;
            c.nch = 4;
            const size_t tl_bytes = sizeof(double2) * TWL_ENTRIES(p->pmax);   // LDS twiddle table behind the buffers
            while (c.nch > 1 && (size_t)c.nch * c.bstride * sizeof(double2) + tl_bytes > lds_budget) c.nch >>= 1;
            if ((size_t)c.nch * c.bstride * sizeof(double2) + tl_bytes > 160 * 1024) {
                corahip_set_error("nside %d: ring FFT of length %d does not fit in LDS", nside, c.bstride);
                return CORAHIP_ENOMEM;
            }
            // (measured and rejected for the P = 4096 class: one channel per 4-wave workgroup, two workgroups per CU,
            //  so that LDS and FP64 phases of different workgroups overlap: 12.8 -> 13.6 ms, the cells are read 4x)
            c.count = (int)kv.second.size();
            if ((rc = dev_upload(&c.d_list, kv.second, s))) return rc;
            p->classes.push_back(c);
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    guard.p = nullptr;
    *out = p;
    return 0;
}

int corahip_sht_plan_rings(const corahip_sht_plan *p, int64_t *host_start, int32_t *host_nphi, double *host_z,
                           double *host_phi0) {
    ARG_CHECK(p != nullptr);
    for (int r = 0; r < p->nring; r++) {
        if (host_start) host_start[r] = p->h_start[r];
        if (host_nphi) host_nphi[r] = p->h_nphi[r];
        if (host_z) host_z[r] = p->h_z[r];
        if (host_phi0) host_phi0[r] = p->h_phi0[r];
    }
    return 0;
}

int corahip_sht_lambda(corahip_ctx *ctx, const corahip_sht_plan *p, int m, int ring_pair, double *out) {
    ARG_CHECK(ctx != nullptr && p != nullptr && out != nullptr);
    ARG_CHECK(m >= 0 && m <= p->lmax && ring_pair >= 0 && ring_pair < p->npair);
    lambda_kernel<<<1, 64, 0, ctx->stream>>>(p->lmax, p->npair, m, ring_pair, p->d_z, p->d_coef, p->d_lstart,
                                             p->d_seed, out);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

static inline int nnu_pad_of(int nnu) { return (nnu + 7) & ~7; }
// K5's register prefetch reads K5_MC * K5_THREADS cells from the start of a row without clamping: the last
// row of the F_m buffer needs that much readable memory behind it
#define K5_TAIL_PAD ((size_t)K5_MC * K5_THREADS * 64)

extern "C" int corahip_alm2map_workspace_bytes(const corahip_sht_plan *p, int nnu, size_t *bytes) {
    ARG_CHECK(p != nullptr && bytes != nullptr && nnu >= 1);
    const size_t G = nnu_pad_of(nnu) / 4;
    *bytes = (size_t)p->nring * G * p->L * 8 * sizeof(double) + K5_TAIL_PAD;
    // an odd number of 4-channel groups cannot be consumed in place (K4 tiles are 16 columns
    // = 2 groups wide): the padded copy of the alm block lives in the workspace too
    if (((nnu + 3) / 4) & 1) *bytes += (size_t)p->nalm * G * 8 * sizeof(double);
    return 0;
}

template <int NT, int RT>
static int launch_legendre(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *alm, double *inter) {
    constexpr int STRIDE = 16 * NT + 8;
    const size_t shm = sizeof(double) * LEG_NBUF * (LEG_KT * STRIDE + 2 * (LEG_KT + 8)) + 16;
    HIP_TRY(hipFuncSetAttribute((const void *)legendre_kernel<NT, RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    const int ntile = (p->npair + LEG_RINGS * RT - 1) / (LEG_RINGS * RT);
    const long nitems = (long)p->L * (ncols / (16 * NT)) * ntile;
    // persistent: as many workgroups as fit (LDS-limited: one per CU for NT = 8)
    const int per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / shm)));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipMemsetAsync(p->d_queue, 0, 64, ctx->stream));
    legendre_kernel<NT, RT><<<grid, 64 * LEG_WAVES, shm, ctx->stream>>>(p->lmax, p->npair, p->nring, ncols, p->d_z,
                                                                       p->d_coef, p->d_lstart, p->d_seed, p->d_lmin,
                                                                       alm, p->d_zeros, inter, p->d_queue);
    LAUNCH_CHECK();
    return 0;
}

// K5 over the F_m cells of `inter` for nnu_valid channels -> maps
static int run_ringfft(corahip_ctx *ctx, const corahip_sht_plan *p, const double *inter, int nnu_chunk_pad, int nnu_valid,
                       double *maps) {
    {
        StageTimer t(ctx, "ringfft");
        const int G = nnu_chunk_pad / 4;
        static const bool class_times = getenv("CORAHIP_K5_TIMES") != nullptr;   // diagnostics: per-class ms on stderr
        for (const auto &c : p->classes) {
            hipEvent_t ce0 = nullptr, ce1 = nullptr;
            if (class_times) {
                (void)hipEventCreate(&ce0);
                (void)hipEventCreate(&ce1);
                (void)hipEventRecord(ce0, ctx->stream);
            }
            const size_t shm = sizeof(double2) * ((size_t)c.nch * c.bstride + TWL_ENTRIES(p->pmax));
            const long nitems = (long)c.count * ((nnu_valid + c.nch - 1) / c.nch);
            const int per_cu = std::max<int>(1, (int)((160 * 1024) / std::max<size_t>(shm, 1)));
            const int k5_threads = c.threads ? c.threads : K5_THREADS;
            dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * std::min(per_cu, 4)));
#define RINGFFT_LAUNCH(NCH, BLU)                                                                                     \
    HIP_TRY(hipFuncSetAttribute((const void *)ringfft_kernel<NCH, BLU>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                160 * 1024));                                                                   \
    ringfft_kernel<NCH, BLU><<<grid, k5_threads, shm, ctx->stream>>>(c.d_list, c.count, p->nside, p->lmax, G, nnu_valid, p->npix,     \
                                                         p->d_nphi, p->d_start, p->d_phi0, inter, maps, p->d_tw, \
                                                         p->pmax, p->d_blu_P, p->d_blu_boff, p->d_blu_foff,      \
                                                         p->d_bchirp, p->d_bfilt, c.bstride, p->d_mcut)
            if (c.P == 0) {
                if (c.nch == 4) { RINGFFT_LAUNCH(4, false); }
                else if (c.nch == 2) { RINGFFT_LAUNCH(2, false); }
                else { RINGFFT_LAUNCH(1, false); }
            } else {
                if (c.nch == 4) { RINGFFT_LAUNCH(4, true); }
                else if (c.nch == 2) { RINGFFT_LAUNCH(2, true); }
                else { RINGFFT_LAUNCH(1, true); }
            }
#undef RINGFFT_LAUNCH
            LAUNCH_CHECK();
            if (class_times) {
                float ms = 0.f;
                (void)hipEventRecord(ce1, ctx->stream);
                (void)hipEventSynchronize(ce1);
                (void)hipEventElapsedTime(&ms, ce0, ce1);
                fprintf(stderr, "K5 class P=%d nch=%d rings=%d: %.3f ms\n", c.P, c.nch, c.count, ms);
                (void)hipEventDestroy(ce0);
                (void)hipEventDestroy(ce1);
            }
#if K5_STAMPS
            {
                unsigned long long hs[8];
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                HIP_TRY(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_k5_stamps), sizeof(hs)));
                fprintf(stderr, "K5 class P=%d nch=%d rings=%d: pre %llu zero %llu fold %llu z %llu fft %llu out %llu mid %llu dit %llu\n", c.P,
                        c.nch, c.count, hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
                unsigned long long z8[8] = {0};
                HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_k5_stamps), z8, sizeof(z8)));
            }
#endif
        }
    }
    return 0;
}

// alm_chunk: [nalm][ncols] with ncols = 2*nnu_pad (multiple of 16)
static int alm2map_chunk(corahip_ctx *ctx, const corahip_sht_plan *p, const double *alm_chunk, int nnu_chunk_pad,
                         int nnu_valid, double *maps, double *inter) {
    const int ncols = 2 * nnu_chunk_pad;
    const int ntile = ncols / 16;
    int rc;
    {
        StageTimer t(ctx, "legendre");
        if (ntile % 8 == 0) rc = launch_legendre<8, 1>(ctx, p, ncols, alm_chunk, inter);
        else if (ntile % 4 == 0) rc = launch_legendre<4, 2>(ctx, p, ncols, alm_chunk, inter);
        else if (ntile % 2 == 0) rc = launch_legendre<2, 2>(ctx, p, ncols, alm_chunk, inter);
        else rc = launch_legendre<1, 2>(ctx, p, ncols, alm_chunk, inter);
        if (rc) return rc;
    }
    return run_ringfft(ctx, p, inter, nnu_chunk_pad, nnu_valid, maps);
}

// gather channel groups [g0, g0+Gc) of alm_dev ([nalm][Gsrc][8]) into a dense [nalm][Gc][8] chunk;
// groups past the source (padding) are zero-filled
__global__ void alm_slice_kernel(const double *__restrict__ src, double *__restrict__ dst, long nalm, int Gsrc,
                                 int g0, int Gc) {
    const long n = nalm * Gc * 4;  // double2 items (4 per group cell)
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const long idx = q / (Gc * 4);
        const int w = (int)(q % (Gc * 4));
        const int gsrc = g0 + (w >> 2);
        double2 v = make_double2(0.0, 0.0);
        if (gsrc < Gsrc) v = *reinterpret_cast<const double2 *>(src + ((size_t)idx * Gsrc + g0) * 8 + 2 * w);
        *reinterpret_cast<double2 *>(dst + (size_t)idx * Gc * 8 + 2 * w) = v;
    }
}

extern "C" int corahip_alm2map(corahip_ctx *ctx, const corahip_sht_plan *p, const double *alm_dev, int nnu,
                               double *maps, void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && p != nullptr && alm_dev != nullptr && maps != nullptr && workspace != nullptr);
    ARG_CHECK(nnu >= 1);
    const int nnu_pad4 = (nnu + 3) & ~3;  // alm_dev layout granularity
    const int Gsrc = nnu_pad4 / 4;
    const int nnu_pad8 = nnu_pad_of(nnu);
    size_t need_full;
    corahip_alm2map_workspace_bytes(p, nnu, &need_full);
    if (nnu_pad4 == nnu_pad8 && workspace_bytes >= need_full) {
        // single pass straight from alm_dev
        return alm2map_chunk(ctx, p, alm_dev, nnu_pad8, nnu, maps, (double *)workspace);
    }
    // chunked: workspace holds [inter for chunk][alm slice for chunk]
    const size_t per8_inter = (size_t)p->nring * 2 * p->L * 8 * sizeof(double);
    const size_t per8_alm = (size_t)p->nalm * 16 * sizeof(double);
    int nchunk8 = (int)(workspace_bytes / (per8_inter + per8_alm));
    if (nchunk8 < 1) {
        corahip_set_error("alm2map workspace too small: %zu bytes, need at least %zu", workspace_bytes,
                          per8_inter + per8_alm);
        return CORAHIP_ENOMEM;
    }
    // prefer chunks that are multiples of 64 channels (NT = 8 tiles)
    if (nchunk8 >= 8) nchunk8 &= ~7;
    const int chunk = nchunk8 * 8;
    double *inter = (double *)workspace;
    double *slice = (double *)((char *)workspace + (size_t)nchunk8 * per8_inter);
    for (int nu0 = 0; nu0 < nnu; nu0 += chunk) {
        const int nvalid = std::min(chunk, nnu - nu0);
        const int cpad8 = nnu_pad_of(nvalid);
        const int Gc = cpad8 / 4;
        alm_slice_kernel<<<2048, 256, 0, ctx->stream>>>(alm_dev, slice, p->nalm, Gsrc, nu0 / 4, Gc);
        LAUNCH_CHECK();
        int rc = alm2map_chunk(ctx, p, slice, cpad8, nvalid, maps + (size_t)nu0 * p->npix, inter);
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// analysis host side
// ------------------------------------------------------------------------------------
extern "C" int corahip_map2alm_workspace_bytes(const corahip_sht_plan *p, int nnu, size_t *bytes) {
    ARG_CHECK(p != nullptr && bytes != nullptr && nnu >= 1);
    const size_t G = nnu_pad_of(nnu) / 4;
    const int ntile = (p->npair + 64 * ADJ_WAVES - 1) / (64 * ADJ_WAVES);
    *bytes = (size_t)p->nring * G * p->L * 8 * sizeof(double)            // G_m cells (the `inter` layout)
             + (size_t)ntile * p->nalm * G * 8 * sizeof(double);         // per-ring-tile partial a_lm
    return 0;
}

template <int NCT>
static int launch_legendre_adj(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *inter,
                               double *part) {
    const size_t shm = sizeof(double) * (size_t)ADJ_WAVES * 64 * 33 + 16;
    HIP_TRY(hipFuncSetAttribute((const void *)legendre_adj_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    const int ntile = (p->npair + 64 * ADJ_WAVES - 1) / (64 * ADJ_WAVES);
    const long nitems = (long)p->L * (ncols / (16 * NCT)) * ntile;
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu));
    HIP_TRY(hipMemsetAsync(p->d_queue, 0, 64, ctx->stream));
    legendre_adj_kernel<NCT><<<grid, 512, shm, ctx->stream>>>(p->lmax, p->npair, p->nring, ncols, p->d_z, p->d_coef,
                                                             p->d_lstart, p->d_seed, p->d_lmin, p->d_mcut, inter, part,
                                                             p->d_queue);
    LAUNCH_CHECK();
    return 0;
}

// maps [nnu][npix] RING -> alm_dev [nalm][nnu_pad8/4][2][4]: ONE weighted quadrature pass (no iteration),
//   a_lm = sum_pix w_ring(pix) (4 pi / npix) map(pix) conj(Y_lm(pix)).
// ring_w: device [2 nside] weights of the north rings incl. equator (mirrored to the south), or NULL = 1.
extern "C" int corahip_map2alm(corahip_ctx *ctx, const corahip_sht_plan *p, const double *maps, int nnu,
                               const double *ring_w, double *alm_dev, void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && p != nullptr && maps != nullptr && alm_dev != nullptr && workspace != nullptr);
    ARG_CHECK(nnu >= 1);
    size_t need;
    corahip_map2alm_workspace_bytes(p, nnu, &need);
    if (workspace_bytes < need) {
        corahip_set_error("map2alm workspace too small: %zu bytes, need %zu (process fewer channels per call)",
                          workspace_bytes, need);
        return CORAHIP_ENOMEM;
    }
    const int nnu_pad8 = nnu_pad_of(nnu);
    const int G = nnu_pad8 / 4;
    const int ncols = 2 * nnu_pad8;
    double *inter = (double *)workspace;
    double *part = inter + (size_t)p->nring * G * p->L * 8;
    {
        StageTimer t(ctx, "ringana");
        const int k5_threads = K5_THREADS;
        for (const auto &c : p->classes) {
            const size_t shm = sizeof(double2) * ((size_t)c.nch * c.bstride + TWL_ENTRIES(p->pmax));
            const long nitems = (long)c.count * ((nnu_pad8 + c.nch - 1) / c.nch);
            dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * 4));
#define RINGANA_LAUNCH(NCH)                                                                                      \
    HIP_TRY(hipFuncSetAttribute((const void *)ringana_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                160 * 1024));                                                                    \
    ringana_kernel<NCH><<<grid, k5_threads, shm, ctx->stream>>>(c.d_list, c.count, p->nside, p->lmax, G, nnu_pad8, p->npix,  \
                                                         p->d_nphi, p->d_start, p->d_phi0, maps, inter, p->d_tw,    \
                                                         p->pmax, p->d_blu_P, p->d_blu_boff, p->d_blu_foff,         \
                                                         p->d_bchirp, p->d_bfilt, c.bstride, p->d_mcut, ring_w, nnu)
            if (c.nch == 4) { RINGANA_LAUNCH(4); }
            else if (c.nch == 2) { RINGANA_LAUNCH(2); }
            else { RINGANA_LAUNCH(1); }
#undef RINGANA_LAUNCH
            LAUNCH_CHECK();
        }
    }
    {
        StageTimer t(ctx, "legendre_adj");
        const int nt16 = ncols / 16;
        int rc;
        if (nt16 % 2 == 0) rc = launch_legendre_adj<2>(ctx, p, ncols, inter, part);
        else rc = launch_legendre_adj<1>(ctx, p, ncols, inter, part);
        if (rc) return rc;
        const int ntile = (p->npair + 64 * ADJ_WAVES - 1) / (64 * ADJ_WAVES);
        const long n = p->nalm * (long)ncols;
        alm_reduce_kernel<<<(int)std::min<long>((n + 255) / 256, 8192), 256, 0, ctx->stream>>>(part, n, ntile, alm_dev);
        LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// polarisation (spin-2) synthesis host side
// ------------------------------------------------------------------------------------
// (g1..g4)(l, m) of legendre_pol_kernel at alm_idx(l, m), long double on the host, once per plan
static int ensure_polc(corahip_ctx *ctx, corahip_sht_plan *p) {
    if (p->d_polc) return 0;
    const int lmax = p->lmax;
    std::vector<double> g((size_t)(p->nalm + 64) * 4, 0.0);
    for (int m = 0; m <= lmax; m++)
        for (int l = std::max(m, 2); l <= lmax; l++) {
            const long double ll = l, mm = m;
            const long double n2 = 2.0L / sqrtl((ll + 2.0L) * (ll + 1.0L) * ll * (ll - 1.0L));
            const long double al = l > m ? sqrtl((4.0L * ll * ll - 1.0L) / (ll * ll - mm * mm)) : 0.0L;
            double *o = &g[(size_t)alm_idx(l, m, lmax) * 4];
            o[0] = (double)(-n2 * (ll - mm * mm));
            o[1] = (double)(-n2 * ll * (ll - 1.0L) / 2.0L);
            o[2] = l > m ? (double)(n2 * (2.0L * ll + 1.0L) / al) : 0.0;
            o[3] = (double)(n2 * mm * (ll - 1.0L));
        }
    HIP_TRY(hipMalloc((void **)&p->d_polc, sizeof(double) * g.size()));
    HIP_TRY(hipMemcpyAsync(p->d_polc, g.data(), sizeof(double) * g.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

template <int NT>
static int launch_legendre_pol(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *alm, double *inter) {
    constexpr int STRIDE = 16 * NT + 8;
    const size_t shm = sizeof(double) * 3 * (32 * STRIDE + 6 * (32 + 8)) + 16;
    HIP_TRY(hipFuncSetAttribute((const void *)legendre_pol_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    const int ntile = (p->npair + LEG_RINGS - 1) / LEG_RINGS;
    const long nitems = (long)p->L * (ncols / (16 * NT)) * ntile;
    const int per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / shm)));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipMemsetAsync(p->d_queue, 0, 64, ctx->stream));
    legendre_pol_kernel<NT><<<grid, 64 * LEG_WAVES, shm, ctx->stream>>>(p->lmax, p->npair, p->nring, ncols, p->d_z, p->d_sth,
                                                                       p->d_coef, p->d_polc, p->d_lstart, p->d_seed,
                                                                       p->d_lmin, alm, p->d_zeros, inter, p->d_queue);
    LAUNCH_CHECK();
    return 0;
}

// alm_dev: nnu = 2 nfreq channels interleaved (E_0, B_0, E_1, B_1, ...) -> maps [nnu, npix] = (Q_0, U_0, Q_1, U_1, ...)
extern "C" int corahip_alm2map_spin2(corahip_ctx *ctx, corahip_sht_plan *p, const double *alm_dev, int nnu, double *maps,
                                     void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && p != nullptr && alm_dev != nullptr && maps != nullptr && workspace != nullptr);
    ARG_CHECK(nnu >= 2 && (nnu & 1) == 0);
    ARG_CHECK(((nnu + 3) & ~3) == nnu_pad_of(nnu));     // the alm_dev group count must be the 8-padded one
    size_t need;
    corahip_alm2map_workspace_bytes(p, nnu, &need);
    if (workspace_bytes < need) {
        corahip_set_error("alm2map_spin2 workspace too small: %zu bytes, need %zu (process fewer channels per call)",
                          workspace_bytes, need);
        return CORAHIP_ENOMEM;
    }
    int rc = ensure_polc(ctx, p);
    if (rc) return rc;
    const int nnu_pad8 = nnu_pad_of(nnu);
    const int ncols = 2 * nnu_pad8;
    double *inter = (double *)workspace;
    {
        StageTimer t(ctx, "legendre_pol");
        const int ntile = ncols / 16;
        if (ntile % 4 == 0) rc = launch_legendre_pol<4>(ctx, p, ncols, alm_dev, inter);
        else if (ntile % 2 == 0) rc = launch_legendre_pol<2>(ctx, p, ncols, alm_dev, inter);
        else rc = launch_legendre_pol<1>(ctx, p, ncols, alm_dev, inter);
        if (rc) return rc;
    }
    return run_ringfft(ctx, p, inter, nnu_pad8, nnu, maps);
}
