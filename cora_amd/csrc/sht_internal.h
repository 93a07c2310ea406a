// sht_internal.h - shared by the translation units of the HEALPix transforms (sht_*.hip):
//   sht_plan.hip      geometry / recurrence / seed / Bluestein tables of a (nside, lmax) plan
//   sht_legendre.hip  K4: Legendre contraction on FP64 MFMA (scalar and spin-2 forms)
//   sht_ringfft.hip   K5 and K5^T: per-ring phase / fold / FFT (synthesis) and FFT / split (analysis)
//   sht_analysis.hip  K4^T: adjoint Legendre contraction + reduction over ring tiles
//   sht_api.hip       C ABI entry points (alm2map, map2alm, alm2map_spin2, workspace sizes)
//
// Replaces hputil.sphtrans_inv_sky -> healpy.alm2map (cora/util/hputil.py:369-391,500-531) and, for the
// "next" rows, healpy.map2alm and the polarised synthesis.  Definition implemented: SURVEY.md Appendix A
// (HEALPix software conventions).
//
// Synthesis = two kernels per pass over a chunk of channels:
//   K4 legendre_kernel : F_m(ring) = sum_l a_lm lambda_lm(cos theta_ring) for every ring pair,
//        as FP64 MFMA (v_mfma_f64_16x16x4_f64): A = lambda (rows = rings, generated in
//        registers by the three-term recurrence), B = a_lm (LDS-staged rows of the
//        [nalm][cols] device layout), even/odd (l-m) accumulated separately so the
//        north ring gets e+o and its southern mirror e-o.
//   K5 ringfft_kernel  : per ring and channel: phase e^{i m phi0}, alias fold onto nphi
//        bins, complex-to-real FFT of length nphi (radix-16/8/4/2 in LDS; Bluestein for the
//        cap rings whose length 4i is not a power of two), pixel store.
// Intermediate F_m layout: inter[ring][g][m][c*4+v] (g = channel/4, v = channel%4,
// c = re/im): 64-byte cells, contiguous in m for K5, 64-byte segments for K4's stores.
#pragma once
#include "common.h"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdlib>
#include <map>

// ------------------------------------------------------------------------------------
struct corahip_sht_plan {
    int nside = 0, lmax = 0, L = 0, npair = 0, nring = 0;
    int cut_exp = 0;                                      // terms of the Legendre sums with |lambda_lm| < 2^cut_exp are dropped
    std::map<int, uint64_t> k4_macro_steps;               // by RT: macro-steps legendre_kernel executes per column group (lazy)
    std::map<std::pair<int, int>, std::pair<int2 *, int>> k4_items;   // by (RT, column groups): legendre_kernel's list of non-empty items (lazy, sht_k4_items)
    std::vector<int32_t> h_lmin;                          // host copy of d_lmin (lazy)
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
    // the scalar synthesis kernel runs the recurrence in its two-instruction form mu_l = (alpha_l x) mu_{l-1} - mu_{l-2},
    // lambda_l = s_l mu_l (s_m = s_{m+1} = 1, s_l = B_l s_{l-2}, alpha_l = A_l s_{l-1} / s_l): its own coefficient and
    // seed tables (the spin-2 and the analysis kernels keep the (A, B) form)
    double2 *d_coefmu = nullptr;                          // [nalm]: (alpha_l, s_l) at alm_idx(l,m)
    double2 *d_seedmu = nullptr;                          // [L][npair]: (mu_{lstart-1}, mu_{lstart})
    double2 *d_seed4 = nullptr;                           // [L][npair][4]: (mu_{R-2}, mu_{R-1}) in front of lane group kq's entry row R (seed_kernel)
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
    // second set of Bluestein filters for the rings whose 2 h - 1 also fits 3/4 of the power-of-two length (P3 = 1536 or
    // 3072): used by the compile-time synthesis kernels only; everything else (analysis, run-time kernel) keeps blu_P
    std::vector<int32_t> h_blu3_P;                        // [nside]: 0 = none
    int64_t *d_blu3_foff = nullptr;
    double2 *d_bfilt3 = nullptr;
    // fold phases of the cap rings for the compile-time Bluestein kernels: e^{i t phi0(ring)}, t < 512, and the steps
    // e^{i 256 phi0}, e^{i 512 phi0}, by north-cap ring number i - 1
    double2 *d_foldph = nullptr, *d_foldstep = nullptr;
    int max_fft_len = 0;                                  // largest LDS FFT buffer (complex elems)
    // K5 launch classes: rings grouped by transform kind/length so each launch sizes its LDS
    struct ring_class {
        int P = 0;        // Bluestein length, 0 = direct power-of-two transform
        int N = 0;        // direct classes: the half length h of the rings (all equal), 0 = mixed (run-time length)
        int P3 = 0;       // Bluestein length the compile-time kernels take for the class, filters in d_bfilt3 (0 = none):
                          // 3 * 2^k (1536, 3072, 6144) where it holds 2 h - 1, or P = 8192 in the compile-time pass order
        int nch = 4;      // channels transformed together per workgroup
        int threads = 0;  // workgroup size (0: K5_THREADS)
        int bstride = 0;  // complex elements per channel buffer in LDS
        int count = 0;
        int32_t *d_list = nullptr;
        std::vector<int32_t> h_list;   // host copy of the ring list (corahip_sht_plan_ring_classes)
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
// K4: Legendre contraction on FP64 MFMA
// ------------------------------------------------------------------------------------
#ifndef LEG_ABLATE
#define LEG_ABLATE 0  // diagnostic builds only (make ablate): 1 no MFMA, 2 no recurrence, 3 no B reads, 4 no epilogue stores
#endif
#ifndef LEG_KT
#define LEG_KT 56     // rows of l per LDS stage (re-tuned after the instruction-count pass: 48 x 3 68.5, 56 x 2 67.7, 48 x 2 68.9, 40 x 3 72.1, 32 x 4 69.8 ms)
#endif
#ifndef LEG_WAVES
#define LEG_WAVES 8
#endif
#define LEG_RINGS (16 * LEG_WAVES)  // ring pairs per workgroup (x RT)
#define LEG_RPM (8 / LEG_WAVES)     // a_lm pieces each wave issues per macro-step (= RPW / (LEG_KT / 8))
#define LMIN_RINGS 128              // granularity of the plan's per-(m, ring block) first-l table
#define ADJ_WAVES 8                 // waves per workgroup of the analysis kernel
#ifndef LEG_NBUF
#define LEG_NBUF 2     // LDS stage ring: one being read + one in flight
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
// same, with the source as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: the row address of an
// a_lm piece is then pure scalar arithmetic (every instruction of K4's macro-step costs issue time)
__device__ static inline void glds16_s(const void *sbase, unsigned lane_byte_off, unsigned lds_byte_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    // (dropping the save / restore of M0 was measured: no gain)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(sbase), "s"(dst)
                 : "memory");
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

// LDS layout of an FFT buffer: logical element i lives at fpad(i); fpad_len(n) slots hold n elements.
// K5_SWZ = 0 (shipped): one spare slot per 8 elements plus 8 per 128.  A simulation of ds_read_b128 /
// ds_write_b128 with their real lane groups (MI355X_MICROARCH.md LDS table: reads are served in 4 groups of 16
// NON-contiguous lanes, writes in 8 x 8) gives this padding 1.5x the conflict-free LDS cycles on every
// radix-16/8/4/2 pass of N = 1024 ... 4096 - the 0.31 = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measured for all
// three instantiations (profiles/r01_k5_lds_pmc.csv): a padded unit-stride run is no longer aligned to the bank
// rows the lane groups assume.
// K5_SWZ = 1: an XOR swizzle of the 16-byte slot inside its 256-byte bank row by parities of the row index
// x = i >> 4 (bit0 = x0^x5, bit1 = x0^x1, bit2 = x0^x2^x4, bit3 = x0^x3^x4), found by hill-climbing in that
// simulation: every pass conflict-free.  Measured: the counter ratio falls to 0.21 / 0.17 / 0.07 (belt / Bluestein
// classes; the rest is the fold, cell and Hermitian stages), the kernel gets 0.7 ms SLOWER (26.9 -> 27.6 ms, A/B on
// one box): the ring FFT is bound by its FP64 VALU work, not by the LDS array, and the swizzle costs four more
// integer operations per element than the padding.  Kept as a switch; the flat-sky line FFT (flatsky.hip), whose
// radix-4 stages ARE LDS-bound, ships its own swizzle.
#ifndef K5_SWZ
#define K5_SWZ 0
#endif
#if K5_SWZ
__host__ __device__ static inline int fpad(int i) {
    const int x = i >> 4;
    int s = (0 - (x & 1)) & 15;        // x0 -> all four bits
    s ^= (x & 14);                     // x1, x2, x3 -> bits 1, 2, 3
    s ^= (0 - ((x >> 4) & 1)) & 12;    // x4 -> bits 2, 3
    s ^= (x >> 5) & 1;                 // x5 -> bit 0
    return i ^ s;
}
__host__ __device__ static inline int fpad_len(int n) { return (n + 15) & ~15; }
#define K5_CH_SKEW 4                   // channel buffers start a quarter bank row apart
#else
__host__ __device__ static inline int fpad(int i) { return i + (i >> 3) + ((i >> 7) << 3); }
__host__ __device__ static inline int fpad_len(int n) { return fpad(n); }
#define K5_CH_SKEW 1
#endif

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
#if K5_TW_4X4
            if (R == 16) {
                // w^r = (w^4)^a w^c for r = 4a + c: 5 multiplies for (w^2, w^3, w^4, w^8, w^12) + 24 to apply them,
                // instead of 3 squarings + 32 selected binary powers (6 complex multiplies fewer per butterfly)
                const double2 w1 = wp[0], w2 = wp[1], w4 = wp[2], w8 = wp[3];
                const double2 w3 = cmul(w2, w1), w12 = cmul(w8, w4);
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    x[(4 * a + 1) % R] = cmul(x[(4 * a + 1) % R], w1);
                    x[(4 * a + 2) % R] = cmul(x[(4 * a + 2) % R], w2);
                    x[(4 * a + 3) % R] = cmul(x[(4 * a + 3) % R], w3);
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    x[(4 + c) % R] = cmul(x[(4 + c) % R], w4);
                    x[(8 + c) % R] = cmul(x[(8 + c) % R], w8);
                    x[(12 + c) % R] = cmul(x[(12 + c) % R], w12);
                }
                return;
            }
#endif
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

#ifndef K5_TW_4X4
#define K5_TW_4X4 0   // 1: radix-16 twiddles as (w^4)^a w^c, 29 complex multiplies per butterfly instead of 35 - measured: no difference (the passes are latency-bound, two waves per SIMD)
#endif
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
#ifndef K5_MC_BLU4
#define K5_MC_BLU4 2   // prefetched cells of the four-channel Bluestein instantiation (2 spills 11 VGPRs)
#endif
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
static inline int nnu_pad_of(int nnu) { return (nnu + 7) & ~7; }
// K5's register prefetch reads K5_MC * K5_THREADS cells from the start of a row without clamping: the last
// row of the F_m buffer needs that much readable memory behind it
// (the compile-time kernels of sht_ringfft_ct.hip prefetch up to 8 x 512 cells)
#define K5_TAIL_PAD ((size_t)8 * 512 * 64)

// ---- cross-translation-unit host functions -------------------------------------------------
// K4: alm [nalm][ncols] -> F_m cells; the launch shape is chosen from the number of 16-column tiles
int sht_legendre(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *alm, double *inter);
// legendre_kernel's work list for (RT ring row tiles per wave, ncg column groups): the (m, column group, ring tile) items with at
// least one contributing l, m-major, as (m | rtile << 15 | cg << 23, first contributing l); cached in the plan
int sht_k4_items(corahip_ctx *ctx, const corahip_sht_plan *p, int rt, int ncg, const int2 **items, int *nitems);
// K4 spin-2: interleaved (E, B) channels -> (Q, U) cells (builds the plan's g table on first use)
int sht_legendre_pol(corahip_ctx *ctx, corahip_sht_plan *p, int ncols, const double *alm, double *inter);
// K5: F_m cells -> maps for nnu_valid channels (nnu_chunk_pad = channels in the cell layout)
int sht_ringfft(corahip_ctx *ctx, const corahip_sht_plan *p, const double *inter, int nnu_chunk_pad, int nnu_valid,
                double *maps);
// K5 with compile-time transform shapes (sht_ringfft_ct.hip): *took = it launched class c (false: the generic kernel
// has to take it); the return value is the error status only (0, CORAHIP_E* or a hipError_t)
int sht_ringfft_ct(corahip_ctx *ctx, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &c, const double *inter,
                   int G, int nnu, double *maps, bool *took);
// plan time: filters of the 3 * 2^k Bluestein lengths (fills p->d_bfilt3; h_blu3_P / d_blu3_foff / d_bchirp must be set)
int sht_blu3_tables(corahip_ctx *ctx, corahip_sht_plan *p, int64_t total);
// creates ctx->stream2 and the fork / join events on first use
int sht_second_stream(corahip_ctx *ctx);
// belt + largest Bluestein class side by side on two streams (sht_ringfft_ct.hip): *took = both launched
int sht_ringfft_ct_pair(corahip_ctx *ctx, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &belt,
                        const corahip_sht_plan::ring_class &cap, const double *inter, int G, int nnu, double *maps, bool *took);
// K5^T: maps -> weighted G_m cells for nnu_pad8 channels (nnu present in `maps`)
int sht_ringana_ct(corahip_ctx *ctx, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &c, const double *maps, int nvalid,
                   int nnu_pad, const double *ring_w, int G, double *inter, bool *took);
int sht_ringana(corahip_ctx *ctx, const corahip_sht_plan *p, const double *maps, int nnu, int nnu_pad8,
                const double *ring_w, double *inter);
// K4^T + reduction over ring tiles: G_m cells -> alm_dev (part: per-ring-tile scratch)
int sht_legendre_adj(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *inter, double *part,
                     double *alm_dev);
