// sht_ringfft_ct.hip - K5 for the ring classes that carry the BASELINE configurations, with everything about the
// transform fixed at compile time (length, pass radices, channels per workgroup):
//   ringfft_direct_ct<N, NCH>  rings whose half length h = N is a power of two (the belt: N = 2 nside)
//   ringfft_blu_ct<P, NCH>     cap rings through a Bluestein convolution of length P (power of two or 3 * 2^k)
//   ringana_direct_ct / ringana_blu_ct   K5^T, the analysis direction (map2alm): the same passes run backwards (round 5)
// The generic kernel of sht_ringfft.hip (run-time lengths) stays for every other class and is the reference
// these are tested against (tests/test_gpu_fullsize.py compares both with the oracle pixel by pixel).
//
// What the fixed shapes buy, measured on the generic kernel's anatomy:
//   * every LDS address of a butterfly is one per-thread base + immediate offsets: fpad(i0 + r q) = fpad(i0) +
//     fpad(r q) whenever the bits of i0 and r q do not overlap, which the pass structure guarantees;
//   * the twiddle a thread needs in a pass depends on tid only (j = tid mod q), not on the ring: the two
//     twiddles of a three-pass transform live in registers for the whole persistent loop - no table load
//     (L2 latency, and in-order vmcnt behind the next item's prefetch) inside any pass;
//   * direct class: the Hermitian -> half-length-complex step is fused into the first pass (the butterfly reads
//     its 16 inputs and their 16 mirror partners, one barrier, then writes) and the last pass stores pixels
//     straight from registers (128-byte segments): four LDS write passes per item instead of five, no
//     bank-conflicted digit-reversed read pass;
//   * Bluestein class: chirp (w^k of the pre-pass is the square of the fold phase: no table, no sincospi), filter and
//     output-chirp values are requested ahead of their use; the first forward pass reads only the non-zero half
//     of the padded input and the last inverse pass forms only the h outputs that exist (so the upper half of
//     the buffer is never zero-filled); last forward pass + filter + first inverse pass stay in registers.
// LDS stores are the expensive operation of these kernels (ds_write_b128: 13 cycles per wave-instruction against
// 4 for ds_read_b128, MI355X_MICROARCH.md section LDS), hence the count of write passes above.
#include "sht_internal.h"

static_assert(K5_SWZ == 0, "the compile-time kernels assume the padded LDS layout");
#define CT_T 512
#ifndef CT_STAMPS
#define CT_STAMPS 0   // diagnostic build (make k5ctstamps): s_memtime per phase of the Bluestein kernel, summed over waves
#endif
#if CT_STAMPS
__device__ unsigned long long g_ct_stamps[12];
#define CTSTAMP(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); ct_acc[k] += _t - ct_last; ct_last = _t; }
#else
#define CTSTAMP(k)
#endif

#include "fft_ct.h"

// ---- register prefetch of the F_m cells of one (ring, NCH channels) item ---------------------------------------
template <int NCH>
struct cell_ct {
    double re[NCH], im[NCH];
};
template <int NCH>
__device__ __forceinline__ static cell_ct<NCH> load_cell_ct(const double *cell, unsigned m) {
    cell_ct<NCH> c;
    if (NCH == 4) {
        const double4 a = *reinterpret_cast<const double4 *>(cell + m * 8u);
        const double4 b = *reinterpret_cast<const double4 *>(cell + m * 8u + 4u);
        c.re[0] = a.x; c.re[1 % NCH] = a.y; c.re[2 % NCH] = a.z; c.re[3 % NCH] = a.w;
        c.im[0] = b.x; c.im[1 % NCH] = b.y; c.im[2 % NCH] = b.z; c.im[3 % NCH] = b.w;
    } else if (NCH == 2) {
        const double2 a = *reinterpret_cast<const double2 *>(cell + m * 8u);
        const double2 b = *reinterpret_cast<const double2 *>(cell + m * 8u + 4u);
        c.re[0] = a.x; c.re[1 % NCH] = a.y;
        c.im[0] = b.x; c.im[1 % NCH] = b.y;
    } else {
        c.re[0] = cell[m * 8u];
        c.im[0] = cell[m * 8u + 4u];
    }
    return c;
}

// phase e^{i m phi0} and fold of one cell onto the bins 0..h of the Hermitian spectrum (see ringfft_kernel)
template <int PK, int NCH, int BS>
__device__ __forceinline__ static void fold_cell(double2 *sm, int m, int n, int h, bool noalias, const double2 ph,
                                                 const cell_ct<NCH> &cv) {
    double *smd = reinterpret_cast<double *>(sm);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const double2 val = cmul(make_double2(cv.re[c], cv.im[c]), ph);
        double *bd = smd + (size_t)c * BS * 2;
        if (noalias) {
            if (m == 0) *reinterpret_cast<double2 *>(bd) = make_double2(val.x, 0.0);             // Re(c_0) only
            else if (m < h) *reinterpret_cast<double2 *>(bd + 2 * fpad(m)) = val;
            else *reinterpret_cast<double2 *>(bd + 2 * fpad(h)) = make_double2(2.0 * val.x, 0.0);  // m == h
        } else {
            const int k = m % n;
            const int kc = (n - k) % n;
            if (m == 0) atomicAdd(&bd[0], val.x);
            else {
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
    }
}

// shared front end of both kernels: zero / fold the ring spectrum of the current item from the prefetched
// registers (cells m = tid + k CT_T, k < MC) and, behind it, cells read in place; then prefetch the next item.
// Ends with a barrier.  X_k, k = 0..h, is then at sm[c BS + fpad(k)].
// TAIL: one more register slot for the cell m = tid + MC CT_T, index clamped to the last cell of the row (the belt
// has lmax + 1 = MC CT_T + 1 cells: read in place, that one cell exposed a whole memory latency per item).
template <int PK, int NCH, int BS, int MC, bool TAIL, int T, int KPN = 4>
struct FrontEnd {
    cell_ct<NCH> pf[MC + (TAIL ? 1 : 0)];
    // PART 0 / 1: first / second half of the cells (the requests of one item are spread over two phases: a burst of all
    // of them at once blocks every wave of the workgroup at the 64 B/clk of the vector memory path); -1: all
    template <int PART = -1>
    __device__ __forceinline__ void prefetch(const double *cell, int L, int tid) {
        constexpr int K0 = PART == 1 ? (MC + 1) / 2 : 0, K1 = PART == 0 ? (MC + 1) / 2 : MC;
#pragma unroll
        for (int k = K0; k < K1; k++) pf[k] = load_cell_ct<NCH>(cell + (size_t)k * T * 8, tid);
        if (TAIL && PART != 0) pf[MC] = load_cell_ct<NCH>(cell, (unsigned)min(tid + MC * T, L - 1));
    }
    // make the compiler wait for the prefetch HERE.  vmcnt completes in order and the pixel stores are conditional, so
    // the compiler cannot count them: a wait for a prefetch register placed after the stores becomes vmcnt(0) and
    // sits out the store acknowledgements of the whole previous item (measured: 7k cycles per item)
    __device__ __forceinline__ void touch() const {
#pragma unroll
        for (int k = 0; k < MC + (TAIL ? 1 : 0); k++)
#pragma unroll
            for (int c = 0; c < NCH; c++) asm volatile("" ::"v"(pf[k].re[c]), "v"(pf[k].im[c]));
    }
    static constexpr int KP = KPN;
    double2 phk[KP];    // e^{i m phi0} at m = tid + k T, k < KP (the Bluestein pre-pass derives e^{i pi k / h} from them)
    // fold with the phase computed here (two sincospi per thread and item: ~150 DP instructions)
    __device__ __forceinline__ void fold(double2 *sm, const double *__restrict__ cell, int Lr, int n, double phi0_over_pi, const int tid) {
        double s, c;
        // (sincospi reduces its argument exactly; an fmod in front of it - the generic kernel has one - is a
        //  slow library loop: the fold took 6.7k cycles per item with it)
        sincospi((double)tid * phi0_over_pi, &s, &c);
        const double2 ph0 = make_double2(c, s);
        sincospi((double)T * phi0_over_pi, &s, &c);
        fold(sm, cell, Lr, n, ph0, make_double2(c, s), tid);
    }
    // fold with ph0 = e^{i tid phi0} and phstep = e^{i T phi0} given: a per-thread constant for the belt (phi0 takes two
    // values there), a plan table for the cap rings (sht_plan.hip: fold_phase_kernel evaluates the same expression)
    __device__ __forceinline__ void fold(double2 *sm, const double *__restrict__ cell, int Lr, int n, const double2 ph0, const double2 phstep, const int tid) {
        const int h = n >> 1;
        const bool noalias = Lr - 1 <= h;
        if (noalias) {
            for (int j = Lr + tid; j <= h; j += T)
#pragma unroll
                for (int c = 0; c < NCH; c++) sm[c * BS + fpad(j)] = make_double2(0.0, 0.0);
        } else {
            for (int j = tid; j <= h; j += T)
#pragma unroll
                for (int c = 0; c < NCH; c++) sm[c * BS + fpad(j)] = make_double2(0.0, 0.0);
            __syncthreads();
        }
        double2 ph = ph0;
        phk[0] = ph;
#pragma unroll
        for (int k = 1; k < KP; k++) phk[k] = cmul(phk[k - 1], phstep);
#pragma unroll
        for (int k = 0; k < MC; k++) {
            if (tid + k * T < Lr) fold_cell<PK, NCH, BS>(sm, tid + k * T, n, h, noalias, ph, pf[k]);
            ph = cmul(ph, phstep);
        }
        if (TAIL) {
            if (tid + MC * T < Lr) fold_cell<PK, NCH, BS>(sm, tid + MC * T, n, h, noalias, ph, pf[MC]);
            ph = cmul(ph, phstep);
        }
        for (int m = tid + (MC + (TAIL ? 1 : 0)) * T; m < Lr; m += T) {   // cells beyond the prefetch window, read in place
            fold_cell<PK, NCH, BS>(sm, m, n, h, noalias, ph, load_cell_ct<NCH>(cell, m));
            ph = cmul(ph, phstep);
        }
    }
};

// blockIdx -> item with the workgroups of one XCD taking the items that share 64-byte cells (K5_XCD_PAIR of
// sht_ringfft.hip)
template <int NCH>
__device__ __forceinline__ static int ct_remap(int v, int nitems) {
    constexpr int SH = NCH == 2 ? 1 : (NCH == 1 ? 2 : 0);
    const bool on = SH > 0 && (nitems & ((8 << SH) - 1)) == 0 && (gridDim.x & ((8 << SH) - 1)) == 0;
    if (!on) return v;
    const int slot = v >> 3, xcd = v & 7;
    return (((slot >> SH) * 8 + xcd) << SH) + (slot & ((1 << SH) - 1));
}

// ------------------------------------------------------------------------------------
// direct class: h = N = 2^k (N >= 2048 so that the first-pass stride is a multiple of the 128-element padding period)
// ------------------------------------------------------------------------------------
template <int N, int NCH, int MC, int T>
__global__ void __launch_bounds__(T)
ringfft_direct_ct(const int32_t *__restrict__ ring_list, int nlist, int lmax, int G, int nnu, long npix,
                  const int64_t *__restrict__ start_a, const double *__restrict__ phi0_a,
                  const double *inter, double *maps, const int32_t *__restrict__ mcut) {   // (not __restrict__: see ringfft_blu_ct)
    constexpr int PK = K5_PK_DIRECT;
    constexpr int R0 = Sch<N>::R0, R1 = Sch<N>::R1, R2 = Sch<N>::R2;
    static_assert(R0 == 16 && R1 == 16, "digit map of the fused store assumes 16 x 16 x R2");
    constexpr int Q0 = N / R0;              // stride of the first pass
    static_assert(Q0 % 128 == 0, "first-pass stride must be a multiple of the padding period");
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    constexpr int n = 2 * N, h = N;
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const int L = lmax + 1;
    const int ngrp = (nnu + NCH - 1) / NCH;
    const int nitems = nlist * ngrp;

    // per-thread twiddles, fixed for the whole kernel
    double2 wH, wA, wB;     // e^{i pi j0 / N}, e^{2 pi i j0 / N}, e^{2 pi i j1 / (N / R0)}
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi((double)j0 / (double)N, &s, &c);
        wH = make_double2(c, s);
        sincospi(2.0 * (double)j0 / (double)N, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    auto cell_ptr = [&](int item) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        return inter + ((size_t)ring * G + (ch0 >> 2)) * L * 8 + (ch0 & 3);
    };
    FrontEnd<PK, NCH, BS, MC, true, T> fe;
    // fold phases e^{i m phi0}: phi0 of a belt ring is 0 or pi / (4 nside) = pi / (2 N) (sht_plan.hip), so e^{i tid phi0}
    // and the step e^{i T phi0} are per-thread constants of the kernel (the same expressions FrontEnd::fold evaluates
    // per item - two sincospi, ~150 DP instructions per thread and item, 7 % of the belt's cycles)
    double2 phS, phstepS;
    {
        const double phs = (M_PI / (2.0 * N)) / M_PI;
        double s, c;
        sincospi((double)tid0 * phs, &s, &c);
        phS = make_double2(c, s);
        sincospi((double)T * phs, &s, &c);
        phstepS = make_double2(c, s);
    }
    int vitem = blockIdx.x;
    if (vitem < nitems) fe.prefetch(cell_ptr(ct_remap<NCH>(vitem, nitems)), L, tid0);
    fe.touch();   // (so that the prefetch is known to be complete on BOTH edges into the loop: no wait in the fold)
    for (; vitem < nitems; vitem += gridDim.x) {
        // the thread index is made opaque once per item: otherwise every LDS / pixel address of every pass (all of them
        // functions of tid only) is hoisted out of this loop as a loop invariant, ~60 registers of them spill, and a
        // scratch reload - a vmcnt-ordered load - in the store pass waits for the whole prefetch of the next item
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int item = ct_remap<NCH>(vitem, nitems);
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const long start = start_a[ring];
        const bool shifted = phi0_a[ring] != 0.0;         // (uniform) the ring starts at pi / (4 nside)
        const int Lr = mcut[ring];
        __syncthreads();                                  // previous item's LDS reads are done
        {
            double2 p0 = shifted ? phS : make_double2(1.0, 0.0), p1 = shifted ? phstepS : make_double2(1.0, 0.0);
            asm volatile("" : "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y));   // (not loop invariants: see tw_apply)
            fe.fold(sm, cell_ptr(item), Lr, n, p0, p1, tid);
        }
        // the next item's cells: two passes (~7k cycles) ahead of the store pass, in front of which they are waited for
        // (unconditional - the last iteration re-reads an item - so that the compiler can COUNT these loads in its
        //  vmcnt waits; behind an `if` it assumes they may be absent and waits for everything instead)
        const double *ncell = cell_ptr(ct_remap<NCH>(min(vitem + (int)gridDim.x, nitems - 1), nitems));
        fe.template prefetch<0>(ncell, L, tid);
        __syncthreads();
        // ---- pass 1 with the Hermitian step: butterfly j0 of channel ch reads X_k, k = j0 + r Q0, and the mirror
        //      partners X_{h-k} = element (Q0 - j0) + (R0 - 1 - r) Q0; Z_k = (X_k + conj X_{h-k}) + i w^k (X_k - conj X_{h-k}),
        //      w^k = e^{i pi j0 / N} e^{i pi r / R0}
        {
            constexpr int TOT = NCH * Q0;
            constexpr int IT = (TOT + T - 1) / T;
            double2 x[IT][R0];
            double2 wh = wH;
            asm volatile("" : "+v"(wh.x), "+v"(wh.y));   // (keeps the 16 products below out of the loop-invariant set: see tw_apply)
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                const double2 *pa = sm + ch * BS + fpad(j0);
                const double2 *pb = sm + ch * BS + fpad(Q0 - j0);
#pragma unroll
                for (int r = 0; r < R0; r++) {
                    const double2 xa = pa[fpc(r * Q0)];
                    const double2 xb = pb[fpc((R0 - 1 - r) * Q0)];
                    const double2 w = cmul(wh, make_double2(kCos16[r], kSin16[r]));
                    const double2 sum = make_double2(xa.x + xb.x, xa.y - xb.y);
                    const double2 dif = make_double2(xa.x - xb.x, xa.y + xb.y);
                    const double2 t = cmul(dif, w);
                    x[it][r] = make_double2(sum.x - t.y, sum.y + t.x);
                }
            }
            __syncthreads();                              // every raw X has been read
            fe.template prefetch<1>(ncell, L, tid);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                double2 *pa = sm + ch * BS + fpad(j0);
                DftR<R0, 1>::run(x[it]);
                tw_apply<R0>(x[it], wA);
#pragma unroll
                for (int r = 0; r < R0; r++) pa[fpc(r * Q0)] = x[it][r];
            }
        }
        __syncthreads();
        ct_pass<PK, N, NCH, BS, Q0, R1, 1, false, T>(sm, wB, tid);
        __syncthreads();
        // ---- last pass (radix R2 on contiguous elements, no twiddles) with the pixel store: butterfly t = 16 k0 + k1
        //      holds the natural indices k0 + 16 k1 + 256 r.  Lane bits: 0-2 = k0 low, 3-5 = k1 low, 6 = k0 high,
        //      7 = k1 high: eight consecutive lanes store 128 contiguous bytes; the LDS reads are 2-way conflicted.
        {
            constexpr int TOT = NCH * 256;
            constexpr int IT = (TOT + T - 1) / T;
            fe.touch();
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx >> 8;
                const int k0 = (idx & 7) | ((idx >> 3) & 8);
                const int k1 = ((idx >> 3) & 7) | ((idx >> 4) & 8);
                const double2 *p = sm + ch * BS + fpad((k0 * 16 + k1) * R2);
                double2 x[R2];
#pragma unroll
                for (int r = 0; r < R2; r++) x[r] = p[fpc(r)];
                DftR<R2, 1>::run(x);
                if (ch0 + ch < nnu) {
                    double *out = maps + (size_t)(ch0 + ch) * npix + start + 2 * (k0 + 16 * k1);
#pragma unroll
                    for (int r = 0; r < R2; r++) *reinterpret_cast<double2 *>(out + 512 * r) = x[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Bluestein class: cap rings, h = 2 i not a power of two, convolution length P >= 2 h - 1
// ------------------------------------------------------------------------------------
template <int P, int NCH, int MC, int T>
__global__ void __launch_bounds__(T)
ringfft_blu_ct(const int32_t *__restrict__ ring_list, int nlist, int nside, int lmax, int G, int nnu, long npix,
               const int32_t *__restrict__ nphi_a, const int64_t *__restrict__ start_a,
               const double *__restrict__ phi0_a, const double *inter, double *maps,
               const int64_t *__restrict__ boff, const int64_t *__restrict__ foff, const double2 *chirp,
               const double2 *filt, const int32_t *__restrict__ mcut, const double2 *foldph, const double2 *foldstep) {
    // (inter, maps, chirp, filt are deliberately NOT __restrict__: the compiler then may not move their loads across the
    //  barriers / pixel stores, and the places where this kernel requests them - one phase ahead of their use, and all
    //  of them completed before the first store - are the places where they are issued; with __restrict__ the chirp
    //  loads of the last pass were sunk to their use and waited out in full)
    constexpr int PK = K5_PK_BLU;
    constexpr int R0 = Sch<P>::R0, R1 = Sch<P>::R1, R2 = Sch<P>::R2;
    constexpr int Q0 = P / R0;
    constexpr int BS = fpc(P) + K5_CH_SKEW;
    constexpr int HALF = (R0 / 2) * Q0;          // the non-zero half of the padded input: h <= HALF
    // pre-pass iterations over the pairs k <= h / 2: h <= HALF, and for a power-of-two P even h <= HALF - 2 (h = HALF
    // is itself a power of two: direct class)
    constexpr int U = (HALF / 2 + ((P & (P - 1)) ? 1 : 0) + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const int L = lmax + 1;
    const int ngrp = (nnu + NCH - 1) / NCH;
    const int nitems = nlist * ngrp;
    const double invP = 1.0 / (double)P;

    double2 wA, wB;     // e^{2 pi i j0 / P}, e^{2 pi i j1 / (P / R0)}
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)P, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    auto cell_ptr = [&](int item) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        return inter + ((size_t)ring * G + (ch0 >> 2)) * L * 8 + (ch0 & 3);
    };
    FrontEnd<PK, NCH, BS, MC, false, T, (U > 4 ? U : 4)> fe;
    double2 cbn[U];    // chirp b_k, k = tid + u T, of the NEXT item (index clamped: unused lanes load a valid slot)
    auto load_cbn = [&](int item, int t) {
        const int ring = ring_list[item / ngrp];
        const int ic = ring + 1 < nside ? ring + 1 : 4 * nside - (ring + 1);
        const double2 *b = chirp + boff[ic - 1];
        const int hh = nphi_a[ring] >> 1;
#pragma unroll
        for (int u = 0; u < U; u++) cbn[u] = b[min(t + u * T, hh - 1)];
    };
    int vitem = blockIdx.x;
    if (vitem < nitems) {
        fe.prefetch(cell_ptr(ct_remap<NCH>(vitem, nitems)), L, tid0);
        load_cbn(ct_remap<NCH>(vitem, nitems), tid0);
    }
    fe.touch();   // complete on both edges into the loop (FrontEnd::touch)
#pragma unroll
    for (int u = 0; u < U; u++) asm volatile("" ::"v"(cbn[u].x), "v"(cbn[u].y));
#if CT_STAMPS
    unsigned long long ct_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ct_last;
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); ct_last = _t; }
#endif
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;                                   // opaque per item: see ringfft_direct_ct
        asm volatile("" : "+v"(tid));
        const int item = ct_remap<NCH>(vitem, nitems);
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const int n = nphi_a[ring];
        const int h = n >> 1;
        const long start = start_a[ring];
        const int icap = ring + 1 < nside ? ring + 1 : 4 * nside - (ring + 1);
        const int Lr = mcut[ring];
        const double2 *bch = chirp + boff[icap - 1];
        // filter values of the register-fused middle (storage positions t R2 + r; the same for every channel), requested
        // first thing: vmcnt completes in order and the pixel stores of the previous item are ahead of these loads;
        // by the middle pass they have long been acknowledged
        // (BPT > 1: one channel per workgroup and fewer threads than middle-stage butterflies - the 256-thread kernels of
        //  P = 8192 / 6144: thread t takes the butterflies t, t + T, ... and holds the filter values of each)
        constexpr int BPT = (P / R2 + T - 1) / T;
        const double2 *f = filt + foff[icap - 1] + (size_t)(tid % (P / R2)) * R2;
        // (the radix-32 / 24 kernels at 512 threads have no registers to hold the filter values across the strided passes:
        //  they request them behind forward pass 1 - the filter of a ring serves all its channel items: L2)
        constexpr bool FL_EARLY = !(R0 > 16 && T == 512);
        double2 fl[BPT][R2];
        if constexpr (FL_EARLY)
#pragma unroll
        for (int q = 0; q < BPT; q++)
#pragma unroll
            for (int r = 0; r < (R2 + 1) / 2; r++) fl[q][r] = f[(size_t)min(q * T, P / R2 - 1 - (int)(tid % (P / R2))) * R2 + r];   // (second half behind the fold: spread requests; clamped: 6144 / 16 = 384 butterflies on 256 threads)
        // fold phases e^{i tid phi0}, e^{i T phi0} of this cap ring from the plan's table (the two sincospi they replace
        // were ~150 DP instructions per thread and item); requested here, consumed behind the barrier
        const double2 fph0 = foldph[(size_t)(icap - 1) * 512 + tid];
        const double2 fphs = foldstep[(icap - 1) * 2 + (T == 512 ? 1 : 0)];
        // chirp b_k of the pre-pass pairs: fetched with the cells (one item ahead, before the previous item's stores)
        double2 cb[U];
#pragma unroll
        for (int u = 0; u < U; u++) cb[u] = cbn[u];
        __syncthreads();                                  // previous item's LDS reads are done
        CTSTAMP(0);
        fe.fold(sm, cell_ptr(item), Lr, n, fph0, fphs, tid);
        CTSTAMP(1);
        if constexpr (FL_EARLY)
#pragma unroll
        for (int q = 0; q < BPT; q++)
#pragma unroll
            for (int r = (R2 + 1) / 2; r < R2; r++) fl[q][r] = f[(size_t)min(q * T, P / R2 - 1 - (int)(tid % (P / R2))) * R2 + r];
        __syncthreads();
        CTSTAMP(2);
        // ---- pre-pass for the pairs (k, h - k), with w = e^{2 pi i / n} and b_{h-k} = b_k, w^{h-k} = -conj(w^k) (h even):
        //        y_k     = b_k [(X_k + conj X_{h-k}) + i w^k (X_k - conj X_{h-k})]
        //        y_{h-k} = b_k [(X_{h-k} + conj X_k) - i conj(w^k) (X_{h-k} - conj X_k)]
        //      w^k = e^{i pi k / h} is the square of the fold phase e^{i k phi0} (phi0 = pi / 2h on a cap ring);
        //      zeros on [h, HALF)
        static_assert(U <= decltype(fe)::KP, "the pre-pass takes w^k from the fold phases that are kept");
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = tid + u * T;
            if (2 * k <= h) {
                const int k2 = h - k;
                const double2 wk = csqr(fe.phk[u]);
                const double2 iw = make_double2(-wk.y, wk.x);                 // i w^k
                const double2 miwc = make_double2(-wk.y, -wk.x);               // -i conj(w^k)
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    double2 *bc = sm + c * BS;
                    const double2 xa = bc[fpad(k)], xb = bc[fpad(k2)];
                    const double2 s1 = make_double2(xa.x + xb.x, xa.y - xb.y), d1 = make_double2(xa.x - xb.x, xa.y + xb.y);
                    const double2 s2 = make_double2(s1.x, -s1.y), d2 = make_double2(-d1.x, d1.y);  // (xb + conj xa), (xb - conj xa)
                    const double2 zk = cmul(cb[u], cadd(s1, cmul(iw, d1)));
                    const double2 zk2 = cmul(cb[u], cadd(s2, cmul(miwc, d2)));
                    bc[fpad(k)] = zk;
                    if (k > 0) {
                        if (k2 != k) bc[fpad(k2)] = zk2;
                    } else {
                        bc[fpad(h)] = make_double2(0.0, 0.0);
                    }
                }
            }
        }
        for (int j = h + 1 + tid; j < HALF; j += T)
#pragma unroll
            for (int c = 0; c < NCH; c++) sm[c * BS + fpad(j)] = make_double2(0.0, 0.0);
        // the next item's cells and pre-pass chirps, in two parts (here and behind forward pass 1): three passes ahead
        // of the last pass, whose wait for its own (younger) chirp loads completes them before the first pixel store
        // is issued - see FrontEnd::touch.  Unconditional (the last iteration re-reads an item): loads behind an `if`
        // cannot be counted by the compiler's vmcnt bookkeeping, and the wait for the filter values in the middle
        // pass would then wait for these as well.
        const int nitem = ct_remap<NCH>(min(vitem + (int)gridDim.x, nitems - 1), nitems);
        fe.template prefetch<0>(cell_ptr(nitem), L, tid);
        __syncthreads();
        CTSTAMP(3);
        // ---- forward pass 1 (sign -), inputs r >= R0 / 2 are the zero padding and are not read
        {
            constexpr int TOT = NCH * Q0;
            constexpr int IT = (TOT + T - 1) / T;
            const double2 w = cconj(wA);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                double2 *p = sm + ch * BS + fpad(j0);
                double2 x[R0];
#pragma unroll
                for (int r = 0; r < R0 / 2; r++) x[r] = p[fpc(r * Q0)];
#pragma unroll
                for (int r = R0 / 2; r < R0; r++) x[r] = make_double2(0.0, 0.0);
                DftR<R0, -1>::run(x);
                tw_apply<R0>(x, w);
#pragma unroll
                for (int r = 0; r < R0; r++) p[fpc(r * Q0)] = x[r];
            }
        }
        fe.template prefetch<1>(cell_ptr(nitem), L, tid);
        load_cbn(nitem, tid);
        if constexpr (!FL_EARLY) {
#pragma unroll
            for (int r = 0; r < R2; r++) fl[0][r] = f[r];
        }
        __syncthreads();
        CTSTAMP(4);
        ct_pass<PK, P, NCH, BS, Q0, R1, -1, false, T>(sm, wB, tid);
        __syncthreads();
        CTSTAMP(5);
        // ---- last forward pass, filter, first inverse pass: R2 contiguous elements, no twiddles, in registers
        {
            // thread -> (channel, butterfly t) with t = tid mod NB in EVERY iteration (the filter registers fl[] belong to
            // that t): CPI whole channels per iteration, threads beyond CPI NB idle (NB = 192 for the 3 * 2^k lengths)
            constexpr int NB = P / R2;
            if constexpr (BPT == 1) {
                constexpr int CPI = T / NB, IT = (NCH + CPI - 1) / CPI;
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const int chl = tid / NB, t = tid - chl * NB;
                    const int ch = it * CPI + chl;
                    if (chl >= CPI || ch >= NCH) break;
                    double2 *p = sm + ch * BS + fpad(t * R2);
                    double2 x[R2];
#pragma unroll
                    for (int r = 0; r < R2; r++) x[r] = p[fpc(r)];
                    DftR<R2, -1>::run(x);
#pragma unroll
                    for (int r = 0; r < R2; r++) x[r] = cmul(x[r], fl[0][r]);
                    DftR<R2, 1>::run(x);
#pragma unroll
                    for (int r = 0; r < R2; r++) p[fpc(r)] = x[r];
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                    for (int q = 0; q < BPT; q++) {
                        if ((NB % T) != 0 && tid + q * T >= NB) break;
                        double2 *p = sm + ch * BS + fpad((tid + q * T) * R2);
                        double2 x[R2];
#pragma unroll
                        for (int r = 0; r < R2; r++) x[r] = p[fpc(r)];
                        DftR<R2, -1>::run(x);
#pragma unroll
                        for (int r = 0; r < R2; r++) x[r] = cmul(x[r], fl[q][r]);
                        DftR<R2, 1>::run(x);
#pragma unroll
                        for (int r = 0; r < R2; r++) p[fpc(r)] = x[r];
                    }
            }
        }
        // chirp of the outputs this thread forms in the last pass: j0 + r Q0 < h, r < R0 / 2 (requested two passes ahead;
        // the radix-32 / 24 kernels have no registers to hold 16 of them across the passes and read them at the store)
        constexpr bool OB_EARLY = R0 <= 16;
        double2 ob[OB_EARLY ? R0 / 2 : 1];
        if constexpr (OB_EARLY) {
            const int j0 = tid & (Q0 - 1);
#pragma unroll
            for (int r = 0; r < R0 / 2; r++) ob[r] = bch[min(j0 + r * Q0, h - 1)];
        }
        __syncthreads();
        CTSTAMP(6);
        ct_pass<PK, P, NCH, BS, Q0, R1, 1, true, T>(sm, wB, tid);
        __syncthreads();
        CTSTAMP(7);
        // ---- last inverse pass (sign +): only the outputs j0 + r Q0 < h exist; times b_j / P, pixel pairs to HBM
        {
            constexpr int TOT = NCH * Q0;
            constexpr int IT = (TOT + T - 1) / T;
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                const double2 *p = sm + ch * BS + fpad(j0);
                double2 x[R0];
#pragma unroll
                for (int r = 0; r < R0; r++) x[r] = p[fpc(r * Q0)];
                tw_apply<R0>(x, wA);
                DftR<R0, 1>::run(x);
                if (ch0 + ch < nnu) {
                    double *out = maps + (size_t)(ch0 + ch) * npix + start;
#pragma unroll
                    for (int r = 0; r < R0 / 2; r++) {
                        const int jj = j0 + r * Q0;
                        if (jj < h) {
                            double2 zv = cmul(x[r], OB_EARLY ? ob[OB_EARLY ? r : 0] : bch[jj]);
                            zv.x *= invP;
                            zv.y *= invP;
                            *reinterpret_cast<double2 *>(out + 2 * jj) = zv;
                        }
                    }
                }
            }
        }
        CTSTAMP(8);
    }
#if CT_STAMPS
    if ((tid0 & 63) == 0)
        for (int k = 0; k < 9; k++) atomicAdd(&g_ct_stamps[k], ct_acc[k]);
#endif
}

// ------------------------------------------------------------------------------------
// K5^T, direct class (the belt rings of map2alm, cora/util/hputil.py:195-234 through healpy.map2alm): the adjoint of
// ringfft_direct_ct pass by pass.  The n = 2 N pixels of a ring are N complex numbers z_j = x_2j + i x_2j+1;
//   pass C^H  radix R2 on the pixels in the lane <-> digit map of the synthesis' fused store (eight lanes load 128
//             contiguous bytes; the loads of the NEXT item are issued behind this pass and complete behind the others),
//   pass B^H, A^H  radix 16 with the conjugate twiddles in front (decimation in time): Z_k in natural order,
//   split     X_m = 1/2 [(Z_m + conj Z_{N-m}) - i e^{-i pi m / N} (Z_m - conj Z_{N-m})], G_m = w_ring (4 pi / npix)
//             e^{-i m phi0} X_m for m < mcut(ring) <= N + 1 (no aliasing on a belt ring), stored as whole 64-byte cells.
// Every twiddle and phase is a per-thread constant of the kernel (phi0 of a belt ring is 0 or pi / (2 N)).
// ------------------------------------------------------------------------------------
template <int N, int NCH, int T>
__global__ void __launch_bounds__(T)
ringana_direct_ct(const int32_t *__restrict__ ring_list, int nlist, int lmax, int G, int nnu, int nvalid, long npix,
                  const int64_t *__restrict__ start_a, const double *__restrict__ phi0_a, const double *maps, double *inter,
                  const int32_t *__restrict__ mcut, const double *__restrict__ ring_w, int nring) {
    constexpr int PK = K5_PK_DIRECT;
    constexpr int R0 = Sch<N>::R0, R1 = Sch<N>::R1, R2 = Sch<N>::R2;
    static_assert(R0 == 16 && R1 == 16, "digit map of the fused load assumes 16 x 16 x R2");
    constexpr int Q0 = N / R0;
    static_assert(Q0 % 128 == 0, "first-pass stride must be a multiple of the padding period");
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    constexpr int TOT0 = NCH * 256, IT0 = (TOT0 + T - 1) / T;
    static_assert(TOT0 % T == 0, "every thread loads whole butterflies");
    constexpr int MO = N / T;                // cells per thread (plus m = N on one thread)
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const int L = lmax + 1;
    const int ngrp = (nnu + NCH - 1) / NCH;
    const int nitems = nlist * ngrp;

    double2 wA, wB;         // e^{2 pi i j0 / N}, e^{2 pi i j1 / (N / R0)} (the passes conjugate them)
    double2 wS, wSstep;     // e^{-i pi tid / N} and its step over T cells: the split twiddle
    double2 phS, phSstep;   // e^{-i tid pi / (2 N)} and its step: the phase of the shifted rings
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)N, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
        sincospi((double)tid0 / (double)N, &s, &c);
        wS = make_double2(c, -s);
        sincospi((double)T / (double)N, &s, &c);
        wSstep = make_double2(c, -s);
        sincospi((double)tid0 / (2.0 * N), &s, &c);
        phS = make_double2(c, -s);
        sincospi((double)T / (2.0 * N), &s, &c);
        phSstep = make_double2(c, -s);
    }
    // the pixels of one item: butterfly idx = tid + it T of pass C^H holds z_j, j = k0 + 16 k1 + 256 r
    double2 pf[IT0][R2];
    auto prefetch = [&](int item, int tid) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const long start = start_a[ring];
#pragma unroll
        for (int it = 0; it < IT0; it++) {
            const int idx = tid + it * T;
            const int ch = idx >> 8;
            const int k0 = (idx & 7) | ((idx >> 3) & 8);
            const int k1 = ((idx >> 3) & 7) | ((idx >> 4) & 8);
            // (a padding channel reads the last valid one: the loads stay unconditional, its cells are zeroed at the store)
            const int chv = min(ch0 + ch, nvalid - 1);
            const double *src = maps + (size_t)chv * npix + start + 2 * (k0 + 16 * k1);
#pragma unroll
            for (int r = 0; r < R2; r++) pf[it][r] = *reinterpret_cast<const double2 *>(src + 512 * r);
        }
    };
    int vitem = blockIdx.x;
    if (vitem < nitems) prefetch(vitem, tid0);
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));          // (addresses of the passes are not loop invariants: see ringfft_direct_ct)
        const int item = vitem;
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const bool shifted = phi0_a[ring] != 0.0;
        const int Lr = mcut[ring];
        const double wr = (ring_w ? ring_w[min(ring, nring - 1 - ring)] : 1.0) * (4.0 * M_PI / (double)npix);
        __syncthreads();                       // previous item's LDS reads are done
        // ---- pass C^H from the prefetched registers
#pragma unroll
        for (int it = 0; it < IT0; it++) {
            const int idx = tid + it * T;
            const int ch = idx >> 8;
            const int k0 = (idx & 7) | ((idx >> 3) & 8);
            const int k1 = ((idx >> 3) & 7) | ((idx >> 4) & 8);
            double2 *p = sm + ch * BS + fpad((k0 * 16 + k1) * R2);
            DftR<R2, -1>::run(pf[it]);
#pragma unroll
            for (int r = 0; r < R2; r++) p[fpc(r)] = pf[it][r];
        }
        // the next item's pixels (unconditional: the last iteration re-reads an item, so that the loads can be counted)
        prefetch(min(vitem + (int)gridDim.x, nitems - 1), tid);
        __syncthreads();
        ct_pass<PK, N, NCH, BS, Q0, R1, -1, true, T>(sm, wB, tid);
        __syncthreads();
        ct_pass<PK, N, NCH, BS, N, R0, -1, true, T>(sm, wA, tid);
        __syncthreads();
        // ---- split + phase + cell store
        {
            double *cell0 = inter + ((size_t)ring * G + (ch0 >> 2)) * L * 8 + (ch0 & 3);
            double2 w = wS, ph = shifted ? phS : make_double2(1.0, 0.0);
            const double2 phstep = shifted ? phSstep : make_double2(1.0, 0.0);
            asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(ph.x), "+v"(ph.y));
            auto emit = [&](int m, const double2 wm, const double2 phm) {
                const int ka = m == N ? 0 : m;                 // Z_N := Z_0
                const int kb = m == 0 ? 0 : N - m;
                double re[NCH], im[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const double2 za = sm[c * BS + fpad(ka)], zb = sm[c * BS + fpad(kb)];
                    const double2 sum = make_double2(za.x + zb.x, za.y - zb.y);
                    const double2 dif = make_double2(za.x - zb.x, za.y + zb.y);
                    const double2 t = cmul(dif, wm);
                    const double2 X = make_double2(0.5 * (sum.x + t.y), 0.5 * (sum.y - t.x));
                    const double2 g = cmul(X, make_double2(phm.x * wr, phm.y * wr));
                    const bool live = ch0 + c < nvalid;
                    re[c] = live ? g.x : 0.0;
                    im[c] = live ? g.y : 0.0;
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
            };
#pragma unroll
            for (int u = 0; u < MO; u++) {
                const int m = tid + u * T;
                if (m < Lr) emit(m, w, ph);
                w = cmul(w, wSstep);
                ph = cmul(ph, phstep);
            }
            // m = N: e^{-i pi} = -1 and e^{-i N phi0} = e^{-i pi / 2} = -i on a shifted ring
            if (tid == 0 && N < Lr) emit(N, make_double2(-1.0, 0.0), shifted ? make_double2(0.0, -1.0) : make_double2(1.0, 0.0));
        }
    }
}

// ------------------------------------------------------------------------------------
// host side: launch one class with the compile-time kernel if there is one for it
// ------------------------------------------------------------------------------------
template <int N, int NCH, int MC, int T>
static int launch_direct(corahip_ctx *ctx, hipStream_t stream, int wg_per_cu, const corahip_sht_plan *p,
                         const corahip_sht_plan::ring_class &c, const double *inter, int G, int nnu, double *maps) {
    constexpr int PK = K5_PK_DIRECT;
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (long)c.count * ((nnu + NCH - 1) / NCH);
    const int per_cu = wg_per_cu > 0 ? wg_per_cu : std::max<int>(1, (int)((160 * 1024) / shm));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    // diagnostics (overlap table): the belt on fewer workgroups than CUs - it is HBM-bound, how many
    // CUs does it need to stream at its rate?
    static const char *belt_wgs = getenv("CORAHIP_K5_BELT_WGS");
    if (belt_wgs && atoi(belt_wgs) > 0) grid.x = (unsigned)std::min<long>(grid.x, atol(belt_wgs));
    HIP_TRY(hipFuncSetAttribute((const void *)ringfft_direct_ct<N, NCH, MC, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    ringfft_direct_ct<N, NCH, MC, T><<<grid, T, shm, stream>>>(c.d_list, c.count, p->lmax, G, nnu, p->npix, p->d_start,
                                                               p->d_phi0, inter, maps, p->d_mcut);
    LAUNCH_CHECK();
    return 0;
}
template <int P, int NCH, int MC, int T>
static int launch_blu(corahip_ctx *ctx, hipStream_t stream, int wg_per_cu, const corahip_sht_plan *p,
                      const corahip_sht_plan::ring_class &c, const double *inter, int G, int nnu, double *maps,
                      const int64_t *d_foff = nullptr, const double2 *d_filt = nullptr) {
    if (!d_foff) {
        d_foff = p->d_blu_foff;
        d_filt = p->d_bfilt;
    }
    constexpr int PK = K5_PK_BLU;
    constexpr int BS = fpc(P) + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (long)c.count * ((nnu + NCH - 1) / NCH);
    const int per_cu = wg_per_cu > 0 ? wg_per_cu : std::max<int>(1, (int)((160 * 1024) / shm));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipFuncSetAttribute((const void *)ringfft_blu_ct<P, NCH, MC, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    ringfft_blu_ct<P, NCH, MC, T><<<grid, T, shm, stream>>>(c.d_list, c.count, p->nside, p->lmax, G, nnu, p->npix, p->d_nphi,
                                                            p->d_start, p->d_phi0, inter, maps, p->d_blu_boff,
                                                            d_foff, p->d_bchirp, d_filt, p->d_mcut, p->d_foldph, p->d_foldstep);
    LAUNCH_CHECK();
#if CT_STAMPS
    {
        unsigned long long hs[12], z[12] = {0};
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_ct_stamps), sizeof(hs)));
        const double per = 1.0 / ((double)nitems * (T / 64));   // cycles per item and wave
        fprintf(stderr, "K5ct P=%d nch=%d items=%ld: cycles/item  top %.0f fold %.0f pf+bar %.0f pre %.0f F1 %.0f F2 %.0f mid %.0f I2 %.0f I3 %.0f\n",
                P, NCH, nitems, hs[0] * per, hs[1] * per, hs[2] * per, hs[3] * per, hs[4] * per, hs[5] * per, hs[6] * per,
                hs[7] * per, hs[8] * per);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ct_stamps), z, sizeof(z)));
    }
#endif
    return 0;
}

// *took = true if the class was launched here, false if the generic kernel has to take it; the return value is the
// error status only (0 = OK, CORAHIP_E* < 0 or a positive hipError_t): hipErrorInvalidValue is 1, so "launched" must
// never share the int with the status
int sht_ringfft_ct(corahip_ctx *ctx, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &c, const double *inter,
                   int G, int nnu, double *maps, bool *took) {
    static const bool off = getenv("CORAHIP_K5_GENERIC") != nullptr;   // diagnostics: force the generic kernel
    *took = false;
    if (off) return 0;
    int rc = -1;
    hipStream_t st = ctx->stream;
    if (c.P == 0) {
        // A/B: the belt as 2-channel items of 256 threads, TWO workgroups (78 KB each) per CU
        static const bool belt2 = getenv("CORAHIP_K5_BELT2") != nullptr;
        if (c.N == 2048 && belt2) rc = launch_direct<2048, 2, 8, 256>(ctx, st, 2, p, c, inter, G, nnu, maps);
        else if (c.N == 2048) rc = launch_direct<2048, 4, 4, 512>(ctx, st, 0, p, c, inter, G, nnu, maps);
        else if (c.N == 4096) rc = launch_direct<4096, 2, 8, 512>(ctx, st, 0, p, c, inter, G, nnu, maps);
        else return 0;
    } else {
        static const bool no3 = getenv("CORAHIP_K5_NO3") != nullptr;   // diagnostics: power-of-two lengths only
        static const bool half = getenv("CORAHIP_K5_HALF") != nullptr;   // A/B: 256-thread workgroups of one channel, two per CU
        // (one channel fills the LDS)
        // Measured at the cfg-5 rank share (128 channels; the generic kernel took 30.6 ms for the two classes): 512 threads
        // (two waves per SIMD, but the radix-32 / 24 butterflies then spill: 268 / 72 bytes) 16.5 / 10.1 ms, 256 threads
        // (one wave per SIMD, no spill) 13.8 / 10.8 ms for P = 8192 / 6144: each takes the faster form.
        // CORAHIP_K5_BIG=256|512 forces one form for both (A/B).
        static const char *bigenv = getenv("CORAHIP_K5_BIG");
        const int big = bigenv ? atoi(bigenv) : 0;
        if (c.P3 == 8192 && big != 512) rc = launch_blu<8192, 1, 16, 256>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P3 == 8192) rc = launch_blu<8192, 1, 8, 512>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P3 == 6144 && big != 256) rc = launch_blu<6144, 1, 8, 512>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P3 == 6144) rc = launch_blu<6144, 1, 16, 256>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (half && c.P3 == 3072 && !no3) rc = launch_blu<3072, 1, 8, 256>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (half && c.P3 != 1536 && c.P == 4096) rc = launch_blu<4096, 1, 8, 256>(ctx, st, 0, p, c, inter, G, nnu, maps);
        else if (c.P3 == 2560 && !no3) rc = launch_blu<2560, 2, 4, 512>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P3 == 3584 && !no3) rc = launch_blu<3584, 2, 4, 512>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P3 == 3072 && !no3) rc = launch_blu<3072, 2, 4, 512>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P3 == 1536 && !no3) rc = launch_blu<1536, 4, 2, 512>(ctx, st, 0, p, c, inter, G, nnu, maps, p->d_blu3_foff, p->d_bfilt3);
        else if (c.P == 4096) rc = launch_blu<4096, 2, 4, 512>(ctx, st, 0, p, c, inter, G, nnu, maps);
        else if (c.P == 2048 && half) rc = launch_blu<2048, 2, 4, 256>(ctx, st, 2, p, c, inter, G, nnu, maps);   // A/B: 2-channel items, two per CU
        else if (c.P == 2048) rc = launch_blu<2048, 4, 2, 512>(ctx, st, 0, p, c, inter, G, nnu, maps);
        else if (c.P == 1024) rc = launch_blu<1024, 4, 2, 256>(ctx, st, 0, p, c, inter, G, nnu, maps);   // (256 threads, two workgroups per CU: 0.61 -> 0.54 ms)
        else return 0;
    }
    if (rc) return rc;
    *took = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// K5^T, Bluestein class (cap rings): Z_k = sum_j z_j e^{-2 pi i j k / h} as the convolution ringfft_blu_ct runs - the
// same five passes on the same chirp / filter tables, applied to conj(z_j) b_j (ringana_kernel's formulation) - with the
// pixels in front (natural order, prefetched one item ahead) and, behind the last inverse pass, Z_k = conj(W_k b_k) / P
// back in LDS for the split X_m = 1/2 [(Z_m + conj Z_{h-m}) - i e^{-i pi m / h} (Z_m - conj Z_{h-m})] and the cell
// store.  Phase e^{-i m phi0} and split twiddle (its square: phi0 = pi / 2h on a cap ring) from the plan's fold table.
// ------------------------------------------------------------------------------------
template <int P, int NCH, int T>
__global__ void __launch_bounds__(T)
ringana_blu_ct(const int32_t *__restrict__ ring_list, int nlist, int nside, int lmax, int G, int nnu, int nvalid, long npix,
               const int32_t *__restrict__ nphi_a, const int64_t *__restrict__ start_a, const double *maps, double *inter,
               const int64_t *__restrict__ boff, const int64_t *__restrict__ foff, const double2 *chirp, const double2 *filt,
               const int32_t *__restrict__ mcut, const double2 *foldph, const double2 *foldstep, const double *__restrict__ ring_w) {
    constexpr int PK = K5_PK_BLU;
    constexpr int R0 = Sch<P>::R0, R1 = Sch<P>::R1, R2 = Sch<P>::R2;
    static_assert(R0 <= 16, "the lengths above 4096 keep the generic kernel");
    constexpr int Q0 = P / R0;
    constexpr int BS = fpc(P) + K5_CH_SKEW;
    constexpr int HALF = (R0 / 2) * Q0;          // h <= HALF: the non-zero half of the padded input
    constexpr int U = (HALF + T - 1) / T;        // pixel pairs per thread and channel
    constexpr int NB = P / R2;
    static_assert(NB <= T, "one middle-stage butterfly per thread and channel group");
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const int L = lmax + 1;
    const int ngrp = (nnu + NCH - 1) / NCH;
    const int nitems = nlist * ngrp;
    const int nring = 4 * nside - 1;
    const double invP = 1.0 / (double)P;

    double2 wA, wB;     // e^{2 pi i j0 / P}, e^{2 pi i j1 / (P / R0)}
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)P, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    double2 pf[NCH][U];   // z_j, j = tid + u T, of the next item (index clamped; zeroed where j >= h at the commit)
    double2 cbn[U];       // chirp b_j of the same positions
    auto prefetch = [&](int item, int t) {
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const int hh = nphi_a[ring] >> 1;
        const long start = start_a[ring];
        const int ic = ring + 1 < nside ? ring + 1 : 4 * nside - (ring + 1);
        const double2 *b = chirp + boff[ic - 1];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = min(t + u * T, hh - 1);
            cbn[u] = b[j];
#pragma unroll
            for (int c = 0; c < NCH; c++)
                pf[c][u] = *reinterpret_cast<const double2 *>(maps + (size_t)min(ch0 + c, nvalid - 1) * npix + start + 2 * j);
        }
    };
    int vitem = blockIdx.x;
    if (vitem < nitems) prefetch(ct_remap<NCH>(vitem, nitems), tid0);
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;                                   // opaque per item: see ringfft_direct_ct
        asm volatile("" : "+v"(tid));
        const int item = ct_remap<NCH>(vitem, nitems);
        const int ring = ring_list[item / ngrp];
        const int ch0 = (item % ngrp) * NCH;
        const int n = nphi_a[ring];
        const int h = n >> 1;
        const int icap = ring + 1 < nside ? ring + 1 : 4 * nside - (ring + 1);
        const int Lr = mcut[ring];
        const double wr = (ring_w ? ring_w[min(ring, nring - 1 - ring)] : 1.0) * (4.0 * M_PI / (double)npix);
        const double2 *bch = chirp + boff[icap - 1];
        const double2 *f = filt + foff[icap - 1] + (size_t)(tid % NB) * R2;
        double2 fl[R2];
#pragma unroll
        for (int r = 0; r < R2; r++) fl[r] = f[r];
        const double2 fph0 = foldph[(size_t)(icap - 1) * 512 + tid];
        const double2 fphs = foldstep[(icap - 1) * 2 + (T == 512 ? 1 : 0)];
        __syncthreads();                                  // previous item's LDS reads are done
        // ---- conj(z_j) b_j at position j, zeros on [h, HALF)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = tid + u * T;
            if (j < HALF) {
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    double2 zv = make_double2(0.0, 0.0);
                    if (j < h) zv = cmul(make_double2(pf[c][u].x, -pf[c][u].y), cbn[u]);
                    sm[c * BS + fpad(j)] = zv;
                }
            }
        }
        // the next item's pixels and chirps (unconditional: the last iteration re-reads an item)
        prefetch(ct_remap<NCH>(min(vitem + (int)gridDim.x, nitems - 1), nitems), tid);
        __syncthreads();
        // ---- forward pass 1 (sign -), inputs r >= R0 / 2 are the zero padding and are not read
        {
            constexpr int TOT = NCH * Q0;
            constexpr int IT = (TOT + T - 1) / T;
            const double2 w = cconj(wA);
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                double2 *p = sm + ch * BS + fpad(j0);
                double2 x[R0];
#pragma unroll
                for (int r = 0; r < R0 / 2; r++) x[r] = p[fpc(r * Q0)];
#pragma unroll
                for (int r = R0 / 2; r < R0; r++) x[r] = make_double2(0.0, 0.0);
                DftR<R0, -1>::run(x);
                tw_apply<R0>(x, w);
#pragma unroll
                for (int r = 0; r < R0; r++) p[fpc(r * Q0)] = x[r];
            }
        }
        __syncthreads();
        ct_pass<PK, P, NCH, BS, Q0, R1, -1, false, T>(sm, wB, tid);
        __syncthreads();
        // ---- last forward pass, filter, first inverse pass in registers
        {
            constexpr int CPI = T / NB, IT = (NCH + CPI - 1) / CPI;
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int chl = tid / NB, t = tid - chl * NB;
                const int ch = it * CPI + chl;
                if (chl >= CPI || ch >= NCH) break;
                double2 *p = sm + ch * BS + fpad(t * R2);
                double2 x[R2];
#pragma unroll
                for (int r = 0; r < R2; r++) x[r] = p[fpc(r)];
                DftR<R2, -1>::run(x);
#pragma unroll
                for (int r = 0; r < R2; r++) x[r] = cmul(x[r], fl[r]);
                DftR<R2, 1>::run(x);
#pragma unroll
                for (int r = 0; r < R2; r++) p[fpc(r)] = x[r];
            }
        }
        // chirp of the outputs this thread forms in the last pass: j0 + r Q0 < h, r < R0 / 2
        double2 ob[R0 / 2];
        {
            const int j0 = tid & (Q0 - 1);
#pragma unroll
            for (int r = 0; r < R0 / 2; r++) ob[r] = bch[min(j0 + r * Q0, h - 1)];
        }
        __syncthreads();
        ct_pass<PK, P, NCH, BS, Q0, R1, 1, true, T>(sm, wB, tid);
        __syncthreads();
        // ---- last inverse pass (sign +): Z_k = conj(W_k b_k) / P for the outputs k = j0 + r Q0 < h, back to position k
        {
            constexpr int TOT = NCH * Q0;
            constexpr int IT = (TOT + T - 1) / T;
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                double2 *p = sm + ch * BS + fpad(j0);
                double2 x[R0];
#pragma unroll
                for (int r = 0; r < R0; r++) x[r] = p[fpc(r * Q0)];
                tw_apply<R0>(x, wA);
                DftR<R0, 1>::run(x);
#pragma unroll
                for (int r = 0; r < R0 / 2; r++) {
                    const double2 zv = cmul(x[r], ob[r]);
                    p[fpc(r * Q0)] = make_double2(zv.x * invP, -zv.y * invP);
                }
            }
        }
        __syncthreads();
        // ---- split + phase + cell store (m >= n aliases back; bins above h are the conjugates of n - k)
        {
            double *cell0 = inter + ((size_t)ring * G + (ch0 >> 2)) * L * 8 + (ch0 & 3);
            double2 ph = cconj(fph0);
            const double2 phstep = cconj(fphs);
            double2 w = csqr(ph);
            const double2 wstep = csqr(phstep);
            for (int m = tid; m < Lr; m += T) {
                const int k = m < n ? m : m % n;
                const bool cj = k > h;
                const int kk = cj ? n - k : k;
                const int ka = kk == h ? 0 : kk;
                const int kb = kk == 0 ? 0 : h - kk;
                const double2 wm = make_double2(w.x, cj ? -w.y : w.y);
                const double2 phm = make_double2(ph.x * wr, ph.y * wr);
                ph = cmul(ph, phstep);
                w = cmul(w, wstep);
                double re[NCH], im[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const double2 za = sm[c * BS + fpad(ka)], zb = sm[c * BS + fpad(kb)];
                    const double2 sum = make_double2(za.x + zb.x, za.y - zb.y);
                    const double2 dif = make_double2(za.x - zb.x, za.y + zb.y);
                    const double2 t = cmul(dif, wm);
                    double2 X = make_double2(0.5 * (sum.x + t.y), 0.5 * (sum.y - t.x));
                    if (cj) X.y = -X.y;
                    const double2 g = cmul(X, phm);
                    const bool live = ch0 + c < nvalid;
                    re[c] = live ? g.x : 0.0;
                    im[c] = live ? g.y : 0.0;
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
}

// K5^T with the compile-time kernel of the class, if there is one (the belt); *took as in sht_ringfft_ct
template <int N, int NCH, int T>
static int launch_ana_direct(corahip_ctx *ctx, hipStream_t stream, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &c,
                             const double *maps, int nvalid, int nnu_pad, const double *ring_w, int G, double *inter) {
    constexpr int PK = K5_PK_DIRECT;
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (long)c.count * ((nnu_pad + NCH - 1) / NCH);
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * std::max<int>(1, (int)((160 * 1024) / shm))));
    HIP_TRY(hipFuncSetAttribute((const void *)ringana_direct_ct<N, NCH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ringana_direct_ct<N, NCH, T><<<grid, T, shm, stream>>>(c.d_list, c.count, p->lmax, G, nnu_pad, nvalid, p->npix, p->d_start, p->d_phi0,
                                                           maps, inter, p->d_mcut, ring_w, 4 * p->nside - 1);
    LAUNCH_CHECK();
    return 0;
}
template <int P, int NCH, int T>
static int launch_ana_blu(corahip_ctx *ctx, hipStream_t stream, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &c,
                          const double *maps, int nvalid, int nnu_pad, const double *ring_w, int G, double *inter, bool blu3) {
    constexpr int PK = K5_PK_BLU;
    constexpr int BS = fpc(P) + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (long)c.count * ((nnu_pad + NCH - 1) / NCH);
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * std::max<int>(1, (int)((160 * 1024) / shm))));
    HIP_TRY(hipFuncSetAttribute((const void *)ringana_blu_ct<P, NCH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ringana_blu_ct<P, NCH, T><<<grid, T, shm, stream>>>(c.d_list, c.count, p->nside, p->lmax, G, nnu_pad, nvalid, p->npix, p->d_nphi,
                                                        p->d_start, maps, inter, p->d_blu_boff, blu3 ? p->d_blu3_foff : p->d_blu_foff,
                                                        p->d_bchirp, blu3 ? p->d_bfilt3 : p->d_bfilt, p->d_mcut, p->d_foldph,
                                                        p->d_foldstep, ring_w);
    LAUNCH_CHECK();
    return 0;
}
int sht_ringana_ct(corahip_ctx *ctx, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &c, const double *maps, int nvalid,
                   int nnu_pad, const double *ring_w, int G, double *inter, bool *took) {
    static const bool off = getenv("CORAHIP_K5_GENERIC") != nullptr;
    static const bool no3 = getenv("CORAHIP_K5_NO3") != nullptr;
    *took = false;
    if (off || nvalid < 1) return 0;
    hipStream_t st = ctx->stream;
    int rc;
#define ANA_ARGS ctx, st, p, c, maps, nvalid, nnu_pad, ring_w, G, inter
    if (c.P == 0 && c.N == 2048) rc = launch_ana_direct<2048, 4, 512>(ANA_ARGS);
    else if (c.P == 0 && c.N == 4096) rc = launch_ana_direct<4096, 2, 512>(ANA_ARGS);
    else if (c.P3 == 2560 && !no3) rc = launch_ana_blu<2560, 2, 512>(ANA_ARGS, true);
    else if (c.P3 == 3584 && !no3) rc = launch_ana_blu<3584, 2, 512>(ANA_ARGS, true);
    else if (c.P3 == 3072 && !no3) rc = launch_ana_blu<3072, 2, 512>(ANA_ARGS, true);
    else if (c.P3 == 1536 && !no3) rc = launch_ana_blu<1536, 4, 512>(ANA_ARGS, true);
    else if (c.P == 4096 && c.P3 <= 4096) rc = launch_ana_blu<4096, 2, 512>(ANA_ARGS, false);
    else if (c.P == 2048) rc = launch_ana_blu<2048, 4, 512>(ANA_ARGS, false);
    else if (c.P == 1024) rc = launch_ana_blu<1024, 4, 256>(ANA_ARGS, false);
    else return 0;
#undef ANA_ARGS
    if (rc) return rc;
    *took = true;
    return 0;
}

int sht_second_stream(corahip_ctx *ctx) {
    if (!ctx->stream2) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    return 0;
}

// The belt (HBM-bound: 2/3 of the pixels, little arithmetic) and the largest Bluestein class (LDS / FP64-bound, a third
// of the traffic) run CONCURRENTLY: each as 256-thread workgroups with half the channels per item (78 KB of LDS), one
// workgroup of each kernel per CU, launched on two streams - the compute-bound items of one kernel fill the memory
// stalls of the other on every CU.  *took = true if it launched both classes (the caller then skips them); the return
// value is the error status only.
int sht_ringfft_ct_pair(corahip_ctx *ctx, const corahip_sht_plan *p, const corahip_sht_plan::ring_class &belt,
                        const corahip_sht_plan::ring_class &cap, const double *inter, int G, int nnu, double *maps, bool *took) {
    *took = false;
    // Measured at cfg 3 (one box, A/B): 18.6 ms paired against 17.0 ms with the two classes one after the other at
    // full width - the half-width kernels lose more per item than the overlap returns - so the pairing is OFF unless
    // CORAHIP_K5_PAIR is set; kept for the record and for other shapes.
    static const bool on = getenv("CORAHIP_K5_PAIR") != nullptr && getenv("CORAHIP_K5_GENERIC") == nullptr;
    if (!on || !(belt.P == 0 && belt.N == 2048 && cap.P == 4096 && cap.P3 == 0)) return 0;
    int rcs = sht_second_stream(ctx);
    if (rcs) return rcs;
    HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    int rc = launch_blu<4096, 1, 8, 256>(ctx, ctx->stream2, 1, p, cap, inter, G, nnu, maps);
    if (!rc) rc = launch_direct<2048, 2, 8, 256>(ctx, ctx->stream, 1, p, belt, inter, G, nnu, maps);
    // joined on the error path as well: nothing may stay on stream2 unordered against the caller's stream
    if (hipEventRecord(ctx->ev_join, ctx->stream2) != hipSuccess || hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) != hipSuccess)
        HIP_TRY(hipStreamSynchronize(ctx->stream2));
    if (rc) return rc;
    *took = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// plan time: Bluestein filters of the 3 * 2^k lengths, in the storage order of the forward passes above
// ------------------------------------------------------------------------------------
template <int P>
__global__ void __launch_bounds__(CT_T)
blu3_filter_kernel(int nside, const int32_t *__restrict__ p3_of, const int64_t *__restrict__ boff, const int64_t *__restrict__ foff3,
                   const double2 *__restrict__ chirp, double2 *__restrict__ filt3) {
    constexpr int PK = K5_PK_BLU;
    constexpr int R0 = Sch<P>::R0, R1 = Sch<P>::R1, R2 = Sch<P>::R2;
    constexpr int Q0 = P / R0, BS = fpc(P) + K5_CH_SKEW;
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int i = blockIdx.x + 1;
    if (p3_of[i - 1] != P) return;
    const int tid = threadIdx.x;
    const int h = 2 * i;
    const double2 *b = chirp + boff[i - 1];
    for (int j = tid; j < P; j += CT_T) {          // conj chirp, wrapped: f_j = conj b_j (j < h), f_{P-j} = conj b_j (0 < j < h)
        double2 v = make_double2(0.0, 0.0);
        if (j < h) v = cconj(b[j]);
        else if (P - j < h) v = cconj(b[P - j]);
        sm[fpad(j)] = v;
    }
    __syncthreads();
    double2 wA, wB;
    {
        double sv, cv;
        sincospi(2.0 * (double)(tid & (Q0 - 1)) / (double)P, &sv, &cv);
        wA = make_double2(cv, sv);
        sincospi(2.0 * (double)(tid & (Q0 / R1 - 1)) / (double)Q0, &sv, &cv);
        wB = make_double2(cv, sv);
    }
    ct_pass<PK, P, 1, BS, P, R0, -1, false, CT_T>(sm, wA, tid);
    __syncthreads();
    ct_pass<PK, P, 1, BS, Q0, R1, -1, false, CT_T>(sm, wB, tid);
    __syncthreads();
    ct_pass<PK, P, 1, BS, Q0 / R1, R2, -1, false, CT_T>(sm, wB, tid);   // (stride 1: no twiddles)
    __syncthreads();
    double2 *f = filt3 + foff3[i - 1];
    for (int j = tid; j < P; j += CT_T) f[j] = sm[fpad(j)];
}

int sht_blu3_tables(corahip_ctx *ctx, corahip_sht_plan *p, int64_t total) {
    (void)total;
    int32_t *d_p3 = nullptr;
    HIP_TRY(hipMalloc((void **)&d_p3, sizeof(int32_t) * p->h_blu3_P.size()));
    HIP_TRY(hipMemcpyAsync(d_p3, p->h_blu3_P.data(), sizeof(int32_t) * p->h_blu3_P.size(), hipMemcpyHostToDevice, ctx->stream));
    const int nb = p->nside - 1;
#define BLU3_LAUNCH(PP)                                                                                                   \
    {                                                                                                                     \
        const size_t shm = sizeof(double2) * (size_t)(fpk<K5_PK_BLU>(PP) + K5_CH_SKEW);                                              \
        HIP_TRY(hipFuncSetAttribute((const void *)blu3_filter_kernel<PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        blu3_filter_kernel<PP><<<nb, CT_T, shm, ctx->stream>>>(p->nside, d_p3, p->d_blu_boff, p->d_blu3_foff, p->d_bchirp, p->d_bfilt3); \
        LAUNCH_CHECK();                                                                                                   \
    }
    BLU3_LAUNCH(1536)
    BLU3_LAUNCH(2560)
    BLU3_LAUNCH(3072)
    BLU3_LAUNCH(3584)
    if (p->nside > 1024) {
        BLU3_LAUNCH(6144)
        BLU3_LAUNCH(8192)
    }
#undef BLU3_LAUNCH
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_p3);
    return 0;
}
