// sht_ringfft.hip - K5 (per-ring phase / alias fold / complex-to-real FFT -> pixels) and K5^T (pixels -> real-to-complex
// FFT -> weighted, phased G_m cells), with their per-class launch loops.  The LDS FFT passes are in sht_internal.h.
#include "sht_internal.h"

#ifndef K5_XCD_PAIR
#define K5_XCD_PAIR 1
#endif

// Persistent workgroups: each loops over work items (ring of the class, NCH consecutive channels),
// all NCH channels transformed together in LDS.  The F_m cells of the NEXT item are fetched into
// registers while the current item is in its FFT passes, so HBM reads overlap the FP64 work and the
// pixel stores of one item drain during the next.  P > 0: Bluestein of length P; P == 0: h = nphi/2
// is a power of two.
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
    constexpr int MC = BLU ? (NCH == 4 ? K5_MC_BLU4 : (K5_MC > 2 ? 2 : K5_MC)) : K5_MC;
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
    // With NCH < 4 the 4 / NCH items of one ring that share every 64-byte cell are consecutive.  Round-robin dispatch
    // puts consecutive workgroups on different XCDs, each L2 then fetches the cells for itself; workgroups b and
    // b + 8 (same XCD, resident together) take such neighbours instead (K5_XCD_PAIR).
    constexpr int SH = NCH == 2 ? 1 : (NCH == 1 ? 2 : 0);             // log2 of the items sharing a cell
    const bool xcd_pair = K5_XCD_PAIR && SH > 0 && (nitems & ((8 << SH) - 1)) == 0 && (gridDim.x & ((8 << SH) - 1)) == 0;
    auto remap = [&](int v) {
        if (!xcd_pair) return v;
        const int slot = v >> 3, xcd = v & 7;
        return (((slot >> SH) * 8 + xcd) << SH) + (slot & ((1 << SH) - 1));
    };
    int vitem = blockIdx.x;
    if (vitem < nitems) prefetch(remap(vitem));
    for (; vitem < nitems; vitem += gridDim.x) {
        const int item = remap(vitem);
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
            for (int j = (noalias ? Lr : 0) + tid; j < flen; j += nt)
#pragma unroll
                for (int c = 0; c < NCH; c++) sm[(size_t)c * bstride + fpad(j)] = make_double2(0.0, 0.0);
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
        if (vitem + (int)gridDim.x < nitems) prefetch(remap(vitem + gridDim.x));
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
    // pixels of the NEXT item are loaded into registers behind the transform of the current one (one workgroup per
    // CU: nothing else hides the HBM latency of the ring's pixels); ANA_PF double2 per channel and thread
    constexpr int ANA_PF = 4;             // covers h <= ANA_PF * blockDim (2048 at 512 threads); longer rings read in place
    double2 pf[NCH][ANA_PF];
    auto prefetch = [&](int it) {
        const int ring = ring_list[it / ngrp];
        const int ch0 = (it % ngrp) * NCH;
        const int h = nphi_a[ring] >> 1;
        const long start = start_a[ring];
#pragma unroll
        for (int u = 0; u < ANA_PF; u++) {
            const int j = tid + u * nt;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                pf[c][u] = make_double2(0.0, 0.0);
                if (j < h && ch0 + c < nvalid)
                    pf[c][u] = *reinterpret_cast<const double2 *>(maps + (size_t)(ch0 + c) * npix + start + 2 * j);
            }
        }
    };
    // (K5_XCD_PAIR as in ringfft_kernel: the items writing the two halves of a 64-byte cell go to one XCD)
    constexpr int SH = NCH == 2 ? 1 : (NCH == 1 ? 2 : 0);
    const bool xcd_pair = K5_XCD_PAIR && SH > 0 && (nitems & ((8 << SH) - 1)) == 0 && (gridDim.x & ((8 << SH) - 1)) == 0;
    auto remap = [&](int v) {
        if (!xcd_pair) return v;
        const int slot = v >> 3, xcd = v & 7;
        return (((slot >> SH) * 8 + xcd) << SH) + (slot & ((1 << SH) - 1));
    };
    if ((int)blockIdx.x < nitems) prefetch(remap(blockIdx.x));
    for (int vitem = blockIdx.x; vitem < nitems; vitem += gridDim.x) {
        const int item = remap(vitem);
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
#pragma unroll
        for (int u = 0; u < ANA_PF; u++) {
            const int j = tid + u * nt;
            if (j < h) {
                const double2 bj = P ? bch[j] : make_double2(1.0, 0.0);
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    double2 zv = pf[c][u];
                    if (P) zv = cmul(make_double2(zv.x, -zv.y), bj);
                    sm[(size_t)c * bstride + fpad(j)] = zv;
                }
            }
        }
        for (int j = tid + ANA_PF * nt; j < h; j += nt) {      // (rings longer than the prefetch window)
            const double2 bj = P ? bch[j] : make_double2(1.0, 0.0);
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                double2 zv = make_double2(0.0, 0.0);
                if (ch0 + c < nvalid) zv = *reinterpret_cast<const double2 *>(maps + (size_t)(ch0 + c) * npix + start + 2 * j);
                if (P) zv = cmul(make_double2(zv.x, -zv.y), bj);
                sm[(size_t)c * bstride + fpad(j)] = zv;
            }
        }
        if (vitem + (int)gridDim.x < nitems) prefetch(remap(vitem + gridDim.x));
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
        // phase e^{-i m phi0} and unpacking twiddle e^{-2 pi i m / n} of this thread's m = tid, tid + nt, ...: one
        // sincospi each, then a rotation by the (exact-argument) step per iteration instead of two sincospi per m
        // (at most ceil(L / nt) = 5 rotations, so the recurrence adds ~1e-16)
        double2 phr, phs, wr_m, ws;
        {
            double sv, cv;
            sincospi(fmod((double)tid * phi0_over_pi, 2.0), &sv, &cv);
            phr = make_double2(cv, -sv);
            sincospi(fmod((double)nt * phi0_over_pi, 2.0), &sv, &cv);
            phs = make_double2(cv, -sv);
            sincospi(2.0 * (double)(tid % n) / (double)n, &sv, &cv);
            wr_m = make_double2(cv, -sv);
            sincospi(2.0 * (double)(nt % n) / (double)n, &sv, &cv);
            ws = make_double2(cv, -sv);
        }
        for (int m = tid; m < Lr; m += nt) {
            const int k = m % n;
            const bool cj = k > h;
            const int kk = cj ? n - k : k;            // 0..h
            const int ka = kk == h ? 0 : kk;          // Z_h := Z_0
            const int kb = kk == 0 ? 0 : h - kk;      // partner h - kk (kk = 0 -> Z_h = Z_0)
            // e^{-2 pi i kk / n}: e^{-2 pi i m / n} has period n in m, and kk = n - k conjugates it
            const double2 w = make_double2(wr_m.x, cj ? -wr_m.y : wr_m.y);
            const double2 ph = make_double2(phr.x * wr, phr.y * wr);   // w_ring area e^{-i m phi0}
            phr = cmul(phr, phs);
            wr_m = cmul(wr_m, ws);
            (void)n_in_table;
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
// K5 over the F_m cells of `inter` for nnu_valid channels -> maps
int sht_ringfft(corahip_ctx *ctx, const corahip_sht_plan *p, const double *inter, int nnu_chunk_pad, int nnu_valid,
                       double *maps) {
    {
        StageTimer t(ctx, "ringfft");
        const int G = nnu_chunk_pad / 4;
        static const bool class_times = getenv("CORAHIP_K5_TIMES") != nullptr;   // diagnostics: per-class ms on stderr
        // the belt and the largest Bluestein class run side by side (two streams, co-resident workgroups)
        const corahip_sht_plan::ring_class *pb = nullptr, *pc = nullptr;
        for (const auto &c : p->classes) {
            if (c.P == 0 && c.N > 0) pb = &c;
            if (c.P == 4096 && c.P3 == 0) pc = &c;
        }
        bool paired = false;
        if (pb && pc && !class_times) {
            const int rcp = sht_ringfft_ct_pair(ctx, p, *pb, *pc, inter, G, nnu_valid, maps, &paired);
            if (rcp) return rcp;
        }
        // The class launches alternate between two streams: every class is a persistent grid that fills the chip, so the
        // two kernels in flight run one after the other EXCEPT for their tails - the workgroups of the next class start on
        // the CUs the finishing one frees (12 launches, 0.1-0.3 ms of tail each on one stream).  CORAHIP_K5_ONE_STREAM=1
        // or the per-class timing switch keep everything on the context's stream.
        static const bool one_stream = getenv("CORAHIP_K5_ONE_STREAM") != nullptr;
        const bool two = !one_stream && !class_times && !K5_STAMPS && p->classes.size() > 1;
        hipStream_t const main_stream = ctx->stream;
        struct Restore {          // ctx->stream is switched per launch below: back to the caller's on every exit path -
            corahip_ctx *c;       // and the caller's stream waits for whatever was put on the second one (an error return
            hipStream_t s;        // in the middle of the loop must not leave class kernels running on stream2 unordered
            bool forked = false;  // against the caller's next launches, or against its freeing of `inter` / `maps`)
            ~Restore() {
                c->stream = s;
                if (forked && (hipEventRecord(c->ev_join, c->stream2) != hipSuccess || hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess))
                    (void)hipStreamSynchronize(c->stream2);
            }
        } restore{ctx, main_stream};
        if (two) {
            int rc2 = sht_second_stream(ctx);
            if (rc2) return rc2;
            HIP_TRY(hipEventRecord(ctx->ev_fork, main_stream));
            HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            restore.forked = true;
        }
        int launch_no = 0;
        for (const auto &c : p->classes) {
            if (paired && (&c == pb || &c == pc)) continue;
            if (two) ctx->stream = (launch_no++ & 1) ? ctx->stream2 : main_stream;   // (restored below; every launch of the loop uses ctx->stream)
            hipEvent_t ce0 = nullptr, ce1 = nullptr;
            if (class_times) {
                (void)hipEventCreate(&ce0);
                (void)hipEventCreate(&ce1);
                (void)hipEventRecord(ce0, ctx->stream);
            }
            bool took = false;   // compile-time kernel for this class?
            const int rct = sht_ringfft_ct(ctx, p, c, inter, G, nnu_valid, maps, &took);
            if (rct) return rct;   // (Restore puts the caller's stream back)
            const size_t shm = sizeof(double2) * ((size_t)c.nch * c.bstride + TWL_ENTRIES(p->pmax));
            const long nitems = (long)c.count * ((nnu_valid + c.nch - 1) / c.nch);
            const int per_cu = std::max<int>(1, (int)((160 * 1024) / std::max<size_t>(shm, 1)));
            const int k5_threads = c.threads ? c.threads : K5_THREADS;
            // (the short classes run narrow workgroups - see the plan - and more of them per CU)
            dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * std::min(per_cu, k5_threads <= 128 ? 8 : 4)));
#define RINGFFT_LAUNCH(NCH, BLU)                                                                                     \
    HIP_TRY(hipFuncSetAttribute((const void *)ringfft_kernel<NCH, BLU>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                160 * 1024));                                                                   \
    ringfft_kernel<NCH, BLU><<<grid, k5_threads, shm, ctx->stream>>>(c.d_list, c.count, p->nside, p->lmax, G, nnu_valid, p->npix,     \
                                                         p->d_nphi, p->d_start, p->d_phi0, inter, maps, p->d_tw, \
                                                         p->pmax, p->d_blu_P, p->d_blu_boff, p->d_blu_foff,      \
                                                         p->d_bchirp, p->d_bfilt, c.bstride, p->d_mcut)
            if (took) {
            } else if (c.P == 0) {
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
                fprintf(stderr, "K5 class P=%d (length %d) rings=%d: %.3f ms\n", c.P, c.P3 ? c.P3 : (c.P ? c.P : c.N), c.count, ms);
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
        ctx->stream = main_stream;
        if (two) {
            HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
            HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
            restore.forked = false;                 // (joined here: nothing left for the guard)
        }
    }
    return 0;
}

int sht_ringana(corahip_ctx *ctx, const corahip_sht_plan *p, const double *maps, int nnu, int nnu_pad8,
                const double *ring_w, double *inter) {
    StageTimer t(ctx, "ringana");
    const int G = nnu_pad8 / 4;
    const int k5_threads = K5_THREADS;
    // The class launches alternate between two streams, as in sht_ringfft: every class is a persistent grid that fills
    // the chip, so two kernels in flight run one after the other except for their tails - the workgroups of the next
    // class start on the CUs the finishing one frees (CORAHIP_K5_ONE_STREAM=1: everything on the context's stream).
    static const bool one_stream = getenv("CORAHIP_K5_ONE_STREAM") != nullptr;
    static const bool class_times = getenv("CORAHIP_K5_TIMES") != nullptr;   // diagnostics: per-class ms on stderr
    const bool two = !one_stream && !class_times && p->classes.size() > 1;
    hipStream_t const main_stream = ctx->stream;
    struct Restore {          // (as in sht_ringfft: the caller's stream back, and joined with the second one, on every exit path)
        corahip_ctx *c;
        hipStream_t s;
        bool forked = false;
        ~Restore() {
            c->stream = s;
            if (forked && (hipEventRecord(c->ev_join, c->stream2) != hipSuccess || hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess))
                (void)hipStreamSynchronize(c->stream2);
        }
    } restore{ctx, main_stream};
    if (two) {
        int rc2 = sht_second_stream(ctx);
        if (rc2) return rc2;
        HIP_TRY(hipEventRecord(ctx->ev_fork, main_stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        restore.forked = true;
    }
    int launch_no = 0;
    for (const auto &c : p->classes) {
        if (two) ctx->stream = (launch_no++ & 1) ? ctx->stream2 : main_stream;
        hipEvent_t ce0 = nullptr, ce1 = nullptr;
        if (class_times) {
            (void)hipEventCreate(&ce0);
            (void)hipEventCreate(&ce1);
            (void)hipEventRecord(ce0, ctx->stream);
        }
        bool took = false;   // compile-time kernel for this class?
        const int rct = sht_ringana_ct(ctx, p, c, maps, nnu, nnu_pad8, ring_w, G, inter, &took);
        if (rct) return rct;
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
        if (took) {
        } else if (c.nch == 4) { RINGANA_LAUNCH(4); }
        else if (c.nch == 2) { RINGANA_LAUNCH(2); }
        else { RINGANA_LAUNCH(1); }
#undef RINGANA_LAUNCH
        LAUNCH_CHECK();
        if (class_times) {
            float ms = 0.f;
            (void)hipEventRecord(ce1, ctx->stream);
            (void)hipEventSynchronize(ce1);
            (void)hipEventElapsedTime(&ms, ce0, ce1);
            fprintf(stderr, "K5^T class P=%d (length %d) nch=%d rings=%d: %.3f ms\n", c.P, c.P3 ? c.P3 : (c.P ? c.P : c.N), c.nch, c.count, ms);
            (void)hipEventDestroy(ce0);
            (void)hipEventDestroy(ce1);
        }
    }
    ctx->stream = main_stream;
    if (two) {
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
        HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        restore.forked = false;
    }
    return 0;
}
