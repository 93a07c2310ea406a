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

#include <type_traits>

#include "rng_dev.h"

// The device stream: the normals of (l, c = re/im, nu', m) and (.., m+1), m even, are the two Box-Muller outputs
// of Philox counter {lo = m/2, hi = l*2F + c*F + nu'} under key = seed, built from the four output words as
// rng_dev.h describes (u1 from 52 bits of (r0, r1), the angle from 60 bits of (r2, r3)).
// A value depends only on (seed, l, c, nu', m): the same for any number of GPUs, and the same whether it is
// materialised in HBM (normals_kernel, stream-order layout) or generated inside K3.  oracle/philox.py
// restates the stream in numpy.
__device__ static inline double2 philox_normal_pair(uint64_t seed, int l, int F, int c, int nup, int mpair,
                                                    const double2 *lg = RNG_LOG_TAB, const double2 *sc = RNG_SC_TAB) {
    const uint64_t ctr = ((uint64_t)((uint32_t)l * 2u * (uint32_t)F + (uint32_t)(c * F + nup)) << 32) | (uint32_t)mpair;
    return philox_boxmuller(ctr, seed, lg, sc);
}

// one thread per (l, c, nu', m-pair): writes the stream-order buffer  g[F l(l+1) + c F(l+1) + nu'(l+1) + m]
__global__ void normals_kernel(uint64_t seed, int lmax, int F, double *__restrict__ g) {
    const int l = blockIdx.y;
    const int lp1 = l + 1, npair = (lp1 + 1) >> 1;
    const long n = 2L * F * npair;
    double *gl = g + (size_t)F * l * lp1;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int mp = (int)(q % npair);
        const int cn = (int)(q / npair);  // c*F + nu'
        const double2 v = philox_normal_pair(seed, l, F, cn / F, cn % F, mp);
        double *dst = gl + (size_t)cn * lp1 + 2 * mp;
        dst[0] = v.x;
        if (2 * mp + 1 < lp1) dst[1] = v.y;
    }
}

// ------------------------------------------------------------------------------------
// K3: per-l GEMM on FP64 MFMA
// ------------------------------------------------------------------------------------
#ifndef DRAW_ABLATE
#define DRAW_ABLATE 0  // diagnostic builds of the fused-RNG kernel: 1 no RNG, 2 no MFMA, 3 no a_lm stores, 4 no staging of T
#endif
#ifndef DRAW_KK_UNROLL
#define DRAW_KK_UNROLL 1   // unroll factor of the k-step loop of a chunk in the fused-RNG kernel
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
            const double *__restrict__ g, int lmax, int F, int nu0, int nnu, int Gout, double *__restrict__ alm) {
    constexpr int NC = 16 * NCT;
    constexpr int STRIDE = DRAW_KC + 2;  // doubles per channel row: 272 B, so 16 consecutive rows hit 16 distinct 16-B slots
    extern __shared__ __attribute__((aligned(16))) double lds[];  // Bs[n][k] = T_l[nu0+col0+n][k0+k], [NC][STRIDE]

    const int l = blockIdx.x;
    const int nrow = 2 * (l + 1);
    const int row0 = blockIdx.y * DRAW_ROWS;
    if (row0 >= nrow) return;
    const int col0 = blockIdx.z * NC;  // local channel index of first column
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    const int lp1 = l + 1;

    // stream offset of this l: sum_{l'<l} 2 F (l'+1) = F l (l+1)
    const double *gl = g + (size_t)F * l * (l + 1);
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
// K3 with the normals generated in registers (device-RNG mode): no 8.6 GB normal buffer is written or
// read.  Wave tile = 32 rows (16 m-pairs of one c: the two Box-Muller outputs feed the even-m and the
// odd-m row tile) x 16*NCT channels; workgroup = 4 waves = 128 rows.
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
#define DRAW_STAMPS 0   // diagnostic build (make k3stamps): s_memtime per phase, summed over the waves that have rows
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

template <int NCT>
__global__ void __launch_bounds__(256, 2)
draw_rng_kernel(const double *__restrict__ T, size_t t_ldl, int t_row0, const int32_t *__restrict__ info,
                const double *__restrict__ zeros, uint64_t seed, int lmax, int F, int nu0, int nnu, int Gout,
                double *__restrict__ alm) {
    constexpr int NC = 16 * NCT;
    constexpr int ROWD = DRAW_KC;            // doubles per channel row in LDS: 256 B, unpadded (DMA is lane-linear)
    constexpr int BUF = NC * ROWD;           // doubles per stage
    // Bs[n][slot' = slot ^ (n & 15)][2]: the 16-byte slots of a row are XOR-swizzled with the row number
    // (applied on the DMA source address), so that 16 rows read at the same k hit 16 distinct slots
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [2][NC][ROWD]
    __shared__ double2 lg_s[257], sc_s[256];   // LDS copies of the Box-Muller tables (rng_dev.h): 8 KB

    // workgroup = 64 values of m x (re, im): waves 0,1 draw the real parts, waves 2,3 the imaginary parts of the
    // SAME m, so that both halves of every 64-byte a_lm cell ([re x4 | im x4]) are written by one workgroup within
    // a short time and merge in L2 (with re and im in different workgroups every cell reached HBM as two 32-byte
    // partial writes: the kernel was bound by that, not by the RNG or the MFMAs - make DRAW_ABLATE builds)
    const int l = blockIdx.x;
    const int lp1 = l + 1;
    const int mb = blockIdx.y;
    const int col0 = blockIdx.z * NC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c_of = wave >> 1;
    const int ri = lane & 15, kq = lane >> 4;
    if (mb * 64 >= lp1) return;                 // no m of this l in the block (half the [l][m-block] grid): uniform, before any barrier / DMA
#if DRAW_STAMPS
    unsigned long long d_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, d_last;
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); d_last = _t; }
#endif
    const int m0 = mb * 64 + (wave & 1) * 32;   // first m of this wave
    const int mpair = (m0 >> 1) + ri;           // this lane's m-pair: rows m = 2 mpair, 2 mpair + 1

    const double *Tl = T + (size_t)l * t_ldl - (size_t)t_row0 * F;  // row nu of T_l at Tl + nu F (rows < t_row0 never read)
    const bool dense = (info == nullptr) || (info[l] != 0);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;

    d4_t acc0[NCT], acc1[NCT];
#pragma unroll
    for (int t = 0; t < NCT; t++) {
        acc0[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
        acc1[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    }
    const int base0 = nu0 + col0;                // first channel of the block
    const int kmax = dense ? F : min(F, base0 + NC);
    const int nchunk = (kmax + DRAW_KC - 1) / DRAW_KC;
    const bool tri_tail = !dense && (base0 % DRAW_KC) == 0;   // the chunks across the block's own channels drop tiles (below)
    const bool wave_has_rows = m0 < lp1;
    const bool full_k = (F % DRAW_KC) == 0;      // rows of T_l are whole 256-byte runs

    // stage chunk c of T_l (rows nu0+col0 .. +NC-1, k in [c KC, c KC + KC)) into buffer c & 1:
    // each wave-instruction moves 4 rows x 16 slots
    auto stage = [&](int c) {
        const int k0 = c * DRAW_KC;
#pragma unroll
        for (int it = 0; it < NC / 16; it++) {      // NC/4 row-quads over 4 waves
            const int rq = wave + 4 * it;           // row quad index
            if (rq >= NC / 4) break;
            // triangular factors: a 16-row tile whose rows all have nu < k0 is zero in this chunk and its MFMAs are
            // skipped below (the same test at kbase >= k0), so its four quads need not be staged at all (a third of the
            // LDS-DMA volume); tile granularity, not quad: a partly-zero tile is still multiplied as a whole
            if (tri_tail && k0 > base0 + 16 * (rq >> 2) + 15) continue;
            const int n = 4 * rq + (lane >> 4);     // row of this lane
            const int slot_dst = lane & 15;
            const int slot_src = slot_dst ^ (n & 15);
            const int nu = nu0 + col0 + n;
            const int k = k0 + 2 * slot_src;
            const double *src = zeros;  // F is even on this path (host wrapper), so k + 1 < F whenever k < F
            if (col0 + n < nnu && nu < F && k + 1 < F) src = Tl + (size_t)nu * F + k;
#if DRAW_ABLATE != 4   // diagnostic 4: no staging of T
            // (LDS-DMA moves only ~10 B/clk per CU - MI355X_MICROARCH.md "ldsdma-fill" - but staging through registers,
            //  global_load_dwordx4 + ds_write_b128 committed before the next barrier, was slower still: 14.0 vs 10.1 ms)
            draw_glds16(src, lds_base + (unsigned)(((c & 1) * BUF + 4 * rq * ROWD) * sizeof(double)));
#else
            (void)src;
#endif
        }
    };
    (void)full_k;

    lg_s[threadIdx.x] = RNG_LOG_TAB[threadIdx.x];                          // (visible after the first chunk's barrier)
    sc_s[threadIdx.x] = RNG_SC_TAB[threadIdx.x];
    if (threadIdx.x == 0) lg_s[256] = RNG_LOG_TAB[256];
    // k-steps of one half (16 nu' = 4 k-steps) of the chunk in buffer `sb`, multiplying the tiles TMIN .. NCT-1
    auto half_steps = [&](auto tmin_c, const double *sb, int k0, int half) {
        constexpr int TMIN = decltype(tmin_c)::value;
#pragma unroll DRAW_KK_UNROLL
        for (int kk = 4 * half; kk < 4 * half + 4; kk++) {
            const int kp = k0 + 4 * kk + kq;
            const int kl = 4 * kk + kq;          // k within the chunk
            // all B operands of the k-step are read up front (one address + immediate offsets)
            double bv[NCT];
            const double *brow = sb + ri * ROWD + 2 * ((kl >> 1) ^ ri) + (kl & 1);
#if DRAW_BEARLY        // the reads in front of the generator chain (their latency behind it), pinned by a scheduling barrier
#pragma unroll
            for (int t = TMIN; t < NCT; t++) bv[t] = brow[16 * t * ROWD];
            __builtin_amdgcn_sched_barrier(0);
#endif
#if DRAW_ABLATE == 1   // diagnostic: no RNG
            double2 a = make_double2(1.0 + kp, 0.5 * mpair);
#else
            // (rows past l and nu' >= F are generated like any other: their products meet staged zeros or are never
            //  stored - no exec masking around the chain)
            double2 a = philox_normal_pair(seed, l, F, c_of, kp, mpair, lg_s, sc_s);
#endif
#if DRAW_STAMPS
            asm volatile("" ::"v"(a.x), "v"(a.y));
            DSTAMP(3);                   // normals of the k-step
#endif
#if !DRAW_BEARLY
#pragma unroll
            for (int t = TMIN; t < NCT; t++) bv[t] = brow[16 * t * ROWD];
#endif
#pragma unroll
            for (int t = TMIN; t < NCT; t++) {
#if DRAW_ABLATE == 2   // diagnostic: no MFMA
                asm volatile("" ::"v"(a.x), "v"(a.y), "v"(bv[t]));
#else
                acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bv[t], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bv[t], acc1[t], 0, 0, 0);
#endif
            }
            DSTAMP(4);                   // B reads + MFMA issue of the k-step
        }
    };
    auto chunk_begin = [&](int c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // chunk c landed; everyone is done with chunk c-1
        DSTAMP(1);                       // wait for the stage + barrier
        if (c + 1 < nchunk) stage(c + 1);
        DSTAMP(2);                       // issue of the next stage
    };

    stage(0);
    DSTAMP(0);                           // prologue + first stage issue
    // Tile t (channels base0 + 16 t .. + 15) of a triangular factor is zero for nu' > base0 + 16 t + 15.  With base0 a
    // multiple of the chunk length the chunks below the block's own channels take every tile and the chunks across them
    // drop one tile per half: that tail is unrolled so that every half knows its tiles at compile time (the per-tile
    // tests inside the k-steps were 45 scalar instructions per k-step, issue time next to the MFMAs; a run-time
    // dispatch per half made the register allocator copy the accumulators between the cases).  A tile that is kept is
    // multiplied as a whole: the entries above the diagonal are stored zeros.  Any other base0 (uneven channel shards)
    // takes every tile up to kmax - correct for the same reason, just not minimal.
    const int c_full = tri_tail ? min(nchunk, base0 / DRAW_KC) : nchunk;
    int c = 0;
    for (; c < c_full; c++) {
        chunk_begin(c);
        if (!wave_has_rows) continue;
        const double *sb = lds + (c & 1) * BUF;
        half_steps(std::integral_constant<int, 0>{}, sb, c * DRAW_KC, 0);
        if (c * DRAW_KC + 16 < kmax) half_steps(std::integral_constant<int, 0>{}, sb, c * DRAW_KC, 1);
    }
    draw_static_for<(NC + DRAW_KC - 1) / DRAW_KC>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (c >= nchunk) return;         // (uniform; also the dense / unaligned case, where c == nchunk here)
        chunk_begin(c);
        if (wave_has_rows) {
            const double *sb = lds + (c & 1) * BUF;
            half_steps(std::integral_constant<int, (2 * J < NCT ? 2 * J : NCT)>{}, sb, c * DRAW_KC, 0);
            if (2 * J + 1 < NCT && c * DRAW_KC + 16 < kmax)
                half_steps(std::integral_constant<int, (2 * J + 1 < NCT ? 2 * J + 1 : NCT)>{}, sb, c * DRAW_KC, 1);
        }
        c++;
    });
    if (!wave_has_rows) return;
    // epilogue: C row i of tile0 is m = m0 + 2 i, of tile1 m = m0 + 2 i + 1; 1/sqrt(2) of complex_std_normal.
    // Lanes (ri, ri ^ 1) hold adjacent channels of the same rows: the even lane takes the m-even row of BOTH channels, the
    // odd lane the m-odd row (one DPP swap per value), so every store is 16 bytes instead of 8 - the a_lm stores are
    // issue-bound (~7 B/clk per CU for 8-byte lanes: 13 % of the kernel's wave cycles by the phase stamps).
    const double sc = 0.70710678118654752440;
    const bool odd = ri & 1;
    auto swap1 = [](double v) {            // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
        return __hiloint2double(hi, lo);
    };
#pragma unroll
    for (int t = 0; t < NCT; t++) {
        const int col = col0 + 16 * t + (ri & ~1);          // first channel of the lane pair
        const bool col_ok = col < 4 * Gout;                 // (4 Gout is a multiple of 4: both channels or none)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = kq + 4 * r;
            const double mine = (odd ? acc1[t][r] : acc0[t][r]) * sc;      // this lane's channel of the row it stores
            const double give = (odd ? acc0[t][r] : acc1[t][r]) * sc;      // the partner's row
            const double got = swap1(give);                                  // the partner's channel of MY row
            const int m = m0 + 2 * i + (odd ? 1 : 0);
            if (col_ok && m < lp1) {
                const long idx = (long)m * (2 * lmax + 1 - m) / 2 + l;
#if DRAW_ABLATE == 3   // diagnostic: no a_lm stores
                if (mine == 1.2345e300)
#endif
                *reinterpret_cast<double2 *>(alm + ((size_t)idx * Gout + (col >> 2)) * 8 + c_of * 4 + (col & 3)) =
                    odd ? make_double2(got, mine) : make_double2(mine, got);
            }
        }
    }
#if DRAW_STAMPS
    DSTAMP(5);                           // epilogue (scale + a_lm stores issued)
    if (lane == 0) {
        for (int k = 0; k < 6; k++) atomicAdd(&g_draw_stamps[k], d_acc[k]);
        atomicAdd(&g_draw_stamps[7], 1ull);
    }
#endif
}

template <int NCT>
static int launch_draw_rng(corahip_ctx *ctx, const double *T, size_t t_ldl, int t_row0, const int32_t *info,
                           uint64_t seed, int lmax, int F, int nu0, int nnu, int Gout, double *alm) {
    constexpr int NC = 16 * NCT;
    const size_t shm = sizeof(double) * 2 * NC * DRAW_KC;
    HIP_TRY(hipFuncSetAttribute((const void *)draw_rng_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    double *zeros = nullptr;
    int rc = corahip_ctx_scratch(ctx, 3, 4096, (void **)&zeros);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(zeros, 0, 4096, ctx->stream));
    dim3 grid(lmax + 1, (lmax + 1 + 63) / 64, (4 * Gout + NC - 1) / NC);
    draw_rng_kernel<NCT><<<grid, 256, shm, ctx->stream>>>(T, t_ldl, t_row0, info, zeros, seed, lmax, F, nu0, nnu, Gout,
                                                          alm);
    LAUNCH_CHECK();
#if DRAW_STAMPS
    {
        unsigned long long hs[8], z[8] = {0};
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_draw_stamps), sizeof(hs)));
        const double per = 1.0 / (double)std::max<unsigned long long>(hs[7], 1);
        fprintf(stderr, "K3 NCT=%d waves=%llu: cycles/wave  prologue %.0f wait+barrier %.0f stage-issue %.0f rng %.0f reads+mfma %.0f epilogue %.0f\n",
                NCT, hs[7], hs[0] * per, hs[1] * per, hs[2] * per, hs[3] * per, hs[4] * per, hs[5] * per);
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
static int launch_draw(corahip_ctx *ctx, const double *T, size_t t_ldl, int t_row0, const int32_t *info,
                       const double *g, int lmax, int F, int nu0, int nnu, int Gout, double *alm) {
    constexpr int NC = 16 * NCT;
    const size_t shm = sizeof(double) * NC * (DRAW_KC + 2);
    HIP_TRY(hipFuncSetAttribute((const void *)draw_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    dim3 grid(lmax + 1, (2 * (lmax + 1) + DRAW_ROWS - 1) / DRAW_ROWS, (4 * Gout + NC - 1) / NC);
    draw_kernel<NCT><<<grid, 256, shm, ctx->stream>>>(T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, Gout, alm);
    LAUNCH_CHECK();
    return 0;
}

extern "C" {

int corahip_normals_philox(corahip_ctx *ctx, uint64_t seed, int lmax, int F, double *g) {
    ARG_CHECK(ctx != nullptr && g != nullptr && lmax >= 0 && F >= 1);
    StageTimer t(ctx, "normals");
    dim3 grid(std::max(1, std::min(64, (F * (lmax + 2) / 2 + 255) / 256)), lmax + 1);
    normals_kernel<<<grid, 256, 0, ctx->stream>>>(seed, lmax, F, g);
    LAUNCH_CHECK();
    return 0;
}

static int draw_host_stream(corahip_ctx *ctx, const double *T, size_t t_ldl, int t_row0, const int32_t *info,
                            const double *g, int lmax, int F, int nu0, int nnu, double *alm_dev) {
    StageTimer t(ctx, "draw");
    const int Gout = (nnu + 3) / 4;
    const int ncol = 4 * Gout;
    if (ncol <= 16) return launch_draw<1>(ctx, T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 32) return launch_draw<2>(ctx, T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 64) return launch_draw<4>(ctx, T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 128) return launch_draw<8>(ctx, T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, Gout, alm_dev);
    return launch_draw<16>(ctx, T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, Gout, alm_dev);
}

static int draw_philox(corahip_ctx *ctx, const double *T, size_t t_ldl, int t_row0, const int32_t *info,
                       uint64_t seed, int lmax, int F, int nu0, int nnu, double *alm_dev) {
    if (F & 1) {
        // odd F: a 16-byte LDS-DMA piece would straddle the end of a T row; materialise the (identical)
        // device stream and use the generic kernel instead
        double *g = nullptr;
        int rc = corahip_ctx_scratch(ctx, 1, sizeof(double) * 2 * (size_t)F * nalm_of(lmax), (void **)&g);
        if (rc) return rc;
        if ((rc = corahip_normals_philox(ctx, seed, lmax, F, g))) return rc;
        return draw_host_stream(ctx, T, t_ldl, t_row0, info, g, lmax, F, nu0, nnu, alm_dev);
    }
    StageTimer t(ctx, "draw");
    const int Gout = (nnu + 3) / 4;
    const int ncol = 4 * Gout;
    if (ncol <= 16) return launch_draw_rng<1>(ctx, T, t_ldl, t_row0, info, seed, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 32) return launch_draw_rng<2>(ctx, T, t_ldl, t_row0, info, seed, lmax, F, nu0, nnu, Gout, alm_dev);
    if (ncol <= 64) return launch_draw_rng<4>(ctx, T, t_ldl, t_row0, info, seed, lmax, F, nu0, nnu, Gout, alm_dev);
    return launch_draw_rng<8>(ctx, T, t_ldl, t_row0, info, seed, lmax, F, nu0, nnu, Gout, alm_dev);
}

int corahip_draw_alm_philox(corahip_ctx *ctx, const double *T, const int32_t *info, uint64_t seed, int lmax, int F,
                            int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_philox(ctx, T, (size_t)F * F, 0, info, seed, lmax, F, nu0, nnu, alm_dev);
}

int corahip_draw_alm_philox_rows(corahip_ctx *ctx, const double *T_rows, const int32_t *info, uint64_t seed, int lmax,
                                 int F, int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T_rows != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_philox(ctx, T_rows, (size_t)nnu * F, nu0, info, seed, lmax, F, nu0, nnu, alm_dev);
}

int corahip_draw_alm(corahip_ctx *ctx, const double *T, const int32_t *info, const double *g, int lmax, int F,
                     int nu0, int nnu, double *alm_dev) {
    ARG_CHECK(ctx != nullptr && T != nullptr && g != nullptr && alm_dev != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    return draw_host_stream(ctx, T, (size_t)F * F, 0, info, g, lmax, F, nu0, nnu, alm_dev);
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
