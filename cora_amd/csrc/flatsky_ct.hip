// flatsky_ct.hip - the flat-sky line transforms (RandomField.getfield, fftutil.rfftn / irfftn: cora/core/gaussianfield.py:
// 102-120, cora/util/fftutil.py:64-87; the cubes of RedshiftCorrelation.realisation, cora/signal/corr.py:562-770) on the
// compile-time FFT passes of the ring transforms (fft_ct.h; sht_ringfft_ct.hip explains what the fixed shapes buy):
//   linec2r_ct / liner2c_ct      contiguous axis, even real length 2 N with a scheduled N (2^k, 3 * 2^k)
//   linec2c_ct                   strided axis of a scheduled length, 8 - 16 neighbouring lines per item; GEN: the input is
//                                generated where it is committed (corahip_randomfield_irfftn)
//   lineblu_c2c_ct / _c2r_ct / _r2c_ct   every other length n <= 2048: Bluestein at a length out of {2, 3, 5} x 2^k >= 2 n - 1,
//                                filter made at plan time by flat_blu_filter_kernel (flat_blu_plan)
// flatsky.hip calls flat_c2r_ct / flat_r2c_ct / flat_c2c_ct / flat_blu_c2c_ct / flat_blu_real_ct first and keeps its
// generic line kernel (radix-4 LDS stages, power-of-two Bluestein) for what they decline: odd real lengths, lengths
// below 17 or above 2048, contiguous complex transforms; CORAHIP_FLAT_GENERIC=1 forces it (A/B, tests/test_flatsky.py).
#include "fft_ct.h"
#include "rng_dev.h"

// ------------------------------------------------------------------------------------
// Flat-sky fields (cora/core/gaussianfield.py:102-120, numpy.fft.irfftn's last axis): the half-complex -> real transform
// of EVEN length 2 N along the contiguous axis as ringfft_direct_ct's three passes - Hermitian step fused into the
// first, pixel-order store fused into the last - on NCH adjacent lines per item.  in: lines of N + 1 complex bins, out:
// lines of 2 N reals, out = scale * irfft (numpy semantics: the imaginary parts of the DC and Nyquist bins are ignored).
// The generic line kernel (flatsky.hip: radix-4 LDS stages, a barrier each) ran this pass at 3.1 TB/s.
// ------------------------------------------------------------------------------------
template <int N, int NCH, int T>
__global__ void __launch_bounds__(T)
linec2r_ct(const double2 *in, double *out, long nlines, double scale) {
    constexpr int PK = 1;                    // (strides down to 16 elements: the one-slot-per-16 padding is additive for them)
    constexpr int R0 = Sch<N>::R0, R1 = Sch<N>::R1, R2 = Sch<N>::R2;
    static_assert((R0 == 16 || R0 == 12) && R1 == 16 && N == R0 * R1 * R2, "digit map of the fused store assumes R0 x 16 x R2");
    constexpr int Q0 = N / R0;
    static_assert(Q0 % 16 == 0, "first-pass stride must be a multiple of the padding period");
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    constexpr int U = NCH * N / T;           // bins 0 .. N-1 per thread (the Nyquist bins: one more load on NCH threads)
    static_assert((NCH * N) % T == 0, "whole loads per thread");
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const long nitems = (nlines + NCH - 1) / NCH;

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
    double2 pf[U], pfn;
    auto prefetch = [&](long item, int tid) {
        const long line0 = item * NCH;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const long line = min(line0 + e / N, nlines - 1);
            pf[u] = in[line * (N + 1) + e % N];
        }
        pfn = in[min(line0 + (tid & (NCH - 1)), nlines - 1) * (N + 1) + N];
    };
    long vitem = blockIdx.x;
    if (vitem < nitems) prefetch(vitem, tid0);
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;                                   // opaque per item: see ringfft_direct_ct
        asm volatile("" : "+v"(tid));
        const long line0 = vitem * NCH;
        __syncthreads();                                  // previous item's LDS reads are done
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e / N, k = e % N;
            double2 v = pf[u];
            if (k == 0) v.y = 0.0;
            sm[c * BS + fpad(k)] = v;
        }
        if (tid < NCH) sm[tid * BS + fpad(N)] = make_double2(pfn.x, 0.0);
        prefetch(min(vitem + (long)gridDim.x, nitems - 1), tid);
        __syncthreads();
        // ---- pass 1 with the Hermitian step (ringfft_direct_ct): Z_k = (X_k + conj X_{N-k}) + i w^k (X_k - conj X_{N-k})
        {
            constexpr int TOT = NCH * Q0;
            constexpr int IT = (TOT + T - 1) / T;
            double2 x[IT][R0];
            double2 wh = wH;
            asm volatile("" : "+v"(wh.x), "+v"(wh.y));
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx / Q0, j0 = idx & (Q0 - 1);
                const double2 *pa = sm + ch * BS + fpad(j0);
                const double2 *pb = sm + ch * BS + fpad(Q0 - j0);
#pragma unroll
                for (int r = 0; r < R0; r++) {
                    const double2 xa = pa[fpc(r * Q0)];
                    const double2 xb = pb[fpc((R0 - 1 - r) * Q0)];
                    const double2 w = cmul(wh, R0 == 16 ? make_double2(kCos16[r % 16], kSin16[r % 16]) : make_double2(kCos12[r % 12], kSin12[r % 12]));
                    const double2 sum = make_double2(xa.x + xb.x, xa.y - xb.y);
                    const double2 dif = make_double2(xa.x - xb.x, xa.y + xb.y);
                    const double2 t = cmul(dif, w);
                    x[it][r] = make_double2(sum.x - t.y, sum.y + t.x);
                }
            }
            __syncthreads();                              // every raw X has been read
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
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
        // ---- last pass (radix R2 on contiguous elements) with the store: butterfly t = 16 k0 + k1 holds the natural
        //      indices k0 + R0 k1 + 16 R0 r; R0 = 16: eight consecutive lanes store 128 contiguous bytes (lane bits as in
        //      ringfft_direct_ct), R0 = 12: lanes along k0
        {
            constexpr int NB2 = R0 * R1;                  // butterflies per line
            constexpr int TOT = NCH * NB2;
            constexpr int IT = (TOT + T - 1) / T;
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx / NB2, q = idx - ch * NB2;
                const int k0 = R0 == 16 ? ((q & 7) | ((q >> 3) & 8)) : q % R0;
                const int k1 = R0 == 16 ? (((q >> 3) & 7) | ((q >> 4) & 8)) : q / R0;
                const double2 *p = sm + ch * BS + fpad((k0 * 16 + k1) * R2);
                double2 x[R2];
#pragma unroll
                for (int r = 0; r < R2; r++) x[r] = p[fpc(r)];
                DftR<R2, 1>::run(x);
                if (line0 + ch < nlines) {
                    double *o = out + (line0 + ch) * (2L * N) + 2 * (k0 + R0 * k1);
#pragma unroll
                    for (int r = 0; r < R2; r++) *reinterpret_cast<double2 *>(o + 2 * NB2 * r) = make_double2(x[r].x * scale, x[r].y * scale);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Flat-sky fields: complex -> complex along a STRIDED axis (element j of line (o, i) at ((o N + j) inner + i)), NCH
// lines that are neighbours along the contiguous axis per item: rows of 16 NCH contiguous bytes (the generic kernel's
// 64 KB tiles hold four lines of 1024: 64-byte segments, 2.8 TB/s).  Natural order in, three DIF passes, the store
// reads through the digit map.  GEN: the input is generated where it is committed - element e = kweight[e] (N(0,1) +
// i N(0,1)), the Box-Muller pair of Philox counter e (randomfield_draw_kernel's values: RandomField.getfield,
// cora/core/gaussianfield.py:115-119) - `in` is then the real k-weight array.  In place allowed.
// ------------------------------------------------------------------------------------
template <int N, int NCH, int T, int SIGN, bool GEN>
__global__ void __launch_bounds__(T)
linec2c_ct(const double *in, double2 *out, long nouter, long inner, double scale, uint64_t seed) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<N>::R0, R1 = Sch<N>::R1, R2 = Sch<N>::R2;
    static_assert((R0 == 16 || R0 == 12) && R1 == 16 && N == R0 * R1 * R2, "R0 x 16 x R2");
    constexpr int Q0 = N / R0;
    static_assert(Q0 % 16 == 0 && T % Q0 == 0, "per-thread twiddles");
    constexpr int BS = fpc(N) + K5_CH_SKEW;
    constexpr int U = NCH * N / T;
    static_assert((NCH * N) % T == 0 && (NCH & (NCH - 1)) == 0, "whole loads per thread");
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    double2 *lg_l = sm + NCH * BS, *sc_l = lg_l + 257;       // GEN: the generator's tables (rng_dev.h)
    const int tid0 = threadIdx.x;
    const double2 *in2 = reinterpret_cast<const double2 *>(in);
    const long chunks = (inner + NCH - 1) / NCH;
    const long ntiles = nouter * chunks;
    if (GEN) {
        for (int k = tid0; k < 257; k += T) lg_l[k] = RNG_LOG_TAB[k];
        for (int k = tid0; k < 256; k += T) sc_l[k] = RNG_SC_TAB[k];
    }
    double2 wA, wB;     // e^{2 pi i j0 / N}, e^{2 pi i j1 / (N / R0)}
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)N, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    // tiles that are neighbours along the contiguous axis share 128-byte lines (the row pitch is odd in 16-byte units):
    // the workgroups of one XCD take eight adjacent tiles at a time (flatsky.hip, FS_PAIR_XCD)
    constexpr int GL = 3;
    const long gmask = (8L << GL) - 1;
    const bool pair_xcd = (ntiles & gmask) == 0 && (gridDim.x & gmask) == 0;
    auto remap = [&](long v) {
        if (!pair_xcd) return v;
        const long slot = v >> 3, xcd = v & 7;
        return (((slot >> GL) * 8 + xcd) << GL) + (slot & ((1 << GL) - 1));
    };
    struct tile_t {
        long base;
        int teff;
    };
    auto tile_of = [&](long v) {
        const long outer = v / chunks, i0 = (v - outer * chunks) * NCH;
        tile_t t;
        t.base = outer * N * inner + i0;
        t.teff = (int)min((long)NCH, inner - i0);
        return t;
    };
    double2 R[U];
    auto prefetch = [&](const tile_t &tl, int tid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e & (NCH - 1), j = e / NCH;
            const long addr = tl.base + (long)j * inner + min(c, tl.teff - 1);
            R[u] = GEN ? make_double2(in[addr], 0.0) : in2[addr];
        }
    };
    long vt = blockIdx.x;
    if (vt >= ntiles) return;
    tile_t cur = tile_of(remap(vt));
    prefetch(cur, tid0);
    while (true) {
        int tid = tid0;                                   // opaque per item: see ringfft_direct_ct
        asm volatile("" : "+v"(tid));
        __syncthreads();                                  // previous item's LDS reads are done (and the tables are filled)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e & (NCH - 1), j = e / NCH;
            double2 v = R[u];
            if (GEN) {
                const long addr = cur.base + (long)j * inner + min(c, cur.teff - 1);
                const double2 z = philox_boxmuller((uint64_t)addr, seed, lg_l, sc_l);
                v = make_double2(z.x * v.x, z.y * v.x);
            }
            sm[c * BS + fpad(j)] = v;
        }
        const long vnext = vt + gridDim.x;
        const tile_t nxt = tile_of(remap(min(vnext, ntiles - 1)));
        prefetch(nxt, tid);                               // (unconditional: the last iteration re-reads a tile)
        __syncthreads();
        ct_pass<PK, N, NCH, BS, N, R0, SIGN, false, T>(sm, wA, tid);
        __syncthreads();
        ct_pass<PK, N, NCH, BS, Q0, R1, SIGN, false, T>(sm, wB, tid);
        __syncthreads();
        // ---- last pass (radix R2 on contiguous elements) with the store: butterfly t = 16 k0 + k1 of line ch holds the
        //      natural indices k0 + R0 k1 + 16 R0 r; the lines of a row leave as 16 NCH contiguous bytes
        {
            constexpr int TOT = NCH * R0 * R1;
            constexpr int IT = (TOT + T - 1) / T;
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                if ((TOT % T) != 0 && idx >= TOT) break;
                const int ch = idx & (NCH - 1), t = idx / NCH;
                const int k0 = t >> 4, k1 = t & 15;
                const double2 *p = sm + ch * BS + fpad(t * R2);
                double2 x[R2];
#pragma unroll
                for (int r = 0; r < R2; r++) x[r] = p[fpc(r)];
                DftR<R2, SIGN>::run(x);
                if (ch < cur.teff) {
                    double2 *o = out + cur.base + (long)(k0 + R0 * k1) * inner + ch;
#pragma unroll
                    for (int r = 0; r < R2; r++) o[(long)(R0 * R1 * r) * inner] = make_double2(x[r].x * scale, x[r].y * scale);
                }
            }
        }
        if (vnext >= ntiles) break;
        vt = vnext;
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------
// linec2c_ct for lengths whose lines do not fit LDS sixteen at a time (N = 1024: 16 x 1024 x 16 bytes): the top radix-2
// stage is done between registers and LDS.  The even rows of a 16-line tile are transformed first (N / 2 points per
// line: three passes) and their result E_k read into registers, then the odd rows (O_k), and X_k = E_k + w^k O_k,
// X_{k + N/2} = E_k - w^k O_k leave as rows of 16 lines = 256 contiguous bytes (8 lines per item: 128-byte rows,
// 3.7 TB/s at 1024^3; sixteen: the rate of the 512-point passes).  Same thread -> (line, bin) map in both halves.
// ------------------------------------------------------------------------------------
template <int N, int NCH, int T, int SIGN, bool GEN>
__global__ void __launch_bounds__(T)
linec2c_h2_ct(const double *in, double2 *out, long nouter, long inner, double scale, uint64_t seed) {
    constexpr int PK = 1;
    constexpr int NH = N / 2;
    constexpr int R0 = Sch<NH>::R0, R1 = Sch<NH>::R1, R2 = Sch<NH>::R2;
    static_assert(R0 == 16 && R1 == 16 && R2 == 2 && NH == 512, "16 x 16 x 2 half transforms");
    constexpr int Q0 = NH / R0;
    static_assert(T % Q0 == 0 && NCH == 16 && T == 512, "thread -> bin map below");
    constexpr int BS = fpc(NH) + K5_CH_SKEW;
    constexpr int U = NCH * NH / T;                          // elements of a half tile per thread
    constexpr int IT = NCH * 256 / T;                        // last-pass butterflies per thread
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    double2 *lg_l = sm + NCH * BS, *sc_l = lg_l + 257;
    const int tid0 = threadIdx.x;
    const double2 *in2 = reinterpret_cast<const double2 *>(in);
    const long chunks = (inner + NCH - 1) / NCH;
    const long ntiles = nouter * chunks;
    if (GEN) {
        for (int k = tid0; k < 257; k += T) lg_l[k] = RNG_LOG_TAB[k];
        for (int k = tid0; k < 256; k += T) sc_l[k] = RNG_SC_TAB[k];
    }
    double2 wA, wB, wK, wK2;   // pass twiddles; w^k of this thread's first butterfly and the step w^2 between its butterflies
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)NH, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
        // butterfly idx = tid + it T of line idx & 15: t = idx >> 4 = 16 k0 + k1, bin k = k0 + 16 k1 = (tid >> 8) + 2 it + 16 ((tid >> 4) & 15)
        const int kb = (tid0 >> 8) + 16 * ((tid0 >> 4) & 15);
        sincospi(2.0 * (double)kb / (double)N, &s, &c);
        wK = make_double2(c, SIGN > 0 ? s : -s);
        sincospi(4.0 / (double)N, &s, &c);
        wK2 = make_double2(c, SIGN > 0 ? s : -s);
    }
    constexpr int GL = 3;
    const long gmask = (8L << GL) - 1;
    const bool pair_xcd = (ntiles & gmask) == 0 && (gridDim.x & gmask) == 0;
    auto remap = [&](long v) {
        if (!pair_xcd) return v;
        const long slot = v >> 3, xcd = v & 7;
        return (((slot >> GL) * 8 + xcd) << GL) + (slot & ((1 << GL) - 1));
    };
    struct tile_t {
        long base;
        int teff;
    };
    auto tile_of = [&](long v) {
        const long outer = v / chunks, i0 = (v - outer * chunks) * NCH;
        tile_t t;
        t.base = outer * N * inner + i0;
        t.teff = (int)min((long)NCH, inner - i0);
        return t;
    };
    double2 R[U];
    // rows 2 jj + par of the tile
    auto prefetch = [&](const tile_t &tl, int par, int tid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e & (NCH - 1), jj = e / NCH;
            const long addr = tl.base + (long)(2 * jj + par) * inner + min(c, tl.teff - 1);
            R[u] = GEN ? make_double2(in[addr], 0.0) : in2[addr];
        }
    };
    auto commit = [&](const tile_t &tl, int par, int tid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e & (NCH - 1), jj = e / NCH;
            double2 v = R[u];
            if (GEN) {
                const long addr = tl.base + (long)(2 * jj + par) * inner + min(c, tl.teff - 1);
                const double2 z = philox_boxmuller((uint64_t)addr, seed, lg_l, sc_l);
                v = make_double2(z.x * v.x, z.y * v.x);
            }
            sm[c * BS + fpad(jj)] = v;
        }
    };
    long vt = blockIdx.x;
    if (vt >= ntiles) return;
    tile_t cur = tile_of(remap(vt));
    prefetch(cur, 0, tid0);
    while (true) {
        int tid = tid0;                                   // opaque per item: see ringfft_direct_ct
        asm volatile("" : "+v"(tid));
        __syncthreads();                                  // previous item's LDS reads are done (and the tables are filled)
        commit(cur, 0, tid);
        prefetch(cur, 1, tid);                            // the odd rows, behind the passes of the even ones
        __syncthreads();
        ct_pass<PK, NH, NCH, BS, NH, R0, SIGN, false, T>(sm, wA, tid);
        __syncthreads();
        ct_pass<PK, NH, NCH, BS, Q0, R1, SIGN, false, T>(sm, wB, tid);
        __syncthreads();
        double2 E[IT][2];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int idx = tid + it * T;
            const int ch = idx & (NCH - 1), t = idx / NCH;
            const double2 *p = sm + ch * BS + fpad(t * R2);
            const double2 x0 = p[0], x1 = p[fpc(1)];
            E[it][0] = cadd(x0, x1);
            E[it][1] = csub(x0, x1);
        }
        __syncthreads();                                  // every E has been read
        commit(cur, 1, tid);
        const long vnext = vt + gridDim.x;
        const tile_t nxt = tile_of(remap(min(vnext, ntiles - 1)));
        prefetch(nxt, 0, tid);                            // (unconditional: the last iteration re-reads a tile)
        __syncthreads();
        ct_pass<PK, NH, NCH, BS, NH, R0, SIGN, false, T>(sm, wA, tid);
        __syncthreads();
        ct_pass<PK, NH, NCH, BS, Q0, R1, SIGN, false, T>(sm, wB, tid);
        __syncthreads();
        {
            double2 w = wK;
            asm volatile("" : "+v"(w.x), "+v"(w.y));
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const int idx = tid + it * T;
                const int ch = idx & (NCH - 1), t = idx / NCH;
                const int k = (t >> 4) + 16 * (t & 15);
                const double2 *p = sm + ch * BS + fpad(t * R2);
                const double2 x0 = p[0], x1 = p[fpc(1)];
                const double2 o0 = cmul(w, cadd(x0, x1));                       // w^k O_k
                const double2 o1s = cmul(w, csub(x0, x1));
                const double2 o1 = SIGN > 0 ? make_double2(-o1s.y, o1s.x) : make_double2(o1s.y, -o1s.x);   // w^(k + N/4) O_(k + N/4)
                if (ch < cur.teff) {
                    double2 *o = out + cur.base + (long)k * inner + ch;
                    const double2 a0 = cadd(E[it][0], o0), b0 = csub(E[it][0], o0), a1 = cadd(E[it][1], o1), b1 = csub(E[it][1], o1);
                    o[0] = make_double2(a0.x * scale, a0.y * scale);
                    o[(long)(N / 4) * inner] = make_double2(a1.x * scale, a1.y * scale);
                    o[(long)(N / 2) * inner] = make_double2(b0.x * scale, b0.y * scale);
                    o[(long)(3 * N / 4) * inner] = make_double2(b1.x * scale, b1.y * scale);
                }
                w = cmul(w, wK2);
            }
        }
        if (vnext >= ntiles) break;
        vt = vnext;
        cur = nxt;
    }
}

template <int N, int NCH, int T>
static int launch_linec2c(corahip_ctx *ctx, const double *in, double *out, long nouter, long inner, int inverse, double scale, bool gen,
                          uint64_t seed) {
    constexpr int PK = 1;
    constexpr int BS = fpc(N) + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * ((size_t)NCH * BS + (gen ? 513 : 0));
    const long ntiles = nouter * ((inner + NCH - 1) / NCH);
    const long per_cu = std::max<long>(1, std::min<long>((160 * 1024) / shm, 2048 / T));
    dim3 grid((unsigned)std::min<long>(ntiles, (long)ctx->num_cu * per_cu));
    double2 *o2 = reinterpret_cast<double2 *>(out);
#define C2C_LAUNCH(SG, GN)                                                                                                  \
    HIP_TRY(hipFuncSetAttribute((const void *)linec2c_ct<N, NCH, T, SG, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    linec2c_ct<N, NCH, T, SG, GN><<<grid, T, shm, ctx->stream>>>(in, o2, nouter, inner, scale, seed)
    if (gen) { C2C_LAUNCH(1, true); }
    else if (inverse) { C2C_LAUNCH(1, false); }
    else { C2C_LAUNCH(-1, false); }
#undef C2C_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
template <int N, int NCH, int T>
static int launch_linec2c_h2(corahip_ctx *ctx, const double *in, double *out, long nouter, long inner, int inverse, double scale, bool gen,
                             uint64_t seed) {
    constexpr int PK = 1;
    constexpr int BS = fpc(N / 2) + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * ((size_t)NCH * BS + (gen ? 513 : 0));
    const long ntiles = nouter * ((inner + NCH - 1) / NCH);
    dim3 grid((unsigned)std::min<long>(ntiles, (long)ctx->num_cu));
    double2 *o2 = reinterpret_cast<double2 *>(out);
#define C2C_LAUNCH(SG, GN)                                                                                                  \
    HIP_TRY(hipFuncSetAttribute((const void *)linec2c_h2_ct<N, NCH, T, SG, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    linec2c_h2_ct<N, NCH, T, SG, GN><<<grid, T, shm, ctx->stream>>>(in, o2, nouter, inner, scale, seed)
    if (gen) { C2C_LAUNCH(1, true); }
    else if (inverse) { C2C_LAUNCH(1, false); }
    else { C2C_LAUNCH(-1, false); }
#undef C2C_LAUNCH
    LAUNCH_CHECK();
    return 0;
}
// a strided complex pass of the flat-sky transforms (inner > 1) for the lengths that have a compile-time schedule;
// gen: inverse pass whose input is generated from the real k-weights `in` (corahip_randomfield_irfftn).  *took = false:
// the generic line kernel takes it
int flat_c2c_ct(corahip_ctx *ctx, const double *in, double *out, long nouter, int n, long inner, int inverse, double scale, bool gen,
                uint64_t seed, bool *took) {
    static const bool off = getenv("CORAHIP_FLAT_GENERIC") != nullptr;
    *took = false;
    if (off || nouter < 1 || inner < 2) return 0;
    int rc;
    if (n == 256 && inner >= 16) rc = launch_linec2c<256, 16, 256>(ctx, in, out, nouter, inner, inverse, scale, gen, seed);
    else if (n == 512 && inner >= 16) rc = launch_linec2c<512, 16, 512>(ctx, in, out, nouter, inner, inverse, scale, gen, seed);
    else if (n == 1024 && inner >= 16 && !getenv("CORAHIP_FLAT_NOH2")) rc = launch_linec2c_h2<1024, 16, 512>(ctx, in, out, nouter, inner, inverse, scale, gen, seed);
    else if (n == 1024 && inner >= 8) rc = launch_linec2c<1024, 8, 512>(ctx, in, out, nouter, inner, inverse, scale, gen, seed);
    else if (n == 384 && inner >= 16) rc = launch_linec2c<384, 16, 512>(ctx, in, out, nouter, inner, inverse, scale, gen, seed);
    else if (n == 768 && inner >= 8) rc = launch_linec2c<768, 8, 512>(ctx, in, out, nouter, inner, inverse, scale, gen, seed);
    else return 0;
    if (rc) return rc;
    *took = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// Flat-sky fields: real -> half-complex along the contiguous axis (numpy.fft.rfftn's first pass, cora/util/fftutil.py:64-75;
// the velocity cube of RedshiftCorrelation.realisation, cora/signal/corr.py:590-599), EVEN length 2 N: ringana_direct_ct's
// passes - z_j = x_2j + i x_2j+1 loaded in the digit order of the first (radix R2) pass, two radix-16 DIT passes, then
// the split X_k = 1/2 [(Z_k + conj Z_{N-k}) - i e^{-i pi k / N} (Z_k - conj Z_{N-k})], k = 0 .. N - on NCH adjacent lines.
// ------------------------------------------------------------------------------------
template <int N, int NCH, int T>
__global__ void __launch_bounds__(T)
liner2c_ct(const double *in, double2 *out, long nlines) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<N>::R0, R1 = Sch<N>::R1, R2 = Sch<N>::R2;
    static_assert((R0 == 16 || R0 == 12) && R1 == 16 && N == R0 * R1 * R2, "R0 x 16 x R2");
    constexpr int Q0 = N / R0;
    static_assert(Q0 % 16 == 0 && T % Q0 == 0, "per-thread twiddles");
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    constexpr int NB2 = R0 * R1;
    constexpr int TOT0 = NCH * NB2, IT0 = (TOT0 + T - 1) / T;
    constexpr int MO = (NCH * N + T - 1) / T;            // output bins 0 .. N-1 per thread (bin N: one more on NCH threads)
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const long nitems = (nlines + NCH - 1) / NCH;
    double2 wA, wB, wS, wSstep;     // (wS: e^{-i pi (tid mod N) / N}, its step over T bins)
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)N, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
        sincospi((double)(tid0 % N) / (double)N, &s, &c);
        wS = make_double2(c, -s);
        sincospi((double)T / (double)N, &s, &c);
        wSstep = make_double2(c, -s);
    }
    double2 pf[IT0 * R2];     // (flat: as [IT0][1] the array stayed in scratch memory)
    auto prefetch = [&](long item, int tid) {
        const long line0 = item * NCH;
#pragma unroll
        for (int it = 0; it < IT0; it++) {
            const int idx = min(tid + it * T, TOT0 - 1);
            const int ch = idx / NB2, q = idx - ch * NB2;
            const int k0 = R0 == 16 ? ((q & 7) | ((q >> 3) & 8)) : q % R0;
            const int k1 = R0 == 16 ? (((q >> 3) & 7) | ((q >> 4) & 8)) : q / R0;
            const double *src = in + min(line0 + ch, nlines - 1) * (2L * N) + 2 * (k0 + R0 * k1);
#pragma unroll
            for (int r = 0; r < R2; r++) pf[it * R2 + r] = *reinterpret_cast<const double2 *>(src + 2 * NB2 * r);
        }
    };
    long vitem = blockIdx.x;
    if (vitem < nitems) prefetch(vitem, tid0);
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const long line0 = vitem * NCH;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT0; it++) {
            const int idx = tid + it * T;
            if ((TOT0 % T) != 0 && idx >= TOT0) break;
            const int ch = idx / NB2, q = idx - ch * NB2;
            const int k0 = R0 == 16 ? ((q & 7) | ((q >> 3) & 8)) : q % R0;
            const int k1 = R0 == 16 ? (((q >> 3) & 7) | ((q >> 4) & 8)) : q / R0;
            double2 *p = sm + ch * BS + fpad((k0 * 16 + k1) * R2);
            double2 x[R2];
#pragma unroll
            for (int r = 0; r < R2; r++) x[r] = pf[it * R2 + r];
            DftR<R2, -1>::run(x);
#pragma unroll
            for (int r = 0; r < R2; r++) p[fpc(r)] = x[r];
        }
        prefetch(min(vitem + (long)gridDim.x, nitems - 1), tid);
        __syncthreads();
        ct_pass<PK, N, NCH, BS, Q0, R1, -1, true, T>(sm, wB, tid);
        __syncthreads();
        ct_pass<PK, N, NCH, BS, N, R0, -1, true, T>(sm, wA, tid);
        __syncthreads();
        auto emit = [&](int c, int k, const double2 w) {       // w = e^{-i pi k / N}
            const int ka = k == N ? 0 : k;                     // Z_N := Z_0
            const int kb = k == 0 ? 0 : N - k;
            const double2 za = sm[c * BS + fpad(ka)], zb = sm[c * BS + fpad(kb)];
            const double2 sum = make_double2(za.x + zb.x, za.y - zb.y);
            const double2 dif = make_double2(za.x - zb.x, za.y + zb.y);
            const double2 t = cmul(dif, w);
            if (line0 + c < nlines) out[(line0 + c) * (N + 1L) + k] = make_double2(0.5 * (sum.x + t.y), 0.5 * (sum.y - t.x));
        };
        // element e = tid + u T is bin k = e mod N of line e / N: the twiddle follows by one rotation per step and a sign
        // per wrap (e^{-i pi (k - N) / N} = -e^{-i pi k / N})
        {
            int c = tid / N, k = tid - c * N;
            double2 w = wS;
            asm volatile("" : "+v"(w.x), "+v"(w.y));
#pragma unroll
            for (int u = 0; u < MO; u++) {
                if ((NCH * N) % T == 0 || c < NCH) emit(c, k, w);
                w = cmul(w, wSstep);
                k += T;
                while (k >= N) {
                    k -= N;
                    c++;
                    w = make_double2(-w.x, -w.y);
                }
            }
        }
        if (tid < NCH) emit(tid, N, make_double2(-1.0, 0.0));
    }
}
template <int N, int NCH, int T>
static int launch_liner2c(corahip_ctx *ctx, const double *in, double *spec, long nlines) {
    constexpr int PK = 1;
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (nlines + NCH - 1) / NCH;
    const long per_cu = std::max<long>(1, std::min<long>((160 * 1024) / shm, 2048 / T));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipFuncSetAttribute((const void *)liner2c_ct<N, NCH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    liner2c_ct<N, NCH, T><<<grid, T, shm, ctx->stream>>>(in, reinterpret_cast<double2 *>(spec), nlines);
    LAUNCH_CHECK();
    return 0;
}
// the contiguous real -> half-complex pass of corahip_rfftn (real length 2 h); *took = false: the generic kernel takes it
int flat_r2c_ct(corahip_ctx *ctx, const double *in, double *spec, long nlines, int h, bool *took) {
    static const bool off = getenv("CORAHIP_FLAT_GENERIC") != nullptr;
    *took = false;
    if (off || nlines < 1) return 0;
    int rc;
    if (h == 256) rc = launch_liner2c<256, 16, 512>(ctx, in, spec, nlines);
    else if (h == 512) rc = launch_liner2c<512, 16, 512>(ctx, in, spec, nlines);
    else if (h == 1024) rc = launch_liner2c<1024, 8, 512>(ctx, in, spec, nlines);
    else if (h == 2048) rc = launch_liner2c<2048, 4, 512>(ctx, in, spec, nlines);
    else if (h == 192) rc = launch_liner2c<192, 16, 512>(ctx, in, spec, nlines);
    else if (h == 384) rc = launch_liner2c<384, 16, 512>(ctx, in, spec, nlines);
    else if (h == 768) rc = launch_liner2c<768, 8, 512>(ctx, in, spec, nlines);
    else if (h == 1536) rc = launch_liner2c<1536, 4, 512>(ctx, in, spec, nlines);
    else return 0;
    if (rc) return rc;
    *took = true;
    return 0;
}

template <int N, int NCH, int T>
static int launch_linec2r(corahip_ctx *ctx, const double *spec, double *out, long nlines, double scale) {
    constexpr int PK = 1;
    constexpr int BS = fpc(N) + 1 + K5_CH_SKEW;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (nlines + NCH - 1) / NCH;
    const long per_cu = std::max<long>(1, std::min<long>((160 * 1024) / shm, 2048 / T));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipFuncSetAttribute((const void *)linec2r_ct<N, NCH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    linec2r_ct<N, NCH, T><<<grid, T, shm, ctx->stream>>>(reinterpret_cast<const double2 *>(spec), out, nlines, scale);
    LAUNCH_CHECK();
    return 0;
}
// the contiguous half-complex -> real pass of corahip_irfftn for the complex lengths that have a compile-time schedule;
// *took = false: the generic line kernel takes it
int flat_c2r_ct(corahip_ctx *ctx, const double *spec, double *out, long nlines, int h, double scale, bool *took) {
    static const bool off = getenv("CORAHIP_FLAT_GENERIC") != nullptr;
    *took = false;
    if (off || nlines < 1) return 0;
    int rc;
    if (h == 256) rc = launch_linec2r<256, 16, 256>(ctx, spec, out, nlines, scale);
    else if (h == 512) rc = launch_linec2r<512, 16, 512>(ctx, spec, out, nlines, scale);
    else if (h == 1024) rc = launch_linec2r<1024, 8, 512>(ctx, spec, out, nlines, scale);
    else if (h == 2048) rc = launch_linec2r<2048, 4, 512>(ctx, spec, out, nlines, scale);
    else if (h == 192) rc = launch_linec2r<192, 16, 256>(ctx, spec, out, nlines, scale);
    else if (h == 384) rc = launch_linec2r<384, 16, 512>(ctx, spec, out, nlines, scale);
    else if (h == 768) rc = launch_linec2r<768, 8, 512>(ctx, spec, out, nlines, scale);
    else if (h == 1536) rc = launch_linec2r<1536, 4, 512>(ctx, spec, out, nlines, scale);
    else return 0;
    if (rc) return rc;
    *took = true;
    return 0;
}


// ------------------------------------------------------------------------------------
// Arbitrary lengths (the comoving boxes of RedshiftCorrelation.realisation are 261 x 316 x 316 and the like,
// cora/signal/corr.py:655-671): Bluestein on the compile-time passes, as ringfft_blu_ct runs the cap rings.
//   X_k = c_k sum_j (x_j c_j) b_{k-j},  c_j = e^{-i pi j^2 / n} (the plan's chirp), b = conj c,
// the convolution at a length P >= 2 n - 1 out of {2, 3, 5} x 2^k: forward pass 1 reads only the non-zero half, last
// forward pass + filter + first inverse pass in registers, the last inverse pass forms only the n outputs that exist.
// Strided axis, NCH neighbouring lines per item; inverse transforms by conjugation at the commit and the store; GEN as in
// linec2c_ct.  Filter (FFT_P of the wrapped b, / P) in the passes' storage order: flat_blu_filter_kernel.
// ------------------------------------------------------------------------------------
template <int P, int T>
__global__ void __launch_bounds__(T)
flat_blu_filter_kernel(int n, const double2 *__restrict__ chirp, double2 *__restrict__ filt) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<P>::R0, R1 = Sch<P>::R1, R2 = Sch<P>::R2;
    constexpr int Q0 = P / R0;
    constexpr int BS = fpc(P) + 1;
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid = threadIdx.x;
    for (int m = tid; m < P; m += T) {
        double2 v = make_double2(0.0, 0.0);
        if (m < n) v = cconj(chirp[m]);
        else if (P - m < n) v = cconj(chirp[P - m]);
        sm[fpad(m)] = v;
    }
    double2 wA, wB;
    {
        const int j0 = tid & (Q0 - 1), j1 = tid & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)P, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    __syncthreads();
    ct_pass<PK, P, 1, BS, P, R0, -1, false, T>(sm, wA, tid);
    __syncthreads();
    ct_pass<PK, P, 1, BS, Q0, R1, -1, false, T>(sm, wB, tid);
    __syncthreads();
    if constexpr (R2 > 1) {
        ct_pass<PK, P, 1, BS, R2, R2, -1, false, T>(sm, make_double2(1.0, 0.0), tid);
        __syncthreads();
    }
    const double invP = 1.0 / (double)P;
    for (int m = tid; m < P; m += T) {
        const double2 v = sm[fpad(m)];
        filt[m] = make_double2(v.x * invP, v.y * invP);
    }
}

// the inner stages of the convolution on NCH lines of P points in LDS (line c at sm + c BS, BS = fpc(P) + 1): forward pass
// 1 (sign -; the inputs r >= R0 / 2 are the zero padding and are not read), forward pass 2, last forward pass + filter +
// first inverse pass in registers (fl: the filter values of this thread's butterfly t = tid mod (P / R2)), inverse pass
// 2.  Starts behind the caller's barrier, ends with one; the caller runs the last inverse pass (radix R0, stride P / R0).
// `hook` runs in front of inverse pass 2: loads that are needed behind the last pass and have no registers to live in
// across all of the passes (the output chirps) are requested there.
struct blu_no_hook {
    __device__ __forceinline__ void operator()() const {}
};
template <int P, int NCH, int T, class Hook = blu_no_hook>
__device__ __forceinline__ static void blu_core(double2 *sm, const int tid, const double2 wA, const double2 wB,
                                                const double2 (&fl)[Sch<P>::R2], Hook &&hook = Hook()) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<P>::R0, R1 = Sch<P>::R1, R2 = Sch<P>::R2;
    constexpr int Q0 = P / R0, BS = fpc(P) + 1, NB = P / R2;
    constexpr int TOTL = NCH * Q0;
    {
        const double2 w = cconj(wA);
        const int idx = tid;
        if (TOTL == T || idx < TOTL) {
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
    hook();      // (loads the caller needs behind the last pass are requested here: one pass ahead)
    __syncthreads();
    ct_pass<PK, P, NCH, BS, Q0, R1, 1, true, T>(sm, wB, tid);
    __syncthreads();
}

template <int P, int NCH, int T, bool GEN>
__global__ void __launch_bounds__(T)
lineblu_c2c_ct(const double *in, double2 *out, long nouter, long inner, int n, double scale, int inverse, const double2 *chirp,
               const double2 *filt, uint64_t seed) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<P>::R0, R1 = Sch<P>::R1, R2 = Sch<P>::R2;
    static_assert(R0 <= 16 && R1 == 16 && P == R0 * R1 * R2, "R0 x 16 x R2");
    constexpr int Q0 = P / R0;
    static_assert((Q0 & (Q0 - 1)) == 0 && Q0 % 16 == 0 && T % Q0 == 0, "per-thread twiddles");
    constexpr int BS = fpc(P) + 1;               // odd: the lines of a row (consecutive lanes at the commit / store) in different banks
    constexpr int HALF = P / 2;                  // n <= HALF: the non-zero half of the padded input
    constexpr int U = (NCH * HALF + T - 1) / T;  // input elements per thread
    constexpr int NB = P / R2;                   // middle-stage butterflies per line
    static_assert(NB <= T, "one middle-stage butterfly per thread and line group");
    constexpr int TOTL = NCH * Q0, ITL = 1;      // first / last pass butterflies: one per thread
    static_assert(TOTL <= T && (T / NCH) % Q0 == 0, "one first / last pass butterfly per thread");
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    double2 *lg_l = sm + NCH * BS, *sc_l = lg_l + 257;
    const int tid0 = threadIdx.x;
    const double2 *in2 = reinterpret_cast<const double2 *>(in);
    const long chunks = (inner + NCH - 1) / NCH;
    const long ntiles = nouter * chunks;
    if (GEN) {
        for (int k = tid0; k < 257; k += T) lg_l[k] = RNG_LOG_TAB[k];
        for (int k = tid0; k < 256; k += T) sc_l[k] = RNG_SC_TAB[k];
    }
    double2 wA, wB, wL;     // e^{2 pi i j0 / P}, e^{2 pi i j1 / (P / R0)}; wL: wA for the last pass' thread -> butterfly map
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / R1 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)P, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
        sincospi(2.0 * (double)((tid0 / NCH) & (Q0 - 1)) / (double)P, &s, &c);
        wL = make_double2(c, s);
    }
    // what depends on the thread only (the length is the same for every line): input chirps, filter values of the
    // middle stage, output chirps of the last pass - loaded once
    double2 cb[U];
#pragma unroll
    for (int u = 0; u < U; u++) cb[u] = chirp[min((tid0 + u * T) / NCH, n - 1)];
    double2 fl[R2];
#pragma unroll
    for (int r = 0; r < R2; r++) fl[r] = filt[(size_t)(tid0 % NB) * R2 + r];
    double2 ob[ITL][R0 / 2];
#pragma unroll
    for (int it = 0; it < ITL; it++)
#pragma unroll
        for (int r = 0; r < R0 / 2; r++) ob[it][r] = chirp[min((tid0 + it * T) / NCH % Q0 + r * Q0, n - 1)];
    constexpr int GL = 3;
    const long gmask = (8L << GL) - 1;
    const bool pair_xcd = (ntiles & gmask) == 0 && (gridDim.x & gmask) == 0;
    auto remap = [&](long v) {
        if (!pair_xcd) return v;
        const long slot = v >> 3, xcd = v & 7;
        return (((slot >> GL) * 8 + xcd) << GL) + (slot & ((1 << GL) - 1));
    };
    struct tile_t {
        long base;
        int teff;
    };
    auto tile_of = [&](long v) {
        const long outer = v / chunks, i0 = (v - outer * chunks) * NCH;
        tile_t t;
        t.base = outer * n * inner + i0;
        t.teff = (int)min((long)NCH, inner - i0);
        return t;
    };
    double2 R[U];
    auto prefetch = [&](const tile_t &tl, int tid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e & (NCH - 1), j = min(e / NCH, n - 1);
            const long addr = tl.base + (long)j * inner + min(c, tl.teff - 1);
            R[u] = GEN ? make_double2(in[addr], 0.0) : in2[addr];
        }
    };
    long vt = blockIdx.x;
    if (vt >= ntiles) return;
    tile_t cur = tile_of(remap(vt));
    prefetch(cur, tid0);
    while (true) {
        int tid = tid0;                                   // opaque per item: see ringfft_direct_ct
        asm volatile("" : "+v"(tid));
        __syncthreads();                                  // previous item's LDS reads are done (and the tables are filled)
        // ---- x_j c_j at position j (conjugated for the inverse transform), zeros on [n, HALF)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = tid + u * T;
            const int c = e & (NCH - 1), j = e / NCH;
            if ((NCH * HALF) % T != 0 && j >= HALF) break;
            double2 v = R[u];
            if (GEN) {
                const long addr = cur.base + (long)min(j, n - 1) * inner + min(c, cur.teff - 1);
                const double2 z = philox_boxmuller((uint64_t)addr, seed, lg_l, sc_l);
                v = make_double2(z.x * v.x, z.y * v.x);
            }
            if (inverse) v.y = -v.y;
            // (opaque: the generated and the loaded input must go through the SAME instructions from here on - fused into
            //  the product above, the chirp multiply contracted differently and the fused call lost its bit-identity with
            //  draw + irfftn)
            asm volatile("" : "+v"(v.x), "+v"(v.y));
            v = j < n ? cmul(v, cb[u]) : make_double2(0.0, 0.0);
            sm[c * BS + fpad(j)] = v;
        }
        const long vnext = vt + gridDim.x;
        const tile_t nxt = tile_of(remap(min(vnext, ntiles - 1)));
        prefetch(nxt, tid);
        __syncthreads();
        blu_core<P, NCH, T>(sm, tid, wA, wB, fl);
        // ---- last inverse pass (sign +) with the store: the outputs k = j0 + r Q0 < n, times c_k; the lines of a row in
        //      consecutive lanes (16 NCH contiguous bytes)
        {
#pragma unroll
            for (int it = 0; it < ITL; it++) {
                const int idx = tid + it * T;
                if ((TOTL % T) != 0 && idx >= TOTL) break;
                const int ch = idx & (NCH - 1), j0 = (idx / NCH) & (Q0 - 1);
                const double2 *p = sm + ch * BS + fpad(j0);
                double2 x[R0];
#pragma unroll
                for (int r = 0; r < R0; r++) x[r] = p[fpc(r * Q0)];
                tw_apply<R0>(x, wL);
                DftR<R0, 1>::run(x);
                if (ch < cur.teff) {
                    double2 *o = out + cur.base + ch;
#pragma unroll
                    for (int r = 0; r < R0 / 2; r++) {
                        const int k = j0 + r * Q0;
                        if (k < n) {
                            double2 v = cmul(x[r], ob[it][r]);
                            if (inverse) v.y = -v.y;
                            o[(long)k * inner] = make_double2(v.x * scale, v.y * scale);
                        }
                    }
                }
            }
        }
        if (vnext >= ntiles) break;
        vt = vnext;
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------
// The contiguous passes of EVEN real length 2 h, h arbitrary, through the same convolution (flatsky.hip's modes 3 / 4 on
// the compile-time passes): c2r packs the pairs (k, h - k) of the half spectrum into conj(Z) c (one complex transform of
// length h; the inverse runs as conj(FFT(conj .))) and stores conj(c_j conv_j) = x_2j + i x_2j+1; r2c transforms z_j =
// x_2j + i x_2j+1 and unpacks X_k, X_{h-k} from Z_k, Z_{h-k}.  Everything that depends on (thread, h) only - LDS offsets
// of the elements a thread commits, its pairs, their twiddles e^{i pi k / h} (the plan's rtw) and chirps - is set up once
// per kernel: h is the same for every line.
// ------------------------------------------------------------------------------------
template <int P, int NCH, int T>
__global__ void __launch_bounds__(T)
lineblu_c2r_ct(const double2 *in, double2 *out, long nlines, int h, double scale, const double2 *chirp, const double2 *filt,
               const double2 *rtw) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<P>::R0, R2 = Sch<P>::R2;
    constexpr int Q0 = P / R0, BS = fpc(P) + 1, HALF = P / 2, NB = P / R2, TOTL = NCH * Q0;
    static_assert(TOTL <= T && T % Q0 == 0 && NB <= T, "one first / last pass butterfly per thread");
    constexpr int UL = (NCH * (HALF + 1) + T - 1) / T;        // input bins per thread
    constexpr int UP = (NCH * (HALF / 2 + 1) + T - 1) / T;    // pairs per thread
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const int hp = (h >> 1) + 1;
    const long nitems = (nlines + NCH - 1) / NCH;
    double2 wA, wB;
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / 16 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)P, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    double2 fl[R2];
#pragma unroll
    for (int r = 0; r < R2; r++) fl[r] = filt[(size_t)(tid0 % NB) * R2 + r];
    int lofs[UL];          // LDS slot of input element tid + u T of the item's chunk (-1: none; bit 30: DC / Nyquist bin)
#pragma unroll
    for (int u = 0; u < UL; u++) {
        const int e = tid0 + u * T;
        lofs[u] = -1;
        if (e < NCH * (h + 1)) {
            const int c = e / (h + 1), k = e - c * (h + 1);
            lofs[u] = (c * BS + fpad(k)) | ((k == 0 || k == h) ? (1 << 30) : 0);
        }
    }
    int pofs[UP], pk[UP];  // pairs: line offset (-1: none), k
    double2 pw[UP];
#pragma unroll
    for (int u = 0; u < UP; u++) {
        const int e = tid0 + u * T;
        pofs[u] = -1;
        pk[u] = 0;
        pw[u] = make_double2(0.0, 0.0);
        if (e < NCH * hp) {
            const int c = e / hp, k = e - c * hp;
            pofs[u] = c * BS;
            pk[u] = k;
            pw[u] = rtw[k];
        }
    }
    double2 R[UL];
    const long nel = nlines * (h + 1L);
    auto prefetch = [&](long item, int tid) {
        const long base = item * NCH * (h + 1L);
#pragma unroll
        for (int u = 0; u < UL; u++) R[u] = in[min(base + tid + u * T, nel - 1)];
    };
    long vitem = blockIdx.x;
    if (vitem < nitems) prefetch(vitem, tid0);
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const long line0 = vitem * NCH;
        // chirps of this thread's pairs (the same for every item, but there are no registers to keep them across the passes)
        double2 pck[UP], pck2[UP];
#pragma unroll
        for (int u = 0; u < UP; u++) {
            pck[u] = chirp[min(pk[u], h - 1)];
            pck2[u] = chirp[min(h - pk[u], h - 1)];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UL; u++)
            if (lofs[u] >= 0) {
                double2 v = R[u];
                if (lofs[u] & (1 << 30)) v.y = 0.0;           // Im of the DC / Nyquist bins is ignored
                sm[lofs[u] & ~(1 << 30)] = v;
            }
        prefetch(min(vitem + (long)gridDim.x, nitems - 1), tid);
        __syncthreads();
        // ---- pairs (k, h - k) -> conj(Z) c in place (flatsky.hip, mode 3); position h (the Nyquist bin) back to zero
#pragma unroll
        for (int u = 0; u < UP; u++)
            if (pofs[u] >= 0) {
                double2 *ln = sm + pofs[u];
                const int k = pk[u], k2 = h - k;
                const double2 xk = ln[fpad(k)], xc = ln[fpad(k2)];
                const double2 E = make_double2(xk.x + xc.x, xk.y - xc.y);
                const double2 O = cmul(make_double2(xk.x - xc.x, xk.y + xc.y), pw[u]);
                const double2 a = make_double2(E.x - O.y, -(E.y + O.x));      // conj(E + i O)
                const double2 b = make_double2(E.x + O.y, -(O.x - E.y));      // conj(conj E + i conj O)
                ln[fpad(k)] = cmul(a, pck[u]);
                if (k == 0) ln[fpad(h)] = make_double2(0.0, 0.0);
                else if (k2 != k) ln[fpad(k2)] = cmul(b, pck2[u]);
            }
        for (int j = h + 1 + tid; j < HALF; j += T)
#pragma unroll
            for (int c = 0; c < NCH; c++) sm[c * BS + fpad(j)] = make_double2(0.0, 0.0);
        __syncthreads();
        double2 ob[R0 / 2];     // chirps of the outputs this thread forms
        blu_core<P, NCH, T>(sm, tid, wA, wB, fl, [&]() {
#pragma unroll
            for (int r = 0; r < R0 / 2; r++) ob[r] = chirp[min((tid & (Q0 - 1)) + r * Q0, h - 1)];
        });
        if (TOTL == T || tid < TOTL) {
            const int ch = tid / Q0, j0 = tid & (Q0 - 1);
            const double2 *p = sm + ch * BS + fpad(j0);
            double2 x[R0];
#pragma unroll
            for (int r = 0; r < R0; r++) x[r] = p[fpc(r * Q0)];
            tw_apply<R0>(x, wA);
            DftR<R0, 1>::run(x);
            if (line0 + ch < nlines) {
                double2 *o = out + (line0 + ch) * (long)h;
#pragma unroll
                for (int r = 0; r < R0 / 2; r++) {
                    const int j = j0 + r * Q0;
                    if (j < h) {
                        const double2 v = cmul(x[r], ob[r]);
                        o[j] = make_double2(v.x * scale, -v.y * scale);    // conj: x_{2j} + i x_{2j+1}
                    }
                }
            }
        }
    }
}

template <int P, int NCH, int T>
__global__ void __launch_bounds__(T)
lineblu_r2c_ct(const double2 *in, double2 *out, long nlines, int h, const double2 *chirp, const double2 *filt, const double2 *rtw) {
    constexpr int PK = 1;
    constexpr int R0 = Sch<P>::R0, R2 = Sch<P>::R2;
    constexpr int Q0 = P / R0, BS = fpc(P) + 1, HALF = P / 2, NB = P / R2, TOTL = NCH * Q0;
    static_assert(TOTL <= T && T % Q0 == 0 && NB <= T, "one first / last pass butterfly per thread");
    constexpr int UL = (NCH * HALF + T - 1) / T;              // z_j per thread
    constexpr int UP = (NCH * (HALF / 2 + 1) + T - 1) / T;    // pairs per thread
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    const int tid0 = threadIdx.x;
    const int hp = (h >> 1) + 1;
    const long nitems = (nlines + NCH - 1) / NCH;
    double2 wA, wB;
    {
        const int j0 = tid0 & (Q0 - 1), j1 = tid0 & (Q0 / 16 - 1);
        double s, c;
        sincospi(2.0 * (double)j0 / (double)P, &s, &c);
        wA = make_double2(c, s);
        sincospi(2.0 * (double)j1 / (double)Q0, &s, &c);
        wB = make_double2(c, s);
    }
    double2 fl[R2];
#pragma unroll
    for (int r = 0; r < R2; r++) fl[r] = filt[(size_t)(tid0 % NB) * R2 + r];
    int lofs[UL], lj[UL];
#pragma unroll
    for (int u = 0; u < UL; u++) {
        const int e = tid0 + u * T;
        lofs[u] = -1;
        lj[u] = 0;
        if (e < NCH * h) {
            const int c = e / h, j = e - c * h;
            lofs[u] = c * BS + fpad(j);
            lj[u] = j;
        }
    }
    int pofs[UP], pk[UP], pline[UP];
    double2 pw[UP];
#pragma unroll
    for (int u = 0; u < UP; u++) {
        const int e = tid0 + u * T;
        pofs[u] = -1;
        pk[u] = pline[u] = 0;
        pw[u] = make_double2(0.0, 0.0);
        if (e < NCH * hp) {
            const int c = e / hp, k = e - c * hp;
            pofs[u] = c * BS;
            pline[u] = c;
            pk[u] = k;
            pw[u] = rtw[k];
        }
    }
    double2 R[UL], lcb[UL];     // z_j of the next item and (requested with them) their chirps
    const long nel = nlines * (long)h;
    auto prefetch = [&](long item, int tid) {
        const long base = item * NCH * (long)h;
#pragma unroll
        for (int u = 0; u < UL; u++) {
            R[u] = in[min(base + tid + u * T, nel - 1)];
            lcb[u] = chirp[lj[u]];
        }
    };
    long vitem = blockIdx.x;
    if (vitem < nitems) prefetch(vitem, tid0);
    for (; vitem < nitems; vitem += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const long line0 = vitem * NCH;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UL; u++)
            if (lofs[u] >= 0) sm[lofs[u]] = cmul(R[u], lcb[u]);
        for (int j = h + tid; j < HALF; j += T)
#pragma unroll
            for (int c = 0; c < NCH; c++) sm[c * BS + fpad(j)] = make_double2(0.0, 0.0);
        prefetch(min(vitem + (long)gridDim.x, nitems - 1), tid);
        __syncthreads();
        double2 ob[R0 / 2];
        blu_core<P, NCH, T>(sm, tid, wA, wB, fl, [&]() {
#pragma unroll
            for (int r = 0; r < R0 / 2; r++) ob[r] = chirp[min((tid & (Q0 - 1)) + r * Q0, h - 1)];
        });
        // ---- last inverse pass: Z_k = c_k conv_k, k < h, back to position k
        if (TOTL == T || tid < TOTL) {
            const int ch = tid / Q0, j0 = tid & (Q0 - 1);
            double2 *p = sm + ch * BS + fpad(j0);
            double2 x[R0];
#pragma unroll
            for (int r = 0; r < R0; r++) x[r] = p[fpc(r * Q0)];
            tw_apply<R0>(x, wA);
            DftR<R0, 1>::run(x);
#pragma unroll
            for (int r = 0; r < R0 / 2; r++) p[fpc(r * Q0)] = cmul(x[r], ob[r]);
        }
        __syncthreads();
        // ---- unpack the pairs: X_k = E_k + W^{-k} O_k, X_{h-k} = conj(E_k - W^{-k} O_k) (flatsky.hip, mode 4)
#pragma unroll
        for (int u = 0; u < UP; u++)
            if (pofs[u] >= 0 && line0 + pline[u] < nlines) {
                const double2 *ln = sm + pofs[u];
                const int k = pk[u], k2 = h - k;
                const double2 zk = ln[fpad(k)], zc = ln[fpad(k2 == h ? 0 : k2)];
                const double2 Ea = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y));
                const double2 Oa = make_double2(0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x));   // (zk - conj zc) / 2i
                const double2 B = cmul(make_double2(pw[u].x, -pw[u].y), Oa);
                double2 *o = out + (line0 + pline[u]) * (long)(h + 1);
                o[k] = make_double2(Ea.x + B.x, Ea.y + B.y);
                if (k2 != k) o[k2] = make_double2(Ea.x - B.x, -(Ea.y - B.y));
            }
    }
}

// the smallest scheduled length that holds the convolution of an n-point line (0: none)
static int flat_blu_length(int n) {
    static const int lens[] = {256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 2560, 3072, 3584, 4096};
    for (int P : lens)
        if (P >= 2 * n - 1) return P;
    return 0;
}
template <int P>
static int blu_filter_build(corahip_ctx *ctx, int n, const double2 *chirp, double2 *filt) {
    constexpr int PK = 1;
    const size_t shm = sizeof(double2) * (fpc(P) + 1);
    HIP_TRY(hipFuncSetAttribute((const void *)flat_blu_filter_kernel<P, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    flat_blu_filter_kernel<P, 256><<<1, 256, shm, ctx->stream>>>(n, chirp, filt);
    LAUNCH_CHECK();
    return 0;
}
// plan time (flatsky.hip, get_linefft_plan of a Bluestein length): the compile-time convolution length for n and its
// filter; *Pct = 0 when no schedule holds it
int flat_blu_plan(corahip_ctx *ctx, int n, const double2 *chirp, int *Pct, double2 **filt_ct) {
    *Pct = 0;
    *filt_ct = nullptr;
    const int P = flat_blu_length(n);
    if (!P || n < 17) return 0;
    double2 *f = nullptr;
    HIP_TRY(hipMalloc((void **)&f, sizeof(double2) * P));
    int rc = -1;
    switch (P) {
#define BLU_CASE(PP) case PP: rc = blu_filter_build<PP>(ctx, n, chirp, f); break;
        BLU_CASE(256) BLU_CASE(320) BLU_CASE(384) BLU_CASE(512) BLU_CASE(640) BLU_CASE(768) BLU_CASE(1024) BLU_CASE(1280)
        BLU_CASE(1536) BLU_CASE(2048) BLU_CASE(2560) BLU_CASE(3072) BLU_CASE(3584) BLU_CASE(4096)
#undef BLU_CASE
    }
    if (rc) {
        (void)hipFree(f);
        return rc;
    }
    *Pct = P;
    *filt_ct = f;
    return 0;
}
template <int P, int NCH, int T>
static int launch_lineblu_c2c(corahip_ctx *ctx, const double *in, double *out, long nouter, long inner, int n, int inverse,
                              double scale, bool gen, uint64_t seed, const double2 *chirp, const double2 *filt) {
    constexpr int PK = 1;
    constexpr int BS = fpc(P) + 1;
    const size_t shm = sizeof(double2) * ((size_t)NCH * BS + (gen ? 513 : 0));
    const long ntiles = nouter * ((inner + NCH - 1) / NCH);
    const long per_cu = std::max<long>(1, std::min<long>((160 * 1024) / shm, 2048 / T));
    dim3 grid((unsigned)std::min<long>(ntiles, (long)ctx->num_cu * per_cu));
    double2 *o2 = reinterpret_cast<double2 *>(out);
    if (gen) {
        HIP_TRY(hipFuncSetAttribute((const void *)lineblu_c2c_ct<P, NCH, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lineblu_c2c_ct<P, NCH, T, true><<<grid, T, shm, ctx->stream>>>(in, o2, nouter, inner, n, scale, 1, chirp, filt, seed);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void *)lineblu_c2c_ct<P, NCH, T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lineblu_c2c_ct<P, NCH, T, false><<<grid, T, shm, ctx->stream>>>(in, o2, nouter, inner, n, scale, inverse, chirp, filt, seed);
    }
    LAUNCH_CHECK();
    return 0;
}
// a strided complex pass of a Bluestein length; Pct / filt_ct from flat_blu_plan.  *took = false: the generic kernel takes it
int flat_blu_c2c_ct(corahip_ctx *ctx, const double *in, double *out, long nouter, int n, long inner, int inverse, double scale, bool gen,
                    uint64_t seed, int Pct, const double2 *chirp, const double2 *filt_ct, bool *took) {
    static const bool off = getenv("CORAHIP_FLAT_GENERIC") != nullptr;
    *took = false;
    if (off || !Pct || !filt_ct || nouter < 1 || inner < 2) return 0;
    int rc;
#define BLU_ARGS ctx, in, out, nouter, inner, n, inverse, scale, gen, seed, chirp, filt_ct
    switch (Pct) {
    case 256: rc = launch_lineblu_c2c<256, 16, 512>(BLU_ARGS); break;
    case 320: rc = launch_lineblu_c2c<320, 16, 512>(BLU_ARGS); break;
    case 384: rc = launch_lineblu_c2c<384, 16, 512>(BLU_ARGS); break;
    case 512: rc = launch_lineblu_c2c<512, 16, 512>(BLU_ARGS); break;
    case 640: rc = launch_lineblu_c2c<640, 8, 512>(BLU_ARGS); break;
    case 768: rc = launch_lineblu_c2c<768, 8, 512>(BLU_ARGS); break;
    case 1024: rc = launch_lineblu_c2c<1024, 8, 512>(BLU_ARGS); break;
    case 1280: rc = launch_lineblu_c2c<1280, 4, 512>(BLU_ARGS); break;
    case 1536: rc = launch_lineblu_c2c<1536, 4, 512>(BLU_ARGS); break;
    case 2048: rc = launch_lineblu_c2c<2048, 4, 512>(BLU_ARGS); break;
    case 2560: rc = launch_lineblu_c2c<2560, 2, 512>(BLU_ARGS); break;
    case 3072: rc = launch_lineblu_c2c<3072, 2, 512>(BLU_ARGS); break;
    case 3584: rc = launch_lineblu_c2c<3584, 2, 512>(BLU_ARGS); break;
    case 4096: rc = launch_lineblu_c2c<4096, 2, 512>(BLU_ARGS); break;
    default: return 0;
    }
#undef BLU_ARGS
    if (rc) return rc;
    *took = true;
    return 0;
}

template <int P, int NCH, int T>
static int launch_lineblu_real(corahip_ctx *ctx, bool c2r, const double *in, double *out, long nlines, int h, double scale,
                               const double2 *chirp, const double2 *filt, const double2 *rtw) {
    constexpr int PK = 1;
    constexpr int BS = fpc(P) + 1;
    const size_t shm = sizeof(double2) * (size_t)NCH * BS;
    const long nitems = (nlines + NCH - 1) / NCH;
    const long per_cu = std::max<long>(1, std::min<long>((160 * 1024) / shm, 2048 / T));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    const double2 *i2 = reinterpret_cast<const double2 *>(in);
    double2 *o2 = reinterpret_cast<double2 *>(out);
    if (c2r) {
        HIP_TRY(hipFuncSetAttribute((const void *)lineblu_c2r_ct<P, NCH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lineblu_c2r_ct<P, NCH, T><<<grid, T, shm, ctx->stream>>>(i2, o2, nlines, h, scale, chirp, filt, rtw);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void *)lineblu_r2c_ct<P, NCH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lineblu_r2c_ct<P, NCH, T><<<grid, T, shm, ctx->stream>>>(i2, o2, nlines, h, chirp, filt, rtw);
    }
    LAUNCH_CHECK();
    return 0;
}
// the contiguous pass of an even real length 2 h whose h needs Bluestein (c2r: half-complex -> real times scale; r2c: real
// -> half-complex); Pct / filt_ct / chirp / rtw of the plan of length h.  *took = false: the generic kernel takes it
int flat_blu_real_ct(corahip_ctx *ctx, bool c2r, const double *in, double *out, long nlines, int h, double scale, int Pct,
                     const double2 *chirp, const double2 *filt_ct, const double2 *rtw, bool *took) {
    static const bool off = getenv("CORAHIP_FLAT_GENERIC") != nullptr;
    *took = false;
    if (off || !Pct || !filt_ct || nlines < 1) return 0;
    int rc;
#define BLU_ARGS ctx, c2r, in, out, nlines, h, scale, chirp, filt_ct, rtw
    switch (Pct) {
    case 256: rc = launch_lineblu_real<256, 16, 512>(BLU_ARGS); break;
    case 320: rc = launch_lineblu_real<320, 16, 512>(BLU_ARGS); break;
    case 384: rc = launch_lineblu_real<384, 16, 512>(BLU_ARGS); break;
    case 512: rc = launch_lineblu_real<512, 16, 512>(BLU_ARGS); break;
    case 640: rc = launch_lineblu_real<640, 8, 512>(BLU_ARGS); break;
    case 768: rc = launch_lineblu_real<768, 8, 512>(BLU_ARGS); break;
    case 1024: rc = launch_lineblu_real<1024, 8, 512>(BLU_ARGS); break;
    case 1280: rc = launch_lineblu_real<1280, 4, 512>(BLU_ARGS); break;
    case 1536: rc = launch_lineblu_real<1536, 4, 512>(BLU_ARGS); break;
    case 2048: rc = launch_lineblu_real<2048, 4, 512>(BLU_ARGS); break;
    case 2560: rc = launch_lineblu_real<2560, 2, 512>(BLU_ARGS); break;
    case 3072: rc = launch_lineblu_real<3072, 2, 512>(BLU_ARGS); break;
    case 3584: rc = launch_lineblu_real<3584, 2, 512>(BLU_ARGS); break;
    case 4096: rc = launch_lineblu_real<4096, 2, 512>(BLU_ARGS); break;
    default: return 0;
    }
#undef BLU_ARGS
    if (rc) return rc;
    *took = true;
    return 0;
}
