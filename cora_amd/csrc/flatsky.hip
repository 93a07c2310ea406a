// flatsky.hip - flat-sky Gaussian random fields: n-dimensional real FFTs and the fused field draw
//
// Replaces, for the flat-sky half of SURVEY 8(f) n4, the numpy transforms behind
// RandomField.getfield (cora/core/gaussianfield.py:102-120: (randn + i randn) kweight -> irfftn),
// fftutil.rfftn / irfftn (cora/util/fftutil.py:64-87) and the mixed ifft/irfft of
// ForegroundMap.getfield (cora/foreground/gaussianfg.py:72-84).
//
// One kernel does a batch of 1-D transforms ("lines") along one axis of a C-contiguous array, a tile of
// T adjacent lines per workgroup staged through LDS so that global accesses stay coalesced whatever the
// stride of the axis:
//   - power-of-two lengths: in-LDS radix-4 (+ one radix-2) decimation-in-time passes on bit-reversed input;
//   - every other length (the reference pads cubes to arbitrary sizes, corr.py:655-671): Bluestein with
//     plan-time chirp and filter tables, DIF forward -> filter (stored bit-reversed) -> DIT, so no
//     permutation pass exists;
//   - inverse transforms by conjugation on load and store; real transforms as a Hermitian-extended
//     (c2r) or zero-imaginary (r2c) complex line - numpy.fft.irfft semantics: the imaginary parts of
//     the DC and Nyquist bins are ignored.
// Line length <= 4096 (LDS holds a tile of P <= 8192 points).
#include <cmath>
#include <complex>
#include <vector>

#include "common.h"
#include "rng_dev.h"

#ifndef FS_ABLATE
#define FS_ABLATE 0   // diagnostic builds (wrong results): 1 no LDS passes, 2 also no bit-reversed commit
#endif
#ifndef FS_PAIR_XCD
#define FS_PAIR_XCD 3   // log2 of the adjacent tiles given to one XCD at a time (0: off); measured 1024^3 axis-1 pass: 7.05 / 6.44 / 6.2 / 6.08 ms for 0 / 1 / 2 / 3
#endif
#ifndef FS_THREADS
#define FS_THREADS 512
#endif
#ifndef FS_LDS_ELEMS
#define FS_LDS_ELEMS 4096   // complex elements per tile (64 KB)
#endif
#define FS_MAXN 4096

__device__ static inline double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
__device__ static inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ static inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ static inline double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }  // a * (-i)
__device__ static inline unsigned brev_n(unsigned i, int logP) {
#if FS_ABLATE == 2
    return i;
#endif
    return logP ? (__brev(i) >> (32 - logP)) : 0u;
}

// LDS layout of a line: element i lives at i ^ S(i >> 4), an XOR swizzle of the 16-byte slot inside its 256-byte
// bank row by parities of the row index x = i >> 4:
//   bit0 = x0^x2^x5, bit1 = x0^x3^x6, bit2 = x0^x1^x7, bit3 = x0^x4.
// Found by simulating ds_read_b128 / ds_write_b128 with their lane groups (MI355X_MICROARCH.md, LDS table:
// reads in 4 groups of 16 non-contiguous lanes, writes in 8 x 8) over every access pattern of this file: the
// bit-reversed commit and all radix-4 stages of P = 256 ... 4096 are conflict-free (stride-8 stages 1.5x); the
// plain layout had 2-4-way conflicts on the stride-1 / stride-4 stages and 8-way on the commit (65 % of the LDS
// cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE), one-slot paddings only move them between stages.  S is
// XOR-linear, so for the elements i0 + r h of a butterfly (the bits of r h are clear in i0)
// S((i0 + r h) >> 4) = S(i0 >> 4) ^ S((r h) >> 4): one evaluation per butterfly plus wave-uniform constants.
__device__ __host__ static inline int swz_bits(int x) {
    int s = (0 - (x & 1)) & 15;                 // x0 -> all four bits
    s ^= (x >> 2) & 3;                          // x2, x3 -> bits 0, 1
    s ^= (x & 2) << 1;                          // x1 -> bit 2
    s ^= (x & 16) >> 1;                         // x4 -> bit 3
    s ^= (x >> 5) & 7;                          // x5, x6, x7 -> bits 0, 1, 2
    return s;
}
__device__ static inline int swz(int i) { return i ^ swz_bits(i >> 4); }
// line t of a tile additionally rotates its slots by t (XOR of the low four bits): the lines of a tile are P apart,
// i.e. in the same banks, and the commit / store of a strided axis touch T lines with consecutive lanes
__device__ static inline int line_x(int t, int P) { return t & 15 & (P - 1); }   // (stays inside the line when P < 16)

// e / d for 0 <= e < 2^23 with rcp = 1.0f / d: one float multiply and a +-1 correction instead of the
// ~35-instruction integer division (the element loops below were 80 % integer division before)
__device__ static inline int fastdiv(int e, int d, float rcp, int &rem) {
    int t = (int)((float)e * rcp);
    rem = e - t * d;
    if (rem < 0) t--, rem += d;
    if (rem >= d) t++, rem -= d;
    return t;
}

// decimation in time, forward sign: position i holds x[bitrev(i)] on entry, X[i] on exit; `lines` lines of P
// `tw` holds e^{-2 pi i k / P} for k < P/4 (only the quarter circle is ever indexed); the kernel keeps it in LDS:
// a twiddle fetched from global memory inside a pass would wait (vmcnt is in-order) for the register prefetch
// of the next tile issued just before the passes.
__device__ __forceinline__ static void lds_dit(double2 *buf, int P, int logP, int lines, const double2 *tw) {
    int h = 1;
    if (logP & 1) {
        for (int w = threadIdx.x; w < lines * (P >> 1); w += FS_THREADS) {
            const int i = (2 * w) & (P - 1);                              // elements i, i + 1 share a bank row
            const int sb = swz_bits(i >> 4) ^ line_x((2 * w) >> logP, P);
            double2 *ln = buf + ((2 * w) & ~(P - 1));
            const double2 a = ln[i ^ sb], b = ln[(i + 1) ^ sb];
            ln[i ^ sb] = cadd(a, b);
            ln[(i + 1) ^ sb] = csub(a, b);
        }
        __syncthreads();
        h = 2;
    }
    const int q = P >> 2, lq = logP - 2;
    int lh = (logP & 1);                        // log2 h: every index below is a shift / mask (no integer division)
    for (; h < P; h <<= 2, lh += 2) {
        const int ls2 = logP - 2 - lh;          // log2 of the twiddle stride P / 4h
        const int c1s = swz_bits(h >> 4), c2s = swz_bits((2 * h) >> 4), c3s = swz_bits((3 * h) >> 4);
        for (int w = threadIdx.x; w < lines * q; w += FS_THREADS) {
            const int t = w >> lq, j = w & (q - 1);
            const int pos = j & (h - 1), grp = j >> lh;
            const int i0 = ((grp << 2) << lh) + pos, s0 = swz_bits(i0 >> 4) ^ line_x(t, P);
            double2 *ln = buf + t * P;
            const int e0 = i0 ^ s0, e1 = (i0 + h) ^ s0 ^ c1s, e2 = (i0 + 2 * h) ^ s0 ^ c2s, e3 = (i0 + 3 * h) ^ s0 ^ c3s;
            const double2 w2 = tw[pos << ls2], w1 = cmul(w2, w2);
            const double2 a = ln[e0], b = cmul(w1, ln[e1]), c = ln[e2], d = cmul(w1, ln[e3]);
            const double2 a1 = cadd(a, b), b1 = csub(a, b);
            const double2 c1 = cmul(w2, cadd(c, d)), d1 = cmul(mul_mi(w2), csub(c, d));
            ln[e0] = cadd(a1, c1);
            ln[e1] = cadd(b1, d1);
            ln[e2] = csub(a1, c1);
            ln[e3] = csub(b1, d1);
        }
        __syncthreads();
    }
}

// decimation in frequency, forward sign: natural order in, position i holds X[bitrev(i)] on exit
__device__ __forceinline__ static void lds_dif(double2 *buf, int P, int logP, int lines, const double2 *tw) {
    const int q = P >> 2, lq = logP - 2;
    const int hmin = (logP & 1) ? 2 : 1;
    int lh = logP - 2;
    for (int h = P >> 2; h >= hmin; h >>= 2, lh -= 2) {
        const int ls2 = logP - 2 - lh;
        const int c1s = swz_bits(h >> 4), c2s = swz_bits((2 * h) >> 4), c3s = swz_bits((3 * h) >> 4);
        for (int w = threadIdx.x; w < lines * q; w += FS_THREADS) {
            const int t = w >> lq, j = w & (q - 1);
            const int pos = j & (h - 1), grp = j >> lh;
            const int i0 = ((grp << 2) << lh) + pos, s0 = swz_bits(i0 >> 4) ^ line_x(t, P);
            double2 *ln = buf + t * P;
            const int e0 = i0 ^ s0, e1 = (i0 + h) ^ s0 ^ c1s, e2 = (i0 + 2 * h) ^ s0 ^ c2s, e3 = (i0 + 3 * h) ^ s0 ^ c3s;
            const double2 w2 = tw[pos << ls2], w1 = cmul(w2, w2);
            const double2 x0 = ln[e0], x1 = ln[e1], x2 = ln[e2], x3 = ln[e3];
            const double2 a1 = cadd(x0, x2), c1 = cmul(w2, csub(x0, x2));
            const double2 b1 = cadd(x1, x3), d1 = cmul(mul_mi(w2), csub(x1, x3));
            ln[e0] = cadd(a1, b1);
            ln[e1] = cmul(w1, csub(a1, b1));
            ln[e2] = cadd(c1, d1);
            ln[e3] = cmul(w1, csub(c1, d1));
        }
        __syncthreads();
    }
    if (logP & 1) {
        for (int w = threadIdx.x; w < lines * (P >> 1); w += FS_THREADS) {
            const int i = (2 * w) & (P - 1);                              // elements i, i + 1 share a bank row
            const int sb = swz_bits(i >> 4) ^ line_x((2 * w) >> logP, P);
            double2 *ln = buf + ((2 * w) & ~(P - 1));
            const double2 a = ln[i ^ sb], b = ln[(i + 1) ^ sb];
            ln[i ^ sb] = cadd(a, b);
            ln[(i + 1) ^ sb] = csub(a, b);
        }
        __syncthreads();
    }
}

struct linefft_args {
    const double *in;
    double *out;
    long nouter, inner;  // lines = nouter * inner; element j of line (o, i) is at ((o n_mem + j) inner + i)
    int n;               // transform length
    int nh;              // n / 2 + 1 (real modes)
    int T;               // lines per tile
    int inverse;         // sign +, for MODE 0
    double scale;
    int P, logP, blu;
    int twl;             // quarter-circle twiddle table kept in LDS behind the tile (P <= 4096)
    int h;               // modes 3/4: complex length n_real / 2 (= n), the real length is 2 h
    const double2 *tw, *chirp, *filt;
    const double2 *rtw;  // modes 3/4: e^{+2 pi i k / (2h)}, k = 0 .. h/2
    uint64_t seed;       // mode 5: Philox key of the generated input
    int gen_lds;         // mode 5: Box-Muller tables in LDS (when they fit beside a second workgroup's tile)
};

// MODE 5: MODE 0, inverse, whose INPUT is generated where it is committed to LDS: element e of the (never materialised)
// spectrum is kweight[e] (N(0,1) + i N(0,1)), the pair being the Box-Muller outputs of Philox counter e - exactly
// what randomfield_draw_kernel writes, so that RandomField.getfield(seed) = (normals x kweight) -> irfftn
// (cora/core/gaussianfield.py:115-119) loses the 16 bytes written and 16 read per element of a separate draw pass:
// A.in = kweight (real), A.out = the spectrum workspace the following passes work on.
// MODE 0: complex -> complex (in place allowed), 1: half-complex -> real (inverse), 2: real -> half-complex
// (1 and 2 transform a full-length complex line and serve odd lengths); 3 / 4: the same two for EVEN real lengths
// 2h through ONE complex transform of length h (A.n = h): c2r packs Z_k = (X_k + conj X_{h-k}) + i W^k (X_k -
// conj X_{h-k}), W = e^{2 pi i / 2h}, whose inverse transform is x_{2j} + i x_{2j+1}; r2c transforms
// z_j = x_{2j} + i x_{2j+1} and unpacks X_k = E_k + W^{-k} O_k, X_{h-k} = conj(E_k - W^{-k} O_k) with
// E_k = (Z_k + conj Z_{h-k}) / 2, O_k = (Z_k - conj Z_{h-k}) / 2i.
//
// Persistent workgroups walk the tiles; the global loads of the NEXT tile are issued into registers before
// the LDS passes of the current one and committed to LDS after its stores, so HBM latency hides behind the
// transform (FS_NLOAD loads in flight per thread).
#define FS_WG_PER_CU (FS_THREADS >= 1024 ? 1 : 2)
#define FS_NLOAD ((FS_LDS_ELEMS + FS_THREADS - 1) / FS_THREADS)
template <int MODE>
__global__ void __launch_bounds__(FS_THREADS, FS_WG_PER_CU * FS_THREADS / 256) linefft_kernel(const linefft_args A) {
    extern __shared__ double2 fs_lds[];
    const int n = A.n, P = A.P, logP = A.logP, T = A.T;
    const bool inv = MODE == 1 || MODE == 5 || (MODE == 0 && A.inverse);
    const int h = A.h, hp = (h >> 1) + 1;      // modes 3/4: pairs (k, h - k), k = 0 .. h/2
    const double2 *in2 = reinterpret_cast<const double2 *>(A.in);
    double2 *out2 = reinterpret_cast<double2 *>(A.out);
    const long chunks = (A.inner + T - 1) / T;  // tiles per outer index (inner > 1)
    const long nlines = A.nouter * A.inner;
    const long ntiles = A.inner == 1 ? (nlines + T - 1) / T : A.nouter * chunks;
    const int nin = MODE == 1 ? A.nh : (MODE == 3 ? n + 1 : n);   // input elements per line (mode 4: double2 = 2 reals)
    const int nout = MODE == 2 ? A.nh : (MODE == 4 ? n + 1 : n);  // output elements per line (mode 3: double2 = 2 reals)

    // LDS behind the tile: quarter-circle twiddles; modes 3/4: W^k, k <= h/2, then one slot per line for X_h
    double2 *twl = fs_lds + T * P;
    double2 *rtwl = twl + (A.twl ? (P >> 2) : 0);
    double2 *xh = rtwl + hp;
    // mode 5: the Box-Muller tables of the generator (rng_dev.h) behind the twiddles: a table fetched from global memory
    // inside the chain waits out an L2 round trip per element (the lesson of K3, round 2)
    double2 *lg_l = twl + (A.twl ? (P >> 2) : 0), *sc_l = lg_l + 257;
    const double2 *lg_t = (MODE == 5 && A.gen_lds) ? lg_l : RNG_LOG_TAB, *sc_t = (MODE == 5 && A.gen_lds) ? sc_l : RNG_SC_TAB;

    struct tile_t {
        long outer, i0;
        int teff;
        float rcp_teff;
    };
    auto tile_of = [&](long tile) {
        tile_t tl;
        if (A.inner == 1) {
            tl.outer = tile * T;  // first line
            tl.i0 = 0;
            tl.teff = (int)min((long)T, nlines - tl.outer);
        } else {
            tl.outer = tile / chunks;
            tl.i0 = (tile - tl.outer * chunks) * T;
            tl.teff = (int)min((long)T, A.inner - tl.i0);
        }
        tl.rcp_teff = 1.0f / (float)tl.teff;
        return tl;
    };
    // element e of a tile -> (line t, index j along the axis, global element offset) for `len` elements per line
    const float rcp_nin = 1.0f / (float)nin, rcp_nout = 1.0f / (float)nout, rcp_hp = 1.0f / (float)hp;
    auto locate = [&](const tile_t &tl, int e, int len, float rcp_len, int &t, int &j) -> long {
        if (A.inner == 1) {
            t = fastdiv(e, len, rcp_len, j);
            return (tl.outer + t) * len + j;
        }
        j = fastdiv(e, tl.teff, tl.rcp_teff, t);
        return (tl.outer * len + j) * A.inner + tl.i0 + t;
    };
    double2 R[FS_NLOAD];
    auto prefetch = [&](const tile_t &tl) {
#pragma unroll
        for (int u = 0; u < FS_NLOAD; u++) {
            const int e = threadIdx.x + u * FS_THREADS;
            if (e < tl.teff * nin) {
                int t, j;
                const long addr = locate(tl, e, nin, rcp_nin, t, j);
                R[u] = (MODE == 2 || MODE == 5) ? make_double2(A.in[addr], 0.0) : in2[addr];
            }
        }
    };
    auto commit = [&](const tile_t &tl) {
        if (A.blu) {
            const float rcp_z = 1.0f / (float)(P - n);
            for (int e = threadIdx.x; e < tl.teff * (P - n); e += FS_THREADS) {
                int j;
                const int t = fastdiv(e, P - n, rcp_z, j);
                fs_lds[t * P + (swz(n + j) ^ line_x(t, P))] = make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < FS_NLOAD; u++) {
            const int e = threadIdx.x + u * FS_THREADS;
            if (e < tl.teff * nin) {
                int t, j;
                const long addr = locate(tl, e, nin, rcp_nin, t, j);
                double2 v = R[u];
                if (MODE == 5) {      // the k-weight arrived in R[u].x: the normals of this element are made here
                    const double2 z = philox_boxmuller((uint64_t)addr, A.seed, lg_t, sc_t);
                    v = make_double2(z.x * v.x, z.y * v.x);
                }
                if (MODE == 3) {
                    // raw spectrum at natural positions; the packing pass below works on pairs in place
                    if (j == h)
                        xh[t] = v;
                    else
                        fs_lds[t * P + (swz(j) ^ line_x(t, P))] = v;
                } else if (MODE == 1) {
                    if (j == 0 || 2 * j == n) v.y = 0.0;
                    // conj of the Hermitian extension: position j gets conj(c_j), position n-j gets c_j
                    const double2 lo = make_double2(v.x, -v.y);
                    const bool mirror = j > 0 && 2 * j < n;
                    if (A.blu) {
                        fs_lds[t * P + (swz(j) ^ line_x(t, P))] = cmul(lo, A.chirp[j]);
                        if (mirror) fs_lds[t * P + (swz(n - j) ^ line_x(t, P))] = cmul(v, A.chirp[n - j]);
                    } else {
                        fs_lds[t * P + (swz(brev_n(j, logP)) ^ line_x(t, P))] = lo;
                        if (mirror) fs_lds[t * P + (swz(brev_n(n - j, logP)) ^ line_x(t, P))] = v;
                    }
                } else {
                    if (inv) v.y = -v.y;
                    if (A.blu)
                        fs_lds[t * P + (swz(j) ^ line_x(t, P))] = cmul(v, A.chirp[j]);
                    else
                        fs_lds[t * P + (swz(brev_n(j, logP)) ^ line_x(t, P))] = v;
                }
            }
        }
    };

    long tile = blockIdx.x;
    if (tile >= ntiles) return;
    if (A.twl)
        for (int k = threadIdx.x; k < (P >> 2); k += FS_THREADS) twl[k] = A.tw[k];
    if (MODE == 3 || MODE == 4)
        for (int k = threadIdx.x; k < hp; k += FS_THREADS) rtwl[k] = A.rtw[k];
    if (MODE == 5 && A.gen_lds) {
        for (int k = threadIdx.x; k < 257; k += FS_THREADS) lg_l[k] = RNG_LOG_TAB[k];
        for (int k = threadIdx.x; k < 256; k += FS_THREADS) sc_l[k] = RNG_SC_TAB[k];
        __syncthreads();
    }
    // Strided axes read and write 16 T-byte segments: two tiles that are neighbours along the contiguous axis share
    // every 128-byte line.  Workgroups b and b + 8 run on the same XCD (round-robin dispatch) at the same time, so
    // they (and b + 16, ...) are given adjacent tiles and a line is fetched into that XCD's L2 once instead of into several L2s.
    constexpr int GL = FS_PAIR_XCD;    // log2 of the tiles per group (1: pairs)
    const long gmask = (8L << GL) - 1;
    const bool pair_xcd = GL > 0 && A.inner != 1 && (ntiles & gmask) == 0 && (gridDim.x & gmask) == 0;
    auto remap = [&](long v) {
        if (!pair_xcd) return v;
        const long slot = v >> 3, xcd = v & 7;
        return (((slot >> GL) * 8 + xcd) << GL) + (slot & ((1 << GL) - 1));
    };
    tile_t cur = tile_of(remap(tile));
    prefetch(cur);
    while (true) {
        commit(cur);
        __syncthreads();
        const long next = tile + gridDim.x;
        tile_t nxt = cur;
        if (next < ntiles) {
            nxt = tile_of(remap(next));
            prefetch(nxt);
        }
        if (MODE == 3) {
            // pack pairs (k, h-k) in place -> conj(Z) (the inverse transform runs as conj(FFT(conj .)))
            for (int e = threadIdx.x; e < cur.teff * hp; e += FS_THREADS) {
                int k;
                const int t = fastdiv(e, hp, rcp_hp, k), k2 = h - k;
                double2 *ln = fs_lds + t * P;
                const int pk = swz(k) ^ line_x(t, P), pk2 = swz(k2 < h ? k2 : 0) ^ line_x(t, P);
                double2 xk = ln[pk], xc = k == 0 ? xh[t] : ln[pk2];
                if (k == 0) xk.y = 0.0, xc.y = 0.0;               // Im of the DC / Nyquist bins is ignored
                const double2 E = make_double2(xk.x + xc.x, xk.y - xc.y);
                const double2 O = cmul(make_double2(xk.x - xc.x, xk.y + xc.y), rtwl[k]);
                double2 a = make_double2(E.x - O.y, -(E.y + O.x));      // conj(E + i O)
                double2 b = make_double2(E.x + O.y, -(O.x - E.y));      // conj(conj E + i conj O)
                if (A.blu) {
                    a = cmul(a, A.chirp[k]);
                    if (k2 < h) b = cmul(b, A.chirp[k2]);
                }
                ln[pk] = a;
                if (k2 != k && k2 < h) ln[pk2] = b;
            }
            __syncthreads();
        }
        // ---- transform --------------------------------------------------------------------------
        if (A.blu) {
            if (A.twl)
                lds_dif(fs_lds, P, logP, cur.teff, twl);
            else
                lds_dif(fs_lds, P, logP, cur.teff, A.tw);
            for (int e = threadIdx.x; e < cur.teff * P; e += FS_THREADS) {
                const int k = e & (P - 1), pe = (e - k) + (swz(k) ^ line_x(e >> logP, P));
                const double2 z = cmul(fs_lds[pe], A.filt[k]);
                fs_lds[pe] = make_double2(z.x, -z.y);
            }
            __syncthreads();
        }
        if (MODE == 3 && !A.blu) {
            // natural order in (the packed pairs), bit-reversed out; the store below reads through the permutation
            if (A.twl)
                lds_dif(fs_lds, P, logP, cur.teff, twl);
            else
                lds_dif(fs_lds, P, logP, cur.teff, A.tw);
        } else if (FS_ABLATE == 0) {
            if (A.twl)
                lds_dit(fs_lds, P, logP, cur.teff, twl);
            else
                lds_dit(fs_lds, P, logP, cur.teff, A.tw);
        }
        // ---- store ------------------------------------------------------------------------------
        // transform value Y_k of line t in natural order
        auto result = [&](int t, int k) {
            double2 v = fs_lds[t * P + (swz((MODE == 3 && !A.blu) ? (int)brev_n(k, logP) : k) ^ line_x(t, P))];
            if (A.blu) v = cmul(make_double2(v.x, -v.y), A.chirp[k]);
            return v;
        };
        if (MODE == 4) {
            for (int e = threadIdx.x; e < cur.teff * hp; e += FS_THREADS) {
                int k;
                const int t = fastdiv(e, hp, rcp_hp, k), k2 = h - k;
                const double2 zk = result(t, k), zc = result(t, k2 == h ? 0 : k2);
                const double2 Ea = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y));
                const double2 Oa = make_double2(0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x));   // (zk - conj zc) / 2i
                const double2 w = rtwl[k];
                const double2 B = cmul(make_double2(w.x, -w.y), Oa);
                double2 *o = out2 + (cur.outer + t) * (long)(h + 1);
                o[k] = make_double2((Ea.x + B.x) * A.scale, (Ea.y + B.y) * A.scale);
                if (k2 != k) o[k2] = make_double2((Ea.x - B.x) * A.scale, -(Ea.y - B.y) * A.scale);
            }
        } else
        for (int e = threadIdx.x; e < cur.teff * nout; e += FS_THREADS) {
            int t, k;
            const long addr = locate(cur, e, nout, rcp_nout, t, k);
            double2 v = result(t, k);
            if (MODE == 3) {
                out2[addr] = make_double2(v.x * A.scale, -v.y * A.scale);   // conj: x_{2j} + i x_{2j+1}
            } else if (MODE == 1) {
                A.out[addr] = v.x * A.scale;
            } else {
                if (inv) v.y = -v.y;
                out2[addr] = make_double2(v.x * A.scale, v.y * A.scale);
            }
        }
        if (next >= ntiles) break;
        __syncthreads();
        tile = next;
        cur = nxt;
    }
}

// spec[e] = (N(0,1) + i N(0,1)) kweight[e], the pair being the Box-Muller outputs of Philox counter e
__global__ void randomfield_draw_kernel(const double *__restrict__ kw, long count, uint64_t seed,
                                        double2 *__restrict__ spec) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
        const double2 z = philox_boxmuller((uint64_t)e, seed);
        const double w = kw[e];
        spec[e] = make_double2(z.x * w, z.y * w);
    }
}

// out[f][m] = aff[m] * sum_c W[f][c] g[c][m]: the frequency mixing of ForegroundMap.getfield
// (cora/foreground/gaussianfg.py:79-82): real normals g, complex angular spectrum aff
__global__ void fg_mix_kernel(const double *__restrict__ W, const double *__restrict__ g,
                              const double2 *__restrict__ aff, int ncorr, long M, double2 *__restrict__ out) {
    const int f = blockIdx.y;
    const double *wr = W + (size_t)f * ncorr;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int c = 0; c < ncorr; c++) s = fma(wr[c], g[(size_t)c * M + m], s);
        const double2 a = aff[m];
        out[(size_t)f * M + m] = make_double2(s * a.x, s * a.y);
    }
}

// spec[e] *= w[e]  (complex spectrum times a real k-space weight: the mu^2 of corr.py:590-599)
__global__ void spec_mul_real_kernel(double2 *__restrict__ spec, const double *__restrict__ w, long count) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
        const double2 v = spec[e];
        const double f = w[e];
        spec[e] = make_double2(v.x * f, v.y * f);
    }
}

// out[z][p] = a[z] df[z][p] (+ b[z] vf[z][p]) + c[z]: the per-slice growth / bias / mean of corr.py:712-726
__global__ void cube_affine_kernel(const double *__restrict__ df, const double *__restrict__ vf,
                                   const double *__restrict__ a, const double *__restrict__ b,
                                   const double *__restrict__ c, long plane, double *__restrict__ out) {
    const int z = blockIdx.y;
    const double az = a[z], bz = vf ? b[z] : 0.0, cz = c[z];
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
        const size_t e = (size_t)z * plane + p;
        // same operation order as the reference: (df a) + (vf b), then + mean
        double v = df[e] * az;
        if (vf) v += vf[e] * bz;
        out[e] = v + cz;
    }
}

// Ray-traced resampling of the comoving cube onto (redshift slice, angle, angle): scipy.ndimage.map_coordinates
// with order = 1 and the default mode 'constant' (corr.py:744-768): trilinear interpolation at
//   (zc[i], (tx[ix] s[i]) / wx (n1 - 1) + (n1 - 1)/2, (ty[iy] s[i]) / wy (n2 - 1) + (n2 - 1)/2),
// exactly 0 when any coordinate lies outside [0, n - 1] (no interpolation towards the fill value).
__global__ void raytrace_kernel(const double *__restrict__ cube, int n0, int n1, int n2, const double *__restrict__ zc,
                                const double *__restrict__ s, const double *__restrict__ tx,
                                const double *__restrict__ ty, double wx, double wy, int numx, int numy,
                                double *__restrict__ out) {
    const int i = blockIdx.y;
    const double cz = zc[i], si = s[i];
    const long plane = (long)numx * numy;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
        const int ix = (int)(p / numy), iy = (int)(p - (long)ix * numy);
        const double cx = (tx[ix] * si) / wx * (n1 - 1.0) + 0.5 * (n1 - 1.0);
        const double cy = (ty[iy] * si) / wy * (n2 - 1.0) + 0.5 * (n2 - 1.0);
        double v = 0.0;
        if (cz >= 0.0 && cz <= n0 - 1.0 && cx >= 0.0 && cx <= n1 - 1.0 && cy >= 0.0 && cy <= n2 - 1.0) {
            const int z0 = (int)cz, x0 = (int)cx, y0 = (int)cy;   // floor: coordinates are >= 0
            const double fz = cz - z0, fx = cx - x0, fy = cy - y0;
            const int z1 = z0 + 1 < n0 ? z0 + 1 : z0, x1 = x0 + 1 < n1 ? x0 + 1 : x0, y1 = y0 + 1 < n2 ? y0 + 1 : y0;
            const double wz[2] = {1.0 - fz, fz}, wxx[2] = {1.0 - fx, fx}, wyy[2] = {1.0 - fy, fy};
            const int zi[2] = {z0, z1}, xi[2] = {x0, x1}, yi[2] = {y0, y1};
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int c = 0; c < 2; c++)
                        v += wz[a] * wxx[b] * wyy[c] * cube[((size_t)zi[a] * n1 + xi[b]) * n2 + yi[c]];
        }
        out[(size_t)i * plane + p] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
typedef std::complex<long double> cld;

static void host_fft(std::vector<cld> &x) {  // forward, power-of-two, recursive (plan time only)
    const size_t n = x.size();
    if (n < 2) return;
    std::vector<cld> ev(n / 2), od(n / 2);
    for (size_t i = 0; i < n / 2; i++) ev[i] = x[2 * i], od[i] = x[2 * i + 1];
    host_fft(ev);
    host_fft(od);
    const long double PI = 3.14159265358979323846264338327950288L;
    for (size_t k = 0; k < n / 2; k++) {
        const long double a = -2.0L * PI * (long double)k / (long double)n;
        const cld t = cld(cosl(a), sinl(a)) * od[k];
        x[k] = ev[k] + t;
        x[k + n / 2] = ev[k] - t;
    }
}

static int get_linefft_plan(corahip_ctx *ctx, int n, const corahip_linefft_plan **out) {
    auto it = ctx->linefft.find(n);
    if (it != ctx->linefft.end()) {
        *out = &it->second;
        return 0;
    }
    const long double PI = 3.14159265358979323846264338327950288L;
    corahip_linefft_plan pl;
    pl.n = n;
    pl.blu = (n & (n - 1)) != 0;
    int P = 1;
    while (P < (pl.blu ? 2 * n - 1 : n)) P <<= 1;
    pl.P = P;
    pl.logP = 0;
    while ((1 << pl.logP) < P) pl.logP++;
    std::vector<double2> tw(P);
    for (int k = 0; k < P; k++) {
        const long double a = -2.0L * PI * (long double)k / (long double)P;
        tw[k] = make_double2((double)cosl(a), (double)sinl(a));
    }
    HIP_TRY(hipMalloc((void **)&pl.tw, sizeof(double2) * P));
    HIP_TRY(hipMemcpy(pl.tw, tw.data(), sizeof(double2) * P, hipMemcpyHostToDevice));
    if (pl.blu) {
        std::vector<double2> chirp(n), filt(P);
        std::vector<cld> b(P, cld(0, 0));
        for (long k = 0; k < n; k++) {
            const long r = (k * k) % (2L * n);  // k^2 mod 2n keeps the angle exact
            const long double a = PI * (long double)r / (long double)n;
            chirp[k] = make_double2((double)cosl(a), (double)-sinl(a));
            const cld bk(cosl(a), sinl(a));
            b[k] = bk;
            if (k) b[P - k] = bk;
        }
        host_fft(b);
        for (int i = 0; i < P; i++) {
            unsigned r = 0;
            for (int bit = 0; bit < pl.logP; bit++) r |= ((i >> bit) & 1u) << (pl.logP - 1 - bit);
            filt[i] = make_double2((double)(b[r].real() / P), (double)(b[r].imag() / P));
        }
        HIP_TRY(hipMalloc((void **)&pl.chirp, sizeof(double2) * n));
        HIP_TRY(hipMemcpy(pl.chirp, chirp.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&pl.filt, sizeof(double2) * P));
        HIP_TRY(hipMemcpy(pl.filt, filt.data(), sizeof(double2) * P, hipMemcpyHostToDevice));
        int rcb = flat_blu_plan(ctx, n, pl.chirp, &pl.Pct, &pl.filt_ct);   // (the compile-time convolution, where a schedule holds 2 n - 1)
        if (rcb) return rcb;
    }
    {   // unpacking twiddles of the real transform of length 2n through this complex plan: e^{+2 pi i k / 2n}, k <= n/2
        std::vector<double2> rtw(n / 2 + 1);
        for (int k = 0; k <= n / 2; k++) {
            const long double a = PI * (long double)k / (long double)n;
            rtw[k] = make_double2((double)cosl(a), (double)sinl(a));
        }
        HIP_TRY(hipMalloc((void **)&pl.rtw, sizeof(double2) * rtw.size()));
        HIP_TRY(hipMemcpy(pl.rtw, rtw.data(), sizeof(double2) * rtw.size(), hipMemcpyHostToDevice));
    }
    auto ins = ctx->linefft.emplace(n, pl);
    *out = &ins.first->second;
    return 0;
}

template <int MODE>
static int launch_linefft(corahip_ctx *ctx, const double *in, double *out, long nouter, int n, long inner,
                          int inverse, double scale, uint64_t seed = 0) {
    ARG_CHECK(n >= 1 && n <= FS_MAXN);
    if (nouter * inner == 0) return 0;
    StageTimer pass_timer(ctx, (MODE == 1 || MODE == 3) ? "fft_c2r" : ((MODE == 2 || MODE == 4) ? "fft_r2c" : (MODE == 5 ? "fft_c2c_draw" : (inner == 1 ? "fft_c2c_contig" : "fft_c2c_strided"))));
    if ((MODE == 0 || MODE == 5) && inner > 1) {     // strided complex pass: the compile-time passes where the length has them
        bool took = false;
        int rct = flat_c2c_ct(ctx, in, out, nouter, n, inner, inverse, scale, MODE == 5, seed, &took);
        if (rct || took) return rct;
    }
    const corahip_linefft_plan *pl;
    int rc = get_linefft_plan(ctx, n, &pl);
    if (rc) return rc;
    if ((MODE == 3 || MODE == 4) && inner == 1 && pl->blu && pl->Pct) {      // even real length 2 n, n arbitrary
        bool took = false;
        int rct = flat_blu_real_ct(ctx, MODE == 3, in, out, nouter, n, scale, pl->Pct, pl->chirp, pl->filt_ct, pl->rtw, &took);
        if (rct || took) return rct;
    }
    if ((MODE == 0 || MODE == 5) && inner > 1 && pl->blu && pl->Pct) {
        bool took = false;
        int rct = flat_blu_c2c_ct(ctx, in, out, nouter, n, inner, inverse, scale, MODE == 5, seed, pl->Pct, pl->chirp, pl->filt_ct, &took);
        if (rct || took) return rct;
    }
    linefft_args A;
    A.in = in;
    A.out = out;
    A.nouter = nouter;
    A.inner = inner;
    A.n = n;
    A.nh = n / 2 + 1;
    A.inverse = inverse;
    A.scale = scale;
    A.P = pl->P;
    A.logP = pl->logP;
    A.blu = pl->blu;
    A.tw = pl->tw;
    A.chirp = pl->chirp;
    A.filt = pl->filt;
    A.rtw = pl->rtw;
    A.h = n;
    A.seed = seed;
    int T = FS_LDS_ELEMS / pl->P;
    T = T < 1 ? 1 : (T > 16 ? 16 : T);
    if (MODE == 3)   // n + 1 input elements per line must fit the register prefetch of a tile
        while (T > 1 && (long)T * (n + 1) > (long)FS_NLOAD * FS_THREADS) T--;
    A.T = T;
    A.twl = pl->P <= 4096;
    size_t shm = sizeof(double2) * ((size_t)T * pl->P + (A.twl ? pl->P / 4 : 0) +
                                    ((MODE == 3 || MODE == 4) ? (size_t)(n / 2 + 1) + T : 0));
    // mode 5: the generator's tables in LDS only where they do not cost the second workgroup of the CU its place
    A.gen_lds = MODE == 5 && 2 * (shm + sizeof(double2) * 513) <= 160 * 1024;
    if (A.gen_lds) shm += sizeof(double2) * 513;
    const long chunks = (inner + T - 1) / T;
    const long ntiles = inner == 1 ? (nouter + T - 1) / T : nouter * chunks;
    const long per_cu = (FS_WG_PER_CU == 2 && shm * 2 <= 160 * 1024) ? 2 : 1;   // launch bounds allow two 8-wave workgroups per CU
    const long maxgrid = (long)ctx->num_cu * per_cu;
    const int grid = (int)(ntiles < maxgrid ? ntiles : maxgrid);
    HIP_TRY(hipFuncSetAttribute((const void *)linefft_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    hipLaunchKernelGGL(linefft_kernel<MODE>, dim3(grid), dim3(FS_THREADS), shm, ctx->stream, A);
    LAUNCH_CHECK();
    return 0;
}

static long prod(const int64_t *d, int a, int b) {
    long p = 1;
    for (int i = a; i < b; i++) p *= d[i];
    return p;
}

// half-complex -> real along the contiguous axis (real length n): even lengths through ONE complex transform of half
// the length - with the compile-time passes where n / 2 has a schedule -, odd lengths as a full-length Hermitian line
static int last_axis_c2r(corahip_ctx *ctx, double *spec, double *out, long nlines, int n) {
    if (n % 2 == 0) {
        bool took = false;
        {
            StageTimer pass_timer(ctx, "fft_c2r");
            int rc = flat_c2r_ct(ctx, spec, out, nlines, n / 2, 1.0 / n, &took);
            if (rc) return rc;
        }
        if (took) return 0;
        return launch_linefft<3>(ctx, spec, out, nlines, n / 2, 1, 1, 1.0 / n);
    }
    return launch_linefft<1>(ctx, spec, out, nlines, n, 1, 1, 1.0 / n);
}

extern "C" {

int corahip_fft_c2c(corahip_ctx *ctx, double *data, int ndim, const int64_t *dims, int axis, int inverse) {
    ARG_CHECK(ctx && data && dims && ndim >= 1 && ndim <= 8 && axis >= 0 && axis < ndim);
    for (int i = 0; i < ndim; i++) ARG_CHECK(dims[i] >= 1);
    StageTimer st(ctx, "flatfft");
    const int n = (int)dims[axis];
    ARG_CHECK(dims[axis] <= FS_MAXN);
    return launch_linefft<0>(ctx, data, data, prod(dims, 0, axis), n, prod(dims, axis + 1, ndim), inverse,
                             inverse ? 1.0 / n : 1.0);
}

int corahip_irfftn(corahip_ctx *ctx, double *spec, int ndim, const int64_t *rdims, int naxes, double *out) {
    ARG_CHECK(ctx && spec && out && rdims && ndim >= 1 && ndim <= 8 && naxes >= 1 && naxes <= ndim);
    int64_t cd[8];
    for (int i = 0; i < ndim; i++) {
        ARG_CHECK(rdims[i] >= 1);
        cd[i] = rdims[i];
    }
    for (int i = ndim - naxes; i < ndim; i++) ARG_CHECK(rdims[i] <= FS_MAXN);
    cd[ndim - 1] = rdims[ndim - 1] / 2 + 1;
    StageTimer st(ctx, "flatfft");
    for (int ax = ndim - naxes; ax < ndim - 1; ax++) {
        const int n = (int)cd[ax];
        int rc = launch_linefft<0>(ctx, spec, spec, prod(cd, 0, ax), n, prod(cd, ax + 1, ndim), 1, 1.0 / n);
        if (rc) return rc;
    }
    const int n = (int)rdims[ndim - 1];
    return last_axis_c2r(ctx, spec, out, prod(rdims, 0, ndim - 1), n);
}

int corahip_randomfield_irfftn(corahip_ctx *ctx, const double *kweight, int ndim, const int64_t *rdims, uint64_t seed,
                               double *spec, double *out) {
    ARG_CHECK(ctx && kweight && spec && out && rdims && ndim >= 1 && ndim <= 8);
    int64_t cd[8];
    for (int i = 0; i < ndim; i++) {
        ARG_CHECK(rdims[i] >= 1 && rdims[i] <= FS_MAXN);
        cd[i] = rdims[i];
    }
    cd[ndim - 1] = rdims[ndim - 1] / 2 + 1;
    if (ndim == 1) {          // (no complex pass to generate in: the two-step form)
        int rc = corahip_randomfield_draw(ctx, kweight, cd[0], seed, spec);
        return rc ? rc : corahip_irfftn(ctx, spec, ndim, rdims, ndim, out);
    }
    StageTimer st(ctx, "flatfft");
    // first pass (axis 0): the spectrum is generated where the pass loads it
    int rc = launch_linefft<5>(ctx, kweight, spec, 1, (int)cd[0], prod(cd, 1, ndim), 1, 1.0 / cd[0], seed);
    if (rc) return rc;
    for (int ax = 1; ax < ndim - 1; ax++) {
        const int n = (int)cd[ax];
        if ((rc = launch_linefft<0>(ctx, spec, spec, prod(cd, 0, ax), n, prod(cd, ax + 1, ndim), 1, 1.0 / n))) return rc;
    }
    const int n = (int)rdims[ndim - 1];
    return last_axis_c2r(ctx, spec, out, prod(rdims, 0, ndim - 1), n);
}

int corahip_rfftn(corahip_ctx *ctx, const double *in, int ndim, const int64_t *rdims, int naxes, double *spec) {
    ARG_CHECK(ctx && spec && in && rdims && ndim >= 1 && ndim <= 8 && naxes >= 1 && naxes <= ndim);
    int64_t cd[8];
    for (int i = 0; i < ndim; i++) {
        ARG_CHECK(rdims[i] >= 1);
        cd[i] = rdims[i];
    }
    for (int i = ndim - naxes; i < ndim; i++) ARG_CHECK(rdims[i] <= FS_MAXN);
    cd[ndim - 1] = rdims[ndim - 1] / 2 + 1;
    StageTimer st(ctx, "flatfft");
    const int nlast = (int)rdims[ndim - 1];
    int rc = 0;
    bool took = false;
    if (nlast % 2 == 0) {      // the compile-time passes where nlast / 2 has a schedule
        StageTimer pass_timer(ctx, "fft_r2c");
        rc = flat_r2c_ct(ctx, in, spec, prod(rdims, 0, ndim - 1), nlast / 2, &took);
        if (rc) return rc;
    }
    if (!took)
        rc = nlast % 2 == 0 ? launch_linefft<4>(ctx, in, spec, prod(rdims, 0, ndim - 1), nlast / 2, 1, 0, 1.0)
                            : launch_linefft<2>(ctx, in, spec, prod(rdims, 0, ndim - 1), nlast, 1, 0, 1.0);
    if (rc) return rc;
    for (int ax = ndim - 2; ax >= ndim - naxes; ax--) {
        rc = launch_linefft<0>(ctx, spec, spec, prod(cd, 0, ax), (int)cd[ax], prod(cd, ax + 1, ndim), 0, 1.0);
        if (rc) return rc;
    }
    return 0;
}

int corahip_randomfield_draw(corahip_ctx *ctx, const double *kweight, int64_t count, uint64_t seed, double *spec) {
    ARG_CHECK(ctx && kweight && spec && count >= 0);
    if (count == 0) return 0;
    StageTimer st(ctx, "flatdraw");
    long blocks = (count + 255) / 256;
    const long cap = (long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(randomfield_draw_kernel, dim3((int)blocks), dim3(256), 0, ctx->stream, kweight, (long)count,
                       seed, reinterpret_cast<double2 *>(spec));
    LAUNCH_CHECK();
    return 0;
}

int corahip_fg_mix(corahip_ctx *ctx, const double *freq_weight, const double *normals, const double *aff, int F,
                   int ncorr, int64_t M, double *out) {
    ARG_CHECK(ctx && freq_weight && normals && aff && out && F >= 1 && ncorr >= 1 && M >= 1);
    StageTimer st(ctx, "fg_mix");
    long bx = (M + 255) / 256;
    if (bx > 65535) bx = 65535;
    hipLaunchKernelGGL(fg_mix_kernel, dim3((unsigned)bx, (unsigned)F), dim3(256), 0, ctx->stream, freq_weight, normals,
                       reinterpret_cast<const double2 *>(aff), ncorr, (long)M, reinterpret_cast<double2 *>(out));
    LAUNCH_CHECK();
    return 0;
}

int corahip_spec_mul_real(corahip_ctx *ctx, double *spec, const double *weight, int64_t count) {
    ARG_CHECK(ctx && spec && weight && count >= 0);
    if (count == 0) return 0;
    StageTimer st(ctx, "spec_mul");
    long blocks = (count + 255) / 256;
    const long cap = (long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(spec_mul_real_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<double2 *>(spec), weight, (long)count);
    LAUNCH_CHECK();
    return 0;
}

int corahip_cube_affine(corahip_ctx *ctx, const double *df, const double *vf, const double *a, const double *b,
                        const double *c, int n0, int64_t plane, double *out) {
    ARG_CHECK(ctx && df && a && c && out && (vf == nullptr || b != nullptr) && n0 >= 1 && n0 <= 65535 && plane >= 1);
    StageTimer st(ctx, "cube_affine");
    long bx = (plane + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(cube_affine_kernel, dim3((unsigned)bx, (unsigned)n0), dim3(256), 0, ctx->stream, df, vf, a, b, c,
                       (long)plane, out);
    LAUNCH_CHECK();
    return 0;
}

int corahip_raytrace_slices(corahip_ctx *ctx, const double *cube, int n0, int n1, int n2, const double *zc,
                            const double *scale, const double *tx, const double *ty, double wx, double wy, int numz,
                            int numx, int numy, double *out) {
    ARG_CHECK(ctx && cube && zc && scale && tx && ty && out);
    ARG_CHECK(n0 >= 1 && n1 >= 1 && n2 >= 1 && numz >= 1 && numz <= 65535 && numx >= 1 && numy >= 1 && wx != 0.0 &&
              wy != 0.0);
    StageTimer st(ctx, "raytrace");
    long bx = ((long)numx * numy + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(raytrace_kernel, dim3((unsigned)bx, (unsigned)numz), dim3(256), 0, ctx->stream, cube, n0, n1, n2,
                       zc, scale, tx, ty, wx, wy, numx, numy, out);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
