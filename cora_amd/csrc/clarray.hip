// clarray.hip - C_l(nu,nu') integration with Romberg channel averaging.
//
// Replaces skysim.clarray (cora/core/skysim.py:10-69).  For the 21cm model the aps
// evaluation RedshiftCorrelation.angular_powerspectrum_fft (cora/signal/corr.py:944-982)
// and its three bilinearmap.interp calls (cora/util/bilinearmap.pyx:14-59) are fused with
// the Romberg reduction (skysim.py:62-67): no [l, F*zint, F*zint] intermediate exists.
//
// K1: for a fixed sub-sample pair (nu_i+a, nu_j+b) the table columns (y0, y0+1) are fixed, so the y
// interpolation and the dd/dv/vv combination collapse, once per sub-pair, into a 1-D profile along
// the k_perp axis x; every multipole then costs one linear interpolation of that profile in LDS.
// The tables are transposed (x contiguous) so the profile is read with coalesced loads: 12x fewer
// table reads than gathering 4 corners x 3 tables per (l, sub-pair).  One workgroup per channel pair
// j >= i, results to a [pair][l] scratch, a tiled transpose scatters them into [l][i][j] and its mirror.
#include "common.h"

#include <cstring>
#include <type_traits>

#ifndef CL_BUILD_BY_ROW
#define CL_BUILD_BY_ROW 1   // profile build with the row loop outside the profile loop (0: the profile-major loops of rounds 1-4, A/B)
#endif
#ifndef CL_BUILD_X4
#define CL_BUILD_X4 1       // dense rows of the row-major build two at a time with 16-byte loads (0: one row per thread, A/B)
#endif
#ifndef CL_ABLATE
#define CL_ABLATE 0  // diagnostic builds: 1 no table loads in the (profile-major) build, 2 interpolation for one multipole per thread only, 3 the row-major build without the loads of its dense part, 4 = 3 + 2
#endif
#define CL_MAXZ 17  // zint <= 17 (zromb <= 4)

#define CL_XS 512      // padded row length of the transposed tables (nkperp <= 511)
#define CL_LPT 9       // multipoles per thread (256 threads x 9 = 2304 >= 2049)

// Transposed, x-padded copy of the three tables: Tt[t][y][x], x contiguous.  For a fixed sub-sample
// pair the y columns (y0, y0+1) are fixed, so the whole l-dependence of the model is a 1-D profile
// along x; with x contiguous that profile is read with coalesced loads.
__global__ void cl_transpose_kernel(const double *__restrict__ dd, const double *__restrict__ dv,
                                    const double *__restrict__ vv, int nkperp, int nkpar, double *__restrict__ tt) {
    __shared__ double tile[32][33];
    const double *src = blockIdx.z == 0 ? dd : (blockIdx.z == 1 ? dv : vv);
    double *dst = tt + (size_t)blockIdx.z * nkpar * CL_XS;
    const int x0 = blockIdx.y * 32, y0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, y = y0 + tx;
        tile[r][tx] = (x < nkperp && y < nkpar) ? src[(size_t)x * nkpar + y] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int y = y0 + r, x = x0 + tx;
        if (y < nkpar && x < CL_XS) dst[(size_t)y * CL_XS + x] = (x < nkperp) ? tile[tx][r] : 0.0;
    }
}

// One workgroup = one channel pair (i, j >= i).  For every sub-sample pair (a, b):
//   profile  s_ab[x] = sum_T (cT0 Tt[T][y0][x] + cT1 Tt[T][y0+1][x])          (x = table row, in LDS)
//   value    C_l    += (1 - wx) s_ab[x0] + wx s_ab[x0+1],  x = (log10 l - log10(xc kperpmin)) xscale
// which is the bilinear lookup of the reference (bilinearmap.pyx:41-59) with the y interpolation and
// the three-table combination (corr.py:980-982) hoisted out of the l loop: 12x fewer table reads.
// Results go to a [pair][l] scratch (coalesced); cl_finish_kernel scatters them into [l][i][j].
// Pairs with i < 0 (padding of a multi-GPU pair shard) are skipped.
// ZINT > 0: the sub-sample count as a compile-time constant (9 = the default oversample of Sky3d): the zint
// interpolations of a multipole are then independent straight-line code whose LDS reads are all in flight together
// (with a run-time trip count every interpolation waited out its own LDS latency); ZINT = 0: any zint.
template <int ZINT>
__global__ void __launch_bounds__(256)
clarray21_kernel(const double *__restrict__ tt, int nkperp, int nkpar, double kperpmin, double xscale, double yscale,
                 const double *__restrict__ chi, const double *__restrict__ pfd, const double *__restrict__ fz,
                 const double *__restrict__ bz, int F, int zint_rt, const double *__restrict__ w,
                 const double *__restrict__ log10l, int nl, int l_base, const int2 *__restrict__ pairs,
                 double *__restrict__ scratch, int nl_total, int l_block) {
    const int zint = ZINT > 0 ? ZINT : zint_rt;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *prof = sm;                       // [zint][CL_XS + 2]
    double *lxcs_s = sm + zint * (CL_XS + 2);   // [zint]
    double *par = lxcs_s + zint;                // [zint^2][8]: c0..c5, lxc of every sub-sample pair
    int *ipar = reinterpret_cast<int *>(par + zint * zint * 8);   // [zint^2][4]: y0, first row, row count
    const int PS = CL_XS + 2;
    const int tid = threadIdx.x;
    const int i = pairs[blockIdx.x].x, j = pairs[blockIdx.x].y;
    if (i < 0) return;
    const double ux = (double)nkperp - 1e-5, uy = (double)nkpar - 1e-5;
    const size_t tsz = (size_t)nkpar * CL_XS;

    double lxs[CL_LPT], acc[CL_LPT];
#pragma unroll
    for (int k = 0; k < CL_LPT; k++) {
        const int li = l_base + tid + 256 * k;
        // (entries past the range repeat its last multipole: every lane of the clamp-free path stays inside the profile)
        lxs[k] = log10l[min(li, nl_total - 1)] * xscale;
        acc[k] = 0.0;
    }
    // rows the l range can touch: x(l_first) <= x <= x(l_last) + 1 (x is monotone in l; log10l ascending)
    const int l_end = min(nl_total, l_base + 256 * CL_LPT);
    const double lx_lo = log10l[l_base] * xscale, lx_hi = log10l[l_end - 1] * xscale;
    const int kmax = (l_end - l_base + 255) / 256;  // multipoles per thread actually present

    // ---- parameters of the zint^2 sub-sample pairs, computed ONCE per workgroup (one thread each) instead of by
    //      every thread in front of every profile: the log10 / divisions were the largest part of the kernel
    int noclamp = 1;
    for (int t = tid; t < zint * zint; t += 256) {
        const int a = t / zint, b = t - a * zint;
        const int za = i * zint + a, zb = j * zint + b;
        const double x1 = chi[za], x2 = chi[zb];
        const double xc = 0.5 * (x1 + x2);
        const double lxc = log10(xc * kperpmin) * xscale;
        double yy = fabs(x2 - x1) * yscale;  // rpar / (pi / kparmax)
        yy = yy < 0.0 ? 0.0 : (yy > uy ? uy : yy);
        int y0 = (int)yy;
        double wy = yy - (double)y0;
        if (y0 + 1 > nkpar - 1) {  // stay in bounds where the reference reads past the table edge
            y0 = nkpar - 2;
            wy = 1.0;
        }
        const double W = w[a] * w[b] * pfd[za] * pfd[zb] / (xc * xc * M_PI);
        const double cdd = W * bz[za] * bz[zb];
        const double cdv = W * (fz[za] * bz[zb] + fz[zb] * bz[za]);
        const double cvv = W * fz[za] * fz[zb];
        double *pp = par + t * 8;
        pp[0] = cdd * (1.0 - wy);
        pp[1] = cdd * wy;
        pp[2] = cdv * (1.0 - wy);
        pp[3] = cdv * wy;
        pp[4] = cvv * (1.0 - wy);
        pp[5] = cvv * wy;
        pp[6] = lxc;
        double xhi = lx_hi - lxc, xlo = lx_lo - lxc;
        xhi = xhi < 0.0 ? 0.0 : (xhi > ux ? ux : xhi);
        xlo = xlo < 0.0 ? 0.0 : (xlo > ux ? ux : xlo);
        ipar[t * 4 + 0] = y0;
        ipar[t * 4 + 1] = (int)xlo;                       // first row needed
        ipar[t * 4 + 2] = min((int)xhi + 2, nkperp);      // rows x0 .. nx-1 are needed
        // does any multipole of this launch other than its FIRST entry (which may be the l = 0 sentinel, 1e-10) need the
        // clamps of bilinearmap.interp?  (x is monotone in l: the second and the last entry bound the rest)
        const double x2nd = (l_base + 1 < l_end ? log10l[l_base + 1] * xscale : lx_hi) - lxc;
        if (!(x2nd >= 0.0 && lx_hi - lxc <= ux)) noclamp = 0;
    }
    // uniform over the workgroup: every sub-sample pair can skip the clamps (true for every configuration cora's
    // frequency ranges produce: 0 <= x <= ~390 of 500 rows for l >= 1)
    const bool all_fast = __syncthreads_and(noclamp) != 0;
    // The first multipoles of a range that starts at low l are many table rows apart (x = log10 l * xscale: l = 1, 2, 3 sit
    // 35 and 21 rows from each other; from l ~ 25 on consecutive l are less than two rows apart): entries 0 .. nsp - 1 get
    // their two rows built individually, the dense row range starts at entry nsp - a third of the table reads of a
    // cfg-3 profile were rows between those first multipoles that no interpolation ever looks at.
    __shared__ int s_nsp;
    __shared__ double s_lxk[32];
    if (tid < 32) s_lxk[tid] = log10l[min(l_base + tid, nl_total - 1)] * xscale;
    if (tid < 64) {
        bool wide = false;
        if (l_base + tid + 1 < l_end) wide = (log10l[l_base + tid + 1] - log10l[l_base + tid]) * xscale >= 2.0;
        const unsigned long long m = __ballot(wide);
        if (tid == 0) s_nsp = m == ~0ull ? 32 : min(32, (int)__builtin_ctzll(~m));
    }

    for (int a = 0; a < zint; a++) {
        __syncthreads();
        // ---- profiles of the zint sub-sample pairs (a, b = 0..zint-1)
#if (CL_ABLATE == 0 || CL_ABLATE >= 3) && CL_BUILD_BY_ROW
        if (ZINT > 0 && all_fast) {
            // Row-major build (round 5): a thread takes table row x for ALL ZINT profiles - over the union of their row
            // ranges; a row outside a profile's own range is never read by the interpolation - so that its 6 ZINT table
            // loads are independent straight-line code in flight together.  With the profile-major loops every (b, x)
            // step waited out its own L2 latency: 18 exposed latencies per a, against 2 here.
            constexpr int ZN = ZINT > 0 ? ZINT : 1;
            const int nsp = min(s_nsp, l_end - l_base);
            const double lxd = log10l[min(l_base + nsp, nl_total - 1)] * xscale;     // first entry of the dense part
            int xlo = nkperp, xhi = 0;
#pragma unroll
            for (int b = 0; b < ZN; b++) {
                const double xx = fmin(fmax(lxd - par[(a * ZN + b) * 8 + 6], 0.0), ux);
                xlo = min(xlo, (int)xx);
                xhi = max(xhi, ipar[(a * ZN + b) * 4 + 2]);
            }
            if (nsp >= l_end - l_base) xlo = xhi;                                    // (every entry is built individually)
            auto row_of = [&](int b, int x, double (&v)[6]) {
                const double *r0 = tt + (size_t)ipar[(a * ZN + b) * 4 + 0] * CL_XS + x, *r1 = r0 + CL_XS;
                v[0] = r0[0];
                v[1] = r1[0];
                v[2] = r0[tsz];
                v[3] = r1[tsz];
                v[4] = r0[2 * tsz];
                v[5] = r1[2 * tsz];
            };
            auto combine = [&](int b, const double (&v)[6]) {
                const double *pp = par + (a * ZN + b) * 8;
                return pp[0] * v[0] + pp[1] * v[1] + pp[2] * v[2] + pp[3] * v[3] + pp[4] * v[4] + pp[5] * v[5];
            };
#if CL_BUILD_X4
            {
                // two rows per thread and 16-byte loads: threads 0 .. 127 build the first (ZN + 1) / 2 profiles, threads
                // 128 .. 255 the others (the table columns past nkperp are zero padding, the profile has room for them)
                constexpr int ZH = (ZN + 1) / 2;
                const int hb = tid >> 7, b0 = hb ? ZH : 0, nb = hb ? ZN - ZH : ZH;
                constexpr int QB = 3;                          // profiles whose loads are in flight together (18 x 16 bytes per thread)
                for (int x = (xlo & ~1) + 2 * (tid & 127); x < xhi; x += 256) {
#pragma unroll
                    for (int q0 = 0; q0 < ZH; q0 += QB) {
                        double2 v[QB][6];
#pragma unroll
                        for (int q = 0; q < QB; q++) {
#if CL_ABLATE >= 3   // diagnostic: the row-major build without its table loads
                            if (q0 + q < nb) {
                                for (int i6 = 0; i6 < 6; i6++) v[q][i6] = make_double2((double)x, 1.0 + i6);
                            } else
#endif
                            if (q0 + q < nb) {
                                const double *r0 = tt + (size_t)ipar[(a * ZN + b0 + q0 + q) * 4 + 0] * CL_XS + x, *r1 = r0 + CL_XS;
                                v[q][0] = *reinterpret_cast<const double2 *>(r0);
                                v[q][1] = *reinterpret_cast<const double2 *>(r1);
                                v[q][2] = *reinterpret_cast<const double2 *>(r0 + tsz);
                                v[q][3] = *reinterpret_cast<const double2 *>(r1 + tsz);
                                v[q][4] = *reinterpret_cast<const double2 *>(r0 + 2 * tsz);
                                v[q][5] = *reinterpret_cast<const double2 *>(r1 + 2 * tsz);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < QB; q++) {
                            if (q0 + q < nb) {
                                const double *pp = par + (a * ZN + b0 + q0 + q) * 8;
                                double2 o;
                                o.x = pp[0] * v[q][0].x + pp[1] * v[q][1].x + pp[2] * v[q][2].x + pp[3] * v[q][3].x + pp[4] * v[q][4].x + pp[5] * v[q][5].x;
                                o.y = pp[0] * v[q][0].y + pp[1] * v[q][1].y + pp[2] * v[q][2].y + pp[3] * v[q][3].y + pp[4] * v[q][4].y + pp[5] * v[q][5].y;
                                *reinterpret_cast<double2 *>(prof + (b0 + q0 + q) * PS + x) = o;
                            }
                        }
                    }
                }
            }
#else
            for (int x = xlo + tid; x < xhi; x += 256) {
                double v[ZN][6];
#pragma unroll
                for (int b = 0; b < ZN; b++) row_of(b, x, v[b]);
#pragma unroll
                for (int b = 0; b < ZN; b++) prof[b * PS + x] = combine(b, v[b]);
            }
#endif
            // the two rows of every early entry, per profile (slot nkperp, if an entry reaches it, repeats row nkperp - 1)
            // (one thread per (entry, profile): its twelve loads are in flight together; 25 x 9 = 225 of them at cfg 3)
            for (int e = tid; e < nsp * ZN; e += 256) {
                const int b = e % ZN, k = e / ZN;
                const double xx = fmin(fmax(s_lxk[k] - par[(a * ZN + b) * 8 + 6], 0.0), ux);
                const int x = (int)xx;
                // rows x and x + 1 by one 16-byte load per table row (8-byte aligned: x has either parity)
                typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
                const double *r0 = tt + (size_t)ipar[(a * ZN + b) * 4 + 0] * CL_XS + x, *r1 = r0 + CL_XS;
                const d2u t0 = *reinterpret_cast<const d2u *>(r0), t1 = *reinterpret_cast<const d2u *>(r1);
                const d2u t2 = *reinterpret_cast<const d2u *>(r0 + tsz), t3 = *reinterpret_cast<const d2u *>(r1 + tsz);
                const d2u t4 = *reinterpret_cast<const d2u *>(r0 + 2 * tsz), t5 = *reinterpret_cast<const d2u *>(r1 + 2 * tsz);
                const double v0[6] = {t0.x, t1.x, t2.x, t3.x, t4.x, t5.x};
                const double v1[6] = {t0.y, t1.y, t2.y, t3.y, t4.y, t5.y};
                const double p0 = combine(b, v0);
                prof[b * PS + x] = p0;
                prof[b * PS + x + 1] = x + 1 < nkperp ? combine(b, v1) : p0;      // (slot nkperp repeats row nkperp - 1)
            }
            if (tid < ZN) lxcs_s[tid] = par[(a * ZN + tid) * 8 + 6];
            // slot nkperp repeats the last row (see below); xhi is uniform, and cora's frequency ranges never reach the
            // last table row - no barrier for a write that does not happen
            if (xhi == nkperp) {
                __syncthreads();
                if (tid < ZN) prof[tid * PS + nkperp] = prof[tid * PS + nkperp - 1];
            }
        } else
#endif
        for (int b = 0; b < zint; b++) {
            const int t = a * zint + b;
            const double *pp = par + t * 8;
            const double c0 = pp[0], c1 = pp[1], c2 = pp[2], c3 = pp[3], c4 = pp[4], c5 = pp[5];
            const int y0 = ipar[t * 4 + 0], x0r = ipar[t * 4 + 1], nx = ipar[t * 4 + 2];
            const double *r0 = tt + (size_t)y0 * CL_XS, *r1 = r0 + CL_XS;
            double *pb = prof + b * PS;
#if CL_ABLATE == 1   // diagnostic: no table loads
            for (int x = x0r + tid; x < nx; x += 256) pb[x] = c0 + c1 * x;
            (void)r0; (void)r1; (void)c2; (void)c3; (void)c4; (void)c5;
#else
            for (int x = x0r + tid; x < nx; x += 256)
                pb[x] = c0 * r0[x] + c1 * r1[x] + c2 * r0[tsz + x] + c3 * r1[tsz + x] + c4 * r0[2 * tsz + x] +
                        c5 * r1[2 * tsz + x];
#endif
            if (tid == 0) lxcs_s[b] = pp[6];
            // slot nkperp repeats the last row: the interpolation reads (x0, x0 + 1) unclamped, which is then the
            // clamped x1 = min(x0 + 1, nkperp - 1) of the reference's intent (the thread that wrote row nkperp - 1, if the
            // profile reaches it, writes it; otherwise the slot is never read: x0 + 1 <= nx - 1 < nkperp)
            if (nx == nkperp && tid == ((nkperp - 1 - x0r) & 255)) pb[nkperp] = pb[nkperp - 1];
        }
        __syncthreads();
        // ---- 1-D interpolation for this thread's multipoles
        double lxc_r[ZINT > 0 ? ZINT : 1];
        if constexpr (ZINT > 0) {
#pragma unroll
            for (int b = 0; b < ZINT; b++) lxc_r[b] = lxcs_s[b];
        }
        // (instruction count matters here: 5.5e9 of these per cfg-3 launch bound the kernel.  fract instead of
        //  int -> double -> subtract, one ds_read2_b64 for the unclamped row pair: 9 VALU + 1 LDS instead of 14 + 2)
        // (the clamps are two of the nine VALU instructions of an interpolation: when no multipole of the launch needs
        //  them - all_fast - only the thread that holds the launch's first entry, the possible l = 0, keeps them)
        auto interp = [&](int b, double lx, double lxc, bool clamp) {
            double xx = lx - lxc;
            if (clamp) xx = fmin(fmax(xx, 0.0), ux);
            const int x0 = (int)xx;
            const double wx = __builtin_amdgcn_fract(xx);
            const double *pr = prof + b * PS + x0;
            const double s0 = pr[0], s1 = pr[1];
            return fma(wx, s1 - s0, s0);
        };
        auto interp_all = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
            for (int k = 0; k < CL_LPT; k++) {
#if CL_ABLATE == 2 || CL_ABLATE == 4   // diagnostic: no interpolation phase
                if (k >= 1) continue;
#endif
                // (uniform: l-sharded callers pass short l ranges; the LAST slot of a range is usually almost empty - 2049 =
                //  8 x 256 + 1: one lane of the workgroup has a ninth multipole - and the waves without one skip it)
                if (k < kmax && (k < kmax - 1 || l_base + tid + 256 * k < nl_total)) {
                    // (FAST: only the first entry of the launch - thread 0, k = 0 - is clamped)
                    const bool clamp = !FAST || (k == 0 && tid == 0);
                    double s = 0.0;
                    if constexpr (ZINT > 0) {
#pragma unroll
                        for (int b = 0; b < ZINT; b++) s += interp(b, lxs[k], lxc_r[b], clamp);
                    } else {
                        for (int b = 0; b < zint; b++) s += interp(b, lxs[k], lxcs_s[b], clamp);
                    }
                    acc[k] += s;
                }
            }
        };
        if (all_fast) interp_all(std::true_type{});
        else interp_all(std::false_type{});
    }
#pragma unroll
    for (int k = 0; k < CL_LPT; k++) {
        const int li = l_base + tid + 256 * k;
        // results: [l / l_block][pair][l % l_block] (l_block = nl: plain [pair][l]; l_block = the l-shard
        // length of a multi-GPU run: one contiguous slab per destination rank of the all-to-all)
        if (li < nl_total)
            scratch[((size_t)(li / l_block) * gridDim.x + blockIdx.x) * l_block + li % l_block] = acc[k];
    }
}

// scratch [slot][l] -> out [l][i][j] and its mirror [l][j][i].  Slot s holds canonical pair
// p = (s % npl) * W + s / npl of the (i, j >= i) enumeration: W = 1, npl = npairs for a single GPU;
// after the all-to-all of a W-rank run slab r = s / npl came from rank r, which integrated pairs r, r + W, ...
// The canonical enumeration of the channel pairs (i, j >= i) runs in BANDS of CL_BAND diagonals: band B holds the
// separations d = j - i in [CL_BAND B, CL_BAND B + CL_BAND), enumerated i-major (for i: for d).  Pairs of nearby
// diagonals have nearly the same radial separation, i.e. read the same few k_par columns of the tables: the ~800
// workgroups in flight then share ~1 MB of table rows in L2 (with the row-major order they spanned every separation at
// once and the rows came from HBM again and again: 43 GB of FETCH_SIZE per launch for 0.4 GB of tables, 28 GB with
// this order, profiles/r02_pmc.json); and consecutive slots are consecutive j of one row i, so the transpose into
// [l][i][j] writes contiguous segments (a plain diagonal-major order made every write an isolated 8 bytes).
#define CL_BAND 32
#ifndef CL_XCD_ORDER
#define CL_XCD_ORDER 1   // 0: plain i-major order inside a band (A/B)
#endif
__host__ __device__ static inline long cl_band_count(int n) {   // pairs of a band that has n rows (n = F - CL_BAND B)
    return n >= CL_BAND ? (long)CL_BAND * (n - (CL_BAND - 1)) + (long)(CL_BAND - 1) * CL_BAND / 2 : (long)n * (n + 1) / 2;
}
__host__ __device__ static inline int2 pair_of_index(long p, int F) {
    int B = 0;
    for (;; B++) {
        const long c = cl_band_count(F - CL_BAND * B);
        if (p < c) break;
        p -= c;
    }
    const int n = F - CL_BAND * B;
    const int nfull = n >= CL_BAND ? n - (CL_BAND - 1) : 0;         // rows with all CL_BAND separations
    int i, dd;
    if (p < (long)nfull * CL_BAND) {
#if CL_XCD_ORDER
        // Round 6: inside a band's full rows index p = 8 s + x is pair (row s / 4, separation 4 x + s % 4).  Workgroups are
        // dealt round-robin over the 8 XCDs, each with its own L2: with the plain i-major order (p = 32 i + d) XCD x got the
        // separations x, x + 8, x + 16, x + 24 of every row - four groups of k_par columns 8 channels (430 columns) apart -
        // now it gets the four ADJACENT separations 4 x .. 4 x + 3 (one group of columns).  Every band starts at a
        // multiple of 8, so p mod 8 is the workgroup's XCD slot; 32 consecutive p still are the 32 separations of one row
        // (cl_finish_kernel's contiguous segment), and rank r of an 8-rank pair shard (pairs r, r + 8, ...) integrates four
        // adjacent separations of every band instead of four spread ones.
        const int xq = (int)(p & 7), sq = (int)(p >> 3);
        i = sq >> 2;
        dd = 4 * xq + (sq & 3);
#else
        i = (int)(p / CL_BAND);
        dd = (int)(p % CL_BAND);
#endif
    } else {
        p -= (long)nfull * CL_BAND;
        int e = n >= CL_BAND ? CL_BAND - 1 : n;                     // entries of the first tail row
        i = nfull;
        while (p >= e) {
            p -= e;
            e--;
            i++;
        }
        dd = (int)p;
    }
    return make_int2(i, i + CL_BAND * B + dd);
}

__global__ void cl_finish_kernel(const double *__restrict__ scratch, long npairs, int W, long npl, int lstride, int nl,
                                 int F, double *__restrict__ out) {
    __shared__ double tile[32][33];
    // tile: 32 slots x 32 l
    const long s0 = (long)blockIdx.x * 32;
    const int l0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long nslot = npl * W;
    for (int r = ty; r < 32; r += 8) {
        const long sl = s0 + r;
        const int l = l0 + tx;
        tile[r][tx] = (sl < nslot && l < nl) ? scratch[(size_t)sl * lstride + l] : 0.0;
    }
    __syncthreads();
    const long sl = s0 + tx;
    const long p = sl < nslot ? (sl % npl) * W + sl / npl : npairs;
    if (p >= npairs) return;
    const int2 ij = pair_of_index(p, F);
    for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r;
        if (l < nl) {
            out[((size_t)l * F + ij.x) * F + ij.y] = tile[tx][r];   // upper triangle (j >= i): consecutive slots, consecutive j
        }
    }
}

// out[l][j][i] = out[l][i][j] for j > i, in 32 x 32 tiles through LDS: both the read of the upper and the write of the
// lower triangle are 256-byte rows (cl_finish_kernel used to write the mirror element by element, F doubles apart:
// 2.9 GB of WRITE_SIZE for 1.07 GB of C_l, profiles/r02_pmc.json).  grid = (upper tiles, l)
__global__ void cl_mirror_kernel(double *__restrict__ out, int F) {
    __shared__ double tile[32][33];
    const int nt = (F + 31) / 32;
    // upper-triangular tile index -> (ti, tj >= ti)
    int ti = 0, rem = blockIdx.x;
    while (rem >= nt - ti) {
        rem -= nt - ti;
        ti++;
    }
    const int tj = ti + rem;
    double *o = out + (size_t)blockIdx.y * F * F;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = ti * 32 + r, j = tj * 32 + tx;
        tile[r][tx] = (i < F && j < F) ? o[(size_t)i * F + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = tj * 32 + r, i = ti * 32 + tx;       // element (j, i) of the lower triangle
        if (j < F && i < F && j > i) o[(size_t)j * F + i] = tile[tx][r];
    }
}
static int launch_mirror(corahip_ctx *ctx, double *out, int nl, int F) {
    const int nt = (F + 31) / 32;
    dim3 grid((unsigned)(nt * (nt + 1) / 2), (unsigned)nl);
    cl_mirror_kernel<<<grid, 256, 0, ctx->stream>>>(out, F);
    LAUNCH_CHECK();
    return 0;
}

// separable model: Bavg[i][j] = sum_ab w_a w_b bcov[i zint + a][j zint + b]; out[l][i][j] = al[l] Bavg[i][j]
__global__ void separable_kernel(const double *__restrict__ al, int nl, const double *__restrict__ bcov, int F, int zint,
                                 const double *__restrict__ w, double *__restrict__ out) {
    const long n = (long)nl * F * F;
    const int nz = F * zint;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int j = (int)(q % F), i = (int)((q / F) % F);
        const long l = q / ((long)F * F);
        double s = 0.0;
        for (int a = 0; a < zint; a++) {
            double t = 0.0;
            for (int b = 0; b < zint; b++) t += w[b] * bcov[(size_t)(i * zint + a) * nz + j * zint + b];
            s += w[a] * t;
        }
        out[q] = al[l] * s;
    }
}

// Romberg reduction of host-evaluated samples clt[l][i][a][j][b]
__global__ void romb_reduce_kernel(const double *__restrict__ clt, int nl, int F, int zint,
                                   const double *__restrict__ w, double *__restrict__ out) {
    const long n = (long)nl * F * F;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int j = (int)(q % F), i = (int)((q / F) % F);
        const long l = q / ((long)F * F);
        const double *base = clt + (((size_t)l * F + i) * zint * F + j) * zint;
        double s = 0.0;
        for (int a = 0; a < zint; a++) {
            double t = 0.0;
            for (int b = 0; b < zint; b++) t += w[b] * base[(size_t)a * F * zint + b];
            s += w[a] * t;
        }
        out[q] = s;
    }
}

// elementwise evaluation of the table model at n independent points (the bare `aps`
// callable, corr.py:953-982): lx = log10(l) (l = 0 -> 1e-10 on the host), chi1/chi2,
// coefficient triples c_dd, c_dv, c_vv WITHOUT the 1/(xc^2 pi) factor.
__global__ void aps21_points_kernel(const double *__restrict__ dd, const double *__restrict__ dv,
                                    const double *__restrict__ vv, int nkperp, int nkpar, double kperpmin,
                                    double xscale, double yscale, long n, const double *__restrict__ lx,
                                    const double *__restrict__ chi1, const double *__restrict__ chi2,
                                    const double *__restrict__ cdd, const double *__restrict__ cdv,
                                    const double *__restrict__ cvv, double *__restrict__ out) {
    const double ux = (double)nkperp - 1e-5, uy = (double)nkpar - 1e-5;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const double x1 = chi1[q], x2 = chi2[q];
        const double xc = 0.5 * (x1 + x2);
        double xx = (lx[q] - log10(xc * kperpmin)) * xscale;
        double yy = fabs(x2 - x1) * yscale;
        xx = xx < 0.0 ? 0.0 : (xx > ux ? ux : xx);
        yy = yy < 0.0 ? 0.0 : (yy > uy ? uy : yy);
        const int x0 = (int)xx, y0 = (int)yy;
        const double wx = xx - (double)x0, wy = yy - (double)y0;
        const int xb = min(x0 + 1, nkperp - 1), yb = min(y0 + 1, nkpar - 1);
        const size_t o00 = (size_t)x0 * nkpar + y0, o01 = (size_t)x0 * nkpar + yb;
        const size_t o10 = (size_t)xb * nkpar + y0, o11 = (size_t)xb * nkpar + yb;
        const double wa = (1.0 - wx) * (1.0 - wy), wb = (1.0 - wx) * wy, wc = wx * (1.0 - wy), wd = wx * wy;
        const double vdd = wa * dd[o00] + wb * dd[o01] + wc * dd[o10] + wd * dd[o11];
        const double vdv = wa * dv[o00] + wb * dv[o01] + wc * dv[o10] + wd * dv[o11];
        const double vvv = wa * vv[o00] + wb * vv[o01] + wc * vv[o10] + wd * vv[o11];
        out[q] = (cdd[q] * vdd + cdv[q] * vdv + cvv[q] * vvv) / (xc * xc * M_PI);
    }
}

extern "C" {

// pairs pair_first, pair_first + pair_step, ... of the canonical (i, j >= i) enumeration -> out_pairs in the
// [l / l_block][k][l % l_block] layout (npl = ceil((npairs - pair_first) / pair_step) slots, padded to npl_pad)
static int clarray21_pairs(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv, int nkperp,
                           int nkpar, double kperpmin, double kperpmax, double kparmax, const double *chi,
                           const double *pfd, const double *f, const double *b, int F, int zint, const double *w,
                           const double *log10l, int nl, int pair_first, int pair_step, long npl_pad, int l_block,
                           double *out_pairs) {
    const size_t tt_bytes = sizeof(double) * 3 * (size_t)nkpar * CL_XS;
    double *tt = nullptr;
    int2 *dpairs = nullptr;
    int rc;
    if ((rc = corahip_ctx_scratch(ctx, 0, tt_bytes, (void **)&tt))) return rc;
    if ((rc = corahip_ctx_scratch(ctx, 2, sizeof(int2) * npl_pad, (void **)&dpairs))) return rc;
    const long key[4] = {F, pair_first, pair_step, npl_pad};
    if (ctx->pairs_ptr != (void *)dpairs || memcmp(key, ctx->pairs_key, sizeof(key)) != 0) {
        // (re)build the pair list of this shard; it stays resident for the following calls
        std::vector<int2> pairs((size_t)npl_pad, make_int2(-1, -1));
        long p = 0, k = 0;
        const long npairs_all = (long)F * (F + 1) / 2;
        for (; p < npairs_all; p++)           // canonical order: bands of diagonals (pair_of_index)
            if (p >= pair_first && (p - pair_first) % pair_step == 0) pairs[(size_t)k++] = pair_of_index(p, F);
        HIP_TRY(hipMemcpyAsync(dpairs, pairs.data(), sizeof(int2) * npl_pad, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // `pairs` (host) must outlive the async copy
        memcpy(ctx->pairs_key, key, sizeof(key));
        ctx->pairs_ptr = (void *)dpairs;
    }
    // the x-contiguous copy of the tables: made on every call unless the caller has pinned exactly these tables
    // (corahip_clarray_tables_pin) and the copy of that generation is still the one in the scratch slot
    const bool pinned = ctx->tt_pinned && ctx->tt_pin[0] == dd && ctx->tt_pin[1] == dv && ctx->tt_pin[2] == vv;
    // (the kept copy was written by work queued on tt_stream: a call on another stream has no ordering with it and
    //  makes its own)
    if (!(pinned && ctx->tt_valid && ctx->tt_stream == ctx->stream && ctx->scratch_bytes[0] == tt_bytes)) {
        dim3 grid((nkpar + 31) / 32, CL_XS / 32, 3);
        cl_transpose_kernel<<<grid, 256, 0, ctx->stream>>>(dd, dv, vv, nkperp, nkpar, tt);
        LAUNCH_CHECK();
        ctx->tt_valid = pinned;
        ctx->tt_stream = ctx->stream;
    }
    const double xscale = (double)(nkperp - 1) / log10(kperpmax / kperpmin);
    const double yscale = kparmax / M_PI;
    const size_t shm = sizeof(double) * ((size_t)zint * (CL_XS + 2) + zint + (size_t)zint * zint * 10);
    auto kern = zint == 9 ? clarray21_kernel<9> : (zint == 5 ? clarray21_kernel<5> : (zint == 3 ? clarray21_kernel<3> : clarray21_kernel<0>));
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int l_base = 0; l_base < nl; l_base += 256 * CL_LPT) {
        kern<<<(unsigned)npl_pad, 256, shm, ctx->stream>>>(tt, nkperp, nkpar, kperpmin, xscale, yscale, chi, pfd, f, b, F, zint, w,
                                                          log10l, nl, l_base, dpairs, out_pairs, nl, l_block);
        LAUNCH_CHECK();
    }
    return 0;
}

int corahip_clarray_tables_pin(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv, uint64_t generation) {
    ARG_CHECK(ctx != nullptr);
    if (dd == nullptr && generation != 0 && !(ctx->tt_pinned && ctx->tt_pin_gen == generation))
        return 0;                  // an owner withdrawing a pin that another generation has replaced: nothing of its own left
    const bool same = ctx->tt_pinned && dd && ctx->tt_pin[0] == dd && ctx->tt_pin[1] == dv && ctx->tt_pin[2] == vv &&
                      ctx->tt_pin_gen == generation;
    if (same) return 0;
    ctx->tt_pin[0] = dd;
    ctx->tt_pin[1] = dv;
    ctx->tt_pin[2] = vv;
    ctx->tt_pin_gen = generation;
    ctx->tt_pinned = dd != nullptr;
    ctx->tt_valid = false;
    return 0;
}

int corahip_clarray_table21cm(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv, int nkperp,
                              int nkpar, double kperpmin, double kperpmax, double kparmax, const double *chi,
                              const double *pfd, const double *f, const double *b, int F, int zint, const double *w,
                              const double *log10l, int nl, double *out) {
    ARG_CHECK(ctx != nullptr && dd && dv && vv && chi && pfd && f && b && w && log10l && out);
    ARG_CHECK(nkperp >= 2 && nkperp < CL_XS && nkpar >= 2 && F >= 1 && zint >= 1 && zint <= CL_MAXZ && nl >= 1);
    ARG_CHECK(kperpmin > 0 && kperpmax > kperpmin && kparmax > 0);
    StageTimer t(ctx, "clarray");
    const long npairs = (long)F * (F + 1) / 2;
    double *scratch = nullptr;
    int rc;
    if ((rc = corahip_ctx_scratch(ctx, 1, sizeof(double) * (size_t)npairs * nl, (void **)&scratch))) return rc;
    if ((rc = clarray21_pairs(ctx, dd, dv, vv, nkperp, nkpar, kperpmin, kperpmax, kparmax, chi, pfd, f, b, F, zint, w,
                              log10l, nl, 0, 1, npairs, nl, scratch)))
        return rc;
    dim3 grid((unsigned)((npairs + 31) / 32), (nl + 31) / 32);
    cl_finish_kernel<<<grid, 256, 0, ctx->stream>>>(scratch, npairs, 1, npairs, nl, nl, F, out);
    LAUNCH_CHECK();
    return launch_mirror(ctx, out, nl, F);
}

int corahip_clarray_table21cm_pairs(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv, int nkperp,
                                    int nkpar, double kperpmin, double kperpmax, double kparmax, const double *chi,
                                    const double *pfd, const double *f, const double *b, int F, int zint,
                                    const double *w, const double *log10l, int nl, int pair_first, int pair_step,
                                    int l_block, double *out_pairs) {
    ARG_CHECK(ctx != nullptr && dd && dv && vv && chi && pfd && f && b && w && log10l && out_pairs);
    ARG_CHECK(nkperp >= 2 && nkperp < CL_XS && nkpar >= 2 && F >= 1 && zint >= 1 && zint <= CL_MAXZ && nl >= 1);
    ARG_CHECK(kperpmin > 0 && kperpmax > kperpmin && kparmax > 0);
    ARG_CHECK(pair_step >= 1 && pair_first >= 0 && pair_first < pair_step && l_block >= 1);
    StageTimer t(ctx, "clarray");
    const long npairs = (long)F * (F + 1) / 2;
    const long npl = (npairs + pair_step - 1) / pair_step;  // slots per rank, equal on all ranks (padded)
    return clarray21_pairs(ctx, dd, dv, vv, nkperp, nkpar, kperpmin, kperpmax, kparmax, chi, pfd, f, b, F, zint, w,
                           log10l, nl, pair_first, pair_step, npl, l_block, out_pairs);
}

int corahip_clarray_pairs_finish(corahip_ctx *ctx, const double *pairs_in, int F, int nranks, int l_stride, int nl,
                                 double *out) {
    ARG_CHECK(ctx != nullptr && pairs_in && out && F >= 1 && nranks >= 1 && nl >= 1 && l_stride >= nl);
    StageTimer t(ctx, "clarray");
    const long npairs = (long)F * (F + 1) / 2;
    const long npl = (npairs + nranks - 1) / nranks;
    dim3 grid((unsigned)((npl * nranks + 31) / 32), (nl + 31) / 32);
    cl_finish_kernel<<<grid, 256, 0, ctx->stream>>>(pairs_in, npairs, nranks, npl, l_stride, nl, F, out);
    LAUNCH_CHECK();
    return launch_mirror(ctx, out, nl, F);
}

int corahip_clarray_separable(corahip_ctx *ctx, const double *al, int nl, const double *bcov, int F, int zint,
                              const double *w, double *out) {
    ARG_CHECK(ctx != nullptr && al && bcov && w && out && nl >= 1 && F >= 1 && zint >= 1);
    StageTimer t(ctx, "clarray");
    const long n = (long)nl * F * F;
    separable_kernel<<<(int)std::min<long>((n + 255) / 256, 4096), 256, 0, ctx->stream>>>(al, nl, bcov, F, zint, w, out);
    LAUNCH_CHECK();
    return 0;
}

int corahip_romb_reduce(corahip_ctx *ctx, const double *clt, int nl, int F, int zint, const double *w, double *out) {
    ARG_CHECK(ctx != nullptr && clt && w && out && nl >= 1 && F >= 1 && zint >= 1);
    StageTimer t(ctx, "clarray");
    const long n = (long)nl * F * F;
    romb_reduce_kernel<<<(int)std::min<long>((n + 255) / 256, 4096), 256, 0, ctx->stream>>>(clt, nl, F, zint, w, out);
    LAUNCH_CHECK();
    return 0;
}

int corahip_aps_table21cm_points(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv, int nkperp,
                                 int nkpar, double kperpmin, double kperpmax, double kparmax, long n,
                                 const double *lx, const double *chi1, const double *chi2, const double *cdd,
                                 const double *cdv, const double *cvv, double *out) {
    ARG_CHECK(ctx != nullptr && dd && dv && vv && lx && chi1 && chi2 && cdd && cdv && cvv && out);
    ARG_CHECK(nkperp >= 2 && nkpar >= 2 && n >= 1 && kperpmin > 0 && kperpmax > kperpmin && kparmax > 0);
    StageTimer t(ctx, "clarray");
    const double xscale = (double)(nkperp - 1) / log10(kperpmax / kperpmin);
    const double yscale = kparmax / M_PI;
    aps21_points_kernel<<<(int)std::min<long>((n + 255) / 256, 4096), 256, 0, ctx->stream>>>(
        dd, dv, vv, nkperp, nkpar, kperpmin, xscale, yscale, n, lx, chi1, chi2, cdd, cdv, cvv, out);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
