// sht_plan.hip - the (nside, lmax) plan of the HEALPix transforms: ring geometry, recurrence coefficients,
// polar seed table (first contributing l and the two recurrence values there), per-(m, ring block) first-l table,
// per-ring m cut-off, twiddles, Bluestein chirps / filters and the K5 launch classes.  See sht_internal.h.
#include "sht_internal.h"

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

// Terms of the Legendre sums with |lambda_lm| < 2^SEED_MIN_EXP are dropped (K4 starts its recurrence at the first l
// that reaches it; K4 neither writes nor K5 reads the F_m cells beyond the per-ring cut-off it implies).  Rounds 2-3: 2^-80 =
// 8.3e-25: the sum of all 2e6 dropped terms of a channel stays below 2e-18 of an O(1) coefficient, two decades
// under fp64 rounding.  (SHT engines truncate in the same spirit - libsharp, as the builder recalls it, by an m limit per
// ring; its source is not in this image and no figure for it is claimed here.)  The exponent is a PLAN parameter
// (corahip_sht_plan_create_ex, default SEED_MIN_EXP): the error of a pixel is bounded by sum_lm |a_lm| 2^cut, so a
// caller with a_lm of extreme dynamic range lowers it (-900 = the oracle's "exactly zero").  The first version used 2^-900:
// with it the rings 513..1023 of nside 1024 kept lmax + 1 > h + 1 cells, so their ring FFT took the aliased
// (LDS-atomic) fold and read 20-45 % more cells; 2^-120 followed (K4 66.2 -> 58.3 ms), 2^-80 took another 1.1 ms off
// K4 (56.4 -> 55.3 ms at cfg 3) with no change in any printed digit of the full-size comparisons.  The oracle keeps 2^-900.
// Round 4: 2^-70 = 8.5e-22 - this library's own choice, justified by the bound below and by measurement.  tools/cut_probe.py
// (cfg 3, unit-variance a_lm, maps against the 2^-900 plan): 2^-120 1.6e-13, 2^-80 3.4e-13, 2^-70 3.6e-13, 2^-60 4.6e-13,
// 2^-40 4.7e-13 of the rms - all of it the rounding of recurrences started at different rows, none of it truncation (the
// bound with every dropped term of a pixel adding coherently, sum_lm |a_lm| 2^cut ~ sqrt(nalm) rms 2^-70, is 1e-18 rms at
// lmax = 2048); K4 52.4 ms at 2^-80, 52.1 at 2^-70, 51.6 at 2^-60, 50.9 at 2^-40: a looser cut would buy another
// 0.5-1.2 ms and is not taken.
#ifndef SEED_MIN_EXP
#define SEED_MIN_EXP (-70)
#endif

// lstart[m][r]: first l at which |lambda_lm(ring r)| >= 2^min_exp (the plan's cut, default SEED_MIN_EXP = -70), with
// the two recurrence values there; terms below (< 8.5e-22 at the default) are dropped.
__global__ void seed_kernel(int lmax, int npair, int min_exp, const double *__restrict__ z, const double *__restrict__ sth,
                            const double *__restrict__ pref, const double2 *__restrict__ coef,
                            const double2 *__restrict__ coefmu, int32_t *__restrict__ lstart, double2 *__restrict__ seed,
                            double2 *__restrict__ seedmu, double2 *__restrict__ seed4) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    int m = blockIdx.y;
    if (r >= npair) return;
    // seed4[o][kq]: the state (mu_{R-2}, mu_{R-1}) of the scaled recurrence in front of row R = R_kq, the first row of the
    // form m + 2 kq + 8 k (k >= 0) that is >= lstart - 1 - the start of the first 8-row window of legendre_kernel's lane
    // group kq in which the ring contributes (its lanes feed rows R, R + 1 of every window).  Every lane then enters at
    // a window start: no per-step injection tests in the kernel.  Rows lstart - 3 .. lstart - 1 appear with their true
    // values (below the cut, i.e. < 2^min_exp: harmless); in front of row m the state is (-mu_m, 0), which makes
    // mu_m = alpha_m x 0 + mu_m and mu_{m+1} = alpha_{m+1} x mu_m come out of the recurrence itself.
    auto emit4 = [&](long o, int ls, const double *lam /* rows ls-3 .. ls+5 */) {
        const double2 *cm = coefmu + alm_idx(0, m, lmax);
        double mu[9];
        for (int i = 0; i < 9; i++) {
            const int row = ls - 3 + i;
            mu[i] = (row >= m && row <= lmax) ? lam[i] / cm[row].y : 0.0;
        }
        for (int i = 0; i < 9; i++)
            if (ls - 3 + i == m - 2) mu[i] = -(lam[m - (ls - 3)] / cm[m].y);   // (s_m = 1)
        for (int kq = 0; kq < 4; kq++) {
            const int t = ls - 1 - m - 2 * kq;
            const int R = m + 2 * kq + (t > 0 ? ((t + 7) >> 3) << 3 : 0);
            const int i0 = R - 2 - (ls - 3);
            seed4[4 * o + kq] = make_double2(mu[i0], mu[i0 + 1]);
        }
    };
    double x = z[r];
    double pm;
    int pe;
    scaled_pow(sth[r], m, pm, pe);
    int t;
    double mant = frexp(pm * pref[m], &t);
    int sc = pe + t;
    if (m & 1) mant = -mant;
    long o = (long)m * npair + r;
    const double2 *cf = coef + alm_idx(0, m, lmax);
    double lam[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // lambda at rows lstart - 3 .. lstart + 5
    auto forward5 = [&](int ls, double q0, double q1, int e) {           // rows ls + 1 .. ls + 5 behind (q0, q1) 2^e
        for (int l = ls + 1; l <= ls + 5; l++) {
            double v = 0.0;
            if (l <= lmax) {
                const double2 c = cf[l];
                v = fma(c.x * x, q1, -(c.y * q0));
            }
            q0 = q1;
            q1 = v;
            lam[3 + l - ls] = ldexp(v, e);
        }
    };
    if (sc >= min_exp) {
        lstart[o] = m;
        seed[o] = make_double2(0.0, ldexp(mant, sc));
        seedmu[o] = make_double2(0.0, ldexp(mant, sc));     // s_m = 1
        lam[3] = ldexp(mant, sc);
        forward5(m, 0.0, mant, sc);
        emit4(o, m, lam);
        return;
    }
    double p0 = 0.0, p1 = mant;  // scaled by 2^sc
    double pm1 = 0.0, pm2 = 0.0; // the two rows in front of p0
    int found = lmax + 1;
    double s0 = 0.0, s1 = 0.0;
    const double lam_m = ldexp(mant, sc);   // (may underflow to 0: row m - 2 is only asked for when lstart <= m + 1)
    for (int l = m + 1; l <= lmax; l++) {
        double2 c = cf[l];
        double v = fma(c.x * x, p1, -(c.y * p0));
        pm2 = pm1;
        pm1 = p0;
        p0 = p1;
        p1 = v;
        if (fabs(p1) > 0x1p100) {
            pm2 *= 0x1p-100;
            pm1 *= 0x1p-100;
            p0 *= 0x1p-100;
            p1 *= 0x1p-100;
            sc += 100;
        }
        if (p1 != 0.0 && sc + ilogb(p1) >= min_exp) {
            found = l;
            s0 = ldexp(p0, sc);
            s1 = ldexp(p1, sc);
            break;
        }
    }
    lstart[o] = found;
    seed[o] = make_double2(s0, s1);
    if (found <= lmax) {
        lam[0] = ldexp(pm2, sc), lam[1] = ldexp(pm1, sc), lam[2] = s0, lam[3] = s1;
        if (m >= found - 3) lam[m - (found - 3)] = lam_m;   // (row m itself, wherever it sits in the window)
        forward5(found, p0, p1, sc);
        emit4(o, found, lam);
    } else {
        for (int kq = 0; kq < 4; kq++) seed4[4 * o + kq] = make_double2(0.0, 0.0);
    }
    // the same pair in the scaled form of the synthesis kernel: mu_l = lambda_l / s_l
    const double2 *cm = coefmu + alm_idx(0, m, lmax);
    seedmu[o] = found <= lmax ? make_double2(s0 / cm[found - 1].y, s1 / cm[found].y) : make_double2(0.0, 0.0);
}

// test hook: lambda_lm for one (m, ring pair), l = m..lmax, as the synthesis kernel forms them: the scaled recurrence
// mu_l = (alpha_l x) mu_{l-1} - mu_{l-2} from the mu seeds, lambda_l = s_l mu_l
__global__ void lambda_kernel(int lmax, int npair, int m, int r, const double *__restrict__ z,
                              const double2 *__restrict__ coefmu, const int32_t *__restrict__ lstart,
                              const double2 *__restrict__ seedmu, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double x = z[r];
    long o = (long)m * npair + r;
    int ls = lstart[o];
    double2 sd = seedmu[o];
    const double2 *cf = coefmu + alm_idx(0, m, lmax);
    double p0 = 0.0, p1 = 0.0;
    for (int l = m; l <= lmax; l++) {
        double2 c = cf[l];
        double v = fma(c.x * x, p1, -p0);
        bool inj = (l == ls);
        v = inj ? sd.y : v;
        p0 = inj ? sd.x : p1;
        p1 = v;
        out[l - m] = l >= ls ? v * c.y : 0.0;
    }
}

// test hook: the same values as legendre_kernel's lane group kq forms them - zero in front of its entry row
// R = m + 2 kq + 8 k >= lstart - 1, the plan's entry state (seed4) there, the scaled recurrence behind it
__global__ void lambda_entry_kernel(int lmax, int npair, int m, int r, int kq, const double *__restrict__ z,
                                    const double2 *__restrict__ coefmu, const int32_t *__restrict__ lstart,
                                    const double2 *__restrict__ seed4, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double x = z[r];
    const long o = (long)m * npair + r;
    const int ls = lstart[o];
    const double2 sd = seed4[4 * o + kq];
    const double2 *cf = coefmu + alm_idx(0, m, lmax);
    const int t = ls - 1 - m - 2 * kq;
    const int R = ls <= lmax ? m + 2 * kq + (t > 0 ? ((t + 7) >> 3) << 3 : 0) : 0x7fffffff;
    double p0 = 0.0, p1 = 0.0;
    for (int l = m; l <= lmax; l++) {
        if (l == R) {
            p0 = sd.x;
            p1 = sd.y;
        }
        const double2 c = cf[l];
        const double v = fma(c.x * x, p1, -p0);
        p0 = p1;
        p1 = v;
        out[l - m] = v * c.y;
    }
}
// per ring: mcut = number of m (from 0) whose lambda_lm reach the plan's cut for some l <= lmax; F_m of the ring
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
// FP64 MFMA instructions legendre_kernel<NT, RT> ISSUES for one column group, summed over (m, ring tile, wave): the
// kernel's own skip logic restated (sht_legendre.hip / leg_stage_body.inc) - an item starts at
// l_begin = m + ((lmin - m) & ~7) with lmin the tile's first contributing l; wave w of the workgroup owns the rings
// tile * 128 RT + 8 j + w (j < 16 RT) and executes the 2 NT RT MFMAs of the macro-step at l0 = l_begin + 8 k iff
// l0 <= lmax and ws_min <= l0 + 7 (ws_min = first contributing l of the wave's rings).  Thread = (m, tile, wave);
// out accumulates the number of executed macro-steps.  The count is what SQ_INSTS_VALU_MFMA_F64 reads for the
// launch (tests/test_gpu_fullsize.py checks it against the committed PMC profile): the roofline fraction of K4 is
// priced on it, not on the algorithmic 8 nside nalm F of SURVEY 8(d), which counts terms nobody has to compute.
__global__ void k4_count_kernel(int lmax, int npair, int rt, const int32_t *__restrict__ lstart,
                                const int32_t *__restrict__ lmin_tab, unsigned long long *__restrict__ out,
                                unsigned long long *__restrict__ out_clean) {
    const int trings = LEG_RINGS * rt;
    const int ntile = (npair + trings - 1) / trings;
    const int ntile128 = (npair + LMIN_RINGS - 1) / LMIN_RINGS;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)(lmax + 1) * ntile * LEG_WAVES;
    unsigned long long n = 0;
    if (idx < total) {
        const int wave = (int)(idx % LEG_WAVES);
        const int tile = (int)((idx / LEG_WAVES) % ntile);
        const int m = (int)(idx / ((long)LEG_WAVES * ntile));
        int lmin = lmax + 1;
        const int t_first = (tile * trings) / LMIN_RINGS;
        const int t_last = min((tile * trings + trings - 1) / LMIN_RINGS, ntile128 - 1);
        for (int t = t_first; t <= t_last; t++) lmin = min(lmin, lmin_tab[m * ntile128 + t]);
        if (lmin <= lmax) {
            int ws_min = lmax + 1;
            for (int j = 0; j < 16 * rt; j++) {
                const int ring = tile * trings + j * LEG_WAVES + wave;
                if (ring < npair) ws_min = min(ws_min, lstart[(long)m * npair + ring]);
            }
            const int l_begin = m + ((lmin - m) & ~7);
            const int k_max = (lmax - l_begin) / 8;
            const int need = ws_min - 7 - l_begin;                 // l0 >= ws_min - 7
            const int k_min = need > 0 ? (need + 7) / 8 : 0;
            if (k_max >= k_min) n = (unsigned long long)(k_max - k_min + 1);
            if (out_clean) {
                // macro-steps of this wave that run in the test-free steady-state stages (sht_legendre.hip: stage_lo && st < st_hi);
                // row_limit is not restated: it only matters for the last m's
                const int KT = LEG_KT;
                const int nstage = (lmax - l_begin) / KT + 1;
                const int st_hi = min((lmax - l_begin + 1) / KT, nstage - LEG_NBUF + 1);
                // ws_maxinj: the last first-contributing l of the wave's rings that lies at or after l_begin + (stagger): bound it by the max lstart <= lmax
                int ws_max = -1;
                for (int j = 0; j < 16 * rt; j++) {
                    const int ring = tile * trings + j * LEG_WAVES + wave;
                    if (ring < npair) {
                        const int ls = lstart[(long)m * npair + ring];
                        if (ls <= lmax) ws_max = max(ws_max, ls);
                    }
                }
                unsigned long long nc = 0;
                for (int st = 0; st < st_hi; st++) {
                    const int ls0 = l_begin + st * KT;
                    if (ws_min <= ls0 && ws_max < ls0) nc += KT / 8;
                }
                if (nc) atomicAdd(out_clean, nc);     // (diagnostics only: one atomic per thread, divergent code - no wave reduction here)
            }
        }
    }
    // wave-level sum, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}
// e^{i t phi0} of north-cap ring blockIdx.x, t < 512, and the steps e^{i 256 phi0}, e^{i 512 phi0}: the expression
// FrontEnd::fold (sht_ringfft_ct.hip) evaluated per item - phi0 / pi as the kernels form it, then sincospi
__global__ void fold_phase_kernel(const double *__restrict__ phi0, double2 *__restrict__ ph, double2 *__restrict__ step) {
    const int r = blockIdx.x, t = threadIdx.x;
    const double pop = phi0[r] / M_PI;
    double s, c;
    sincospi((double)t * pop, &s, &c);
    ph[(size_t)r * 512 + t] = make_double2(c, s);
    if (t < 2) {
        sincospi((double)(256 << t) * pop, &s, &c);
        step[r * 2 + t] = make_double2(c, s);
    }
}
// Bluestein tables for cap ring i (h = 2i not a power of two): chirp b_j = e^{i pi j^2/h}, j < h,
// and filt = FFT_P(conj chirp wrapped), stored in the digit-reversed order fft_dif produces.
__global__ void __launch_bounds__(256)
bluestein_table_kernel(const int32_t *__restrict__ blu_P, const int64_t *__restrict__ boff,
                       const int64_t *__restrict__ foff, double2 *__restrict__ chirp,
                       double2 *__restrict__ filt, const double2 *__restrict__ tw, int pmax, int tl_off) {
    extern __shared__ __attribute__((aligned(16))) double2 fbuf[];   // [fpad_len(maxlen) + 1] then the twiddle table
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
    for (int j = threadIdx.x; j < fpad_len(P); j += blockDim.x) fbuf[j] = make_double2(0.0, 0.0);
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

int sht_k4_items(corahip_ctx *ctx, const corahip_sht_plan *cp, int rt, int ncg, const int2 **items, int *nitems) {
    corahip_sht_plan *p = const_cast<corahip_sht_plan *>(cp);      // (lazy caches of the plan)
    const auto key = std::make_pair(rt, ncg);
    auto it = p->k4_items.find(key);
    if (it == p->k4_items.end()) {
        const int ntile128 = (p->npair + LMIN_RINGS - 1) / LMIN_RINGS;
        if (p->h_lmin.empty()) {
            p->h_lmin.resize((size_t)p->L * ntile128);
            HIP_TRY(hipMemcpyAsync(p->h_lmin.data(), p->d_lmin, sizeof(int32_t) * p->h_lmin.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        const int trings = LEG_RINGS * rt;
        const int ntile = (p->npair + trings - 1) / trings;
        ARG_CHECK(p->lmax < (1 << 15) && ntile <= 256 && ncg <= 256);
        std::vector<int2> list;
        list.reserve((size_t)p->L * ncg * ntile);
        for (int m = 0; m <= p->lmax; m++)
            for (int cg = 0; cg < ncg; cg++)
                for (int t = 0; t < ntile; t++) {
                    int lmin = p->lmax + 1;
                    const int t_first = (t * trings) / LMIN_RINGS;
                    const int t_last = std::min((t * trings + trings - 1) / LMIN_RINGS, ntile128 - 1);
                    for (int t128 = t_first; t128 <= t_last; t128++) lmin = std::min(lmin, p->h_lmin[(size_t)m * ntile128 + t128]);
                    if (lmin <= p->lmax) list.push_back(make_int2(m | (t << 15) | (cg << 23), lmin));
                }
        int2 *d = nullptr;
        HIP_TRY(hipMalloc((void **)&d, sizeof(int2) * std::max<size_t>(1, list.size())));
        HIP_TRY(hipMemcpyAsync(d, list.data(), sizeof(int2) * list.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        it = p->k4_items.emplace(key, std::make_pair(d, (int)list.size())).first;
    }
    *items = it->second.first;
    *nitems = it->second.second;
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
    (void)hipFree(p->d_coefmu);
    (void)hipFree(p->d_seedmu);
    (void)hipFree(p->d_seed4);
    for (auto &kv : p->k4_items) (void)hipFree(kv.second.first);
    (void)hipFree(p->d_tw);
    (void)hipFree(p->d_zeros);
    (void)hipFree(p->d_lmin);
    (void)hipFree(p->d_queue);
    (void)hipFree(p->d_mcut);
    (void)hipFree(p->d_polc);
    (void)hipFree(p->d_blu_P);
    (void)hipFree(p->d_blu_boff);
    (void)hipFree(p->d_blu_foff);
    (void)hipFree(p->d_blu3_foff);
    (void)hipFree(p->d_bfilt3);
    (void)hipFree(p->d_bchirp);
    (void)hipFree(p->d_bfilt);
    (void)hipFree(p->d_foldph);
    (void)hipFree(p->d_foldstep);
    for (auto &c : p->classes) (void)hipFree(c.d_list);
    delete p;
    return 0;
}

int corahip_sht_plan_create(corahip_ctx *ctx, int nside, int lmax, corahip_sht_plan **out) {
    return corahip_sht_plan_create_ex(ctx, nside, lmax, 0, out);
}

int corahip_sht_plan_cut_exp(const corahip_sht_plan *p, int *cut_exp) {
    ARG_CHECK(p != nullptr && cut_exp != nullptr);
    *cut_exp = p->cut_exp;
    return 0;
}

int corahip_sht_plan_k4_mfma_count(corahip_ctx *ctx, const corahip_sht_plan *p, int nnu, uint64_t *mfma_instructions) {
    ARG_CHECK(ctx != nullptr && p != nullptr && mfma_instructions != nullptr && nnu >= 1);
    // the launch shape sht_legendre picks for this many channels (16-column tiles: 8 channels x re/im)
    const int ncols = 2 * nnu_pad_of(nnu);
    const int nt16 = ncols / 16;
    const int NT = nt16 % 8 == 0 ? 8 : (nt16 % 4 == 0 ? 4 : (nt16 % 2 == 0 ? 2 : 1));
    const int RT = NT == 8 ? 1 : 2;
    auto it = p->k4_macro_steps.find(RT);
    if (it == p->k4_macro_steps.end()) {
        HIP_TRY(hipSetDevice(ctx->device));
        unsigned long long *d_n = nullptr, h_n = 0;
        HIP_TRY(hipMalloc((void **)&d_n, 2 * sizeof(h_n)));
        HIP_TRY(hipMemsetAsync(d_n, 0, 2 * sizeof(h_n), ctx->stream));
        static const bool dbg = getenv("CORAHIP_K4_COUNT_DEBUG") != nullptr;   // diagnostics: share of the steady-state stages
        const int ntile = (p->npair + LEG_RINGS * RT - 1) / (LEG_RINGS * RT);
        const long total = (long)p->L * ntile * LEG_WAVES;
        k4_count_kernel<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(p->lmax, p->npair, RT, p->d_lstart, p->d_lmin, d_n, dbg ? d_n + 1 : nullptr);
        LAUNCH_CHECK();
        HIP_TRY(hipMemcpyAsync(&h_n, d_n, sizeof(h_n), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (dbg) {
            unsigned long long h_c = 0;
            HIP_TRY(hipMemcpy(&h_c, d_n + 1, sizeof(h_c), hipMemcpyDeviceToHost));
            fprintf(stderr, "K4 macro-steps per column group (RT %d): %llu executed, %llu of them in steady-state stages (%.3f)\n", RT,
                    h_n, h_c, (double)h_c / (double)std::max<unsigned long long>(h_n, 1));
        }
        (void)hipFree(d_n);
        it = const_cast<corahip_sht_plan *>(p)->k4_macro_steps.emplace(RT, (uint64_t)h_n).first;
    }
    *mfma_instructions = it->second * (uint64_t)(2 * NT * RT) * (uint64_t)(ncols / (16 * NT));
    return 0;
}

int corahip_sht_plan_create_ex(corahip_ctx *ctx, int nside, int lmax, int cut_exp, corahip_sht_plan **out) {
    ARG_CHECK(ctx != nullptr && out != nullptr);
    ARG_CHECK(nside >= 1 && is_pow2(nside) && nside <= 8192);
    ARG_CHECK(lmax >= 0 && lmax <= 16384);
    ARG_CHECK(cut_exp <= 0 && cut_exp >= -1000);
    if (cut_exp == 0) cut_exp = SEED_MIN_EXP;
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
    p->cut_exp = cut_exp;
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
        // scaled form: s_m = s_{m+1} = 1, s_l = B_l s_{l-2}; alpha_l = A_l s_{l-1} / s_l (alpha_m = 0: never used, the
        // recurrence starts from the seed at l >= m).  s_l stays within [~0.15, 1] (it tends to sqrt(A_inf / A_{m+1})).
        std::vector<double2> cmu(p->nalm + 32, make_double2(0.0, 0.0));
        for (int m = 0; m <= lmax; m++) {
            long double aprev = 0.0L, s1 = 1.0L, s2 = 1.0L;    // A_{l-1}, s_{l-1}, s_{l-2}
            for (int l = m; l <= lmax; l++) {
                const long o = alm_idx(l, m, lmax);
                if (l == m) {
                    cmu[o] = make_double2(0.0, 1.0);
                    continue;
                }
                const long double ll = l, mm = m;
                const long double al = sqrtl((4.0L * ll * ll - 1.0L) / (ll * ll - mm * mm));
                const long double sl = l == m + 1 ? 1.0L : (al / aprev) * s2;
                cmu[o] = make_double2((double)(al * s1 / sl), (double)sl);
                aprev = al;
                s2 = s1;
                s1 = sl;
            }
        }
        if ((rc = dev_upload(&p->d_coefmu, cmu, s))) return rc;
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
    HIP_TRY(hipMalloc((void **)&p->d_seedmu, sizeof(double2) * (size_t)p->L * p->npair));
    HIP_TRY(hipMalloc((void **)&p->d_seed4, sizeof(double2) * 4 * (size_t)p->L * p->npair));
    {
        dim3 grid((p->npair + 63) / 64, p->L);
        seed_kernel<<<grid, 64, 0, s>>>(lmax, p->npair, cut_exp, p->d_z, p->d_sth, d_pref, p->d_coef, p->d_coefmu, p->d_lstart,
                                        p->d_seed, p->d_seedmu, p->d_seed4);
        LAUNCH_CHECK();
    }
    {
        const int ntile = (p->npair + LMIN_RINGS - 1) / LMIN_RINGS;
        HIP_TRY(hipMalloc((void **)&p->d_queue, 1024));   // 8 queue heads, 128 bytes apart
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
            const int tl_off = fpad_len(maxlen) + 1;
            const size_t shm = sizeof(double2) * (size_t)(tl_off + TWL_ENTRIES(p->pmax));
            HIP_TRY(hipFuncSetAttribute((const void *)bluestein_table_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            bluestein_table_kernel<<<nside - 1, 256, shm, s>>>(p->d_blu_P, p->d_blu_boff, p->d_blu_foff,
                                                                p->d_bchirp, p->d_bfilt, p->d_tw, p->pmax, tl_off);
            LAUNCH_CHECK();
        }
    }
    HIP_TRY(hipMalloc((void **)&p->d_foldph, sizeof(double2) * 512 * (size_t)nside));
    HIP_TRY(hipMalloc((void **)&p->d_foldstep, sizeof(double2) * 2 * (size_t)nside));
    if (nside > 1) {
        fold_phase_kernel<<<nside - 1, 512, 0, s>>>(p->d_phi0, p->d_foldph, p->d_foldstep);
        LAUNCH_CHECK();
    }
    // 3 * 2^k Bluestein lengths for the compile-time synthesis kernels: 3 P / 4 where it still holds 2 h - 1
    {
        p->h_blu3_P.assign(nside, 0);
        std::vector<int64_t> fo3(nside, 0);
        int64_t nf3 = 0;
        for (int i = 1; i < nside; i++) {
            const int h = 2 * i;
            if (is_pow2(h)) continue;
            int P = 1;
            while (P < 2 * h - 1) P <<= 1;
            // the length the compile-time kernels take for this ring, with a filter in THEIR storage order: 3 P / 4 where
            // it still holds 2 h - 1, and P = 8192 itself (their 16 x 16 x 32 schedule differs from the generic passes)
            const int P3 = 3 * (P / 4);
            int alt = 0;
            if (P == 4096 && 2560 >= 2 * h - 1 && !getenv("CORAHIP_K5_NO57")) alt = 2560;          // 5 * 2^9
            else if ((P3 == 1536 || P3 == 3072 || P3 == 6144) && P3 >= 2 * h - 1) alt = P3;
            else if (P == 4096 && 3584 >= 2 * h - 1 && !getenv("CORAHIP_K5_NO57")) alt = 3584;     // 7 * 2^9
            else if (P == 8192) alt = P;
            if (alt) {
                p->h_blu3_P[i - 1] = alt;
                fo3[i - 1] = nf3;
                nf3 += alt;
            }
        }
        if ((rc = dev_upload(&p->d_blu3_foff, fo3, s))) return rc;
        HIP_TRY(hipMalloc((void **)&p->d_bfilt3, sizeof(double2) * std::max<int64_t>(1, nf3)));
        if (nf3 && (rc = sht_blu3_tables(ctx, p, nf3))) return rc;
    }
    // K5 ring classes
    {
        // key: Bluestein length P, or -h for the belt (direct transform of the one length 2 nside: its own class so
        // that the compile-time kernel can take it); the few power-of-two cap rings share the run-time class 0
        std::map<int, std::vector<int32_t>> by_len;
        for (int r = 0; r < nring; r++) {
            const int i = r + 1;
            int icap = 0;
            if (i < nside) icap = i;
            else if (i > 3 * nside) icap = 4 * nside - i;
            int P = icap ? 0 : -2 * nside;
            if (icap) {
                const int h = 2 * icap;
                if (!is_pow2(h)) {
                    P = 1;
                    while (P < 2 * h - 1) P <<= 1;
                    // rings with a compile-time length of their own (3 P / 4: code 1; P in the compile-time order: 2;
                    // 5 P / 8: 3; 7 P / 8: 4)
                    const int alt = p->h_blu3_P[icap - 1];
                    P = 8 * P + (alt == 0 ? 0 : (alt == P ? 2 : (alt == 3 * (P / 4) ? 1 : (alt == 5 * (P / 8) ? 3 : 4))));
                }
            }
            by_len[P].push_back(r);
        }
        size_t lds_budget = 160 * 1024;
        if (getenv("CORAHIP_K5_LDS_KB")) lds_budget = (size_t)atoi(getenv("CORAHIP_K5_LDS_KB")) * 1024;
        for (auto &kv : by_len) {
            corahip_sht_plan::ring_class c;
            c.P = kv.first > 0 ? kv.first / 8 : 0;
            {
                const int code = kv.first > 0 ? (kv.first & 7) : 0;
                c.P3 = code == 1 ? 3 * (c.P / 4) : (code == 2 ? c.P : (code == 3 ? 5 * (c.P / 8) : (code == 4 ? 7 * (c.P / 8) : 0)));
            }
            c.N = kv.first < 0 ? -kv.first : 0;
            c.bstride = fpad_len(c.P ? c.P : 2 * nside + 1) + K5_CH_SKEW;
            c.nch = 4;
            const size_t tl_bytes = sizeof(double2) * TWL_ENTRIES(p->pmax);   // LDS twiddle table behind the buffers
            while (c.nch > 1 && (size_t)c.nch * c.bstride * sizeof(double2) + tl_bytes > lds_budget) c.nch >>= 1;
            if ((size_t)c.nch * c.bstride * sizeof(double2) + tl_bytes > 160 * 1024) {
                corahip_set_error("nside %d: ring FFT of length %d does not fit in LDS", nside, c.bstride);
                return CORAHIP_ENOMEM;
            }
            // (measured and rejected for the P = 4096 class: one channel per 4-wave workgroup, two workgroups per CU,
            //  so that LDS and FP64 phases of different workgroups overlap: 12.8 -> 13.6 ms, the cells are read 4x)
            // The short Bluestein classes (P <= 512: 0.5 % of the pixels at nside 1024, but six launches in a row) run
            // workgroups of 64 / 128 threads instead of 512: a 512-point transform of four channels has 128 radix-16
            // butterflies per pass - most of a 512-thread workgroup only attended the barriers (P = 512 0.41 -> 0.19 ms,
            // P = 256 0.19 -> 0.05 ms, K5 -0.5 ms at cfg 3).
            if (c.P > 0 && c.P <= 512) c.threads = c.P >= 512 ? 128 : 64;
            c.count = (int)kv.second.size();
            c.h_list = kv.second;
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

int corahip_sht_plan_ring_classes(const corahip_sht_plan *p, int32_t *host_len) {
    ARG_CHECK(p != nullptr && host_len != nullptr);
    for (const auto &c : p->classes)
        for (int32_t r : c.h_list) host_len[r] = c.P3 ? c.P3 : c.P;   // (the 3 * 2^k kernels take the classes that admit them)
    return 0;
}

int corahip_sht_lambda(corahip_ctx *ctx, const corahip_sht_plan *p, int m, int ring_pair, double *out) {
    ARG_CHECK(ctx != nullptr && p != nullptr && out != nullptr);
    ARG_CHECK(m >= 0 && m <= p->lmax && ring_pair >= 0 && ring_pair < p->npair);
    lambda_kernel<<<1, 64, 0, ctx->stream>>>(p->lmax, p->npair, m, ring_pair, p->d_z, p->d_coefmu, p->d_lstart,
                                             p->d_seedmu, out);
    LAUNCH_CHECK();
    return 0;
}

int corahip_sht_lambda_entry(corahip_ctx *ctx, const corahip_sht_plan *p, int m, int ring_pair, int kq, double *out) {
    ARG_CHECK(ctx != nullptr && p != nullptr && out != nullptr);
    ARG_CHECK(m >= 0 && m <= p->lmax && ring_pair >= 0 && ring_pair < p->npair && kq >= 0 && kq < 4);
    lambda_entry_kernel<<<1, 64, 0, ctx->stream>>>(p->lmax, p->npair, m, ring_pair, kq, p->d_z, p->d_coefmu, p->d_lstart,
                                                   p->d_seed4, out);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
