/* oracle/sht_ref.c - CPU restatement of the spherical-harmonic synthesis that
 * the reference delegates to healpy.alm2map (cora/util/hputil.py:388-391).
 *
 * TEST INFRASTRUCTURE ONLY (oracle): used by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  Never linked into the product library.
 *
 * healpy is a third-party dependency absent from /root/reference, so this
 * follows the published definition:
 *   T(theta,phi) = sum_m c_m Re[F_m(theta) e^{i m phi}],  c_0 = 1, c_m = 2,
 *   F_m(theta)   = sum_{l>=m} a_lm lambda_lm(cos theta),
 *   lambda_lm    = sqrt((2l+1)/(4pi) (l-m)!/(l+m)!) P_l^m  (Condon-Shortley phase).
 * This file does the Legendre part (F_m for every ring pair); the per-ring
 * phase / alias fold / inverse real FFT is done in oracle/sht.py with numpy.
 *
 * Recurrence (per m, marching in l), with a power-of-two scale exponent so the
 * sin^m(theta) seed cannot underflow:
 *   lambda_mm     = (-1)^m sqrt((2m+1)!!/(4pi (2m)!!)) sin^m(theta)
 *   lambda_lm     = alpha_lm (x lambda_{l-1,m} - lambda_{l-2,m}/alpha_{l-1,m}),
 *   alpha_lm      = sqrt((4l^2-1)/(l^2-m^2)).
 * North/south symmetry lambda_lm(-x) = (-1)^{l+m} lambda_lm(x): even and odd
 * (l-m) partial sums are accumulated once per ring pair.
 */
#include <math.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* alm: packed healpy order idx(l,m) = m(2 lmax+1-m)/2 + l, interleaved (re,im).
 * z, sth: cos/sin(theta) of the `npair` northern rings (equator included).
 * fn, fs: [npair][lmax+1][2] outputs (F_m on the north ring, and on its mirror).
 *
 * Per (ring, m): phase 1 marches the recurrence with a power-of-two scale exponent
 * (no accumulation) until |lambda| >= 2^-900 - terms below that are < 1e-270 of the
 * sum and are dropped, as libsharp does; phase 2 is the plain recurrence. */
void oracle_legendre_synth(int lmax, int npair, const double *z, const double *sth,
                           const double *alm, double *fn, double *fs)
{
    const int L = lmax + 1;
    const long nalm = (long)L * (L + 1) / 2;
    /* log2 of |lambda_mm| prefactor */
    double *lp = (double *)malloc(sizeof(double) * L);
    lp[0] = -0.5 * log2(4.0 * M_PI);
    for (int m = 1; m < L; m++)
        lp[m] = lp[m - 1] + 0.5 * log2((2.0 * m + 1.0) / (2.0 * m));
    /* alpha_lm and 1/alpha_lm at idx(l,m); alpha_mm = inf -> 1/alpha = 0 */
    double *al = (double *)malloc(sizeof(double) * nalm);
    double *ial = (double *)malloc(sizeof(double) * nalm);
#pragma omp parallel for schedule(dynamic, 16)
    for (int m = 0; m < L; m++) {
        long base = (long)m * (2 * lmax + 1 - m) / 2;
        const double m2 = (double)m * m;
        al[base + m] = 0.0;
        ial[base + m] = 0.0;
        for (int l = m + 1; l < L; l++) {
            double ll = l;
            double a = sqrt((4.0 * ll * ll - 1.0) / (ll * ll - m2));
            al[base + l] = a;
            ial[base + l] = 1.0 / a;
        }
    }

#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < npair; r++) {
        const double x = z[r];
        const double l2s = log2(sth[r]);
        for (int m = 0; m < L; m++) {
            const long base = (long)m * (2 * lmax + 1 - m) / 2;
            const double *a = alm + 2 * base; /* a[2*l], a[2*l+1] is a_lm */
            const double *A = al + base, *IA = ial + base;
            double L2 = lp[m] + m * l2s;
            int sc = (int)floor(L2);
            double lam = exp2(L2 - sc);
            if (m & 1) lam = -lam;
            double lam_prev = 0.0;
            int l = m;
            /* phase 1: scaled, no accumulation */
            while (l < L && sc + ilogb(lam) < -900) {
                double nxt = (l + 1 < L) ? A[l + 1] * (x * lam - lam_prev * IA[l]) : 0.0;
                lam_prev = lam;
                lam = nxt;
                l++;
                if (fabs(lam) > 0x1p300) { lam *= 0x1p-300; lam_prev *= 0x1p-300; sc += 300; }
                if (lam == 0.0) break;
            }
            double fer = 0, fei = 0, f_or = 0, foi = 0;
            if (l < L && lam != 0.0) {
                lam = ldexp(lam, sc);
                lam_prev = ldexp(lam_prev, sc);
                /* phase 2: plain recurrence, even/odd (l-m) sums */
                for (; l < L; l++) {
                    double ar = a[2 * l], ai = a[2 * l + 1];
                    if (((l - m) & 1) == 0) { fer += ar * lam; fei += ai * lam; }
                    else                    { f_or += ar * lam; foi += ai * lam; }
                    if (l + 1 < L) {
                        double nxt = A[l + 1] * (x * lam - lam_prev * IA[l]);
                        lam_prev = lam;
                        lam = nxt;
                    }
                }
            }
            long o = 2 * ((long)r * L + m);
            fn[o] = fer + f_or; fn[o + 1] = fei + foi;
            fs[o] = fer - f_or; fs[o + 1] = fei - foi;
        }
    }
    free(lp); free(al); free(ial);
}


/* The same sums as oracle_legendre_synth with the loops re-cut for the host CPU, used where the oracle is TIMED
 * (bench.py's cpu_baseline): a block of RB consecutive ring pairs is taken through the l recurrence together, so that
 * the coefficients alpha_lm and the a_lm of a multipole are loaded once per block instead of once per ring and the
 * inner loop over the block's rings vectorises (the arrangement libsharp, healpy's engine, uses; target_clones
 * selects AVX-512 / AVX2 / baseline code at load time).  Per ring and m the scaled phase 1 of oracle_legendre_synth
 * finds the first l with |lambda| >= 2^-900 and the two recurrence values there; phase 2 starts every ring of the
 * block at its own l by injecting those values.  Checked against oracle_legendre_synth in tests/test_oracle.py. */
#define RB 8
__attribute__((target_clones("avx512f", "avx2", "default")))
void oracle_legendre_synth_blocked(int lmax, int npair, const double *z, const double *sth,
                                   const double *alm, double *fn, double *fs)
{
    const int L = lmax + 1;
    const long nalm = (long)L * (L + 1) / 2;
    double *lp = (double *)malloc(sizeof(double) * L);
    lp[0] = -0.5 * log2(4.0 * M_PI);
    for (int m = 1; m < L; m++)
        lp[m] = lp[m - 1] + 0.5 * log2((2.0 * m + 1.0) / (2.0 * m));
    double *al = (double *)malloc(sizeof(double) * nalm);
    double *ial = (double *)malloc(sizeof(double) * nalm);
#pragma omp parallel for schedule(dynamic, 16)
    for (int m = 0; m < L; m++) {
        long base = (long)m * (2 * lmax + 1 - m) / 2;
        const double m2 = (double)m * m;
        al[base + m] = 0.0;
        ial[base + m] = 0.0;
        for (int l = m + 1; l < L; l++) {
            double ll = l;
            double a = sqrt((4.0 * ll * ll - 1.0) / (ll * ll - m2));
            al[base + l] = a;
            ial[base + l] = 1.0 / a;
        }
    }
    const int nblk = (npair + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int blk = 0; blk < nblk; blk++) {
        const int r0 = blk * RB;
        double x[RB], l2s[RB];
        for (int q = 0; q < RB; q++) {
            const int r = r0 + q < npair ? r0 + q : npair - 1;     /* (a padded lane repeats the last ring) */
            x[q] = z[r];
            l2s[q] = log2(sth[r]);
        }
        for (int m = 0; m < L; m++) {
            const long base = (long)m * (2 * lmax + 1 - m) / 2;
            const double *a = alm + 2 * base;
            const double *A = al + base, *IA = ial + base;
            int ls[RB];
            double s0[RB], s1[RB];
            int lmin = L;
            for (int q = 0; q < RB; q++) {                          /* phase 1, scalar per ring */
                double L2 = lp[m] + m * l2s[q];
                int sc = (int)floor(L2);
                double lam = exp2(L2 - sc);
                if (m & 1) lam = -lam;
                double lam_prev = 0.0;
                int l = m;
                while (l < L && sc + ilogb(lam) < -900) {
                    double nxt = (l + 1 < L) ? A[l + 1] * (x[q] * lam - lam_prev * IA[l]) : 0.0;
                    lam_prev = lam;
                    lam = nxt;
                    l++;
                    if (fabs(lam) > 0x1p300) { lam *= 0x1p-300; lam_prev *= 0x1p-300; sc += 300; }
                    if (lam == 0.0) break;
                }
                if (l < L && lam != 0.0) {
                    ls[q] = l;
                    s1[q] = ldexp(lam, sc);
                    s0[q] = ldexp(lam_prev, sc);
                } else {
                    ls[q] = L;
                    s0[q] = s1[q] = 0.0;
                }
                if (ls[q] < lmin) lmin = ls[q];
            }
            double lam[RB], lamp[RB], er[RB], ei[RB], orr[RB], oi[RB];
            for (int q = 0; q < RB; q++) lam[q] = lamp[q] = er[q] = ei[q] = orr[q] = oi[q] = 0.0;
            for (int l = lmin; l < L; l++) {                        /* phase 2, the block's rings together */
                const double ar = a[2 * l], ai = a[2 * l + 1];
                const double Anext = l + 1 < L ? A[l + 1] : 0.0, IAl = IA[l];
                const int even = ((l - m) & 1) == 0;
#pragma omp simd
                for (int q = 0; q < RB; q++) {
                    const int start = l == ls[q];
                    const double lq = start ? s1[q] : lam[q];
                    const double pq = start ? s0[q] : lamp[q];
                    if (even) { er[q] += ar * lq; ei[q] += ai * lq; }
                    else      { orr[q] += ar * lq; oi[q] += ai * lq; }
                    lam[q] = Anext * (x[q] * lq - pq * IAl);
                    lamp[q] = lq;
                }
            }
            for (int q = 0; q < RB && r0 + q < npair; q++) {
                long o = 2 * ((long)(r0 + q) * L + m);
                fn[o] = er[q] + orr[q]; fn[o + 1] = ei[q] + oi[q];
                fs[o] = er[q] - orr[q]; fs[o + 1] = ei[q] - oi[q];
            }
        }
    }
    free(lp); free(al); free(ial);
}

/* Adjoint of oracle_legendre_synth (the Legendre part of healpy.map2alm, which the reference
 * reaches through hputil.sphtrans_real, cora/util/hputil.py:195-234):
 *   a_lm = sum_rings lambda_lm(z_r) [G_m(north r) + (-1)^{l+m} G_m(south mirror of r)]
 * gn, gs: [npair][lmax+1][2] weighted ring spectra (gs of an unpaired equator ring = 0).
 * alm out: packed healpy order, interleaved (re,im).  Same two-phase recurrence as above. */
void oracle_legendre_anal(int lmax, int npair, const double *z, const double *sth,
                          const double *gn, const double *gs, double *alm)
{
    const int L = lmax + 1;
    double *lp = (double *)malloc(sizeof(double) * L);
    lp[0] = -0.5 * log2(4.0 * M_PI);
    for (int m = 1; m < L; m++)
        lp[m] = lp[m - 1] + 0.5 * log2((2.0 * m + 1.0) / (2.0 * m));
#pragma omp parallel for schedule(dynamic, 4)
    for (int m = 0; m < L; m++) {
        const long base = (long)m * (2 * lmax + 1 - m) / 2;
        const double m2 = (double)m * m;
        double *A = (double *)malloc(sizeof(double) * (L + 1));
        double *IA = (double *)malloc(sizeof(double) * (L + 1));
        A[m] = 0.0; IA[m] = 0.0;
        for (int l = m + 1; l < L; l++) {
            double ll = l;
            A[l] = sqrt((4.0 * ll * ll - 1.0) / (ll * ll - m2));
            IA[l] = 1.0 / A[l];
        }
        double *out = alm + 2 * base;
        for (int l = m; l < L; l++) { out[2 * l] = 0.0; out[2 * l + 1] = 0.0; }
        for (int r = 0; r < npair; r++) {
            const double x = z[r];
            const long o = 2 * ((long)r * L + m);
            const double er = gn[o] + gs[o], ei = gn[o + 1] + gs[o + 1];       /* even l-m */
            const double orr = gn[o] - gs[o], oi = gn[o + 1] - gs[o + 1];      /* odd l-m  */
            double L2 = lp[m] + m * log2(sth[r]);
            int sc = (int)floor(L2);
            double lam = exp2(L2 - sc);
            if (m & 1) lam = -lam;
            double lam_prev = 0.0;
            int l = m;
            while (l < L && sc + ilogb(lam) < -900) {
                double nxt = (l + 1 < L) ? A[l + 1] * (x * lam - lam_prev * IA[l]) : 0.0;
                lam_prev = lam;
                lam = nxt;
                l++;
                if (fabs(lam) > 0x1p300) { lam *= 0x1p-300; lam_prev *= 0x1p-300; sc += 300; }
                if (lam == 0.0) break;
            }
            if (l < L && lam != 0.0) {
                lam = ldexp(lam, sc);
                lam_prev = ldexp(lam_prev, sc);
                for (; l < L; l++) {
                    if (((l - m) & 1) == 0) { out[2 * l] += er * lam;  out[2 * l + 1] += ei * lam; }
                    else                    { out[2 * l] += orr * lam; out[2 * l + 1] += oi * lam; }
                    if (l + 1 < L) {
                        double nxt = A[l + 1] * (x * lam - lam_prev * IA[l]);
                        lam_prev = lam;
                        lam = nxt;
                    }
                }
            }
        }
        free(A); free(IA);
    }
    free(lp);
}

/* normalised lambda_lm(x) for one (m, x), l = m..lmax, same recurrence
 * (used by the tests to check against mpmath / scipy spot values). */
void oracle_lambda_lm(int lmax, int m, double x, double sthv, double *out)
{
    double L2 = -0.5 * log2(4.0 * M_PI);
    for (int k = 1; k <= m; k++) L2 += 0.5 * log2((2.0 * k + 1.0) / (2.0 * k));
    L2 += m * log2(sthv);
    int sc = (int)floor(L2);
    double lam = exp2(L2 - sc);
    if (m & 1) lam = -lam;
    double lam_prev = 0.0, inv_alpha_prev = 0.0;
    const double m2 = (double)m * m;
    for (int l = m; l <= lmax; l++) {
        out[l - m] = (sc > -1070) ? ldexp(lam, sc) : 0.0;
        double lp1 = l + 1.0;
        double alpha = sqrt((4.0 * lp1 * lp1 - 1.0) / (lp1 * lp1 - m2));
        double nxt = alpha * (x * lam - lam_prev * inv_alpha_prev);
        lam_prev = lam;
        lam = nxt;
        inv_alpha_prev = 1.0 / alpha;
        if (fabs(lam) > 0x1p300) { lam *= 0x1p-300; lam_prev *= 0x1p-300; sc += 300; }
    }
}


/* Spin-2 (polarisation) synthesis, Legendre part: what the reference obtains from healpy.alm2map([T, E, B], nside)
 * for Q and U through hputil.sphtrans_inv_real_pol (cora/util/hputil.py:394-432).  Convention (Zaldarriaga & Seljak
 * 1997, the "COSMO" convention HEALPix documents):  Q +- iU = - sum_lm (a^E_lm +- i a^B_lm) (+-2)Y_lm, with
 *   (+2)Y_lm = (W_lm - X_lm) e^{i m phi},  (-2)Y_lm = (W_lm + X_lm) e^{i m phi},
 *   W_lm = 2 N_l [ -((l - m^2)/sin^2 + l(l-1)/2) lambda_lm + (l+m) (cos/sin^2) c_lm lambda_{l-1,m} ],
 *   X_lm = 2 N_l (m/sin^2) [ (l-1) cos lambda_lm - (l+m) c_lm lambda_{l-1,m} ],
 *   N_l = 1/sqrt((l+2)(l+1)l(l-1)),  c_lm = sqrt((2l+1)/(2l-1) (l-m)/(l+m))
 * (checked against the eth-operator definition of the spin-weighted harmonics in tests/test_oracle.py), so that
 *   Q_m = - sum_l (E_lm W_lm - i B_lm X_lm),   U_m = - sum_l (B_lm W_lm + i E_lm X_lm).
 * W has the parity (-1)^{l+m} of lambda under theta -> pi - theta, X the opposite one.
 * alme, almb: packed healpy order, interleaved (re, im); outputs [npair][lmax+1][2] for Q and U on the north ring
 * and on its mirror.  PARITY UNPINNED against healpy (absent). */
void oracle_legendre_synth_spin2(int lmax, int npair, const double *z, const double *sth, const double *alme,
                                 const double *almb, double *qn, double *qs, double *un, double *us)
{
    const int L = lmax + 1;
    double *lp = (double *)malloc(sizeof(double) * L);
    lp[0] = -0.5 * log2(4.0 * M_PI);
    for (int m = 1; m < L; m++)
        lp[m] = lp[m - 1] + 0.5 * log2((2.0 * m + 1.0) / (2.0 * m));
#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < npair; r++) {
        const double x = z[r], s2 = sth[r] * sth[r];
        const double l2s = log2(sth[r]);
        double *A = (double *)malloc(sizeof(double) * (L + 1));
        double *IA = (double *)malloc(sizeof(double) * (L + 1));
        for (int m = 0; m < L; m++) {
            const long base = (long)m * (2 * lmax + 1 - m) / 2;
            const double *ae = alme + 2 * base, *ab = almb + 2 * base;
            const double m2 = (double)m * m;
            A[m] = 0.0; IA[m] = 0.0;
            for (int l = m + 1; l < L; l++) {
                double ll = l;
                A[l] = sqrt((4.0 * ll * ll - 1.0) / (ll * ll - m2));
                IA[l] = 1.0 / A[l];
            }
            double L2 = lp[m] + m * l2s;
            int sc = (int)floor(L2);
            double lam = exp2(L2 - sc);
            if (m & 1) lam = -lam;
            double lam_prev = 0.0;
            int l = m;
            while (l < L && sc + ilogb(lam) < -900) {
                double nxt = (l + 1 < L) ? A[l + 1] * (x * lam - lam_prev * IA[l]) : 0.0;
                lam_prev = lam;
                lam = nxt;
                l++;
                if (fabs(lam) > 0x1p300) { lam *= 0x1p-300; lam_prev *= 0x1p-300; sc += 300; }
                if (lam == 0.0) break;
            }
            /* sums of W a and X a over even / odd (l - m), for the four real columns (Re E, Im E, Re B, Im B) */
            double we[4] = {0, 0, 0, 0}, wo[4] = {0, 0, 0, 0}, xe[4] = {0, 0, 0, 0}, xo[4] = {0, 0, 0, 0};
            if (l < L && lam != 0.0) {
                lam = ldexp(lam, sc);
                lam_prev = ldexp(lam_prev, sc);
                for (; l < L; l++) {
                    if (l >= 2) {
                        const double ll = l;
                        const double N2 = 2.0 / sqrt((ll + 2.0) * (ll + 1.0) * ll * (ll - 1.0));
                        const double cl = (l > m) ? (2.0 * ll + 1.0) * IA[l] : 0.0;      /* (l+m) c_lm = (2l+1)/A_l */
                        const double W = N2 * (-((ll - m2) / s2 + ll * (ll - 1.0) / 2.0) * lam + cl * x / s2 * lam_prev);
                        const double X = N2 * m / s2 * ((ll - 1.0) * x * lam - cl * lam_prev);
                        const double col[4] = {ae[2 * l], ae[2 * l + 1], ab[2 * l], ab[2 * l + 1]};
                        double *w = ((l - m) & 1) ? wo : we, *xx = ((l - m) & 1) ? xo : xe;
                        for (int c = 0; c < 4; c++) { w[c] += W * col[c]; xx[c] += X * col[c]; }
                    }
                    if (l + 1 < L) {
                        double nxt = A[l + 1] * (x * lam - lam_prev * IA[l]);
                        lam_prev = lam;
                        lam = nxt;
                    }
                }
            }
            /* north: W_e + W_o, X_e + X_o; south: W_e - W_o, X_o - X_e */
            for (int hemi = 0; hemi < 2; hemi++) {
                double sw[4], sx[4];
                for (int c = 0; c < 4; c++) {
                    sw[c] = hemi ? we[c] - wo[c] : we[c] + wo[c];
                    sx[c] = hemi ? xo[c] - xe[c] : xe[c] + xo[c];
                }
                double *q = (hemi ? qs : qn) + 2 * ((long)r * L + m);
                double *u = (hemi ? us : un) + 2 * ((long)r * L + m);
                /* Q = -(E W - i B X): Re = -(ReE W + ImB X), Im = -(ImE W - ReB X) */
                q[0] = -(sw[0] + sx[3]);
                q[1] = -(sw[1] - sx[2]);
                /* U = -(B W + i E X): Re = -(ReB W - ImE X), Im = -(ImB W + ReE X) */
                u[0] = -(sw[2] - sx[1]);
                u[1] = -(sw[3] + sx[0]);
            }
        }
        free(A); free(IA);
    }
    free(lp);
}

/* cora/util/bilinearmap.pyx:14-59 restated (OpenMP over points like the reference's prange): clip to
 * [0, n - 1e-5], truncate to the lower corner, four-corner weights.  n points, table arr [nx][ny] row-major. */
void oracle_bilinear_interp(const double *arr, long nx, long ny, const double *x, const double *y, long n,
                            double *out)
{
    const double ux = (double)nx - 1e-5, uy = (double)ny - 1e-5;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        double xx = x[i] < 0.0 ? 0.0 : (x[i] > ux ? ux : x[i]);
        double yy = y[i] < 0.0 ? 0.0 : (y[i] > uy ? uy : y[i]);
        long x0 = (long)xx, y0 = (long)yy;
        long x1 = x0 + 1, y1 = y0 + 1;
        /* the reference reads arr[x1][y1] unchecked; x1 = nx / y1 = ny can only occur within 1e-5 of the
         * upper clip, which the hot path never reaches - clamp instead of reading out of bounds */
        long xc = x1 < nx ? x1 : nx - 1, yc = y1 < ny ? y1 : ny - 1;
        double wa = ((double)x1 - xx) * ((double)y1 - yy);
        double wb = ((double)x1 - xx) * (yy - (double)y0);
        double wc = (xx - (double)x0) * ((double)y1 - yy);
        double wd = (xx - (double)x0) * (yy - (double)y0);
        out[i] = wa * arr[x0 * ny + y0] + wb * arr[x0 * ny + yc] + wc * arr[xc * ny + y0] + wd * arr[xc * ny + yc];
    }
}

/* ------------------------------------------------------------------------------------
 * Ring stage of the synthesis in C/OpenMP ("Ring FFT with aliasing"): per ring the phase
 * e^{i m phi0}, the alias fold onto n = nphi bins and the length-n inverse DFT.  Same arithmetic as
 * oracle/sht.py:ring_synthesis (which stays the readable definition and is what the tests compare this with);
 * this version exists so that the oracle, where it is TIMED as the CPU baseline, runs on all host cores instead of
 * a Python loop over 4 nside rings.  Lengths that are powers of two use an iterative radix-2 FFT, all others
 * Bluestein's chirp convolution on top of it.
 * ------------------------------------------------------------------------------------ */
typedef struct { double re, im; } cplx;

static void fft_pow2(cplx *a, int n, int sign, const cplx *tw /* e^{+2 pi i k/n}, k < n/2 */)
{
    for (int i = 1, j = 0; i < n; i++) {               /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { cplx t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                cplx w = tw[k * step];
                if (sign < 0) w.im = -w.im;
                cplx u = a[i + k], v = a[i + k + len / 2];
                cplx t = { v.re * w.re - v.im * w.im, v.re * w.im + v.im * w.re };
                a[i + k].re = u.re + t.re; a[i + k].im = u.im + t.im;
                a[i + k + len / 2].re = u.re - t.re; a[i + k + len / 2].im = u.im - t.im;
            }
    }
}

/* fn, fs: [npair][L][2] F_m of the north rings and of their southern mirrors (oracle_legendre_synth);
 * start, nphi, phi0: [nring] ring geometry; map: [npix] RING ordered output. */
void oracle_ring_synth(int nside, int lmax, const double *fn, const double *fs, const long *start, const int *nphi,
                       const double *phi0, double *map)
{
    const int L = lmax + 1, nring = 4 * nside - 1, npair = 2 * nside;
    int pmax = 1;
    while (pmax < 8 * nside) pmax <<= 1;                /* >= 2 n - 1 for every ring */
    cplx *tw = (cplx *)malloc(sizeof(cplx) * (pmax / 2));
    for (int k = 0; k < pmax / 2; k++) {
        tw[k].re = cos(2.0 * M_PI * k / pmax);
        tw[k].im = sin(2.0 * M_PI * k / pmax);
    }
#pragma omp parallel
    {
        cplx *X = (cplx *)malloc(sizeof(cplx) * pmax);
        cplx *A = (cplx *)malloc(sizeof(cplx) * pmax);
        cplx *B = (cplx *)malloc(sizeof(cplx) * pmax);
        cplx *W = (cplx *)malloc(sizeof(cplx) * pmax);
        cplx *twl = (cplx *)malloc(sizeof(cplx) * (pmax / 2));
#pragma omp for schedule(dynamic, 4)
        for (int r = 0; r < nring; r++) {
            const int n = nphi[r];
            const double *f = r < npair ? fn + 2 * (long)r * L : fs + 2 * (long)(nring - 1 - r) * L;
            for (int k = 0; k < n; k++) X[k].re = X[k].im = 0.0;
            for (int m = 0; m < L; m++) {
                const double c = cos(m * phi0[r]), s = sin(m * phi0[r]);
                const double cr = f[2 * m] * c - f[2 * m + 1] * s, ci = f[2 * m] * s + f[2 * m + 1] * c;
                if (m == 0) { X[0].re += cr; continue; }
                const int k = m % n, kc = (n - k) % n;
                X[k].re += cr; X[k].im += ci;
                X[kc].re += cr; X[kc].im -= ci;
            }
            double *out = map + start[r];
            if ((n & (n - 1)) == 0) {                   /* power of two: T_j = sum_k X_k e^{+2 pi i jk/n} */
                if (n == 1) { out[0] = X[0].re; continue; }
                for (int k = 0; k < n / 2; k++) twl[k] = tw[k * (pmax / n)];
                fft_pow2(X, n, +1, twl);
                for (int j = 0; j < n; j++) out[j] = X[j].re;
                continue;
            }
            /* Bluestein: jk = (j^2 + k^2 - (j-k)^2)/2, w_k = e^{i pi k^2/n} */
            int P = 1;
            while (P < 2 * n - 1) P <<= 1;
            for (int k = 0; k < P / 2; k++) twl[k] = tw[k * (pmax / P)];
            for (int k = 0; k < n; k++) {
                const long q = ((long)k * k) % (2L * n);
                W[k].re = cos(M_PI * q / n); W[k].im = sin(M_PI * q / n);
            }
            for (int k = 0; k < P; k++) A[k].re = A[k].im = B[k].re = B[k].im = 0.0;
            for (int k = 0; k < n; k++) {
                A[k].re = X[k].re * W[k].re - X[k].im * W[k].im;
                A[k].im = X[k].re * W[k].im + X[k].im * W[k].re;
                B[k].re = W[k].re; B[k].im = -W[k].im;
                if (k) B[P - k] = B[k];
            }
            fft_pow2(A, P, -1, twl);
            fft_pow2(B, P, -1, twl);
            for (int k = 0; k < P; k++) {
                const double re = A[k].re * B[k].re - A[k].im * B[k].im, im = A[k].re * B[k].im + A[k].im * B[k].re;
                A[k].re = re; A[k].im = im;
            }
            fft_pow2(A, P, +1, twl);
            for (int j = 0; j < n; j++) out[j] = (A[j].re * W[j].re - A[j].im * W[j].im) / P;
        }
        free(X); free(A); free(B); free(W); free(twl);
    }
    free(tw);
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
