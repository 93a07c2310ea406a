// tables21.hip - K0: the three lookup tables of the flat-sky 21cm model on the device.
//
// Replaces the one-off setup of RedshiftCorrelation.angular_powerspectrum_fft (cora/signal/corr.py:909-942):
//   kperp = logspace(kperpmin, kperpmax, 500), kpar = linspace(0, kparmax, 32768)
//   dd = P(k) sinc^2(kpar w / 2 pi), dv = dd mu^2, vv = dd mu^4,  k^2 = kpar^2 + kperp^2, mu^2 = kpar^2 / k^2
//   table = scipy.fftpack.dct(., type=1) * kparmax / (2 nkpar)     along kpar
// with P(k) = exp(-k^2 / 2 k*^2) LogInterpolater(ps_z1.5.dat)(k) for Corr21cm (cora/signal/corr21cm.py:24-29): a
// natural cubic spline in (log k, log P) with bisection lookup and end-slope extrapolation
// (cora/util/cubicspline.pyx:126-175,254-288).
//
// The DCT-I of N = nkpar points is the real DFT of the even extension of length M = 2 (N - 1) = 65534, i.e. one
// complex DFT of length H = N - 1 = 32767 = 7 * 31 * 151 on z_n = e_2n + i e_2n+1 plus the usual split.  H has no
// factor 2, so instead of padding to a power of two (Bluestein at 65536 would need a two-pass out-of-LDS FFT) the
// transform uses the Good-Thomas prime-factor map: with pairwise coprime factors the length-H DFT is EXACTLY a
// 7 x 31 x 151 tensor DFT - three batched small dense DFTs, no twiddles between them, sums of at most 151 terms
// (rounding ~1e-15) - at 189 complex multiply-adds per element it is a few milliseconds for the 1500 rows.
// Any nkpar whose nkpar - 1 splits into pairwise coprime prime powers <= TAB_MAXP works the same way.
#include "common.h"

#include <cmath>

#define TAB_MAXP 2048      // largest prime-power factor of nkpar - 1 (its root-of-unity table lives in LDS)
#define TAB_MAXF 6         // at most this many coprime factors

// ------------------------------------------------------------------------------------
// P(k) tables
// ------------------------------------------------------------------------------------
// natural cubic spline with the reference's conventions; knots in LDS
__device__ static inline double spline_eval(const double *xs, const double *ys, const double *y2, int n, double xv) {
    if (xv < xs[0]) {
        const double h0 = xs[1] - xs[0];
        return ((ys[1] - ys[0]) / h0 - h0 * y2[1] / 6) * (xv - xs[0]) + ys[0];
    }
    if (xv >= xs[n - 1]) {
        const double h1 = xs[n - 1] - xs[n - 2];
        return ((ys[n - 1] - ys[n - 2]) / h1 + h1 * y2[n - 2] / 6) * (xv - xs[n - 1]) + ys[n - 1];
    }
    int lo = 0, hi = n - 1;               // xs[lo] <= xv < xs[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] <= xv) lo = mid;
        else hi = mid;
    }
    const double h = xs[hi] - xs[lo];
    const double a = (xs[hi] - xv) / h, b = (xv - xs[lo]) / h;
    return a * ys[lo] + b * ys[hi] + (a * a * a - a) * h * h / 6 * y2[lo] + (b * b * b - b) * h * h / 6 * y2[hi];
}

// mode 0: dd from the spline (loglog: exp(spline(log k)); kstar > 0: Gaussian cut-off); mode 1: dd given (host callable)
__global__ void __launch_bounds__(256)
ps_table_kernel(int mode, const double *__restrict__ kx, const double *__restrict__ ky, const double *__restrict__ ky2,
                int nknot, int loglog, double kstar, const double *__restrict__ kperp, int nkperp,
                const double *__restrict__ kpar, int nkpar, double freq_window, const double *__restrict__ dd_in,
                double *__restrict__ dd, double *__restrict__ dv, double *__restrict__ vv) {
    extern __shared__ double knots[];   // [3][nknot]
    if (mode == 0) {
        for (int i = threadIdx.x; i < nknot; i += blockDim.x) {
            knots[i] = kx[i];
            knots[nknot + i] = ky[i];
            knots[2 * nknot + i] = ky2[i];
        }
        __syncthreads();
    }
    const long n = (long)nkperp * nkpar;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int i = (int)(q / nkpar), j = (int)(q - (long)i * nkpar);
        const double kp = kpar[j], kt = kperp[i];
        const double k = sqrt(kp * kp + kt * kt);          // (kpar**2 + kperp**2) ** 0.5
        const double mu2 = (kp * kp) / (k * k);            // kpar**2 / k**2
        double d;
        if (mode == 0) {
            double p = loglog ? exp(spline_eval(knots, knots + nknot, knots + 2 * nknot, nknot, log(k)))
                              : spline_eval(knots, knots + nknot, knots + 2 * nknot, nknot, k);
            if (kstar > 0.0) p = exp(-0.5 * (k * k) / (kstar * kstar)) * p;
            double win = 1.0;
            if (freq_window != 0.0) {                      // np.sinc(kpar w / 2 pi) ** 2
                const double xw = kp * freq_window / (2.0 * M_PI);
                const double sc = xw == 0.0 ? 1.0 : sinpi(xw) / (M_PI * xw);
                win = sc * sc;
            }
            d = p * win;
            dd[q] = d;
        } else {
            d = dd_in[q];
        }
        dv[q] = d * mu2;
        vv[q] = d * (mu2 * mu2);
    }
}

// ------------------------------------------------------------------------------------
// DCT-I of the rows of a [nrows][n] array through a prime-factor DFT of length H = n - 1
// ------------------------------------------------------------------------------------
struct pfa_desc {
    int nf;                 // number of coprime factors
    int dims[TAB_MAXF];     // N_i, tensor index t = ((n_0 N_1 + n_1) N_2 + n_2) ...
    int mstep[TAB_MAXF];    // H / N_i: input index n = sum n_i (H / N_i) mod H
};

// z_n = e_2n + i e_2n+1 of the even extension e (e_m = x_m, m <= H; e_m = x_{2H - m} above), gathered into the tensor
// order of the prime-factor input map
__global__ void dct_pack_kernel(const double *__restrict__ x, long nrows, int n, pfa_desc d, double2 *__restrict__ T) {
    const int H = n - 1;
    const long tot = nrows * H;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
        const long row = q / H;
        int t = (int)(q - row * H);
        long nn = 0;
        for (int f = d.nf - 1; f >= 0; f--) {
            const int ni = t % d.dims[f];
            t /= d.dims[f];
            nn += (long)ni * d.mstep[f];
        }
        const int nidx = (int)(nn % H);
        const int m0 = 2 * nidx, m1 = 2 * nidx + 1;                       // m1 <= 2H - 1
        const double *xr = x + row * n;
        const double e0 = xr[m0 <= H ? m0 : 2 * H - m0];
        const double e1 = xr[m1 <= H ? m1 : 2 * H - m1];
        T[q] = make_double2(e0, e1);
    }
}
// one tensor axis: out[.., k, ..] = sum_n e^{-2 pi i n k / p} in[.., n, ..]; axis size p, stride s (product of the later
// dims); one thread per output element, roots of unity in LDS
__global__ void __launch_bounds__(256)
pfa_stage_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, long nrows, int H, int p, int s) {
    __shared__ double2 w[TAB_MAXP];
    for (int i = threadIdx.x; i < p; i += blockDim.x) {
        double sv, cv;
        sincospi(2.0 * (double)i / (double)p, &sv, &cv);
        w[i] = make_double2(cv, -sv);
    }
    __syncthreads();
    const long tot = nrows * H;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
        const long row = q / H;
        const int t = (int)(q - row * H);
        const int inner = t % s;
        const int k = (t / s) % p;
        const int outer = t / (s * p);
        const double2 *src = in + row * H + (long)outer * p * s + inner;
        double ar = 0.0, ai = 0.0;
        int idx = 0;
        for (int nn = 0; nn < p; nn++) {
            const double2 v = src[(long)nn * s];
            const double2 ww = w[idx];
            ar = fma(v.x, ww.x, fma(-v.y, ww.y, ar));
            ai = fma(v.x, ww.y, fma(v.y, ww.x, ai));
            idx += k;
            if (idx >= p) idx -= p;
        }
        out[q] = make_double2(ar, ai);
    }
}
// y_k = Re E_k, E_k = 1/2 [(Z_k + conj Z_{H-k}) - i e^{-2 pi i k / M} (Z_k - conj Z_{H-k})], Z_H = Z_0, k = 0..H; Z_k sits
// at the tensor position of the residues (k mod N_0, k mod N_1, ...) - the CRT output map of the prime-factor DFT
__global__ void dct_split_kernel(const double2 *__restrict__ Z, long nrows, int n, pfa_desc d, double scale,
                                 double *__restrict__ y) {
    const int H = n - 1;
    const long tot = nrows * n;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
        const long row = q / n;
        const int k = (int)(q - row * n);
        auto pos = [&](int kk) {
            int t = 0;
            for (int f = 0; f < d.nf; f++) t = t * d.dims[f] + kk % d.dims[f];
            return t;
        };
        const double2 *Zr = Z + row * H;
        const double2 za = Zr[pos(k % H)];
        const double2 zb = Zr[pos((H - k) % H)];
        double sv, cv;
        sincospi((double)k / (double)H, &sv, &cv);                 // 2 pi k / M = pi k / H
        // (za + conj zb) - i (c - i s) (za - conj zb): real part
        const double sr = za.x + zb.x;
        const double dr = za.x - zb.x, di = za.y + zb.y;
        // -i (c - i s)(dr + i di) = -i (c dr + s di + i (c di - s dr)) = (c di - s dr) - i (c dr + s di)
        y[q] = 0.5 * (sr + (cv * di - sv * dr)) * scale;
    }
}

static int pfa_factor(int H, pfa_desc *d) {
    d->nf = 0;
    int rem = H;
    for (int p = 2; (long)p * p <= rem; p++) {
        if (rem % p) continue;
        int q = 1;
        while (rem % p == 0) {
            rem /= p;
            q *= p;
        }
        if (d->nf == TAB_MAXF || q > TAB_MAXP) return -1;
        d->dims[d->nf++] = q;
    }
    if (rem > 1) {
        if (d->nf == TAB_MAXF || rem > TAB_MAXP) return -1;
        d->dims[d->nf++] = rem;
    }
    for (int f = 0; f < d->nf; f++) d->mstep[f] = H / d->dims[f];
    return 0;
}

extern "C" {

int corahip_ps_table21cm(corahip_ctx *ctx, const double *knots_x, const double *knots_y, const double *knots_y2, int nknot,
                         int loglog, double kstar, const double *kperp, int nkperp, const double *kpar, int nkpar,
                         double freq_window, const double *dd_in, double *dd, double *dv, double *vv) {
    ARG_CHECK(ctx != nullptr && kperp != nullptr && kpar != nullptr && dv != nullptr && vv != nullptr);
    ARG_CHECK(nkperp >= 1 && nkpar >= 2);
    const int mode = dd_in != nullptr ? 1 : 0;
    if (mode == 0) ARG_CHECK(knots_x != nullptr && knots_y != nullptr && knots_y2 != nullptr && nknot >= 4 && dd != nullptr);
    ARG_CHECK(mode == 1 || (size_t)nknot * 3 * sizeof(double) <= 64 * 1024);
    StageTimer t(ctx, "tables");
    const size_t shm = mode == 0 ? (size_t)nknot * 3 * sizeof(double) : 0;
    ps_table_kernel<<<ctx->num_cu * 8, 256, shm, ctx->stream>>>(mode, knots_x, knots_y, knots_y2, nknot, loglog, kstar, kperp,
                                                               nkperp, kpar, nkpar, freq_window, dd_in, dd, dv, vv);
    LAUNCH_CHECK();
    return 0;
}

int corahip_dct1_workspace_bytes(long nrows, int n, size_t *bytes) {
    ARG_CHECK(bytes != nullptr && nrows >= 1 && n >= 3);
    *bytes = 2 * (size_t)nrows * (size_t)(n - 1) * sizeof(double2);
    return 0;
}

int corahip_dct1_rows(corahip_ctx *ctx, double *data, long nrows, int n, double scale, void *workspace,
                      size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && data != nullptr && workspace != nullptr && nrows >= 1 && n >= 3);
    size_t need;
    corahip_dct1_workspace_bytes(nrows, n, &need);
    if (workspace_bytes < need) {
        corahip_set_error("dct1_rows workspace too small: %zu bytes, need %zu", workspace_bytes, need);
        return CORAHIP_ENOMEM;
    }
    const int H = n - 1;
    pfa_desc d;
    if (pfa_factor(H, &d)) {
        corahip_set_error("dct1_rows: n - 1 = %d has a prime-power factor above %d (or more than %d factors)", H, TAB_MAXP,
                          TAB_MAXF);
        return CORAHIP_EINVAL;
    }
    StageTimer t(ctx, "tables");
    double2 *A = (double2 *)workspace, *B = A + (size_t)nrows * H;
    const int blocks = ctx->num_cu * 8;
    dct_pack_kernel<<<blocks, 256, 0, ctx->stream>>>(data, nrows, n, d, A);
    LAUNCH_CHECK();
    int s = H;
    for (int f = 0; f < d.nf; f++) {
        s /= d.dims[f];
        pfa_stage_kernel<<<blocks, 256, 0, ctx->stream>>>(A, B, nrows, H, d.dims[f], s);
        LAUNCH_CHECK();
        std::swap(A, B);
    }
    dct_split_kernel<<<blocks, 256, 0, ctx->stream>>>(A, nrows, n, d, scale, data);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
