// corrfunc.hip - xi(r) -> C_l(chi, chi'): replaces corrfunc.corr_to_clarray
// (cora/signal/corrfunc.py:290-400) for correlation functions given as cubic-spline tables
// (cora/util/cubicspline.pyx Interpolater / LogInterpolater / SinhInterpolater), and its Legendre
// projection  C_l = sum_m (w_m 4 pi / wsum) P_l(mu_m) xi_m  (corrfunc.py:387-397) for any input.
//
//   xi_table_kernel      [mu][i][j >= i]: radial-bin average of the spline at r(mu, x_i + a, x_j + b)
//   legendre_matrix      lm[l][m] = weight_m P_l(mu_m) by the three-term recurrence (one thread per node)
//   dgemm_nn_kernel      C[L x N] = A[L x M] B[M x N] on FP64 MFMA, 128 x 128 tiles, LDS-staged 16-deep K chunks
#include "common.h"

#include "rng_dev.h"   // fast_log01 (valid for any positive normal argument), fast_sqrt_pos

// natural cubic spline with end-slope extrapolation, as cubicspline.pyx:126-175.  Knot tables in LDS:
// xs, ys, y2 and per interval 1/h and h^2/6 (no divisions per evaluation); the interval comes from a uniform
// look-up grid over [xs[0], xs[n-1]] (lut[g] = last knot at or before the left edge of cell g) walked forward to
// the reference's bisection result - the last knot with xs[k] <= x - instead of ~log2(n) dependent LDS reads.
struct spline_lds {
    const double *xs, *ys, *y2, *invh, *h2o6;
    const int *lut;
    int n, nlut;
    double x0, inv_dx;
};
__device__ static inline double spline_eval(const spline_lds &S, double x) {
    const int n = S.n;
    if (x < S.xs[0]) {
        const double h = S.xs[1] - S.xs[0];
        return ((S.ys[1] - S.ys[0]) / h - h * S.y2[1] / 6.0) * (x - S.xs[0]) + S.ys[0];
    }
    if (x >= S.xs[n - 1]) {
        const double h = S.xs[n - 1] - S.xs[n - 2];
        return ((S.ys[n - 1] - S.ys[n - 2]) / h + h * S.y2[n - 2] / 6.0) * (x - S.xs[n - 1]) + S.ys[n - 1];
    }
    int g = (int)((x - S.x0) * S.inv_dx);
    g = g < 0 ? 0 : (g >= S.nlut ? S.nlut - 1 : g);
    int kl = S.lut[g];
    while (kl > 0 && S.xs[kl] > x) kl--;            // (rounding of the cell index at a cell edge)
    while (kl + 2 < n && S.xs[kl + 1] <= x) kl++;
    const double ih = S.invh[kl];
    const double a = (S.xs[kl + 1] - x) * ih, b = (x - S.xs[kl]) * ih;
    return a * S.ys[kl] + b * S.ys[kl + 1] + ((a * a * a - a) * S.y2[kl] + (b * b * b - b) * S.y2[kl + 1]) * S.h2o6[kl];
}

// asinh(u), u >= 0: log(u + sqrt(u^2 + 1)) with the short log / sqrt of rng_dev.h (the argument is >= 1, where
// fast_log01 keeps full relative accuracy); below 2^-6 the odd series (the log would cancel)
__device__ static inline double fast_asinh_pos(double u) {
    const double u2 = u * u;
    const double big = fast_log01(u + fast_sqrt_pos(u2 + 1.0));
    const double small = u * (1.0 + u2 * (-1.0 / 6.0 + u2 * (3.0 / 40.0 + u2 * (-15.0 / 336.0 + u2 * (105.0 / 3456.0)))));
    return u < 0.015625 ? small : big;
}
// sinh(y) for |y| < ~700, branch-free: E = expm1(|y|) = 2^k p + (2^k - 1) with p = expm1(r), r = |y| - k ln2
// (|r| <= ln2 / 2, degree-13 Taylor polynomial: remainder 4e-18), sinh = (E + E / (E + 1)) / 2
__device__ static inline double fast_expm1_pos(double ay) {   // ay >= 0
    const double kf = __builtin_rint(ay * 1.44269504088896340736);
    const double r = fma(-kf, 1.90821492927058770002e-10, fma(-kf, 6.93147180369123816490e-01, ay));   // ln2 hi / lo
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p * r, r, r);                       // expm1(r)
    const int k = (int)kf;
    const double two_k = __builtin_amdgcn_ldexp(1.0, k);
    return fma(two_k, p, two_k - 1.0);
}
__device__ static inline double fast_sinh(double y) {
    const double E = fast_expm1_pos(fabs(y));
    const double s = 0.5 * (E + E / (E + 1.0));
    return y < 0.0 ? -s : s;
}
// e^y, |y| < ~700 (e^{-inf} = 0)
__device__ static inline double fast_exp(double y) {
    const double ay = fabs(y);
    if (!(ay < 700.0)) return y < 0.0 ? 0.0 : exp(y);
    const double E1 = fast_expm1_pos(ay) + 1.0;
    return y < 0.0 ? 1.0 / E1 : E1;
}

// kind: 0 plain, 1 log-log (exp(spline(log r))), 2 sinh (f_t sinh(spline(asinh(r / x_t))))
__global__ void __launch_bounds__(256)
xi_table_kernel(const double *__restrict__ kx, const double *__restrict__ ky, const double *__restrict__ ky2, int nk,
                int nlut, int kind, double x_t, double f_t, const double *__restrict__ mu, int nm,
                const double *__restrict__ xa, const double *__restrict__ xw, int F, int xint,
                double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sx = sm, *sy = sm + nk, *s2 = sm + 2 * nk, *sih = sm + 3 * nk, *sh2 = sm + 4 * nk;
    int *slut = reinterpret_cast<int *>(sm + 5 * nk);
    for (int t = threadIdx.x; t < nk; t += blockDim.x) {
        sx[t] = kx[t];
        sy[t] = ky[t];
        s2[t] = ky2[t];
        if (t + 1 < nk) {
            const double h = kx[t + 1] - kx[t];
            sih[t] = 1.0 / h;
            sh2[t] = (h * h) / 6.0;
        }
    }
    __syncthreads();
    spline_lds S;
    S.xs = sx, S.ys = sy, S.y2 = s2, S.invh = sih, S.h2o6 = sh2, S.lut = slut, S.n = nk, S.nlut = nlut;
    S.x0 = sx[0];
    const double dx = (sx[nk - 1] - sx[0]) / (double)nlut;
    S.inv_dx = 1.0 / dx;
    for (int g = threadIdx.x; g < nlut; g += blockDim.x) {
        const double xl = S.x0 + g * dx;        // left edge of the cell: bisection for the last knot <= xl
        int kl = 0, kh = nk;
        while (kh - kl > 1) {
            const int kn = (kh + kl) >> 1;
            if (sx[kn] > xl) kh = kn;
            else kl = kn;
        }
        slut[g] = kl;
    }
    __syncthreads();
    const double inv_xt = 1.0 / x_t;
    const long npair = (long)F * (F + 1) / 2;
    const long total = (long)nm * npair;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int m = (int)(q / npair);
        long p = q - (long)m * npair;
        // (i, j >= i) from the row-major pair index
        int i = (int)(((2.0 * F + 1.0) - sqrt((2.0 * F + 1.0) * (2.0 * F + 1.0) - 8.0 * (double)p)) * 0.5);
        i = max(0, min(i, F - 1));
        while (i > 0 && (long)i * F - (long)i * (i - 1) / 2 > p) i--;
        while ((long)(i + 1) * F - (long)(i + 1) * i / 2 <= p) i++;
        const int j = i + (int)(p - ((long)i * F - (long)i * (i - 1) / 2));
        const double om = 1.0 - mu[m];
        double acc = 0.0;
        for (int a = 0; a < xint; a++) {
            const double x1 = xa[i * xint + a];
            double row = 0.0;
            for (int b = 0; b < xint; b++) {
                const double x2 = xa[j * xint + b];
                const double dx12 = x1 - x2;
                const double r = sqrt(dx12 * dx12 + 2.0 * x1 * x2 * om);
                double v;
                if (kind == 1) v = fast_exp(spline_eval(S, r > 0.0 ? fast_log01(r) : -INFINITY));
                else if (kind == 2) v = f_t * fast_sinh(spline_eval(S, fast_asinh_pos(r * inv_xt)));
                else v = spline_eval(S, r);
                row += xw[b] * v;
            }
            acc += xw[a] * row;
        }
        out[((size_t)m * F + i) * F + j] = acc;
        out[((size_t)m * F + j) * F + i] = acc;
    }
}

// lm[l][m] = wt[m] * P_l(mu[m]), l = 0..lmax (row stride ldm >= nm, padding columns zero)
__global__ void legendre_matrix_kernel(const double *__restrict__ mu, const double *__restrict__ wt, int nm, int lmax,
                                       int ldm, double *__restrict__ lm) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= ldm) return;
    const bool ok = m < nm;
    const double x = ok ? mu[m] : 0.0, w = ok ? wt[m] : 0.0;
    double p0 = 1.0, p1 = x;
    lm[m] = w;
    if (lmax >= 1) lm[(size_t)ldm + m] = w * x;
    for (int l = 2; l <= lmax; l++) {
        const double p2 = ((2.0 * l - 1.0) * x * p1 - (l - 1.0) * p0) / (double)l;
        p0 = p1;
        p1 = p2;
        lm[(size_t)l * ldm + m] = w * p2;
    }
}

// C[Mr x N] = A[Mr x K] B[K x N], row-major (lda, ldb, ldc), K a multiple of 16 (callers pad with zeros).
// Workgroup = 4 waves = 128 x 128 tile of C; wave = 64 x 64 = 4 x 4 MFMA tiles (128 accumulator VGPRs).
// K runs in 16-deep chunks through a double-buffered LDS stage: As[128][17] (row i, k), Bs[16][132] (k, col).
#define GM_T 128
#define GM_K 16
__global__ void __launch_bounds__(256)
dgemm_nn_kernel(const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb, double *__restrict__ C,
                int ldc, int Mr, int N, int K) {
    constexpr int AS = GM_K + 1, BS = GM_T + 4;
    __shared__ double As[2][GM_T * AS];
    __shared__ double Bs[2][GM_K * BS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.y * GM_T, col0 = blockIdx.x * GM_T;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;   // wave's 64 x 64 sub-tile
    d4_t acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[u][v] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // global -> registers -> LDS, split so that the loads of chunk c+1 are in flight while chunk c is multiplied
    double ra[GM_T * GM_K / 256], rb[GM_K * GM_T / 256];
    auto gload = [&](int kc) {
#pragma unroll
        for (int u = 0; u < GM_T * GM_K / 256; u++) {
            const int e = tid + 256 * u;
            const int r = e / GM_K, k = e % GM_K;     // A tile: 128 rows x 16 k (128-byte runs per row)
            const int gr = row0 + r;
            ra[u] = gr < Mr ? A[(size_t)gr * lda + kc + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < GM_K * GM_T / 256; u++) {
            const int e = tid + 256 * u;
            const int k = e / GM_T, c = e % GM_T;     // B tile: 16 k x 128 columns (coalesced along the columns)
            const int gc = col0 + c;
            rb[u] = gc < N ? B[(size_t)(kc + k) * ldb + gc] : 0.0;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < GM_T * GM_K / 256; u++) {
            const int e = tid + 256 * u;
            As[buf][(e / GM_K) * AS + e % GM_K] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < GM_K * GM_T / 256; u++) {
            const int e = tid + 256 * u;
            Bs[buf][(e / GM_T) * BS + e % GM_T] = rb[u];
        }
    };
    const int nchunk = K / GM_K;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload((c + 1) * GM_K);
        const double *as = As[buf], *bs = Bs[buf];
#pragma unroll
        for (int ks = 0; ks < GM_K / 4; ks++) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = as[(wr + 16 * u + ri) * AS + 4 * ks + kq];
#pragma unroll
            for (int v = 0; v < 4; v++) b[v] = bs[(4 * ks + kq) * BS + wc + 16 * v + ri];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
        }
        if (c + 1 < nchunk) lstore(buf ^ 1);   // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // C/D layout: column = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int gc = col0 + wc + 16 * v + ri;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gr = row0 + wr + 16 * u + kq + 4 * r;
                if (gr < Mr && gc < N) C[(size_t)gr * ldc + gc] = acc[u][v][r];
            }
        }
}

extern "C" {

int corahip_xi_table_average(corahip_ctx *ctx, const double *knots_x, const double *knots_y, const double *knots_y2,
                             int nk, int kind, double x_t, double f_t, const double *mu, int nm, const double *xa,
                             const double *xw, int F, int xint, double *out) {
    ARG_CHECK(ctx != nullptr && knots_x && knots_y && knots_y2 && mu && xa && xw && out);
    ARG_CHECK(nk >= 4 && nk <= 6000 && kind >= 0 && kind <= 2 && nm >= 1 && F >= 1 && xint >= 1);
    StageTimer t(ctx, "xi_average");
    const int nlut = 4 * nk;
    const size_t shm = sizeof(double) * 5 * (size_t)nk + sizeof(int) * (size_t)nlut;
    HIP_TRY(hipFuncSetAttribute((const void *)xi_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const long total = (long)nm * F * (F + 1) / 2;
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)ctx->num_cu * 8);
    xi_table_kernel<<<blocks, 256, shm, ctx->stream>>>(knots_x, knots_y, knots_y2, nk, nlut, kind, x_t, f_t, mu, nm, xa, xw,
                                                      F, xint, out);
    LAUNCH_CHECK();
    return 0;
}

int corahip_legendre_project(corahip_ctx *ctx, const double *mu, const double *wt, int nm, int lmax, const double *xi,
                             long ncol, double *out) {
    ARG_CHECK(ctx != nullptr && mu && wt && xi && out && nm >= 1 && lmax >= 0 && ncol >= 1);
    StageTimer t(ctx, "legendre_project");
    const int L = lmax + 1;
    const int Kp = (nm + GM_K - 1) / GM_K * GM_K;       // the GEMM runs K in chunks of 16: lm gets zero columns up to Kp
    double *lm = nullptr;
    int rc = corahip_ctx_scratch(ctx, 4, sizeof(double) * (size_t)L * Kp, (void **)&lm);
    if (rc) return rc;
    legendre_matrix_kernel<<<(Kp + 255) / 256, 256, 0, ctx->stream>>>(mu, wt, nm, lmax, Kp, lm);
    LAUNCH_CHECK();
    ARG_CHECK(ncol <= 0x7fffffffL);
    dim3 grid((unsigned)((ncol + GM_T - 1) / GM_T), (L + GM_T - 1) / GM_T);
    const double *Bop = xi;
    if (Kp != nm) {
        // rows nm..Kp-1 of the B operand must exist (and be zero): zero-padded copy of xi
        double *pad = nullptr;
        if ((rc = corahip_ctx_scratch(ctx, 5, sizeof(double) * (size_t)Kp * ncol, (void **)&pad))) return rc;
        HIP_TRY(hipMemcpyAsync(pad, xi, sizeof(double) * (size_t)nm * ncol, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemsetAsync(pad + (size_t)nm * ncol, 0, sizeof(double) * (size_t)(Kp - nm) * ncol, ctx->stream));
        Bop = pad;
    }
    dgemm_nn_kernel<<<grid, 256, 0, ctx->stream>>>(lm, Kp, Bop, (int)ncol, out, (int)ncol, L, (int)ncol, Kp);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
