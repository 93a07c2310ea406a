// corrfunc.hip - xi(r) -> C_l(chi, chi'): replaces corrfunc.corr_to_clarray
// (cora/signal/corrfunc.py:290-400) for correlation functions given as cubic-spline tables
// (cora/util/cubicspline.pyx Interpolater / LogInterpolater / SinhInterpolater), and its Legendre
// projection  C_l = sum_m (w_m 4 pi / wsum) P_l(mu_m) xi_m  (corrfunc.py:387-397) for any input.
//
//   xi_table_kernel      [mu][i][j >= i]: radial-bin average of the spline at r(mu, x_i + a, x_j + b)
//   legendre_matrix      lm[l][m] = weight_m P_l(mu_m) by the three-term recurrence (one thread per node)
//   dgemm_nn_kernel      C[L x N] = A[L x M] B[M x N] on FP64 MFMA, 128 x 128 tiles, LDS-staged 16-deep K chunks
#include "common.h"

// natural cubic spline with end-slope extrapolation, as cubicspline.pyx:126-175 (knots in LDS)
__device__ static inline double spline_eval(const double *__restrict__ xs, const double *__restrict__ ys,
                                            const double *__restrict__ y2, int n, double x) {
    if (x < xs[0]) {
        const double h = xs[1] - xs[0];
        return ((ys[1] - ys[0]) / h - h * y2[1] / 6.0) * (x - xs[0]) + ys[0];
    }
    if (x >= xs[n - 1]) {
        const double h = xs[n - 1] - xs[n - 2];
        return ((ys[n - 1] - ys[n - 2]) / h + h * y2[n - 2] / 6.0) * (x - xs[n - 1]) + ys[n - 1];
    }
    int kl = 0, kh = n;
    while (kh - kl > 1) {   // bisection exactly as the reference: interval [kl, kl+1) with xs[kl] <= x
        const int kn = (kh + kl) >> 1;
        if (xs[kn] > x) kh = kn;
        else kl = kn;
    }
    const double h = xs[kl + 1] - xs[kl];
    const double a = (xs[kl + 1] - x) / h, b = (x - xs[kl]) / h;
    return a * ys[kl] + b * ys[kl + 1] + ((a * a * a - a) * y2[kl] + (b * b * b - b) * y2[kl + 1]) * (h * h) / 6.0;
}

// kind: 0 plain, 1 log-log (exp(spline(log r))), 2 sinh (f_t sinh(spline(asinh(r / x_t))))
__global__ void __launch_bounds__(256)
xi_table_kernel(const double *__restrict__ kx, const double *__restrict__ ky, const double *__restrict__ ky2, int nk,
                int kind, double x_t, double f_t, const double *__restrict__ mu, int nm,
                const double *__restrict__ xa, const double *__restrict__ xw, int F, int xint,
                double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sx = sm, *sy = sm + nk, *s2 = sm + 2 * nk;
    for (int t = threadIdx.x; t < nk; t += blockDim.x) {
        sx[t] = kx[t];
        sy[t] = ky[t];
        s2[t] = ky2[t];
    }
    __syncthreads();
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
                const double dx = x1 - x2;
                const double r = sqrt(dx * dx + 2.0 * x1 * x2 * om);
                double v;
                if (kind == 1) v = exp(spline_eval(sx, sy, s2, nk, log(r)));
                else if (kind == 2) v = f_t * sinh(spline_eval(sx, sy, s2, nk, asinh(r / x_t)));
                else v = spline_eval(sx, sy, s2, nk, r);
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
    const size_t shm = sizeof(double) * 3 * (size_t)nk;
    HIP_TRY(hipFuncSetAttribute((const void *)xi_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const long total = (long)nm * F * (F + 1) / 2;
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)ctx->num_cu * 8);
    xi_table_kernel<<<blocks, 256, shm, ctx->stream>>>(knots_x, knots_y, knots_y2, nk, kind, x_t, f_t, mu, nm, xa, xw, F,
                                                      xint, out);
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
