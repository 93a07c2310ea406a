// clarray.hip - C_l(nu,nu') integration with Romberg channel averaging.
//
// Replaces skysim.clarray (cora/core/skysim.py:10-69).  For the 21cm model the aps
// evaluation RedshiftCorrelation.angular_powerspectrum_fft (cora/signal/corr.py:944-982)
// and its three bilinearmap.interp calls (cora/util/bilinearmap.pyx:14-59) are fused with
// the Romberg reduction (skysim.py:62-67): no [l, F*zint, F*zint] intermediate exists.
//
// K1 mapping: one workgroup = (16x16 tile of channel pairs with jt >= it, 256 consecutive l).
// Lanes run over l, so for a given sub-sample pair every lane shares the table columns
// (y0, y0+1) and touches only the 1-3 adjacent table rows that 64 consecutive l span:
// gathers are near-broadcast and served from L1/L2 (the per-l working set of the three
// 131 MB tables is a ~3.5 MB band each).  Per-sub-pair constants (log10(xc kperpmin), y
// split, prefactors x Romberg weights) are computed once per (i, j-tile) into LDS.
#include "common.h"

#define CL_TI 16
#define CL_TJ 16
#define CL_MAXZ 17  // zint <= 17 (zromb <= 4)

#define CL_ISPLIT 4
#define CL_TJH 8    // channels of the j tile whose constants are resident in LDS at a time

// per sub-sample pair (i a, j b): everything that does not depend on l.  The y interpolation
// weights are folded into the coefficients: value = sum_T (cT0 T[x][y0] + cT1 T[x][y0+1]).
struct cl_pair_const {
    double lxcs;             // log10(xc * kperpmin) * xscale
    double c[6];             // {dd,dv,vv} x {(1-wy), wy} x W x model factor; W = w_a w_b pfD_a pfD_b/(xc^2 pi)
    unsigned y0, pad;        // floor of the clipped y
};

struct __attribute__((aligned(8))) cl_d2 {  // two adjacent table entries (8-byte aligned 16-byte load)
    double a, b;
};

__global__ void __launch_bounds__(256)
clarray21_kernel(const double *__restrict__ dd, const double *__restrict__ dv, const double *__restrict__ vv,
                 int nkperp, int nkpar, double kperpmin, double xscale, double yscale,
                 const double *__restrict__ chi, const double *__restrict__ pfd, const double *__restrict__ fz,
                 const double *__restrict__ bz, int F, int zint, const double *__restrict__ w,
                 const double *__restrict__ log10l, int nl, const int2 *__restrict__ tiles,
                 double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cl_pair_const *pc = reinterpret_cast<cl_pair_const *>(smem);  // [zint][CL_TJH][zint]
    const int tid = threadIdx.x;
    const int it = tiles[blockIdx.x].x, jt = tiles[blockIdx.x].y;
    const int li = blockIdx.y * 256 + tid;
    const bool l_ok = li < nl;
    const double lxs = (l_ok ? log10l[li] : 0.0) * xscale;
    const double ux = (double)nkperp - 1e-5, uy = (double)nkpar - 1e-5;
    const int nsub = zint * CL_TJH * zint;
    const unsigned xlast = (unsigned)(nkperp - 1);

    // blockIdx.z splits the 16 rows of the tile over CL_ISPLIT workgroups (more, shorter workgroups:
    // 136 tile pairs x 9 l-chunks alone leave the second wave of workgroups 40 % empty at F = 256)
    for (int ii = blockIdx.z * (CL_TI / CL_ISPLIT); ii < (blockIdx.z + 1) * (CL_TI / CL_ISPLIT); ii++) {
        const int i = it * CL_TI + ii;
        if (i >= F) break;
        double acc[CL_TJ];
#pragma unroll
        for (int jj = 0; jj < CL_TJ; jj++) acc[jj] = 0.0;
#pragma unroll
        for (int half = 0; half < CL_TJ / CL_TJH; half++) {
            __syncthreads();
            // constants of channel i against CL_TJH channels of the j tile
            for (int q = tid; q < nsub; q += 256) {
                const int b = q % zint, jj = (q / zint) % CL_TJH, a = q / (zint * CL_TJH);
                const int j = jt * CL_TJ + half * CL_TJH + jj;
                cl_pair_const c;
                c.lxcs = 0.0;
#pragma unroll
                for (int u = 0; u < 6; u++) c.c[u] = 0.0;
                c.y0 = 0;
                c.pad = 0;
                if (j < F) {
                    const int za = i * zint + a, zb = j * zint + b;
                    const double x1 = chi[za], x2 = chi[zb];
                    const double xc = 0.5 * (x1 + x2);
                    const double rpar = fabs(x2 - x1);
                    c.lxcs = log10(xc * kperpmin) * xscale;
                    double yy = rpar * yscale;  // rpar / (pi / kparmax)
                    yy = yy < 0.0 ? 0.0 : (yy > uy ? uy : yy);
                    unsigned y0 = (unsigned)yy;
                    double wy = yy - (double)y0;
                    if (y0 + 1 > (unsigned)(nkpar - 1)) {  // keep the 16-byte pair load in bounds (the
                        y0 = (unsigned)(nkpar - 2);        // reference reads out of bounds here)
                        wy = 1.0;
                    }
                    c.y0 = y0;
                    const double W = w[a] * w[b] * pfd[za] * pfd[zb] / (xc * xc * M_PI);
                    const double cdd = W * bz[za] * bz[zb];
                    const double cdv = W * (fz[za] * bz[zb] + fz[zb] * bz[za]);
                    const double cvv = W * fz[za] * fz[zb];
                    c.c[0] = cdd * (1.0 - wy);
                    c.c[1] = cdd * wy;
                    c.c[2] = cdv * (1.0 - wy);
                    c.c[3] = cdv * wy;
                    c.c[4] = cvv * (1.0 - wy);
                    c.c[5] = cvv * wy;
                }
                pc[q] = c;
            }
            __syncthreads();
            if (l_ok) {
                for (int a = 0; a < zint; a++) {
#pragma unroll
                    for (int jj = 0; jj < CL_TJH; jj++) {
                        double s = 0.0;
                        for (int b = 0; b < zint; b++) {
                            const cl_pair_const &c = pc[(a * CL_TJH + jj) * zint + b];
                            double xx = lxs - c.lxcs;
                            xx = xx < 0.0 ? 0.0 : (xx > ux ? ux : xx);
                            const unsigned x0 = (unsigned)xx;
                            const double wx = xx - (double)x0;
                            const unsigned x1 = x0 + 1 > xlast ? xlast : x0 + 1;
                            const unsigned o0 = x0 * (unsigned)nkpar + c.y0, o1 = x1 * (unsigned)nkpar + c.y0;
                            const cl_d2 d0 = *reinterpret_cast<const cl_d2 *>(dd + o0);
                            const cl_d2 v0 = *reinterpret_cast<const cl_d2 *>(dv + o0);
                            const cl_d2 q0 = *reinterpret_cast<const cl_d2 *>(vv + o0);
                            const cl_d2 d1 = *reinterpret_cast<const cl_d2 *>(dd + o1);
                            const cl_d2 v1 = *reinterpret_cast<const cl_d2 *>(dv + o1);
                            const cl_d2 q1 = *reinterpret_cast<const cl_d2 *>(vv + o1);
                            const double s0 = c.c[0] * d0.a + c.c[1] * d0.b + c.c[2] * v0.a + c.c[3] * v0.b +
                                              c.c[4] * q0.a + c.c[5] * q0.b;
                            const double s1 = c.c[0] * d1.a + c.c[1] * d1.b + c.c[2] * v1.a + c.c[3] * v1.b +
                                              c.c[4] * q1.a + c.c[5] * q1.b;
                            s += s0 + wx * (s1 - s0);
                        }
                        acc[half * CL_TJH + jj] += s;
                    }
                }
            }
        }
        if (l_ok) {
            double *orow = out + ((size_t)li * F + i) * F + (size_t)jt * CL_TJ;
#pragma unroll
            for (int jj = 0; jj < CL_TJ; jj++)
                if (jt * CL_TJ + jj < F) orow[jj] = acc[jj];
        }
    }
}

// fill everything below the diagonal from the computed upper part: C[l][i][j] = C[l][j][i], i > j
// (tiles with jt > it were not computed at all; inside diagonal tiles this makes the block
// exactly symmetric instead of symmetric to rounding as in the reference).
__global__ void clarray_mirror_kernel(double *__restrict__ out, int nl, int F) {
    const long n = (long)nl * F * F;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const int j = (int)(q % F), i = (int)((q / F) % F);
        const long l = q / ((long)F * F);
        if (i > j) out[q] = out[(l * F + j) * F + i];
    }
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

int corahip_clarray_table21cm(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv, int nkperp,
                              int nkpar, double kperpmin, double kperpmax, double kparmax, const double *chi,
                              const double *pfd, const double *f, const double *b, int F, int zint, const double *w,
                              const double *log10l, int nl, double *out) {
    ARG_CHECK(ctx != nullptr && dd && dv && vv && chi && pfd && f && b && w && log10l && out);
    ARG_CHECK(nkperp >= 2 && nkpar >= 2 && F >= 1 && zint >= 1 && zint <= CL_MAXZ && nl >= 1);
    ARG_CHECK(kperpmin > 0 && kperpmax > kperpmin && kparmax > 0);
    StageTimer t(ctx, "clarray");
    const int nt = (F + CL_TI - 1) / CL_TI;
    std::vector<int2> tiles;
    for (int it = 0; it < nt; it++)
        for (int jt = it; jt < nt; jt++) tiles.push_back(make_int2(it, jt));
    int2 *dtiles = nullptr;
    HIP_TRY(hipMalloc((void **)&dtiles, sizeof(int2) * tiles.size()));
    HIP_TRY(hipMemcpyAsync(dtiles, tiles.data(), sizeof(int2) * tiles.size(), hipMemcpyHostToDevice, ctx->stream));
    const double xscale = (double)(nkperp - 1) / log10(kperpmax / kperpmin);
    const double yscale = kparmax / M_PI;
    const size_t shm = sizeof(cl_pair_const) * (size_t)zint * CL_TJH * zint;
    HIP_TRY(hipFuncSetAttribute((const void *)clarray21_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    dim3 grid((unsigned)tiles.size(), (nl + 255) / 256, CL_ISPLIT);
    clarray21_kernel<<<grid, 256, shm, ctx->stream>>>(dd, dv, vv, nkperp, nkpar, kperpmin, xscale, yscale, chi, pfd, f,
                                                      b, F, zint, w, log10l, nl, dtiles, out);
    LAUNCH_CHECK();
    if (F > 1) {
        const long n = (long)nl * F * F;
        clarray_mirror_kernel<<<(int)std::min<long>((n + 255) / 256, 4096), 256, 0, ctx->stream>>>(out, nl, F);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // dtiles lifetime (cold path)
    (void)hipFree(dtiles);
    return 0;
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
