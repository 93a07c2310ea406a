// sht_polana.hip - spin-2 ANALYSIS ((Q, U) maps -> (E, B) a_lm; healpy.map2alm of a T, Q, U triple as
// hputil.sphtrans_real_pol calls it, cora/util/hputil.py:274-323) composed from the scalar analysis kernels.
//
// With W_lm = (g1 r1 + g2) lambda_l + g3 r2 lambda_{l-1},  X_lm = g4 r2 lambda_l - m g3 r1 lambda_{l-1}
// (r1 = 1/sin^2, r2 = cos/sin^2 of the ring; g(l, m): the plan's spin-2 table, see sht_legendre.hip) the ring sums
//   sum_rings W_lm F = g1 A[r1 F]_lm + g2 A[F]_lm + g3 A[r2 F]_{l-1,m},
//   sum_rings X_lm F = g4 A[r2 F]_lm - m g3 A[r1 F]_{l-1,m}
// are combinations of SCALAR quadrature passes A[.] (ringana + K4^T, sht_ringfft.hip / sht_analysis.hip) of the map
// scaled ring by ring, because r1, r2 depend on the ring only and the sums run over north and south rings separately
// (the parity signs of lambda_{l-1}, r2 are the ring's own).  So
//   E_lm = -(sum W Q~ - i sum X U~),   B_lm = -(sum W U~ + i sum X Q~)
// need six scalar transforms per (Q, U) pair: [Q, r1 Q, r2 Q, U, r1 U, r2 U] - three times the scalar cost per map
// where a dedicated two-operand kernel (as on the synthesis side) would need two, and no new MFMA kernel.
#include "sht_internal.h"

int sht_ensure_polc(corahip_ctx *ctx, corahip_sht_plan *p);

// maps_out[6 f + 3 s + k][pix] = (1, r1, r2)[k] * maps_in[2 f + s][pix], s = 0 (Q), 1 (U); block = (ring, input channel)
__global__ void __launch_bounds__(256)
spin2_ring_scale_kernel(const double *__restrict__ in, long npix, int nring, const int64_t *__restrict__ start,
                        const int32_t *__restrict__ nphi, const double *__restrict__ z, double *__restrict__ out) {
    const int ring = blockIdx.x, ch = blockIdx.y;
    const int rn = min(ring, nring - 1 - ring);
    const double zz = ring == rn ? z[rn] : -z[rn];
    const double r1 = 1.0 / ((1.0 - zz) * (1.0 + zz)), r2 = zz * r1;
    const double *src = in + (size_t)ch * npix + start[ring];
    const int f = ch >> 1, s = ch & 1;
    double *dst = out + (size_t)(6 * f + 3 * s) * npix + start[ring];
    for (int j = threadIdx.x; j < nphi[ring]; j += blockDim.x) {
        const double v = src[j];
        dst[j] = v;
        dst[(size_t)npix + j] = r1 * v;
        dst[2 * (size_t)npix + j] = r2 * v;
    }
}

__device__ static inline double2 ld_alm(const double *__restrict__ a, long idx, int G, int ch) {
    const double *p = a + ((size_t)idx * G + (ch >> 2)) * 8 + (ch & 3);
    return make_double2(p[0], p[4]);
}

// (E, B)_f at every (l, m) from the six scalar transforms of field f; alm layouts [nalm][G][c][v]
__global__ void spin2_combine_kernel(const double *__restrict__ a6, int G6, const double *__restrict__ polc, int lmax,
                                     int nf, int Gout, double *__restrict__ out) {
    const long nalm = nalm_of(lmax);
    const long total = nalm * nf;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long idx = q / nf;
        const int f = (int)(q - idx * nf);
        // (l, m) of the packed index: m-major, idx = m (2 lmax + 1 - m) / 2 + l
        int m = (int)(((2.0 * lmax + 3.0) - sqrt((2.0 * lmax + 3.0) * (2.0 * lmax + 3.0) - 8.0 * (double)idx)) * 0.5);
        m = max(0, min(m, lmax));
        while (m > 0 && alm_idx(m, m, lmax) > idx) m--;
        while (m < lmax && alm_idx(m + 1, m + 1, lmax) <= idx) m++;
        const int l = m + (int)(idx - alm_idx(m, m, lmax));
        double2 E = make_double2(0.0, 0.0), B = make_double2(0.0, 0.0);
        if (l >= 2) {
            const double *g = polc + (size_t)idx * 4;
            const double g1 = g[0], g2 = g[1], g3 = g[2], g4 = g[3], mg3 = (double)m * g3;
            const bool lower = l - 1 >= m;
            double2 sw[2], sx[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int c0 = 6 * f + 3 * s;
                const double2 t0 = ld_alm(a6, idx, G6, c0), t1 = ld_alm(a6, idx, G6, c0 + 1), t2 = ld_alm(a6, idx, G6, c0 + 2);
                double2 p1 = make_double2(0.0, 0.0), p2 = p1;
                if (lower) p1 = ld_alm(a6, idx - 1, G6, c0 + 1), p2 = ld_alm(a6, idx - 1, G6, c0 + 2);
                sw[s] = make_double2(g1 * t1.x + g2 * t0.x + g3 * p2.x, g1 * t1.y + g2 * t0.y + g3 * p2.y);
                sx[s] = make_double2(g4 * t2.x - mg3 * p1.x, g4 * t2.y - mg3 * p1.y);
            }
            // E = -(SW_Q - i SX_U),  B = -(SW_U + i SX_Q)
            E = make_double2(-(sw[0].x + sx[1].y), -(sw[0].y - sx[1].x));
            B = make_double2(-(sw[1].x - sx[0].y), -(sw[1].y + sx[0].x));
        }
        double *oe = out + ((size_t)idx * Gout + ((2 * f) >> 2)) * 8 + ((2 * f) & 3);
        oe[0] = E.x, oe[4] = E.y;
        oe[1] = B.x, oe[5] = B.y;     // channel 2 f + 1 sits next to 2 f in the same cell
    }
}

extern "C" {

int corahip_spin2_ring_scale(corahip_ctx *ctx, corahip_sht_plan *plan, const double *maps_qu, int nfields,
                             double *maps6) {
    ARG_CHECK(ctx && plan && maps_qu && maps6 && nfields >= 1 && 2 * nfields <= 65535);
    StageTimer t(ctx, "spin2_scale");
    dim3 grid((unsigned)plan->nring, (unsigned)(2 * nfields));
    hipLaunchKernelGGL(spin2_ring_scale_kernel, grid, dim3(256), 0, ctx->stream, maps_qu, plan->npix, plan->nring,
                       plan->d_start, plan->d_nphi, plan->d_z, maps6);
    LAUNCH_CHECK();
    return 0;
}

int corahip_spin2_combine(corahip_ctx *ctx, corahip_sht_plan *plan, const double *alm6_dev, int g6, int nfields,
                          double *alm_eb_dev, int gout) {
    ARG_CHECK(ctx && plan && alm6_dev && alm_eb_dev && nfields >= 1);
    ARG_CHECK(4 * g6 >= 6 * nfields && 4 * gout >= 2 * nfields);
    int rc = sht_ensure_polc(ctx, plan);
    if (rc) return rc;
    StageTimer t(ctx, "spin2_combine");
    const int G6 = g6, Gout = gout;
    HIP_TRY(hipMemsetAsync(alm_eb_dev, 0, sizeof(double) * (size_t)plan->nalm * Gout * 8, ctx->stream));
    const long total = plan->nalm * nfields;
    long blocks = (total + 255) / 256;
    const long cap = (long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(spin2_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, alm6_dev, G6, plan->d_polc,
                       plan->lmax, nfields, Gout, alm_eb_dev);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
