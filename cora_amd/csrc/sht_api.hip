// sht_api.hip - C ABI entry points of the HEALPix transforms: workspace sizes, alm2map (with channel chunking),
// map2alm (one weighted quadrature pass) and the spin-2 synthesis.  See include/corahip.h and sht_internal.h.
#include "sht_internal.h"

extern "C" int corahip_alm2map_workspace_bytes(const corahip_sht_plan *p, int nnu, size_t *bytes) {
    ARG_CHECK(p != nullptr && bytes != nullptr && nnu >= 1);
    const size_t G = nnu_pad_of(nnu) / 4;
    *bytes = (size_t)p->nring * G * p->L * 8 * sizeof(double) + K5_TAIL_PAD;
    // an odd number of 4-channel groups cannot be consumed in place (K4 tiles are 16 columns
    // = 2 groups wide): the padded copy of the alm block lives in the workspace too
    if (((nnu + 3) / 4) & 1) *bytes += (size_t)p->nalm * G * 8 * sizeof(double);
    return 0;
}
// alm_chunk: [nalm][ncols] with ncols = 2*nnu_pad (multiple of 16)
static int alm2map_chunk(corahip_ctx *ctx, const corahip_sht_plan *p, const double *alm_chunk, int nnu_chunk_pad,
                         int nnu_valid, double *maps, double *inter) {
    int rc = sht_legendre(ctx, p, 2 * nnu_chunk_pad, alm_chunk, inter);
    if (rc) return rc;
    return sht_ringfft(ctx, p, inter, nnu_chunk_pad, nnu_valid, maps);
}

// gather channel groups [g0, g0+Gc) of alm_dev ([nalm][Gsrc][8]) into a dense [nalm][Gc][8] chunk;
// groups past the source (padding) are zero-filled
__global__ void alm_slice_kernel(const double *__restrict__ src, double *__restrict__ dst, long nalm, int Gsrc,
                                 int g0, int Gc) {
    const long n = nalm * Gc * 4;  // double2 items (4 per group cell)
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        const long idx = q / (Gc * 4);
        const int w = (int)(q % (Gc * 4));
        const int gsrc = g0 + (w >> 2);
        double2 v = make_double2(0.0, 0.0);
        if (gsrc < Gsrc) v = *reinterpret_cast<const double2 *>(src + ((size_t)idx * Gsrc + g0) * 8 + 2 * w);
        *reinterpret_cast<double2 *>(dst + (size_t)idx * Gc * 8 + 2 * w) = v;
    }
}

extern "C" int corahip_alm2map(corahip_ctx *ctx, const corahip_sht_plan *p, const double *alm_dev, int nnu,
                               double *maps, void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && p != nullptr && alm_dev != nullptr && maps != nullptr && workspace != nullptr);
    ARG_CHECK(nnu >= 1);
    const int nnu_pad4 = (nnu + 3) & ~3;  // alm_dev layout granularity
    const int Gsrc = nnu_pad4 / 4;
    const int nnu_pad8 = nnu_pad_of(nnu);
    size_t need_full;
    corahip_alm2map_workspace_bytes(p, nnu, &need_full);
    if (nnu_pad4 == nnu_pad8 && workspace_bytes >= need_full) {
        // single pass straight from alm_dev
        return alm2map_chunk(ctx, p, alm_dev, nnu_pad8, nnu, maps, (double *)workspace);
    }
    // chunked: workspace holds [inter for chunk][alm slice for chunk]
    const size_t per8_inter = (size_t)p->nring * 2 * p->L * 8 * sizeof(double);
    const size_t per8_alm = (size_t)p->nalm * 16 * sizeof(double);
    // K5's register prefetch over-reads up to K5_TAIL_PAD bytes behind the last F_m row: those bytes are the alm
    // slice and, when the slice is smaller than the pad (small plans), the reserve kept free at the end
    const size_t usable = workspace_bytes > K5_TAIL_PAD ? workspace_bytes - K5_TAIL_PAD : 0;
    int nchunk8 = (int)(usable / (per8_inter + per8_alm));
    if (nchunk8 < 1) {
        corahip_set_error("alm2map workspace too small: %zu bytes, need at least %zu", workspace_bytes,
                          per8_inter + per8_alm + K5_TAIL_PAD);
        return CORAHIP_ENOMEM;
    }
    // prefer chunks that are multiples of 64 channels (NT = 8 tiles)
    if (nchunk8 >= 8) nchunk8 &= ~7;
    const int chunk = nchunk8 * 8;
    double *inter = (double *)workspace;
    double *slice = (double *)((char *)workspace + (size_t)nchunk8 * per8_inter);
    for (int nu0 = 0; nu0 < nnu; nu0 += chunk) {
        const int nvalid = std::min(chunk, nnu - nu0);
        const int cpad8 = nnu_pad_of(nvalid);
        const int Gc = cpad8 / 4;
        alm_slice_kernel<<<2048, 256, 0, ctx->stream>>>(alm_dev, slice, p->nalm, Gsrc, nu0 / 4, Gc);
        LAUNCH_CHECK();
        int rc = alm2map_chunk(ctx, p, slice, cpad8, nvalid, maps + (size_t)nu0 * p->npix, inter);
        if (rc) return rc;
    }
    return 0;
}
// ------------------------------------------------------------------------------------
// analysis host side
// ------------------------------------------------------------------------------------
extern "C" int corahip_map2alm_workspace_bytes(const corahip_sht_plan *p, int nnu, size_t *bytes) {
    ARG_CHECK(p != nullptr && bytes != nullptr && nnu >= 1);
    const size_t G = nnu_pad_of(nnu) / 4;
    const int ntile = (p->npair + 64 * ADJ_WAVES - 1) / (64 * ADJ_WAVES);
    *bytes = (size_t)p->nring * G * p->L * 8 * sizeof(double)            // G_m cells (the `inter` layout)
             + (size_t)ntile * p->nalm * G * 8 * sizeof(double);         // per-ring-tile partial a_lm
    return 0;
}
// maps [nnu][npix] RING -> alm_dev [nalm][nnu_pad8/4][2][4]: ONE weighted quadrature pass (no iteration),
//   a_lm = sum_pix w_ring(pix) (4 pi / npix) map(pix) conj(Y_lm(pix)).
// ring_w: device [2 nside] weights of the north rings incl. equator (mirrored to the south), or NULL = 1.
extern "C" int corahip_map2alm(corahip_ctx *ctx, const corahip_sht_plan *p, const double *maps, int nnu,
                               const double *ring_w, double *alm_dev, void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && p != nullptr && maps != nullptr && alm_dev != nullptr && workspace != nullptr);
    ARG_CHECK(nnu >= 1);
    size_t need;
    corahip_map2alm_workspace_bytes(p, nnu, &need);
    if (workspace_bytes < need) {
        corahip_set_error("map2alm workspace too small: %zu bytes, need %zu (process fewer channels per call)",
                          workspace_bytes, need);
        return CORAHIP_ENOMEM;
    }
    const int nnu_pad8 = nnu_pad_of(nnu);
    const int G = nnu_pad8 / 4;
    double *inter = (double *)workspace;
    double *part = inter + (size_t)p->nring * G * p->L * 8;
    int rc = sht_ringana(ctx, p, maps, nnu, nnu_pad8, ring_w, inter);
    if (rc) return rc;
    return sht_legendre_adj(ctx, p, 2 * nnu_pad8, inter, part, alm_dev);
}

// alm_dev: nnu = 2 nfreq channels interleaved (E_0, B_0, E_1, B_1, ...) -> maps [nnu, npix] = (Q_0, U_0, Q_1, U_1, ...)
extern "C" int corahip_alm2map_spin2(corahip_ctx *ctx, corahip_sht_plan *p, const double *alm_dev, int nnu, double *maps,
                                     void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && p != nullptr && alm_dev != nullptr && maps != nullptr && workspace != nullptr);
    ARG_CHECK(nnu >= 2 && (nnu & 1) == 0);
    ARG_CHECK(((nnu + 3) & ~3) == nnu_pad_of(nnu));     // the alm_dev group count must be the 8-padded one
    size_t need;
    corahip_alm2map_workspace_bytes(p, nnu, &need);
    if (workspace_bytes < need) {
        corahip_set_error("alm2map_spin2 workspace too small: %zu bytes, need %zu (process fewer channels per call)",
                          workspace_bytes, need);
        return CORAHIP_ENOMEM;
    }
    const int nnu_pad8 = nnu_pad_of(nnu);
    double *inter = (double *)workspace;
    int rc = sht_legendre_pol(ctx, p, 2 * nnu_pad8, alm_dev, inter);
    if (rc) return rc;
    return sht_ringfft(ctx, p, inter, nnu_pad8, nnu, maps);
}
