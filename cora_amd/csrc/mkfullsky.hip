// mkfullsky.hip - the fused convenience entry point of the drop-in boundary: C_l(nu,nu') -> sky maps in ONE call.
//
// Replaces the body of skysim.mkfullsky (cora/core/skysim.py:72-136) for a single process: per-l jitter + root
// (:115-119), the complex normals of nputil.complex_std_normal (:120; cora/util/nputil.py:104-125), a_lm = T_l g_l
// (:121) and hputil.sphtrans_inv_sky (:130; cora/util/hputil.py:500-531) - by chaining the library's own entry points
// (corahip_factor_batched, corahip_draw_alm* / corahip_draw_alm_numpy, corahip_alm2map / corahip_alm_dev_to_square) on
// buffers cut from one caller-owned workspace.  Nothing here that a caller could not do with those calls.
#include "sht_internal.h"

#include <algorithm>

namespace {
struct mk_layout {
    size_t off_T, off_info, off_alm, off_sht, total_min, total_full;
};
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

int layout_of(const corahip_sht_plan *p, int F, int nnu, int rng_kind, int alms, mk_layout &lo) {
    const size_t L = p->L, nalm = p->nalm;
    const size_t G = ((size_t)nnu + 3) / 4;
    lo.off_T = 0;
    lo.off_info = up256(sizeof(double) * L * F * F);
    lo.off_alm = lo.off_info + up256(sizeof(int32_t) * L);
    // (the PCG64 / MT19937 kinds keep no normal buffer here: corahip_draw_alm_numpy generates the stream range by range
    //  into the library's own ring)
    (void)rng_kind;
    lo.off_sht = lo.off_alm + up256(sizeof(double) * nalm * G * 8);
    size_t full = 0;
    if (!alms) {
        int rc = corahip_alm2map_workspace_bytes(p, nnu, &full);
        if (rc) return rc;
    }
    lo.total_full = lo.off_sht + full;
    // smallest synthesis workspace corahip_alm2map accepts (it then works through the channels in chunks of 8)
    const size_t chunk8 = alms ? 0 : (size_t)p->nring * 2 * p->L * 8 * sizeof(double) + (size_t)p->nalm * 16 * sizeof(double) + K5_TAIL_PAD;
    lo.total_min = std::min(lo.total_full, lo.off_sht + chunk8);
    return 0;
}
}  // namespace

extern "C" {

int corahip_mkfullsky_workspace_bytes(const corahip_sht_plan *plan, int F, int nu0, int nnu, int rng_kind, int alms,
                                      size_t *bytes) {
    ARG_CHECK(plan != nullptr && bytes != nullptr && F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    ARG_CHECK(rng_kind == CORAHIP_RNG_STREAM || rng_kind == CORAHIP_RNG_PHILOX || rng_kind == CORAHIP_RNG_PCG64 ||
              rng_kind == CORAHIP_RNG_MT19937);
    mk_layout lo;
    int rc = layout_of(plan, F, nnu, rng_kind, alms, lo);
    if (rc) return rc;
    *bytes = lo.total_full;
    return 0;
}

int corahip_mkfullsky(corahip_ctx *ctx, const corahip_sht_plan *plan, const double *C, int F, corahip_rng *rng, int nu0,
                      int nnu, int alms, double *out, void *workspace, size_t workspace_bytes) {
    ARG_CHECK(ctx != nullptr && plan != nullptr && C != nullptr && rng != nullptr && out != nullptr && workspace != nullptr);
    ARG_CHECK(F >= 1 && nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    ARG_CHECK(rng->kind == CORAHIP_RNG_STREAM || rng->kind == CORAHIP_RNG_PHILOX || rng->kind == CORAHIP_RNG_PCG64 ||
              rng->kind == CORAHIP_RNG_MT19937);
    ARG_CHECK(rng->kind != CORAHIP_RNG_STREAM || rng->stream != nullptr);
    ARG_CHECK(rng->kind != CORAHIP_RNG_MT19937 || rng->legacy != nullptr);
    mk_layout lo;
    int rc = layout_of(plan, F, nnu, rng->kind, alms, lo);
    if (rc) return rc;
    if (workspace_bytes < lo.total_min) {
        corahip_set_error("mkfullsky workspace too small: %zu bytes, need at least %zu (%zu for one synthesis pass)", workspace_bytes,
                          lo.total_min, lo.total_full);
        return CORAHIP_ENOMEM;
    }
    char *ws = (char *)workspace;
    double *T = (double *)(ws + lo.off_T);
    int32_t *info = (int32_t *)(ws + lo.off_info);
    double *alm = (double *)(ws + lo.off_alm);
    const int lmax = plan->lmax, L = plan->L;
    // numpy's own stream depends on the generator alone: its device passes (and the first two ranges of normals) are
    // enqueued on the generator stream BEFORE the factorisation and run beside it (round 6)
    corahip_draw_pending *pending = nullptr;
    const bool numpy_stream = rng->kind == CORAHIP_RNG_PCG64 || rng->kind == CORAHIP_RNG_MT19937;
    if (numpy_stream && (rc = corahip_draw_alm_numpy_prepare(ctx, rng, lmax, F, 0, &pending))) return rc;
    // skysim.py:115-119: C_l + I max(diag) 1e-14 -> Cholesky, eigen root where that fails (nputil.py:51-101, threshold 1e-16)
    if ((rc = corahip_factor_batched(ctx, C, L, F, 1e-14, 1e-16, T, info))) {
        if (pending) (void)corahip_draw_alm_numpy_end(ctx, pending, rng);     // (given up: the generator stays as it was)
        return rc;
    }
    // skysim.py:120-121
    if (rng->kind == CORAHIP_RNG_PHILOX) {
        rc = corahip_draw_alm_philox(ctx, T, info, rng->seed, lmax, F, nu0, nnu, alm);
    } else if (rng->kind == CORAHIP_RNG_STREAM) {
        rc = corahip_draw_alm(ctx, T, info, rng->stream, lmax, F, nu0, nnu, alm);
    } else {
        // numpy's own stream (PCG64 + ziggurat, or the legacy MT19937 + polar method), continued on the device range by
        // range: enqueued here, the generator state read back (rng->state / rng->legacy left where numpy would leave
        // them) once the synthesis has been enqueued behind it
        const corahip_chanset set = {1, nnu, {nu0, 0}};
        rc = corahip_draw_alm_numpy_run(ctx, pending, T, 0, info, &set, alm);
    }
    if (rc) {
        if (pending) (void)corahip_draw_alm_numpy_end(ctx, pending, rng);     // (frees the session)
        return rc;
    }
    // skysim.py:123-130
    if (alms) rc = corahip_alm_dev_to_square(ctx, alm, lmax, nnu, out);
    else rc = corahip_alm2map(ctx, plan, alm, nnu, out, ws + lo.off_sht, workspace_bytes - lo.off_sht);
    if (pending) {
        const int rc2 = corahip_draw_alm_numpy_end(ctx, pending, rng);
        if (!rc) rc = rc2;
    }
    return rc;
}

}  // extern "C"
