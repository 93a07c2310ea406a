// drawstream.hip - K3 fed with numpy's OWN normal stream, generated on the device one range of multipoles at a time.
//
// The reference draws inside its l loop (cora/core/skysim.py:114-121: for every l `complex_std_normal((numz, l + 1), rng)`,
// cora/util/nputil.py:121-125, then `np.dot(trans, gaussvars)`): at no time do more than 2 F (l + 1) normals exist.
// Rounds 1-4 of this library materialised the whole stream of a realisation first (16 F nalm bytes: 8.6 GB at cfg 3,
// 137 GB at cfg 5 - more than the rest of a rank's working set).  Here the stream is produced the way it is consumed:
//
//   prepare   the generator's count + scan passes over the WHOLE stream (they are what solves the prefix problem of the
//             ziggurat / the compaction of the polar method, npnormal.hip / mtlegacy.hip) - tables of ~0.3 bytes per normal
//   ranges    the multipoles are cut into ranges of <= slot bytes of normals (whole l, at least one); the emit pass of
//             range r writes slot r % 2 of a two-slot ring on the generator stream while K3 (draw.hip, the persistent
//             MFMA kernel reading its A operands from the slot) consumes range r - 1 on the context's stream; two events
//             per slot order "emitted -> drawn -> refilled"
//   finish    the generator state numpy would be left in: ONE read-back, which the caller places where it likes
//             (corahip_draw_alm_numpy_end) - behind the launches of the synthesis, so that the host never waits with an
//             empty queue in the middle of a step
//
// Every rank of a frequency-sharded job runs this for its own rows of the factors (rows = 1: T is the row block
// [L, nnu, F] the all-to-all delivered, no zero-padded [L, F, F] stack).
#include "stream_internal.h"

#include <algorithm>
#include <cstdlib>

namespace {

int ring_setup(corahip_ctx *ctx) {
    if (!ctx->gen_stream) {
        // highest priority: when the generator runs beside the kernels that make the factors (corahip_draw_alm_numpy_prepare)
        // its small latency-bound launches - the jump tree of the legacy stream - get the CUs that come free first
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        static const bool flat = getenv("CORAHIP_GEN_PRIO_OFF") != nullptr;     // A/B: default priority
        HIP_TRY(hipStreamCreateWithPriority(&ctx->gen_stream, hipStreamNonBlocking, flat ? 0 : prio_hi));
        for (auto &e : ctx->ev_ring) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return 0;
}

}  // namespace

struct corahip_draw_pending {
    zig_session *zs = nullptr;
    mt_session *ms = nullptr;
    // the ranges of the session (corahip_draw_alm_numpy_prepare): range r = multipoles l_first[r] .. l_first[r + 1] - 1 =
    // stream elements bounds[r] .. bounds[r + 1]; `emitted` ranges are already in their ring slots (or on their way)
    std::vector<int> l_first;
    std::vector<unsigned long long> bounds;
    unsigned long long slot_elems = 0;
    double *ring = nullptr;
    int lmax = 0, F = 0, emitted = 0;
    bool ran = false;
};

static void pending_free(corahip_draw_pending *p) {
    if (!p) return;
    if (p->zs) zig_stream_free(p->zs);
    if (p->ms) mt_stream_free(p->ms);
    delete p;
}

#define DS_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            corahip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            (void)hipStreamSynchronize(B);                                                                   \
            (void)hipStreamSynchronize(A);                                                                   \
            if (own) {                                                                                       \
                ctx->draw_pending = nullptr;                                                                 \
                pending_free(pd);                                                                            \
            }                                                                                                \
            return (int)_e;                                                                                  \
        }                                                                                                    \
    } while (0)

// emit range r into its ring slot on the generator stream (behind the draw of the range that held the slot before)
static int emit_range(corahip_ctx *ctx, corahip_draw_pending *pd, int r, bool own) {
    hipStream_t A = ctx->stream, B = ctx->gen_stream;
    hipEvent_t ev_emit[2] = {ctx->ev_ring[1], ctx->ev_ring[2]}, ev_drawn[2] = {ctx->ev_ring[3], ctx->ev_ring[4]};
    const int sl = r & 1;
    double *slot = pd->ring + (size_t)sl * pd->slot_elems;
    if (r >= 2) DS_TRY(hipStreamWaitEvent(B, ev_drawn[sl], 0));       // the slot's previous range has been drawn
    const int rc = pd->zs ? zig_stream_emit_range(ctx, B, pd->zs, r, slot) : mt_stream_emit_range(ctx, B, pd->ms, r, slot);
    if (rc) return rc;
    DS_TRY(hipEventRecord(ev_emit[sl], B));
    return 0;
}

// The generator's part of a draw that does not depend on the factors: ranges, ring, the generator's own prepare passes
// (count + scan / jump tree + count) and the emit passes of the first two ranges (both ring slots are free) - all on the
// generator stream.  A caller that issues this BEFORE the launches that make the factors (K1, K2) lets the generator run
// beside them; the stream is a function of the generator alone (cora/util/nputil.py:121-125 draws it inside mkfullsky,
// but nothing it draws depends on the covariance).
static int draw_numpy_prepare(corahip_ctx *ctx, const corahip_rng *rng, int lmax, int F, size_t ring_bytes,
                              corahip_draw_pending **pending) {
    ARG_CHECK(ctx != nullptr && rng != nullptr && pending != nullptr);
    ARG_CHECK(lmax >= 0 && F >= 1);
    ARG_CHECK(rng->kind == CORAHIP_RNG_PCG64 || rng->kind == CORAHIP_RNG_MT19937);
    ARG_CHECK(rng->kind != CORAHIP_RNG_MT19937 || rng->legacy != nullptr);
    *pending = nullptr;
    if (ctx->draw_pending) {
        corahip_set_error("draw_alm_numpy: the previous session has not been ended (corahip_draw_alm_numpy_end): its "
                          "generator tables and ring are still in use");
        return CORAHIP_ESTATE;
    }
    int rc = ring_setup(ctx);
    if (rc) return rc;
    const unsigned long long total = (unsigned long long)F * (lmax + 1) * (lmax + 2);
    if (ring_bytes == 0) {
        // Default (measured at cfg 3, tools/seeded_probe.py, whole step): every range is an emit launch, an event hop and a
        // K3 launch with its own ramp and tail, and the emit pass cannot run BESIDE K3 (223 VGPRs x 2 waves per SIMD leave
        // no room for its 107) - the draw pipeline alone takes 12.65 ms as one range, 12.97 / 13.31 / 14.42 ms with a ring
        // of 2 GB / 1 GB / 512 MB.  But the kernel BEHIND the draw pays for the footprint of the normals: K4's first run
        // after a draw that streamed through 8.6 GB takes 53.5 ms, after a 2 GB ring 52.4, after 512 MB 51.8 (its usual
        // time).  The step is shortest with a ring of 2 GiB (88.3 ms; 89.1 as one range, 89.2 at 512 MB): that is the
        // default while the stream is a small part of the device memory; a stream beyond 1/8 of it (cfg 5: 137 GB, where
        // every range carries 4x the MFMA work per byte) goes through a ring of 1/16 of the memory.  CORAHIP_RING_MB overrides.
        const char *e = getenv("CORAHIP_RING_MB");
        if (e) ring_bytes = (size_t)std::max(1L, atol(e)) << 20;
        else if (8 * total <= ctx->total_mem / 8) ring_bytes = (size_t)2 << 30;
        else ring_bytes = std::max<size_t>(ctx->total_mem / 16, (size_t)2 << 30);
    }
    corahip_draw_pending *pd = new corahip_draw_pending();
    pd->lmax = lmax;
    pd->F = F;
    // ranges of whole multipoles: l contributes 2 F (l + 1) normals; a slot holds at least the largest l
    const unsigned long long per_lmax = 2ull * F * (lmax + 1);
    pd->slot_elems = std::min(total, std::max<unsigned long long>(ring_bytes / 16, per_lmax));
    {
        unsigned long long fill = 0;
        for (int l = 0; l <= lmax; l++) {
            const unsigned long long nl = 2ull * F * (l + 1);
            if (l == 0 || fill + nl > pd->slot_elems) {
                pd->l_first.push_back(l);
                pd->bounds.push_back((unsigned long long)F * l * (l + 1));
                fill = 0;
            }
            fill += nl;
        }
        pd->bounds.push_back(total);
        pd->l_first.push_back(lmax + 1);
    }
    const int nr = (int)pd->bounds.size() - 1;
    const int64_t n = (int64_t)total;
    if ((rc = corahip_ctx_scratch(ctx, 7, sizeof(double) * (nr > 1 ? 2 : 1) * (size_t)pd->slot_elems, (void **)&pd->ring))) {
        pending_free(pd);
        return rc;
    }
    hipStream_t A = ctx->stream, B = ctx->gen_stream;
    const bool own = true;
    // The generator stream never starts before the context's stream has reached this call (the ring and the generator
    // tables may still be read by a draw queued there: a caller that pipelines realisations without synchronising)
    ctx->draw_pending = pd;
    DS_TRY(hipEventRecord(ctx->ev_ring[0], A));
    DS_TRY(hipStreamWaitEvent(B, ctx->ev_ring[0], 0));
    if (rng->kind == CORAHIP_RNG_PCG64) rc = zig_stream_prepare(ctx, B, rng->state, rng->inc, n, pd->bounds, &pd->zs);
    else {
        corahip_mt_state st0 = *rng->legacy;       // (prepare reads it; the caller's copy is rewritten by _end only)
        rc = mt_stream_prepare(ctx, B, &st0, n, pd->bounds, &pd->ms);
    }
    // both ring slots are free: the first two ranges are emitted right away (they too run beside whatever the caller
    // enqueues on the context's stream before it hands the factors over)
    for (int r = 0; r < std::min(nr, 2) && !rc; r++) {
        rc = emit_range(ctx, pd, r, own);
        if (!rc) pd->emitted = r + 1;
    }
    if (rc) {
        (void)hipStreamSynchronize(B);
        ctx->draw_pending = nullptr;
        pending_free(pd);
        return rc;
    }
    *pending = pd;
    return 0;
}

// K3 of every range against the factors, on the context's stream, the remaining emit passes on the generator stream
static int draw_numpy_run(corahip_ctx *ctx, corahip_draw_pending *pd, const double *T, int rows, const int32_t *info,
                          const corahip_chanset *set, double *alm_dev, bool own) {
    ARG_CHECK(ctx != nullptr && pd != nullptr && T != nullptr && alm_dev != nullptr && set != nullptr);
    ARG_CHECK(ctx->draw_pending == pd && !pd->ran);
    hipStream_t A = ctx->stream, B = ctx->gen_stream;
    hipEvent_t ev_emit[2] = {ctx->ev_ring[1], ctx->ev_ring[2]}, ev_drawn[2] = {ctx->ev_ring[3], ctx->ev_ring[4]};
    const int nr = (int)pd->bounds.size() - 1;
    int rc = 0;
    pd->ran = true;
    {
        StageTimer t(ctx, "draw");
        for (int r = 0; r < nr && !rc; r++) {
            const int sl = r & 1;
            double *slot = pd->ring + (size_t)sl * pd->slot_elems;
            if (r >= pd->emitted) {
                rc = emit_range(ctx, pd, r, own);
                if (rc) break;
                pd->emitted = r + 1;
            }
            DS_TRY(hipStreamWaitEvent(A, ev_emit[sl], 0));
            rc = corahip_draw_range(ctx, A, T, rows, info, slot, (size_t)pd->bounds[r], pd->l_first[r], pd->l_first[r + 1] - 1, pd->lmax,
                                    pd->F, set, alm_dev);
            if (rc) break;
            DS_TRY(hipEventRecord(ev_drawn[sl], A));
        }
    }
    if (rc) {
        (void)hipStreamSynchronize(B);
        (void)hipStreamSynchronize(A);
        if (own) {
            ctx->draw_pending = nullptr;
            pending_free(pd);
        }
        return rc;
    }
    // (the last draw waited for the last emit: everything of the generator stream is behind the context stream's tail)
    return 0;
}
#undef DS_TRY

static int draw_numpy_begin(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, const corahip_rng *rng, int lmax,
                            int F, const corahip_chanset *set, double *alm_dev, size_t ring_bytes,
                            corahip_draw_pending **pending) {
    ARG_CHECK(ctx != nullptr && T != nullptr && rng != nullptr && alm_dev != nullptr && pending != nullptr && set != nullptr);
    corahip_draw_pending *pd = nullptr;
    int rc = draw_numpy_prepare(ctx, rng, lmax, F, ring_bytes, &pd);
    if (rc) return rc;
    *pending = nullptr;
    if ((rc = draw_numpy_run(ctx, pd, T, rows, info, set, alm_dev, true))) return rc;
    *pending = pd;
    return 0;
}

extern "C" {

int corahip_draw_alm_numpy_begin(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, const corahip_rng *rng,
                                 int lmax, int F, int nu0, int nnu, double *alm_dev, size_t ring_bytes,
                                 corahip_draw_pending **pending) {
    ARG_CHECK(nu0 >= 0 && nnu >= 1 && nu0 + nnu <= F);
    const corahip_chanset set = {1, nnu, {nu0, 0}};
    return draw_numpy_begin(ctx, T, rows, info, rng, lmax, F, &set, alm_dev, ring_bytes, pending);
}

int corahip_draw_alm_numpy_begin_set(corahip_ctx *ctx, const double *T_rows, const int32_t *info, const corahip_rng *rng,
                                     int lmax, int F, const corahip_chanset *set, double *alm_dev, size_t ring_bytes,
                                     corahip_draw_pending **pending) {
    return draw_numpy_begin(ctx, T_rows, 1, info, rng, lmax, F, set, alm_dev, ring_bytes, pending);
}

int corahip_draw_alm_numpy_prepare(corahip_ctx *ctx, const corahip_rng *rng, int lmax, int F, size_t ring_bytes,
                                   corahip_draw_pending **pending) {
    return draw_numpy_prepare(ctx, rng, lmax, F, ring_bytes, pending);
}

int corahip_draw_alm_numpy_run(corahip_ctx *ctx, corahip_draw_pending *pending, const double *T, int rows, const int32_t *info,
                               const corahip_chanset *set, double *alm_dev) {
    ARG_CHECK(set != nullptr && pending != nullptr);
    // (a failure leaves the session to the caller: corahip_draw_alm_numpy_end frees it)
    return draw_numpy_run(ctx, pending, T, rows, info, set, alm_dev, false);
}

int corahip_draw_alm_numpy_end(corahip_ctx *ctx, corahip_draw_pending *pd, corahip_rng *rng) {
    ARG_CHECK(ctx != nullptr && pd != nullptr && rng != nullptr);
    ARG_CHECK(ctx->draw_pending == pd);
    int rc;
    if (!pd->ran) {
        // a prepared session that is given up (the factors could not be made): nothing was drawn, the generator stays
        (void)hipStreamSynchronize(ctx->gen_stream);
        ctx->draw_pending = nullptr;
        pending_free(pd);
        return 0;
    }
    if (pd->zs) {
        uint64_t n_raw = 0, after[2];
        rc = zig_stream_finish(ctx, ctx->stream, pd->zs, &n_raw);
        if (!rc) rc = corahip_pcg64_advance(rng->state, rng->inc, n_raw, after);
        if (!rc) {
            rng->state[0] = after[0];              // the generator as numpy would leave it
            rng->state[1] = after[1];
        }
    } else {
        rc = rng->legacy ? mt_stream_finish(ctx, ctx->stream, pd->ms, rng->legacy) : CORAHIP_EINVAL;
    }
    if (rc) (void)hipStreamSynchronize(ctx->gen_stream);
    ctx->draw_pending = nullptr;
    pending_free(pd);
    return rc;
}

int corahip_draw_alm_numpy(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, corahip_rng *rng, int lmax, int F,
                           int nu0, int nnu, double *alm_dev, size_t ring_bytes) {
    corahip_draw_pending *pd = nullptr;
    int rc = corahip_draw_alm_numpy_begin(ctx, T, rows, info, rng, lmax, F, nu0, nnu, alm_dev, ring_bytes, &pd);
    if (rc) return rc;
    return corahip_draw_alm_numpy_end(ctx, pd, rng);
}

}  // extern "C"
