// ctx.hip - context, error string, timers, device memory helpers of libcorahip.so
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";

void corahip_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int corahip_ctx_scratch(corahip_ctx *ctx, int slot, size_t bytes, void **out) {
    if (ctx->scratch_bytes[slot] < bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch[slot]) HIP_TRY(hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
        HIP_TRY(hipMalloc(&ctx->scratch[slot], bytes));
        ctx->scratch_bytes[slot] = bytes;
    }
    *out = ctx->scratch[slot];
    return 0;
}

extern "C" {

int corahip_abi_version(void) { return CORAHIP_ABI_VERSION; }

int corahip_abi_minor(void) { return CORAHIP_ABI_MINOR; }

const char *corahip_last_error(void) { return g_err; }

int corahip_device_count(int *count) {
    ARG_CHECK(count != nullptr);
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

int corahip_ctx_create(int device_id, corahip_ctx **out) {
    ARG_CHECK(out != nullptr);
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    ARG_CHECK(device_id >= 0 && device_id < n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        corahip_set_error("device %d is %s; libcorahip is built for gfx950 (MI355X) only", device_id,
                          prop.gcnArchName);
        return CORAHIP_ESTATE;
    }
    corahip_ctx *c = new corahip_ctx();
    c->device = device_id;
    c->num_cu = prop.multiProcessorCount;
    c->total_mem = prop.totalGlobalMem;
    HIP_TRY(hipEventCreate(&c->t0));
    HIP_TRY(hipEventCreate(&c->t1));
    *out = c;
    return 0;
}

int corahip_ctx_destroy(corahip_ctx *ctx) {
    if (!ctx) return 0;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending) {
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    (void)hipEventDestroy(ctx->t0);
    (void)hipEventDestroy(ctx->t1);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
        (void)hipEventDestroy(ctx->ev_fork);
        (void)hipEventDestroy(ctx->ev_join);
    }
    if (ctx->gen_stream) {
        (void)hipStreamSynchronize(ctx->gen_stream);
        (void)hipStreamDestroy(ctx->gen_stream);
        for (auto &e : ctx->ev_ring)
            if (e) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < CORAHIP_NSCRATCH; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->draw_slot_tab) (void)hipFree(ctx->draw_slot_tab);
    for (auto &kv : ctx->linefft) {
        if (kv.second.tw) (void)hipFree(kv.second.tw);
        if (kv.second.chirp) (void)hipFree(kv.second.chirp);
        if (kv.second.filt) (void)hipFree(kv.second.filt);
        if (kv.second.filt_ct) (void)hipFree(kv.second.filt_ct);
        if (kv.second.rtw) (void)hipFree(kv.second.rtw);
    }
    delete ctx;
    return 0;
}

int corahip_ctx_set_stream(corahip_ctx *ctx, void *hip_stream) {
    ARG_CHECK(ctx != nullptr);
    ctx->stream = (hipStream_t)hip_stream;
    return 0;
}

int corahip_ctx_sync(corahip_ctx *ctx) {
    ARG_CHECK(ctx != nullptr);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int corahip_timer_begin(corahip_ctx *ctx) {
    ARG_CHECK(ctx != nullptr);
    HIP_TRY(hipEventRecord(ctx->t0, ctx->stream));
    return 0;
}

int corahip_timer_end(corahip_ctx *ctx, float *elapsed_ms) {
    ARG_CHECK(ctx != nullptr && elapsed_ms != nullptr);
    HIP_TRY(hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->t1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    return 0;
}

int corahip_profile_enable(corahip_ctx *ctx, int enable) {
    ARG_CHECK(ctx != nullptr);
    ctx->profile = enable != 0;
    return 0;
}

static int drain_pending(corahip_ctx *ctx) {
    if (ctx->pending.empty()) return 0;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.e0, p.e1));
        auto &e = ctx->prof[p.name];
        e.total_ms += ms;
        e.launches += 1;
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    ctx->pending.clear();
    return 0;
}

int corahip_profile_get(corahip_ctx *ctx, const char *name, double *total_ms, int *launches) {
    ARG_CHECK(ctx != nullptr && name != nullptr);
    int rc = drain_pending(ctx);
    if (rc) return rc;
    auto it = ctx->prof.find(name);
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return 0;
}

int corahip_profile_reset(corahip_ctx *ctx) {
    ARG_CHECK(ctx != nullptr);
    int rc = drain_pending(ctx);
    if (rc) return rc;
    ctx->prof.clear();
    return 0;
}

int corahip_malloc(corahip_ctx *ctx, size_t bytes, void **dptr) {
    ARG_CHECK(ctx != nullptr && dptr != nullptr);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(dptr, bytes));
    return 0;
}

int corahip_free(corahip_ctx *ctx, void *dptr) {
    ARG_CHECK(ctx != nullptr);
    HIP_TRY(hipFree(dptr));
    return 0;
}

int corahip_memcpy_h2d(corahip_ctx *ctx, void *dst, const void *host_src, size_t bytes) {
    ARG_CHECK(ctx != nullptr);
    HIP_TRY(hipMemcpyAsync(dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int corahip_memcpy_d2h(corahip_ctx *ctx, void *host_dst, const void *src, size_t bytes) {
    ARG_CHECK(ctx != nullptr);
    HIP_TRY(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
