// common.h - internals shared by the translation units of libcorahip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/corahip.h"

void corahip_set_error(const char *fmt, ...);

struct corahip_prof_entry {
    double total_ms = 0.0;
    int launches = 0;
};

struct corahip_pending_event {
    std::string name;
    hipEvent_t e0, e1;
};

// tables of one line-FFT length of the flat-sky engine (flatsky.hip); device pointers
struct corahip_linefft_plan {
    int n = 0, P = 0, logP = 0, blu = 0;
    int Pct = 0;               // Bluestein lengths: convolution length of the compile-time passes (flatsky_ct.hip), 0 = none
    double2 *filt_ct = nullptr;   // [Pct] its filter, in the passes' storage order
    double2 *tw = nullptr;     // [P]  exp(-2 pi i k / P)
    double2 *chirp = nullptr;  // [n]  exp(-i pi k^2 / n)                       (Bluestein lengths only)
    double2 *filt = nullptr;   // [P]  FFT_P(wrapped conj chirp) / P, bit-reversed (Bluestein lengths only)
    double2 *rtw = nullptr;    // [n/2 + 1]  e^{+2 pi i k / 2n}: (un)packing of a real transform of length 2n
};

// contiguous c2r pass of the flat-sky transforms with the compile-time FFT passes of sht_ringfft_ct.hip
struct corahip_ctx;
int flat_c2r_ct(corahip_ctx *ctx, const double *spec, double *out, long nlines, int h, double scale, bool *took);
int flat_blu_plan(corahip_ctx *ctx, int n, const double2 *chirp, int *Pct, double2 **filt_ct);
int flat_blu_c2c_ct(corahip_ctx *ctx, const double *in, double *out, long nouter, int n, long inner, int inverse, double scale, bool gen,
                    uint64_t seed, int Pct, const double2 *chirp, const double2 *filt_ct, bool *took);
int flat_blu_real_ct(corahip_ctx *ctx, bool c2r, const double *in, double *out, long nlines, int h, double scale, int Pct,
                     const double2 *chirp, const double2 *filt_ct, const double2 *rtw, bool *took);
int flat_r2c_ct(corahip_ctx *ctx, const double *in, double *spec, long nlines, int h, bool *took);
int flat_c2c_ct(corahip_ctx *ctx, const double *in, double *out, long nouter, int n, long inner, int inverse, double scale, bool gen,
                uint64_t seed, bool *took);

#define CORAHIP_NSCRATCH 10
struct corahip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;                 // second stream for kernels that run beside those of `stream` (K5 pair)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // the l-range pipeline of the numpy-stream draw (drawstream.hip): its generator stream and the ring's events
    unsigned *draw_slot_tab = nullptr;             // K3: (l, m block) of every work slot of 0 .. draw_slot_lmax (draw.hip)
    int draw_slot_lmax = -1;
    bool mt_plist_ready = false;                   // scratch slot 9 holds the position lists (mtlegacy.hip)
    hipStream_t gen_stream = nullptr;
    hipEvent_t ev_ring[8] = {};
    // the session of corahip_draw_alm_numpy_begin that has not seen its _end yet: its generator tables (scratch slot 6)
    // and ring (slot 7) are in use by queued launches, so a second _begin and the whole-stream generators
    // (corahip_normals_pcg64 / _mt19937_legacy, which share slot 6) return CORAHIP_ESTATE until _end has run
    const void *draw_pending = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profile = false;
    std::map<std::string, corahip_prof_entry> prof;
    std::vector<corahip_pending_event> pending;
    int num_cu = 256;
    size_t total_mem = 0;                          // device memory (bytes)
    // grow-only device scratch slots owned by the context (freed by ctx_destroy)
    // slots: 0 K1 transposed tables, 1 K1 pair results / odd-F normal stream, 2 K1 pair list, 3 zeros,
    //        4 Legendre matrix of legendre_project, 5 its zero-padded operand, 6 block tables of normals_pcg64 /
    //        segment tables of normals_mt19937_legacy, 7 the two-slot ring of the l-range pipeline (drawstream.hip),
    //        8 barrier words of the cooperative Cholesky, 9 position lists of the MT19937 jump polynomials
    void *scratch[CORAHIP_NSCRATCH] = {};
    size_t scratch_bytes[CORAHIP_NSCRATCH] = {};
    // K1 transposed tables resident in scratch slot 0: valid for the pinned (dd, dv, vv, generation) only
    // (corahip_clarray_tables_pin: the caller vouches that the tables do not change under that generation)
    const void *tt_pin[3] = {nullptr, nullptr, nullptr};
    uint64_t tt_pin_gen = 0;
    bool tt_pinned = false, tt_valid = false;
    hipStream_t tt_stream = nullptr;               // stream the kept copy was made on
    // K1 pair list resident in scratch slot 2: (F, first, step, slots, device pointer it was written to)
    long pairs_key[4] = {-1, -1, -1, -1};
    void *pairs_ptr = nullptr;
    // flat-sky line-FFT tables by transform length
    std::map<int, corahip_linefft_plan> linefft;
};

// returns a device buffer of at least `bytes` for `slot`, reallocating only when it must grow
int corahip_ctx_scratch(corahip_ctx *ctx, int slot, size_t bytes, void **out);

// RAII: HIP-event pair around the launches of one named stage when profiling is on.
struct StageTimer {
    corahip_ctx *ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char *name;
    hipStream_t stream;
    // `on`: the stream the timed launches go to (default: the context's; the events of another stream must have completed
    // by the time the context's stream is drained - true for the side streams of the library, which it always joins)
    StageTimer(corahip_ctx *c, const char *n, hipStream_t on = nullptr, bool other = false) : ctx(c), name(n), stream(other ? on : c->stream) {
        if (ctx->profile) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, stream);
        }
    }
    ~StageTimer() {
        if (ctx->profile) {
            (void)hipEventRecord(e1, stream);
            ctx->pending.push_back({name, e0, e1});
        }
    }
};

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            corahip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                              __FILE__, __LINE__);                                          \
            return (int)_e;                                                                 \
        }                                                                                   \
    } while (0)

#define ARG_CHECK(cond)                                                                     \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            corahip_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);   \
            return CORAHIP_EINVAL;                                                          \
        }                                                                                   \
    } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

__host__ __device__ static inline long nalm_of(int lmax) { return (long)(lmax + 1) * (lmax + 2) / 2; }
// healpy packed index (m-major): idx(l,m) = m(2 lmax+1-m)/2 + l
__host__ __device__ static inline long alm_idx(int l, int m, int lmax) {
    return (long)m * (2 * lmax + 1 - m) / 2 + l;
}

typedef double d4_t __attribute__((ext_vector_type(4)));
