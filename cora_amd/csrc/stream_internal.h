// stream_internal.h - the pieces of the l-range pipeline of the numpy-stream draw (drawstream.hip): numpy's normal
// stream is generated on the device one RANGE of multipoles at a time into a two-slot ring and consumed by K3 range by
// range, as the reference consumes it inside its l loop (cora/core/skysim.py:114-121, cora/util/nputil.py:121-125) -
// the 16 F nalm bytes of a whole realisation's normals never exist.  Not part of the C ABI.
#pragma once
#include "common.h"

#include <vector>

// ---- K3 of one range (draw.hip) -----------------------------------------------------------------------------------
// a_lm of the multipoles l_lo .. l_hi from `gslot`, whose element 0 is element g_off = F l_lo (l_lo + 1) of the
// stream-order buffer; T full [L, F, F] (rows = 0) or the rank's row block [L, nnu, F] (rows = 1) for the channels
// of `set` (one block, or the two chunks of a folded shard); on `stream`.
int corahip_draw_range(corahip_ctx *ctx, hipStream_t stream, const double *T, int rows, const int32_t *info, const double *gslot,
                       size_t g_off, int l_lo, int l_hi, int lmax, int F, const corahip_chanset *set, double *alm_dev);

// ---- numpy's PCG64 + ziggurat stream in ranges (npnormal.hip) -------------------------------------------------------
// prepare: seek + count + scan of the whole stream of n normals on `stream` (every block's entry state and first
// ordinal), then the first block of every range [bounds[r], bounds[r + 1]) by a search on the device - no host
// synchronisation.  emit_range: the normals with ordinals in range r to slot[ordinal - bounds[r]].  finish: reads the
// status back (synchronises `stream`), returns the raw draws the n normals consumed.
struct zig_session;
int zig_stream_prepare(corahip_ctx *ctx, hipStream_t stream, const uint64_t state[2], const uint64_t inc[2], int64_t n,
                       const std::vector<unsigned long long> &bounds, zig_session **out);
int zig_stream_emit_range(corahip_ctx *ctx, hipStream_t stream, zig_session *s, int r, double *slot);
int zig_stream_finish(corahip_ctx *ctx, hipStream_t stream, zig_session *s, uint64_t *n_raw);
void zig_stream_free(zig_session *s);

// ---- numpy's legacy MT19937 + polar-method stream in ranges (mtlegacy.hip) -----------------------------------------
// prepare: jump tree, count pass (which also keeps a snapshot of the generator every MT_SUB_BLOCKS blocks, so that the
// emit pass can start anywhere with fine granularity) and scan; emit_range / finish as above.  `state` is updated by
// finish to the state numpy would be left in.
struct mt_session;
int mt_stream_prepare(corahip_ctx *ctx, hipStream_t stream, corahip_mt_state *state, int64_t n,
                      const std::vector<unsigned long long> &bounds, mt_session **out);
int mt_stream_emit_range(corahip_ctx *ctx, hipStream_t stream, mt_session *s, int r, double *slot);
int mt_stream_finish(corahip_ctx *ctx, hipStream_t stream, mt_session *s, corahip_mt_state *state);
void mt_stream_free(mt_session *s);
