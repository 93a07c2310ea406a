/* corahip.h - C ABI of libcorahip.so: the MI355X (gfx950) implementation of cora's
 * Gaussian-sky hot path (C_l(nu,nu') integration -> per-l factor -> correlated
 * a_lm draw -> HEALPix synthesis).
 *
 * The reference (radiocosmology/cora) has NO FFI on this path: its boundary is the
 * Python call surface of cora/core/skysim.py, cora/util/nputil.py and
 * cora/util/hputil.py.  Each entry point below names the reference code it replaces
 * (file:line relative to the reference root); cora_amd/ mirrors the Python surface
 * on top of this ABI, and INTEGRATION.md shows the ctypes stub a cora maintainer
 * would add.
 *
 * Conventions
 *  - every function returns int: 0 = OK, <0 = invalid argument (CORAHIP_E*),
 *    >0 = hipError_t of the failing runtime call.  Nothing throws across the ABI.
 *  - corahip_last_error() gives a thread-local human-readable message.
 *  - all array arguments are DEVICE pointers unless the name says `host_`; they are
 *    caller-owned (the Python side allocates them as torch tensors; hosts without
 *    torch can use corahip_malloc/free/memcpy_*).  Arrays are C-contiguous,
 *    float64 unless stated.
 *  - work is enqueued on the context's stream (corahip_ctx_set_stream) and is
 *    asynchronous; corahip_ctx_sync waits for it.
 *  - one context per GPU/process; a context is not thread-safe.
 */
#ifndef CORAHIP_H
#define CORAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORAHIP_ABI_VERSION 1
#define CORAHIP_ABI_MINOR 4      /* additions since version 1: 1 = normals_pcg64, pcg64_advance, draw_alm_rows, mkfullsky, mkfullsky_workspace_bytes, abi_minor, normals_mt19937_legacy; 2 = sht_lambda_entry (test hook); 3 = draw_alm_numpy, draw_alm_numpy_begin / _end, corahip_chanset: draw_alm_philox_rows_set, draw_alm_numpy_begin_set, randomfield_irfftn; 4 = glibc_exp (test hook), draw_alm_numpy_prepare / _run */

#define CORAHIP_EINVAL (-1)   /* bad argument / shape */
#define CORAHIP_ENOMEM (-2)   /* workspace too small / allocation refused */
#define CORAHIP_ESTATE (-3)   /* object used in the wrong state */

typedef struct corahip_ctx corahip_ctx;
typedef struct corahip_sht_plan corahip_sht_plan;

/* ---- library / context ------------------------------------------------------------ */
int corahip_abi_version(void);
int corahip_abi_minor(void);      /* entry points added since the version was cut, see CORAHIP_ABI_MINOR */
const char *corahip_last_error(void);
int corahip_device_count(int *count);
int corahip_ctx_create(int device_id, corahip_ctx **ctx);
int corahip_ctx_destroy(corahip_ctx *ctx);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream */
int corahip_ctx_set_stream(corahip_ctx *ctx, void *hip_stream);
int corahip_ctx_sync(corahip_ctx *ctx);
/* HIP-event timing of everything enqueued between begin and end on the ctx stream (bench.py) */
int corahip_timer_begin(corahip_ctx *ctx);
int corahip_timer_end(corahip_ctx *ctx, float *elapsed_ms);
/* per-kernel accumulated HIP-event timings (enable = 1 records an event pair around every
 * kernel launch of the library; names: "clarray","factor","normals","draw","legendre","ringfft") */
int corahip_profile_enable(corahip_ctx *ctx, int enable);
int corahip_profile_get(corahip_ctx *ctx, const char *name, double *total_ms, int *launches);
int corahip_profile_reset(corahip_ctx *ctx);

/* ---- plain device memory helpers (for hosts without torch) ----------------------- */
int corahip_malloc(corahip_ctx *ctx, size_t bytes, void **dptr);
int corahip_free(corahip_ctx *ctx, void *dptr);
int corahip_memcpy_h2d(corahip_ctx *ctx, void *dst, const void *host_src, size_t bytes);
int corahip_memcpy_d2h(corahip_ctx *ctx, void *host_dst, const void *src, size_t bytes);

/* ---- K1: C_l(nu,nu') integration --------------------------------------------------
 * Replaces skysim.clarray (cora/core/skysim.py:10-69) for the two model families
 * cora ships, plus the bare Romberg reduction for arbitrary host callables.         */

/* 21cm flat-sky table model: RedshiftCorrelation.angular_powerspectrum_fft evaluation
 * (cora/signal/corr.py:944-982) with bilinearmap.interp (cora/util/bilinearmap.pyx:14-59)
 * fused with the Romberg channel average of clarray (skysim.py:41-67).
 *   dd,dv,vv   [nkperp, nkpar] lookup tables (corr.py:936-940)
 *   chi,pfd,f,b [F*zint]  per sub-sample comoving distance, prefactor*D(z)/D(z_ps),
 *                         growth rate, bias (corr.py:944-951), sub-sample s of channel i
 *                         at index i*zint+s (skysim.py:47-49 ordering)
 *   w          [zint]     normalised Romberg weights (sum = 1); zint = 1, w = {1} is zromb=0
 *   log10l     [nl]       log10 of the multipoles to evaluate (l = 0 passed as 1e-10, corr.py:957)
 *   out        [nl, F, F]
 */
int corahip_clarray_table21cm(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv,
                              int nkperp, int nkpar, double kperpmin, double kperpmax, double kparmax,
                              const double *chi, const double *pfd, const double *f, const double *b,
                              int F, int zint, const double *w, const double *log10l, int nl,
                              double *out);

/* Multi-GPU form of the same integration (the reference shards clarray's output over l with
 * caput.mpiarray, skysim.py:97-103; the table lookups of a channel pair are shared by all l, so here
 * the PAIRS are sharded instead and the l-shards are assembled by one all-to-all):
 *   table21cm_pairs integrates, for all nl multipoles, the channel pairs p = pair_first + k pair_step
 *   (k = 0 .. npl-1, npl = ceil(F(F+1)/2 / pair_step); p enumerates the pairs (i, j >= i) in bands of 32 diagonals, i-major inside a band) and writes
 *   out_pairs [ceil(nl / l_block)][npl][l_block]: slab q is what the rank owning l in [q l_block, (q+1) l_block)
 *   receives.  pairs_finish takes the received slabs [nranks][npl][l_stride] (slab r from rank r) and
 *   scatters them into out [nl, F, F] and its mirror.                                              */
int corahip_clarray_table21cm_pairs(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv,
                                    int nkperp, int nkpar, double kperpmin, double kperpmax, double kparmax,
                                    const double *chi, const double *pfd, const double *f, const double *b,
                                    int F, int zint, const double *w, const double *log10l, int nl,
                                    int pair_first, int pair_step, int l_block, double *out_pairs);
int corahip_clarray_pairs_finish(corahip_ctx *ctx, const double *pairs_in, int F, int nranks, int l_stride,
                                 int nl, double *out);

/* The table integrators work from an x-contiguous copy of (dd, dv, vv) that they make on every call (0.15 ms at
 * 500 x 32768).  A caller that keeps the tables unchanged between calls - they are the per-model cache of the reference,
 * cora/signal/corr.py:909-942 - pins them: the copy made by the next call is then reused as long as the SAME pointers
 * are passed and no other pin is made.  `generation` distinguishes table sets that happen to reuse device addresses
 * (increment it whenever the tables are rebuilt or rewritten).  The pin must not outlive the tables: dd = NULL removes it -
 * with generation != 0 only if that generation is still the one pinned (an owner releasing its own pin when its tables
 * are freed or replaced; it cannot remove a later owner's), with generation = 0 unconditionally.  The kept copy is tied
 * to the stream it was made on; a call on another stream makes its own. */
int corahip_clarray_tables_pin(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv,
                               uint64_t generation);

/* the bare aps callable at n independent points (corr.py:953-982): lx = log10(l), chi1, chi2,
 * and the coefficient triples b1 b2 P, (f1 b2 + f2 b1) P, f1 f2 P with P = D1 D2 pf1 pf2
 * (the 1/(xc^2 pi) factor is applied by the kernel).  All arrays [n]. */
int corahip_aps_table21cm_points(corahip_ctx *ctx, const double *dd, const double *dv, const double *vv,
                                 int nkperp, int nkpar, double kperpmin, double kperpmax, double kparmax,
                                 long n, const double *lx, const double *chi1, const double *chi2,
                                 const double *cdd, const double *cdv, const double *cvv, double *out);

/* separable model C_l = A_l * B(nu,nu') (cora/foreground/gaussianfg.py:40-41):
 *   al [nl], bcov [F*zint, F*zint] (sub-sampled frequency covariance), w [zint] -> out [nl,F,F] */
int corahip_clarray_separable(corahip_ctx *ctx, const double *al, int nl, const double *bcov, int F,
                              int zint, const double *w, double *out);

/* Romberg reduction of host-evaluated samples (skysim.py:62-67):
 *   clt [nl, F, zint, F, zint] -> out [nl, F, F] = sum_ab w_a w_b clt                */
int corahip_romb_reduce(corahip_ctx *ctx, const double *clt, int nl, int F, int zint, const double *w,
                        double *out);

/* ---- K2: per-l matrix root ---------------------------------------------------------
 * Replaces the loop body of mkfullsky (skysim.py:115-119) and
 * nputil.matrix_root_manynull(truncate=False) (cora/util/nputil.py:51-101):
 *   Cm = C_l + I * max(diag C_l) * jitter_rel;  T_l = chol_lower(Cm);  if a pivot is
 *   not positive (LAPACK potrf failure) -> symmetric eigen-decomposition, eigenvalues
 *   < max * eig_thresh set to 0, T_l = V sqrt(Lambda).
 *   C [nl,F,F] -> T [nl,F,F];  info [nl] int32: 0 = Cholesky, 1 = eigen branch.      */
int corahip_factor_batched(corahip_ctx *ctx, const double *C, int nl, int F, double jitter_rel,
                           double eig_thresh, double *T, int32_t *info);

/* ---- K3: correlated draw -----------------------------------------------------------
 * Normal stream layout ("stream order"; nputil.py:104-125 called from
 * skysim.py:120): for l = 0..lmax: F*(l+1) reals [nu'][m] then F*(l+1) imags [nu'][m];
 * total 2*F*nalm doubles.  The 1/sqrt(2) of complex_std_normal is applied by draw_alm.
 * normals_philox fills that buffer with the device stream: the elements (l, c = 0 / 1, nu', m) are the two Box-Muller
 * outputs of Philox4x32-10 counter {m, l*F + nu'} under key `seed` (independent of the GPU count; the mapping of the
 * four output words to the pair is specified in oracle/philox.py and cora_amd/csrc/rng_dev.h). */
int corahip_normals_philox(corahip_ctx *ctx, uint64_t seed, int lmax, int F, double *g);

/* The REFERENCE's own seeded stream on the device: the next n values of
 * numpy.random.Generator(PCG64).standard_normal - what `rng.standard_normal(shape)` of cora/util/nputil.py:121-125
 * returns for the `rng = default_rng(seed)` of cora/signal/lss.py:449-450 - bit for bit (fast-path and wedge samples
 * are one exact multiply; tail samples use glibc's log1p, the wedge test glibc's exp, both restated operation by
 * operation), written to g[0..n) in draw
 * order: with n = 2*F*nalm the "stream order" buffer draw_alm consumes.
 *   state, inc   host: the 128-bit PCG64 state and increment as {high 64 bits, low 64 bits}
 *                (rng.bit_generator.state["state"]["state" | "inc"])
 *   n_raw        host, out: the number of raw 64-bit draws the n normals consumed (1.0215 n on average: the ziggurat
 *                takes a data-dependent number per sample); corahip_pcg64_advance(state, inc, *n_raw) is the state
 *                numpy would be left in.
 * Synchronises the context's stream (n_raw is read back).  numpy's algorithm: PCG64 = pcg_setseq_128_xsl_rr_64,
 * 256-strip ziggurat of numpy/random/src/distributions/distributions.c; restated in oracle/npnormal.py. */
int corahip_normals_pcg64(corahip_ctx *ctx, const uint64_t host_state[2], const uint64_t host_inc[2], int64_t n,
                          double *g, uint64_t *host_n_raw);
/* Test hook of that stream's one libm call in the accept path: y[i] = exp(x[i]) as the wedge test of the ziggurat
 * evaluates it on the device - glibc's table-driven exp (sysdeps/ieee754/dbl-64/e_exp.c) in the evaluation order of its
 * FMA build, the libm numpy's random_standard_normal calls (reached from cora/util/nputil.py:125) - for |x| < 512.
 * x, y device arrays; asynchronous on the context's stream.  tests/test_gpu_npnormal.py sweeps it against the host's exp. */
int corahip_glibc_exp(corahip_ctx *ctx, const double *x, int64_t n, double *y);
/* host arithmetic: the PCG64 state after `delta` steps (numpy's bit_generator.advance) */
int corahip_pcg64_advance(const uint64_t host_state[2], const uint64_t host_inc[2], uint64_t delta,
                          uint64_t host_out_state[2]);

/* numpy's LEGACY normal stream on the device: the next n values of np.random.standard_normal - what the reference
 * draws when it is called WITHOUT a generator (rng=None in cora/util/nputil.py:121-123; Sky3d.getsky(),
 * cora/core/maps.py:235-237): the global MT19937 state + the polar method of numpy's legacy_gauss with its one cached
 * value.  host_state: np.random.get_state(legacy=False) as a struct - key (624 words), pos, has_gauss, gauss - in; the
 * state numpy would be left in out (an equivalent (key, pos) pair: the 624-word block the generator is in and the
 * position inside it).  Which attempts of the polar method are accepted - and with it the state - is numpy's exactly;
 * the values take glibc's log restated operation by operation (its FMA build, csrc/mtlegacy.hip glibc_log_fma) and
 * correctly rounded sqrt / divisions: numpy's bit for bit where numpy's libm is that routine, within 4 ulp elsewhere.
 * Synchronises the context's stream.  Algorithm: csrc/mtlegacy.hip (MT19937 cut into segments by jump-ahead polynomials
 * over GF(2), csrc/mt_jump.inc); restated in oracle/mtlegacy.py. */
typedef struct corahip_mt_state {
    uint32_t key[624];
    int32_t pos, has_gauss;
    double gauss;
} corahip_mt_state;
int corahip_normals_mt19937_legacy(corahip_ctx *ctx, corahip_mt_state *host_state, int64_t n, double *g);

/* a_lm(nu) = sum_nu' T_l[nu,nu'] g_lm(nu')  (skysim.py:121) for channels nu0 <= nu < nu0+nnu.
 *   T [lmax+1, F, F], info [lmax+1] (from factor_batched; NULL = treat all as dense),
 *   g stream order (above), alm_dev: device a_lm layout [nalm][nnu_pad/4][2][4]
 *   (packed healpy index idx = m(2 lmax+1-m)/2 + l; channel nu0+4g+v at [g][c][v], c = re/im;
 *   nnu_pad = nnu rounded up to a multiple of 4, padding channels are written as 0).  */
int corahip_draw_alm(corahip_ctx *ctx, const double *T, const int32_t *info, const double *g, int lmax,
                     int F, int nu0, int nnu, double *alm_dev);

/* draw_alm with only the rank's rows of the factors resident: T_rows [lmax+1, nnu, F] = rows nu0 .. nu0+nnu-1 of every T_l
 * (see draw_alm_philox_rows) */
int corahip_draw_alm_rows(corahip_ctx *ctx, const double *T_rows, const int32_t *info, const double *g, int lmax,
                          int F, int nu0, int nnu, double *alm_dev);

/* draw_alm with the device stream generated in registers (same values as normals_philox(seed) followed by
 * draw_alm, without the 16*F*nalm-byte normal buffer) */
int corahip_draw_alm_philox(corahip_ctx *ctx, const double *T, const int32_t *info, uint64_t seed, int lmax,
                            int F, int nu0, int nnu, double *alm_dev);

/* draw_alm_philox with only the rank's rows of the factors resident: T_rows [lmax+1, nnu, F] holds rows
 * nu0 .. nu0+nnu-1 of every T_l (what a frequency-sharded rank receives from the all-to-all of the
 * l-sharded factor stack: 1/N of the all-gather traffic and memory).                                */
int corahip_draw_alm_philox_rows(corahip_ctx *ctx, const double *T_rows, const int32_t *info, uint64_t seed,
                                 int lmax, int F, int nu0, int nnu, double *alm_dev);

/* A rank's channels as one block or as the two chunks of a FOLDED frequency shard: local channel c < chunk_nnu is global
 * channel nu0[0] + c, local channel chunk_nnu + c is nu0[1] + c (nchunks = 2; nu0[1] >= nu0[0] + chunk_nnu; chunk_nnu a
 * multiple of 4; F even).  The draw's cost grows with the channel index (T_l is lower triangular: channel nu takes nu + 1
 * terms), so a frequency shard made of a low and a high chunk - rank r of N takes chunks r and 2 N - 1 - r of 2 N - costs
 * every rank the same (cora_amd.parallel, `fold=True`; the reference's contiguous split, cora/core/skysim.py:132-134,
 * stays the default).  T_rows [lmax+1, nchunks * chunk_nnu, F] holds the rows of the local channels in local order;
 * alm_dev and the maps made from it are in local order too. */
typedef struct corahip_chanset {
    int32_t nchunks, chunk_nnu;
    int32_t nu0[2];
} corahip_chanset;
int corahip_draw_alm_philox_rows_set(corahip_ctx *ctx, const double *T_rows, const int32_t *info, uint64_t seed,
                                     int lmax, int F, const corahip_chanset *set, double *alm_dev);

/* ---- the whole of skysim.mkfullsky in one call (cora/core/skysim.py:72-136, single process) --------------------
 * C [L, F, F] (device; L = lmax + 1 of `plan`) -> maps of channels nu0 .. nu0+nnu-1: jitter + root per l (:115-119),
 * complex normals (:120), a_lm = T_l g_l (:121), HEALPix synthesis (:130).  A chain of the entry points above
 * (factor_batched, normals_pcg64 / draw_alm*, alm2map) on buffers cut from ONE caller-owned device workspace.
 *   rng   which normals (host struct):
 *           CORAHIP_RNG_PCG64   the reference's seeded call, rng = numpy.random.default_rng(seed): `state`, `inc` are
 *                               bit_generator.state["state"]["state" | "inc"] as {high, low} 64-bit words; numpy's
 *                               sequence is continued on the device and `state` is UPDATED to the state numpy would be
 *                               left in (the call synchronises the stream for that);
 *           CORAHIP_RNG_STREAM  `stream`: DEVICE pointer to 2 F nalm normals in stream order (any generator, drawn by
 *                               the caller in the reference's order - e.g. rng=None, numpy's legacy global state);
 *           CORAHIP_RNG_MT19937 the reference called WITHOUT a generator (rng=None: numpy's legacy global state, what
 *                               Sky3d.getsky() draws from): `legacy` points to np.random.get_state(legacy=False) as a
 *                               corahip_mt_state; continued on the device (corahip_normals_mt19937_legacy) and UPDATED;
 *           CORAHIP_RNG_PHILOX  the library's counter-based stream under `seed` (not numpy's numbers).
 *   alms  0: out = maps [nnu, npix] RING (skysim.py:130-136);  1: out = a_lm [nnu, 1, L, L] complex128, m > l zero (:123-125)
 *   workspace  >= corahip_mkfullsky_workspace_bytes(...) for one synthesis pass; a smaller one (down to the factors,
 *              the a_lm and one 8-channel synthesis chunk) is worked through in chunks; the PCG64 / MT19937 kinds draw
 *              through corahip_draw_alm_numpy (below): their normals live in the library's own ring, range by range;
 *              CORAHIP_ENOMEM with the sizes in corahip_last_error() below that.                                      */
#define CORAHIP_RNG_STREAM 0
#define CORAHIP_RNG_PHILOX 1
#define CORAHIP_RNG_PCG64 2
#define CORAHIP_RNG_MT19937 3
typedef struct corahip_rng {
    int32_t kind, reserved;
    const double *stream;
    uint64_t seed;
    uint64_t state[2], inc[2];
    corahip_mt_state *legacy;
} corahip_rng;
int corahip_mkfullsky_workspace_bytes(const corahip_sht_plan *plan, int F, int nu0, int nnu, int rng_kind, int alms,
                                      size_t *bytes);
int corahip_mkfullsky(corahip_ctx *ctx, const corahip_sht_plan *plan, const double *C, int F, corahip_rng *host_rng,
                      int nu0, int nnu, int alms, double *out, void *workspace, size_t workspace_bytes);

/* draw_alm with numpy's OWN normal stream generated on the device RANGE BY RANGE, as the reference consumes it inside its
 * l loop (cora/core/skysim.py:114-121, cora/util/nputil.py:121-125): normals_pcg64 / normals_mt19937_legacy followed by
 * draw_alm (rows = 0: T [lmax+1, F, F]) or draw_alm_rows (rows = 1: T_rows [lmax+1, nnu, F]), bit for bit, WITHOUT the
 * 16 F nalm-byte stream buffer (8.6 GB at F = 256, lmax = 2048; 137 GB at F = 1024, lmax = 4096): the generator's
 * count + scan passes cover the whole stream, its emit pass fills one slot of a two-slot ring (library-owned, ring_bytes
 * in total, 0 = the default: 2 GiB - the measured optimum of the whole step at F = 256, lmax = 2048 - while the
 * stream is <= 1/8 of the device memory, else 1/16 of the memory, or CORAHIP_RING_MB; a slot is never smaller than the normals of l = lmax) per range
 * of multipoles on a second stream while K3 consumes the other slot.
 *   host_rng  (the struct of corahip_mkfullsky above) kind CORAHIP_RNG_PCG64 (state, inc as in corahip_normals_pcg64) or CORAHIP_RNG_MT19937 (legacy state);
 *             UPDATED to the state numpy would be left in.  Synchronises the context's stream (that read-back).
 * _begin / _end: the same in two halves.  _begin enqueues everything and returns without waiting; the caller goes on
 * enqueueing (the synthesis of the a_lm) and calls _end when it wants the generator state: _end waits for the context's
 * stream, updates host_rng and frees `pending`.  Between the two, no other numpy-stream draw may be started on the
 * context (enforced: a second _begin, corahip_normals_pcg64 and corahip_normals_mt19937_legacy return CORAHIP_ESTATE
 * while a session is pending - they share its device tables) and host_rng->legacy (MT19937 kind) must stay valid. */
typedef struct corahip_draw_pending corahip_draw_pending;
int corahip_draw_alm_numpy(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, corahip_rng *host_rng,
                           int lmax, int F, int nu0, int nnu, double *alm_dev, size_t ring_bytes);
int corahip_draw_alm_numpy_begin(corahip_ctx *ctx, const double *T, int rows, const int32_t *info, const corahip_rng *host_rng,
                                 int lmax, int F, int nu0, int nnu, double *alm_dev, size_t ring_bytes,
                                 corahip_draw_pending **pending);
int corahip_draw_alm_numpy_end(corahip_ctx *ctx, corahip_draw_pending *pending, corahip_rng *host_rng);
/* _begin in its two halves.  _prepare is the part of a draw that does NOT depend on the factors - ranges, ring, the
 * generator's own passes (count + scan / jump tree + count) and the emit passes of the first two ranges, all on the
 * library's generator stream: a caller that issues it BEFORE the launches that make the factors (C_l integration,
 * factorisation) lets the generator run beside them (the stream is a function of the generator alone; the reference draws
 * it inside mkfullsky, cora/util/nputil.py:121-125, but nothing it draws depends on the covariance).  _run enqueues K3 of
 * every range against the factors (T full [L, F, F] with rows = 0 - `set` then names one block - or the row block of the
 * set's channels with rows = 1) and the remaining emit passes; _end as above.  A prepared session that is given up
 * (no _run) is freed by _end, which then leaves host_rng untouched.  _begin = _prepare + _run. */
int corahip_draw_alm_numpy_prepare(corahip_ctx *ctx, const corahip_rng *host_rng, int lmax, int F, size_t ring_bytes,
                                   corahip_draw_pending **pending);
int corahip_draw_alm_numpy_run(corahip_ctx *ctx, corahip_draw_pending *pending, const double *T, int rows, const int32_t *info,
                               const corahip_chanset *set, double *alm_dev);
/* _begin for a channel set (the struct above; T_rows = the row block of its channels) */
int corahip_draw_alm_numpy_begin_set(corahip_ctx *ctx, const double *T_rows, const int32_t *info, const corahip_rng *host_rng,
                                     int lmax, int F, const corahip_chanset *set, double *alm_dev, size_t ring_bytes,
                                     corahip_draw_pending **pending);

/* ---- frequency sharding for callers that pass the messages themselves -----------------------------
 * The reference distributes this path with caput.mpiarray over MPI (cora/core/skysim.py:97-110: C_l and the a_lm
 * buffer split over l; :125-134: allgather / redistribute to a frequency split).  These entry points give such a
 * caller the sharded path without torch.distributed: the only exchange is ONE all-to-all of factor row blocks.
 *   shard_plan        the split both sides use: multipoles [l_lo, l_hi) (contiguous blocks of l_shard, the last
 *                     ones shorter / empty), channels [nu0, nu0 + nnu); rows_exchange = 1 iff F % world == 0 (the
 *                     row-block all-to-all applies; otherwise all-gather the [L, F, F] factor stack instead).
 *                     Pure host arithmetic - any other contiguous l split (caput's) works with pack / unpack too.
 *   factor_rows_pack  T_local [n_local, F, F] (this rank's factors, corahip_factor_batched of its C_l block) ->
 *                     send [world][l_stride][F / world][F]: slab q = the rows of rank q's channels, l rows
 *                     n_local .. l_stride - 1 zero (l_stride = the largest block of any rank: equal slabs)
 *   (caller)          all-to-all: slab q of every rank goes to rank q -> recv [world][l_stride][nnu][F], slab r
 *                     from rank r; all-gather of the info flags (int32 [n_local] each)
 *   factor_rows_unpack recv + host_counts [world <= 64] (multipoles each rank holds, in rank order) ->
 *                     T_rows [sum counts][nnu][F]: what corahip_draw_alm_philox_rows consumes
 * then corahip_draw_alm_philox_rows(T_rows, info_all, seed, lmax, F, nu0, nnu) and corahip_alm2map: every rank
 * draws from the same counter-based stream for its own channels, the a_lm are never exchanged, the maps stay
 * frequency-sharded (the reference's MPIArray.wrap(sky, axis=0), skysim.py:132-134). */
typedef struct corahip_shard {
    int32_t l_lo, l_hi;      /* this rank integrates / factors multipoles [l_lo, l_hi) */
    int32_t l_shard, l_pad;  /* padded block length (equal on all ranks), l_shard * world >= L */
    int32_t nu0, nnu;        /* this rank draws and synthesises channels [nu0, nu0 + nnu) */
    int32_t rows_exchange;   /* 1: F % world == 0, factor row blocks go by all-to-all */
    int32_t L;
} corahip_shard;
int corahip_shard_plan(int L, int F, int rank, int world, corahip_shard *out);
int corahip_factor_rows_pack(corahip_ctx *ctx, const double *T_local, int n_local, int l_stride, int F, int world,
                             double *send);
int corahip_factor_rows_unpack(corahip_ctx *ctx, const double *recv, const int32_t *host_counts, int world,
                               int l_stride, int nnu, int F, double *T_rows);

/* layout converters between alm_dev and the reference's arrays:
 *   square  [nnu, 1, L, L] complex128 as returned by mkfullsky(alms=True) (skysim.py:108-125)
 *   packed  [nnu, nalm]   complex128 healpy order, as pack_alm produces (hputil.py:124-152)  */
int corahip_alm_dev_to_square(corahip_ctx *ctx, const double *alm_dev, int lmax, int nnu, double *square);
int corahip_alm_packed_to_dev(corahip_ctx *ctx, const double *packed, int lmax, int nnu, double *alm_dev);

/* ---- K4/K5: HEALPix synthesis ------------------------------------------------------
 * Replaces hputil.sphtrans_inv_sky / sphtrans_inv_real -> healpy.alm2map
 * (cora/util/hputil.py:369-391,500-531) for nnu channels at once.                   */
int corahip_sht_plan_create(corahip_ctx *ctx, int nside, int lmax, corahip_sht_plan **plan);
/* ... with the truncation of the Legendre sums as a parameter: terms with |lambda_lm(theta)| < 2^cut_exp are dropped
 * (-1000 <= cut_exp < 0; 0 = the default, -70).  The error this leaves in a pixel is bounded by
 * 2 sum_lm |a_lm| 2^cut_exp (1.7e-21 per unit coefficient at the default - the library's own choice; what the engine
 * behind healpy.alm2map, cora/util/hputil.py:388-391, truncates is not pinned, healpy is absent): lower it for a_lm whose
 * dynamic range exceeds ~1e10; -900 keeps every term fp64 can represent (what the oracle does). */
int corahip_sht_plan_create_ex(corahip_ctx *ctx, int nside, int lmax, int cut_exp, corahip_sht_plan **plan);
int corahip_sht_plan_cut_exp(const corahip_sht_plan *plan, int *cut_exp);
int corahip_sht_plan_destroy(corahip_ctx *ctx, corahip_sht_plan *plan);
/* number of v_mfma_f64_16x16x4_f64 instructions (2048 flop each) the Legendre kernel of corahip_alm2map ISSUES
 * for one pass over nnu channels with this plan - computed from the plan's first-contributing-l tables, i.e. what
 * the SQ_INSTS_VALU_MFMA_F64 counter reads for the launch.  bench.py prices K4's roofline fraction on it. */
int corahip_sht_plan_k4_mfma_count(corahip_ctx *ctx, const corahip_sht_plan *plan, int nnu,
                                   uint64_t *mfma_instructions);
/* bytes of scratch alm2map needs to process `nnu` channels in one pass */
int corahip_alm2map_workspace_bytes(const corahip_sht_plan *plan, int nnu, size_t *bytes);
/* alm_dev [nalm][nnu_pad/4][2][4] -> maps [nnu, 12 nside^2] RING order.
 * workspace: >= corahip_alm2map_workspace_bytes(plan, nnu_chunk) for some nnu_chunk (multiple
 * of 4) <= nnu_pad; channels are processed in chunks that fit.                        */
int corahip_alm2map(corahip_ctx *ctx, const corahip_sht_plan *plan, const double *alm_dev, int nnu,
                    double *maps, void *workspace, size_t workspace_bytes);

/* ---- analysis: the quadrature pass of healpy.map2alm --------------------
 * Replaces what hputil.sphtrans_real / sphtrans_sky / sph_ps obtain from healpy.map2alm
 * (cora/util/hputil.py:195-234,460-497,607-619) for nnu channels at once:
 *   a_lm = sum_pix w_ring(pix) (4 pi / npix) map(pix) conj(Y_lm(pix))
 * maps [nnu, 12 nside^2] RING -> alm_dev [nalm][nnu_pad/4][2][4] (nnu_pad = nnu rounded up to 8).
 * ring_w: device [2 nside] quadrature weights of the north rings incl. the equator (the south mirrors
 * them; what healpy reads from weight_ring_n*.fits for use_weights=True), NULL = uniform.
 * healpy's iter=N refinement alm += A(map - S alm) is composed by the caller from alm2map/map2alm. */
int corahip_map2alm_workspace_bytes(const corahip_sht_plan *plan, int nnu, size_t *bytes);
int corahip_map2alm(corahip_ctx *ctx, const corahip_sht_plan *plan, const double *maps, int nnu,
                    const double *ring_w, double *alm_dev, void *workspace, size_t workspace_bytes);

/* ---- polarisation: spin-2 synthesis ----------------------------------------
 * Replaces the Q, U part of healpy.alm2map([T, E, B], nside) behind hputil.sphtrans_inv_real_pol
 * (cora/util/hputil.py:394-432):  Q +- iU = - sum_lm (a^E_lm +- i a^B_lm) (+-2)Y_lm  (Zaldarriaga & Seljak 1997).
 * alm_dev holds nnu = 2 nfreq channels interleaved (E_0, B_0, E_1, B_1, ...) in the usual layout (nnu_pad8/4 groups
 * must equal nnu_pad4/4, i.e. nnu mod 8 in {0, 5, 6, 7}); maps [nnu, npix] RING = (Q_0, U_0, Q_1, U_1, ...).
 * T (and V) go through corahip_alm2map.  Workspace: corahip_alm2map_workspace_bytes(plan, nnu).             */
int corahip_alm2map_spin2(corahip_ctx *ctx, corahip_sht_plan *plan, const double *alm_dev, int nnu,
                          double *maps, void *workspace, size_t workspace_bytes);

/* ---- spin-2 analysis ((Q, U) -> (E, B); cora/util/hputil.py:274-323 sphtrans_real_pol -> healpy.map2alm) ----
 * Composed from scalar quadrature passes (csrc/sht_polana.hip): spin2_ring_scale writes, for every field f with
 * maps (Q_f, U_f) = channels (2f, 2f+1) of maps_qu [2 nfields, npix], the six maps
 * [Q, r1 Q, r2 Q, U, r1 U, r2 U] (r1 = 1/sin^2 theta, r2 = cos theta / sin^2 theta of the pixel's ring) as channels
 * 6f .. 6f+5 of maps6; after corahip_map2alm(maps6, 6 nfields) spin2_combine forms
 *   E = -(sum_rings W Q~ - i X U~),  B = -(sum_rings W U~ + i X Q~)
 * into alm_eb_dev [nalm][gout][2][4] with (E_f, B_f) = channels (2f, 2f+1), the layout corahip_alm2map_spin2
 * consumes; alm6_dev is [nalm][g6][2][4] (g6, gout: channel groups of four, padding channels zero).  One call sequence = one quadrature pass; healpy's iter = N is
 * alm += A(map - S alm) with corahip_alm2map_spin2 as S, composed by the caller.                      */
int corahip_spin2_ring_scale(corahip_ctx *ctx, corahip_sht_plan *plan, const double *maps_qu, int nfields,
                             double *maps6);
int corahip_spin2_combine(corahip_ctx *ctx, corahip_sht_plan *plan, const double *alm6_dev, int g6, int nfields,
                          double *alm_eb_dev, int gout);

/* ---- K0: the 21cm lookup tables (SURVEY 8 row a4 / section 2a "K0") ------------------------------
 * Replaces the one-off setup of RedshiftCorrelation.angular_powerspectrum_fft (cora/signal/corr.py:909-942)
 * and the spline evaluation behind it (cora/util/cubicspline.pyx:126-175,254-288).
 * ps_table21cm: dd, dv = dd mu^2, vv = dd mu^4 on the [nkperp][nkpar] grid (kperp, kpar: device arrays, as numpy's
 *   logspace / linspace made them).  dd_in == NULL: dd = P(k) sinc^2(kpar freq_window / 2 pi) with P(k) a natural
 *   cubic spline, knots (x, y, y'') [nknot] on the device - loglog != 0: exp(spline(log k)) (LogInterpolater) - times
 *   exp(-k^2 / 2 kstar^2) when kstar > 0 (cora/signal/corr21cm.py:24-29).  dd_in != NULL: dd_in is dd as the host
 *   evaluated it (any other ps_vv callable); only dv and vv are made (dd may be NULL).
 * dct1_rows: scipy.fftpack.dct(x, type=1) * scale of every row of data [nrows][n], in place, as one complex DFT
 *   of length n - 1 by the prime-factor (Good-Thomas) map: n - 1 must split into at most 6 pairwise coprime prime
 *   powers <= 2048 (nkpar = 32768: 32767 = 7 * 31 * 151).  workspace: dct1_workspace_bytes(nrows, n). */
int corahip_ps_table21cm(corahip_ctx *ctx, const double *knots_x, const double *knots_y, const double *knots_y2,
                         int nknot, int loglog, double kstar, const double *kperp, int nkperp, const double *kpar,
                         int nkpar, double freq_window, const double *dd_in, double *dd, double *dv, double *vv);
int corahip_dct1_workspace_bytes(long nrows, int n, size_t *bytes);
int corahip_dct1_rows(corahip_ctx *ctx, double *data, long nrows, int n, double scale, void *workspace,
                      size_t workspace_bytes);

/* ---- xi(r) -> C_l(chi, chi') ----------------------------------------------
 * Replaces corrfunc.corr_to_clarray (cora/signal/corrfunc.py:290-400).
 * xi_table_average: for every Gauss-Legendre node mu_m and channel pair (i, j) the radial-bin average
 *   sum_ab xw_a xw_b xi(r(mu_m, xa[i xint + a], xa[j xint + b])),  r^2 = (x - x')^2 + 2 x x' (1 - mu)
 *   (corrfunc.py:368-381) of a correlation function given as a natural cubic spline with end-slope
 *   extrapolation (cora/util/cubicspline.pyx:126-231): knots (x, y, y'') [nk];
 *   kind 0: spline(r); 1: exp(spline(log r)) (LogInterpolater); 2: f_t sinh(spline(asinh(r / x_t)))
 *   (SinhInterpolater).  out [nm, F, F].
 * legendre_project: out[l, n] = sum_m wt[m] P_l(mu[m]) xi[m, n], l = 0..lmax (corrfunc.py:387-397, where
 *   wt = w 4 pi / wsum); xi [nm, ncol], out [lmax+1, ncol]: Legendre matrix by recurrence + FP64 MFMA GEMM. */
int corahip_xi_table_average(corahip_ctx *ctx, const double *knots_x, const double *knots_y,
                             const double *knots_y2, int nk, int kind, double x_t, double f_t,
                             const double *mu, int nm, const double *xa, const double *xw, int F, int xint,
                             double *out);
int corahip_legendre_project(corahip_ctx *ctx, const double *mu, const double *wt, int nm, int lmax,
                             const double *xi, long ncol, double *out);

/* ---- flat-sky Gaussian fields (flat-sky half) ------------------------------
 * n-dimensional FFTs over C-contiguous device arrays, complex = interleaved (re, im) float64; every
 * transformed axis has length <= 4096, any length (powers of two directly, others by Bluestein).
 * fft_c2c:   in-place transform of one axis; inverse = 0: sum x e^{-2 pi i jk/n}; 1: e^{+...} / n
 *            (numpy.fft.fft / ifft along `axis`).
 * irfftn:    numpy.fft.irfftn over the LAST naxes axes (cora/util/fftutil.py:80-87 with naxes = ndim;
 *            the ifft + irfft of cora/foreground/gaussianfg.py:82-84 with naxes = 2): spec has shape
 *            rdims with the last axis rdims[-1]/2 + 1 and is OVERWRITTEN; out has shape rdims.
 * rfftn:     numpy.fft.rfftn over the last naxes axes (fftutil.py:64-77); in real rdims, spec as above.
 * randomfield_draw: spec[e] = (g1 + i g2) kweight[e] with (g1, g2) the two Box-Muller normals of
 *            Philox4x32-10 counter e under key = seed - the device form of
 *            RandomField.getfield's `randn + 1j randn` (cora/core/gaussianfield.py:115-116); count
 *            complex elements.  Follow with irfftn for the field.
 * fg_mix:    out[f, m] = aff[m] sum_c freq_weight[f, c] normals[c, m]  (complex [F, M]; aff complex [M],
 *            normals real [ncorr, M]) - the tensordot of ForegroundMap.getfield
 *            (cora/foreground/gaussianfg.py:79-82); follow with irfftn(naxes = 2).                 */
int corahip_fft_c2c(corahip_ctx *ctx, double *data, int ndim, const int64_t *dims, int axis, int inverse);
int corahip_irfftn(corahip_ctx *ctx, double *spec, int ndim, const int64_t *rdims, int naxes, double *out);
int corahip_rfftn(corahip_ctx *ctx, const double *in, int ndim, const int64_t *rdims, int naxes, double *spec);
int corahip_randomfield_draw(corahip_ctx *ctx, const double *kweight, int64_t count, uint64_t seed,
                             double *spec);
int corahip_fg_mix(corahip_ctx *ctx, const double *freq_weight, const double *normals, const double *aff,
                   int F, int ncorr, int64_t M, double *out);
/* randomfield_draw + irfftn (all axes) in one call, the spectrum GENERATED where the first pass loads it - the same
 * values, bit for bit, without writing and re-reading the 16 bytes per element of a separate draw pass: kweight real
 * [rdims[0], ..., rdims[-1]/2 + 1], spec a workspace of that many complex elements, out real rdims.  The whole of
 * RandomField.getfield with the device stream (cora/core/gaussianfield.py:102-120).                              */
int corahip_randomfield_irfftn(corahip_ctx *ctx, const double *kweight, int ndim, const int64_t *rdims, uint64_t seed,
                               double *spec, double *out);
/* The redshift-space cube of RedshiftCorrelation.realisation / Corr21cm.getfield (cora/signal/corr.py:562-770,
 * cora/signal/corr21cm.py:241-257) on top of the transforms above:
 * spec_mul_real:   spec[e] *= weight[e] (complex x real; the mu^2 factor of corr.py:590-599), count elements.
 * cube_affine:     out[z, p] = a[z] df[z, p] + b[z] vf[z, p] + c[z] (corr.py:712-726); vf (and b) may be NULL.
 * raytrace_slices: scipy.ndimage.map_coordinates(cube, order=1, mode='constant') at the coordinates of
 *                  corr.py:744-768: out[i, ix, iy] = cube(zc[i], (tx[ix] scale[i]) / wx (n1-1) + (n1-1)/2,
 *                  (ty[iy] scale[i]) / wy (n2-1) + (n2-1)/2), 0 outside [0, n-1]; cube [n0, n1, n2],
 *                  out [numz, numx, numy].                                                         */
int corahip_spec_mul_real(corahip_ctx *ctx, double *spec, const double *weight, int64_t count);
int corahip_cube_affine(corahip_ctx *ctx, const double *df, const double *vf, const double *a, const double *b,
                        const double *c, int n0, int64_t plane, double *out);
int corahip_raytrace_slices(corahip_ctx *ctx, const double *cube, int n0, int n1, int n2, const double *zc,
                            const double *scale, const double *tx, const double *ty, double wx, double wy,
                            int numz, int numx, int numy, double *out);

/* ring geometry of the plan (host arrays of length 4 nside - 1), for tests */
int corahip_sht_plan_rings(const corahip_sht_plan *plan, int64_t *host_start, int32_t *host_nphi,
                           double *host_z, double *host_phi0);
/* ring-FFT class of every ring (host array of length 4 nside - 1), for tests and per-class reports: 0 = direct
 * transform (belt, power-of-two cap rings), else the Bluestein length the synthesis kernel of that ring runs
 * (a power of two, or 3 * 2^k for the rings whose 2 h - 1 fits it) */
int corahip_sht_plan_ring_classes(const corahip_sht_plan *plan, int32_t *host_len);
/* normalised associated Legendre values lambda_lm(cos theta_ring) the synthesis uses
 * (device recurrence incl. the polar seed table), l = m..lmax -> out [lmax-m+1] (device) */
int corahip_sht_lambda(corahip_ctx *ctx, const corahip_sht_plan *plan, int m, int ring_pair, double *out);
/* the same values as lane group kq (0..3) of the synthesis kernel forms them: zero in front of the group's entry row
 * R = m + 2 kq + 8 k >= lstart - 1, the plan's entry state there (four per (m, ring)), the recurrence behind it; equal
 * to corahip_sht_lambda from row max(R, lstart) on, and below the plan's cut before it.  For tests. */
int corahip_sht_lambda_entry(corahip_ctx *ctx, const corahip_sht_plan *plan, int m, int ring_pair, int kq, double *out);

#ifdef __cplusplus
}
#endif
#endif /* CORAHIP_H */
