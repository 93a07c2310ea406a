// sht_legendre.hip - K4: the Legendre contraction of the synthesis on FP64 MFMA, scalar (legendre_kernel) and
// spin-2 (legendre_pol_kernel) forms, and their launchers.  See sht_internal.h.
#include "sht_internal.h"

#ifndef LEG_ST_UNROLL
#define LEG_ST_UNROLL 1   // unroll factor of the stage loop (3 would make the LDS ring offsets immediates)
#endif
#ifndef LEG_RING_BLOCKS
#define LEG_RING_BLOCKS 0   // (measured neutral: 56.6 vs 56.5 ms, off) 1: a wave takes CONSECUTIVE rings of the tile (its own first contributing l), the two waves of a SIMD complementary blocks; 0: rings interleaved over the waves
#endif
#ifndef LEG_MS_UNROLL
#define LEG_MS_UNROLL 7   // (= LEG_KT / 8 at the shipped stage length; only the rolled head / tail stages use it) macro-step loop of a stage fully unrolled (loop counters and pointer increments become immediates): 71.0 -> 69.5 ms; factors 2 and 3: no change
#endif

#ifndef LEG_STAMPS
#define LEG_STAMPS 0      // diagnostic build (make k4stamps): s_memtime per phase of the item loop, summed over the workgroups' first waves
#endif
#if LEG_STAMPS
__device__ unsigned long long g_leg_stamps[16];
#define LEG_STAMP(acc)                                                                       \
    {                                                                                        \
        unsigned long long _t;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");           \
        acc += _t - leg_last;                                                                \
        leg_last = _t;                                                                       \
    }
#else
#define LEG_STAMP(acc)
#endif

// Lane roles (wave = 16 rings x 4 k-slots, the A operand of v_mfma_f64_16x16x4_f64):
// lane (ri = lane&15, kq = lane>>4) runs the recurrence of ring ri STAGGERED by 2 kq steps, so that
// at every macro-step (8 consecutive l, base l0) the first two values it produces are exactly the
// ones its k-slot must feed: lambda at l0+2kq (even l-m -> north+south accumulator) and l0+2kq+1
// (odd).  No cross-lane movement, no selects; the price is that the recurrence coefficients are
// no longer wave-uniform (4 distinct rows per step) - they are staged through LDS with the a_lm
// rows and read with one broadcast ds_read_b128 per step.
// The recurrence runs in its SCALED two-instruction form (round 3): lambda_l = s_l mu_l with s_m = s_{m+1} = 1,
// s_l = B_l s_{l-2}, so that  mu_l = (alpha_l x) mu_{l-1} - mu_{l-2},  alpha_l = A_l s_{l-1} / s_l  - a multiply and an
// fma per step instead of two multiplies and an fma - and only the two values a lane feeds to the MFMAs are scaled back
// (lambda = s mu): 18 instead of 24 DP instructions per macro-step.  `coef` holds (alpha_l, s_l), `seed` the mu pair at
// the first contributing l (sht_plan.hip: d_coefmu, d_seedmu); rows past lmax are zeros: s = 0 makes their A operand 0.
// Round 4: a lane ENTERS AT A WINDOW START.  `seed` is the plan's d_seed4: per (m, ring) four states (mu_{R-2},
// mu_{R-1}), one per lane group kq, in front of the first row R = m + 2 kq + 8 k >= lstart - 1 (seed_kernel).  The
// head stages' macro-steps then carry one compare + two selects per ring (is this window the entry?) instead of a
// compare + three selects per ROW (19.2k against 16.4k cycles per stage, 13 % of the kernel's time was in head
// stages), and the item prologue no longer walks lanes forward from the first contributing row with dependent loads.
// NT = 16-column tiles per wave, RT = 16-ring row tiles per wave (RT x NT x 2 parities = 16 accumulator
// tiles = 128 VGPRs in both shipped shapes: <8,1> for >= 128 columns, <4,2> for 64-column shards, where a
// second, independent recurrence per lane keeps the recurrence : MFMA ratio of the wide shape).
template <int NT, int RT>
__global__ void __launch_bounds__(64 * LEG_WAVES, LEG_WAVES <= 4 ? 2 : 1)
legendre_kernel(int lmax, int npair, int nring, int ncols, const double *__restrict__ z,
                const double2 *__restrict__ coef, const int32_t *__restrict__ lstart,
                const double2 *__restrict__ seed, const int2 *__restrict__ items, int nitems,
                const double *__restrict__ alm, const double *__restrict__ zeros, double *__restrict__ inter,
                unsigned *__restrict__ queue) {
    constexpr int TCOLS = 16 * NT;          // columns of this block
    constexpr int STRIDE = TCOLS + 8;       // LDS row stride (doubles): 2 rows apart = 128 B mod 256
    constexpr int CROWS = LEG_KT + 8;       // coefficient rows per stage (staggered lanes look 6 ahead)
    constexpr int STAGE = LEG_KT * STRIDE + 2 * CROWS;  // doubles per stage: a_lm rows + (A,B) pairs
    constexpr int RPW = LEG_KT / LEG_WAVES;             // a_lm rows each wave moves per stage
    constexpr int PIECES = RPW + 1;                     // LDS-DMA pieces per wave per stage (+ coefficients)
    constexpr int TRINGS = LEG_RINGS * RT;              // ring pairs per workgroup
    constexpr int RPM = RPW / (LEG_KT / 8);             // a_lm pieces each wave issues per macro-step
    static_assert(RPM * (LEG_KT / 8) == RPW && RPM >= 1 && RPM == LEG_RPM, "whole a_lm pieces per macro-step");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int *const s_next = reinterpret_cast<int *>(lds + LEG_NBUF * STAGE);  // next work item, two slots used in turn (carved after the ring)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ri = lane & 15, kq = lane >> 4;
    const int d = 2 * kq;
    const int L = lmax + 1;
    const int G = ncols >> 3;
    const long last_row = nalm_of(lmax) - 1;
    const unsigned lds_base_bytes = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;
    const bool odd_lane = lane & 1;

    // Rings of a tile -> waves.  Interleaved (ring = 8 j + wave, round 1) every wave sees the tile's whole range of first
    // contributing l, i.e. every wave starts at the tile's EARLIEST ring and multiplies zeros for the others.  With
    // consecutive blocks a wave starts at its own block's first l (its skip tests are per wave); so that the SIMDs
    // still finish a stage together, the two waves of a SIMD (waves w and w + 4) take complementary blocks (b, 7 - b):
    // the first l grows monotonically over the tile, so every SIMD gets the same total.
    auto ring_in_tile = [&](int j) {     // j-th ring (0 .. 16 RT - 1) of this wave
        if (LEG_RING_BLOCKS && LEG_WAVES == 8) return (wave < 4 ? wave : 11 - wave) * (16 * RT) + j;
        return j * LEG_WAVES + wave;
    };
    // Persistent workgroups.  Work item = (m, column group, ring tile); consecutive items are the ring
    // tiles of one a_lm slice, so the workgroups running at the same time share slices in L2 (they are
    // dealt over all 8 XCDs; packing a slice group onto ONE XCD was measured 24 % slower: every resident
    // workgroup of the XCD then hits the same 1-2 L2 channels in lock step; pairing the two ring tiles of a slice on
    // one XCD - LEG_XCD_PAIR of rounds 2-3 - was neutral and is gone).  Small m (long K) first.
    // `items` is the plan's list of the NON-EMPTY items in that order (sht_plan.hip: sht_k4_items; 27 % of the
    // (m, ring tile) pairs of cfg 3 have no contributing l at all), each with its first contributing l: decoding an
    // item is one 8-byte scalar load (rounds 1-3: two integer divisions and dependent loads from the first-l table).
    struct item_t {
        int m, cg, rtile, l_begin, nstage;
        long base_m;
        const double *src0;   // a_lm row (base_m + l_begin) of this column group (wave-uniform)
        int row_limit;        // rows behind src0 that exist: the last m's stage overhang is clamped to the last row
    };
    auto decode = [&](int it) {
        item_t w;
        const int2 e = items[it];                 // (m | rtile << 15 | cg << 23, first contributing l): wave-uniform
        w.m = e.x & 0x7fff;
        w.rtile = (e.x >> 15) & 0xff;
        w.cg = (e.x >> 23) & 0xff;
        const int lmin = e.y;
        w.l_begin = w.m + ((lmin - w.m) & ~7);
        w.nstage = (lmax - w.l_begin) / LEG_KT + 1;
        w.base_m = alm_idx(0, w.m, lmax);
        w.src0 = alm + (size_t)w.cg * TCOLS + (size_t)(w.base_m + w.l_begin) * ncols;
        w.row_limit = (int)(last_row - (w.base_m + w.l_begin));
        return w;
    };
    // LDS-DMA pieces: every wave issues exactly PIECES per stage (counted vmcnt): RPW a_lm rows (rows past
    // lmax are never used - their lambda is 0 - so any valid row is read, keeping the address scalar) and
    // the CROWS coefficient pairs (all waves write the same bytes).
    auto issue_row = [&](const item_t &w, int st, int rr) {
        const int row = wv + LEG_WAVES * rr;
        int r = st * LEG_KT + row;
        r = r < w.row_limit ? r : w.row_limit;
        // wave-uniform: scalar address arithmetic only (the byte offset of a row fits 32 bits: < 2^12 rows of < 2^13 B... x 8)
        const char *src = reinterpret_cast<const char *>(w.src0) + (unsigned)(r * ncols) * 8u;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % LEG_NBUF) * STAGE + row * STRIDE) * sizeof(double));
        if (lane < 8 * NT) glds16_s(src, 16u * lane, dst);
    };
    // The a_lm pieces of the steady-state stages ("fast" form: 3 instructions instead of 13).  Every instruction of
    // the macro-step loop costs issue time next to the MFMAs: the generic issue_row spends 8 scalar
    // instructions on the row address (clamp, multiply, 64-bit add, LDS offset) and 4 on saving / setting / restoring
    // M0.  Here the source is a per-STAGE scalar base + a per-lane byte offset that lives in a VGPR for the whole
    // kernel (one per piece of a stage: voff[p] = 16 lane + 8 p ncols LEG_WAVES), the LDS destination a per-stage
    // scalar + an immediate, written straight into M0 (nothing else in this kernel reads M0: the other LDS-DMA
    // helpers set it themselves and the compiler does not use it on gfx950; checked in the ISA).  Only taken when
    // every row of the refilled stage exists (no clamping): the stage loop's `refill_clean`.
    unsigned voff[RPW];
#pragma unroll
    for (int p = 0; p < RPW; p++) voff[p] = 16u * lane + (unsigned)(p * LEG_WAVES * ncols) * 8u;
    auto issue_row_fast = [&](const char *rsrc, unsigned rdst, auto pc) {
        constexpr int P = decltype(pc)::value;
        constexpr int IMM = P * LEG_WAVES * STRIDE * 8;
        const unsigned vo = voff[P];     // (asm operands inside a generic lambda do not capture: name a local)
        const bool on = NT == 8 || lane < 8 * NT;
        if (on)
            asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :
                         : "v"(vo), "s"(rsrc), "s"(rdst), "n"(IMM)
                         : "memory", "scc");     // (s_add_u32 writes SCC: undeclared, it ate the stage loop's compare; M0 cannot be
                                                 //  declared: hipcc treats it as reserved - "may not be preserved" - so the
                                                 //  contract stays the comment above, checked in the ISA by tools/asm_of.sh)
    };
    auto issue_coef = [&](const item_t &w, int st) {
        const int l = w.l_begin + st * LEG_KT + lane;
        const double *src = (l <= lmax) ? reinterpret_cast<const double *>(coef + w.base_m + l) : zeros;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % LEG_NBUF) * STAGE + LEG_KT * STRIDE) * sizeof(double));
        if (lane < CROWS) glds16(src, dst);
        if constexpr (CROWS > 64) {            // (stages of 64 rows: the 72 coefficient rows take a second instruction)
            const int l2 = l + 64;
            const double *src2 = (l2 <= lmax) ? reinterpret_cast<const double *>(coef + w.base_m + l2) : zeros;
            if (lane < CROWS - 64) glds16(src2, dst + 64 * 16);
        }
    };
    auto issue_stage = [&](const item_t &w, int st) {
        issue_coef(w, st);
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) issue_row(w, st, rr);
    };

    // dynamic work queue (one atomic per item, fetched one item ahead): items differ a lot in length
    // (polar ring tiles start late, large m is short), a static assignment left ~10 % on the table.
    // The first gridDim.x entries are pre-assigned; the queue starts behind them.
    const int nwg = (int)gridDim.x;
    auto clamp_item = [&](unsigned t) { return t < (unsigned)nitems ? (int)t : nitems; };
    // ring state of an item: cos(theta), first contributing l and the lane group's entry state (plan: d_seed4)
    auto load_rings = [&](const item_t &w, double (&x)[RT], int (&my_ls)[RT], double2 (&sd)[RT]) {
#pragma unroll
        for (int q = 0; q < RT; q++) {
            const int ring = w.rtile * TRINGS + ring_in_tile(ri + 16 * q);
            x[q] = 0.0;
            my_ls[q] = lmax + 1;
            sd[q] = make_double2(0.0, 0.0);
            if (ring < npair) {
                x[q] = z[ring];
                const long o = (long)w.m * npair + ring;
                my_ls[q] = lstart[o];
                sd[q] = seed[4 * o + kq];
            }
        }
    };
    int item = (int)blockIdx.x;
    if (item >= nitems) return;
    item_t w = decode(item);
    double x[RT];
    int my_ls[RT];
    double2 sd[RT];
    // order of the vector-memory operations at an item boundary (vmcnt retires in order): queue atomic, ring state,
    // first stage(s) of the next item, THEN the epilogue stores of the current one; the next item's prologue then has
    // nothing to load.  (Measured and dropped, round 4: a first-stage wait of vmcnt(<stores of a full tile>) that lets
    // the stores drain behind the next item's first macro-steps: +0.45 ms - the refill pieces of that stage queue behind
    // the stores; and an epilogue with a DPP exchange and scalar-base addressing, half the VALU instructions and no
    // LDS permutes: no change - the epilogue runs at the CU's store path, 32 B/clk for these 64-byte segments.)
    if (tid == 0) s_next[0] = clamp_item(nwg + atomicAdd(queue, 1u));
    load_rings(w, x, my_ls, sd);
#pragma unroll
    for (int st = 0; st < LEG_NBUF - 1; st++)
        if (st < w.nstage) issue_stage(w, st);
    int slot = 0;

#if LEG_STAMPS
    unsigned long long leg_last, t_pro = 0, t_head = 0, t_clean = 0, t_tail = 0, t_bnd = 0, t_epi = 0, t_bar2 = 0;
    unsigned long long n_head = 0, n_clean = 0, n_tail = 0, n_items = 0, t_tbar = 0, n_tail_ms = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(leg_last)::"memory");
#endif
    for (;;) {
        const int m = w.m;
        d4_t acce[RT][NT], acco[RT][NT];
#pragma unroll
        for (int q = 0; q < RT; q++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acce[q][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
                acco[q][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            }
        {
            // rings are dealt to the waves interleaved (ring = tile base + 8 (ri + 16 q) + wave) so that every
            // wave of the workgroup has the same mix of first-contributing l and reaches the barriers together
            double p0[RT], p1[RT];
            int inj_l[RT];
            int ls_min = lmax + 1;
#pragma unroll
            for (int q = 0; q < RT; q++) {
                ls_min = min(ls_min, my_ls[q]);
                // the lane's state is zero until its entry row (a window start of this lane group: >= l_begin + d)
                p0[q] = 0.0;
                p1[q] = 0.0;
                const int t = my_ls[q] - 1 - m - d;
                inj_l[q] = my_ls[q] <= lmax ? m + d + (t > 0 ? ((t + 7) >> 3) << 3 : 0) : 0x7fffffff;
            }
            // wave-uniform bounds of the start rows: the skip tests of the macro-step loop become scalar compares
            // (every vector instruction there costs issue time next to the MFMAs)
            int ws_min = ls_min, ws_maxinj = -1;
#pragma unroll
            for (int q = 0; q < RT; q++) ws_maxinj = max(ws_maxinj, inj_l[q] == 0x7fffffff ? -1 : inj_l[q]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ws_min = min(ws_min, __shfl_xor(ws_min, o));
                ws_maxinj = max(ws_maxinj, __shfl_xor(ws_maxinj, o));
            }
            ws_min = __builtin_amdgcn_readfirstlane(ws_min);
            ws_maxinj = __builtin_amdgcn_readfirstlane(ws_maxinj);
            // The ring's loads (z, first row, seed pair) are CONSUMED here.  Left to the compiler, the wait for the seed pair
            // lands at its only use - the injection branch inside the macro-step loop - and where its registers are
            // reused in the epilogue, as an s_waitcnt vmcnt(0): vmcnt is in order, so that wait also drains every LDS-DMA
            // piece in flight (the stage being refilled; in the epilogue the next item's first stages, just requested).
#pragma unroll
            for (int q = 0; q < RT; q++) asm volatile("" ::"v"(x[q]), "v"(sd[q].x), "v"(sd[q].y), "v"(my_ls[q]));

            // Stage loop in three consecutive pieces (same body, leg_stage_body.inc): the head stages in which rings of
            // the wave still start, the steady-state stages - every ring runs, none starts, every row <= lmax - whose
            // macro-steps carry no tests at all and are fully unrolled (every scalar compare / branch / loop increment
            // costs issue time next to the MFMAs), and the tail stage that overhangs lmax.
            // A stage is "clean" when (lo) every ring of the wave already runs and none starts in it - true from some
            // stage on - and (hi) all its rows are <= lmax, it refills the ring and every row of the stage it refills
            // exists (the fast pieces do not clamp) - true up to some stage.  Both bounds are found once per item, so
            // the steady-state loop's own condition is one scalar compare.
            auto stage_lo = [&](int st) {
                const int ls = w.l_begin + st * LEG_KT;
                return (ws_min <= ls) && (ws_maxinj < ls);
            };
            int st_hi = min(min((lmax - w.l_begin + 1) / LEG_KT, (w.row_limit + 1) / LEG_KT - LEG_NBUF + 1),
                            w.nstage - LEG_NBUF + 1);
            st_hi = __builtin_amdgcn_readfirstlane(st_hi);
            int st = 0;
            LEG_STAMP(t_pro);
#if LEG_STAMPS
            int st_mark = 0;
#endif
#define LEG_CLEAN 2
#define LEG_MS_PRAGMA _Pragma("unroll")
            for (; st < st_hi && !stage_lo(st); st++) {
#include "leg_stage_body.inc"
            }
#undef LEG_CLEAN
#undef LEG_MS_PRAGMA
            LEG_STAMP(t_head);
#if LEG_STAMPS
            n_head += st - st_mark, st_mark = st;
#endif
#define LEG_CLEAN 1
#define LEG_MS_PRAGMA _Pragma("unroll")
            for (; st < st_hi; st++) {
#include "leg_stage_body.inc"
            }
#undef LEG_CLEAN
#undef LEG_MS_PRAGMA
            LEG_STAMP(t_clean);
#if LEG_STAMPS
            n_clean += st - st_mark, st_mark = st;
#endif
#define LEG_CLEAN 0
#define LEG_MS_PRAGMA _Pragma("unroll 1")
            for (; st < w.nstage; st++) {
#include "leg_stage_body.inc"
            }
#undef LEG_CLEAN
#undef LEG_MS_PRAGMA
            LEG_STAMP(t_tail);
#if LEG_STAMPS
            n_tail += st - st_mark, n_items++;
#endif
        }

        // ---- next item: queue fetch for the one after it, its ring state and its first stages now, in front of this
        //      item's epilogue stores (see the note on the order above)
        const int cur_rtile = w.rtile, cur_cg = w.cg, cur_m = w.m;
        __syncthreads();  // all waves are done reading the stage ring; s_next[slot] (written during the previous boundary) is visible
        item = __builtin_amdgcn_readfirstlane(s_next[slot]);
        const bool have_next = item < nitems;
        unsigned fetched = 0;
        double xn[RT];
        int lsn[RT];
        double2 sdn[RT];
        if (have_next) {
            if (tid == 0) fetched = atomicAdd(queue, 1u);     // (its value is only needed behind the epilogue)
            w = decode(item);
            load_rings(w, xn, lsn, sdn);
#pragma unroll
            for (int st = 0; st < LEG_NBUF - 1; st++)
                if (st < w.nstage) issue_stage(w, st);
        }
        LEG_STAMP(t_bnd);

        // ---- epilogue: north = even + odd, south mirror = even - odd.  Adjacent lanes (columns n, n+1 of the
        //      same rows) swap one value each so that every lane stores 16 bytes: half the store instructions.
        //      (Tiles with no contributing l at all are not in the item list: K5 never reads cells with m >= mcut(ring).)
        {
#pragma unroll
            for (int q = 0; q < RT; q++)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int col = cur_cg * TCOLS + 16 * t + (ri & ~1);  // even column of the lane pair
                    const int g = col >> 3, cv = col & 7;
#pragma unroll
                    for (int rp = 0; rp < 2; rp++) {
                        const int r0 = 2 * rp, r1 = 2 * rp + 1;
                        const double n0 = acce[q][t][r0] + acco[q][t][r0], n1 = acce[q][t][r1] + acco[q][t][r1];
                        const double s0 = acce[q][t][r0] - acco[q][t][r0], s1 = acce[q][t][r1] - acco[q][t][r1];
                        // even lane keeps row r0 and sends its r1 value; odd lane keeps row r1 and sends its r0 value
                        const double nrecv = __shfl_xor(odd_lane ? n0 : n1, 1);
                        const double srecv = __shfl_xor(odd_lane ? s0 : s1, 1);
                        const int rr = odd_lane ? r1 : r0;
                        const int ro = cur_rtile * TRINGS + ring_in_tile(kq + 4 * rr + 16 * q);
#if LEG_ABLATE == 4  // diagnostic: no epilogue stores (unless a value is absurd: keeps the arithmetic alive)
                        if (ro < npair && n0 == 1.2345e300) {
#else
                        if (ro < npair) {
#endif
                            const double2 nv = odd_lane ? make_double2(nrecv, n1) : make_double2(n0, nrecv);
                            *reinterpret_cast<double2 *>(inter + (((size_t)ro * G + g) * L + cur_m) * 8 + cv) = nv;
                            const int rs = nring - 1 - ro;
                            if (rs != ro) {
                                const double2 sv = odd_lane ? make_double2(srecv, s1) : make_double2(s0, srecv);
                                *reinterpret_cast<double2 *>(inter + (((size_t)rs * G + g) * L + cur_m) * 8 + cv) = sv;
                            }
                        }
                    }
                }
        }
        LEG_STAMP(t_epi);
        if (!have_next) break;
        // the other slot: its last readers passed the barrier above
        if (tid == 0) s_next[slot ^ 1] = clamp_item(nwg + fetched);
        slot ^= 1;
#pragma unroll
        for (int q = 0; q < RT; q++) x[q] = xn[q], my_ls[q] = lsn[q], sd[q] = sdn[q];
        LEG_STAMP(t_bar2);
    }
#if LEG_STAMPS
    if (tid == 0) {
        atomicAdd(&g_leg_stamps[0], t_pro), atomicAdd(&g_leg_stamps[1], t_head), atomicAdd(&g_leg_stamps[2], t_clean);
        atomicAdd(&g_leg_stamps[3], t_tail), atomicAdd(&g_leg_stamps[4], t_bnd), atomicAdd(&g_leg_stamps[5], t_epi);
        atomicAdd(&g_leg_stamps[6], t_bar2), atomicAdd(&g_leg_stamps[7], n_head), atomicAdd(&g_leg_stamps[8], n_clean);
        atomicAdd(&g_leg_stamps[9], n_tail), atomicAdd(&g_leg_stamps[10], n_items);
        atomicAdd(&g_leg_stamps[11], t_tbar), atomicAdd(&g_leg_stamps[12], n_tail_ms);
    }
#endif
}

// K4 for polarisation (spin 2): (E, B) -> (Q, U), what healpy.alm2map([T, E, B]) does for Q and U behind
// hputil.sphtrans_inv_real_pol (cora/util/hputil.py:394-432).  Same structure as legendre_kernel (persistent
// workgroups + queue, staggered-lane recurrence, LDS-DMA stage ring, even/odd parity accumulators), with the two
// A operands per parity derived from the scalar recurrence:
//     W_lm = (g1 r1 + g2) lambda_l + g3 r2 lambda_{l-1},     X_lm = g4 r2 lambda_l - m g3 r1 lambda_{l-1},
//     r1 = 1/sin^2, r2 = cos/sin^2 per ring (lane),  (g1..g4)(l, m) from the plan's table (staged through LDS):
//     g1 = -N2 (l - m^2), g2 = -N2 l(l-1)/2, g3 = N2 (2l+1)/A_l, g4 = N2 m (l-1), N2 = 2/sqrt((l+2)(l+1)l(l-1)).
// The kernel accumulates S_W = sum_l W a and S_X = sum_l X a for the natural columns; channels are interleaved
// (E_f, B_f) so that one a_lm cell [re x4 | im x4] holds (Re E, Re B, .., Im E, Im B, ..) and
//     Re Q = -(S_W[ReE] + S_X[ImB]),  Im Q = -(S_W[ImE] - S_X[ReB]),  Re U = -(S_W[ReB] - S_X[ImE]),  Im U = -(S_W[ImB] + S_X[ReE])
// is a lane-xor-5 exchange in the epilogue; Q_f, U_f leave in the cell positions of E_f, B_f.  W has the parity
// (-1)^{l+m} of lambda under theta -> pi - theta, X the opposite: north = (W_e + W_o, X_e + X_o), south = (W_e - W_o, X_o - X_e).
template <int NT>
__global__ void __launch_bounds__(64 * LEG_WAVES)
legendre_pol_kernel(int lmax, int npair, int nring, int ncols, const double *__restrict__ z,
                    const double *__restrict__ sth, const double2 *__restrict__ coef,
                    const double *__restrict__ polc, const int32_t *__restrict__ lstart,
                    const double2 *__restrict__ seed, const int32_t *__restrict__ lmin_tab,
                    const double *__restrict__ alm, const double *__restrict__ zeros, double *__restrict__ inter,
                    unsigned *__restrict__ queue) {
    constexpr int KT = 32;                  // l rows per stage (the 4 accumulator sets leave room for NT = 4 only)
    constexpr int NBUF = 3;
    constexpr int TCOLS = 16 * NT;
    constexpr int STRIDE = TCOLS + 8;
    constexpr int CROWS = KT + 8;
    constexpr int STAGE = KT * STRIDE + 2 * CROWS + 4 * CROWS;   // a_lm rows + (A, B) pairs + (g1..g4) rows
    constexpr int RPW = KT / LEG_WAVES;
    constexpr int PIECES = RPW + 3;         // a_lm rows + the (A, B) piece + two pieces of the g table (see issue_coef)
    constexpr int TRINGS = LEG_RINGS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int &s_next = *reinterpret_cast<int *>(lds + NBUF * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ri = lane & 15, kq = lane >> 4;
    const int d = 2 * kq;
    const int L = lmax + 1;
    const int G = ncols >> 3;
    const int ntile128 = (npair + LMIN_RINGS - 1) / LMIN_RINGS;
    const int ntile = (npair + TRINGS - 1) / TRINGS;
    const int ncg = ncols / TCOLS;
    const int nitems = L * ncg * ntile;
    const long last_row = nalm_of(lmax) - 1;
    const unsigned lds_base_bytes = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds;

    struct item_t {
        int m, cg, rtile, l_begin, nstage;
        long base_m;
        const double *src0;   // a_lm row (base_m + l_begin) of this column group (wave-uniform)
        int row_limit;        // rows behind src0 that exist
    };
    auto decode = [&](int it) {
        item_t w;
        const int gidx = it / ntile;
        w.rtile = it - gidx * ntile;
        w.m = gidx / ncg;
        w.cg = gidx - w.m * ncg;
        int lmin = lmax + 1;
        const int t_first = (w.rtile * TRINGS) / LMIN_RINGS;
        const int t_last = min((w.rtile * TRINGS + TRINGS - 1) / LMIN_RINGS, ntile128 - 1);
        for (int t128 = t_first; t128 <= t_last; t128++) lmin = min(lmin, lmin_tab[w.m * ntile128 + t128]);
        w.l_begin = w.m + ((lmin - w.m) & ~7);
        w.nstage = lmin <= lmax ? (lmax - w.l_begin) / KT + 1 : 0;
        w.base_m = alm_idx(0, w.m, lmax);
        w.src0 = alm + (size_t)w.cg * TCOLS + (size_t)(w.base_m + w.l_begin) * ncols;
        w.row_limit = (int)(last_row - (w.base_m + w.l_begin));
        return w;
    };
    auto issue_row = [&](const item_t &w, int st, int rr) {
        const int row = wv + LEG_WAVES * rr;
        int r = st * KT + row;
        r = r < w.row_limit ? r : w.row_limit;
        const char *src = reinterpret_cast<const char *>(w.src0) + (unsigned)(r * ncols) * 8u;   // scalar arithmetic only
        const unsigned dst = lds_base_bytes + (unsigned)(((st % NBUF) * STAGE + row * STRIDE) * sizeof(double));
        if (lane < 8 * NT) glds16_s(src, 16u * lane, dst);
    };
    // coefficient pieces of a stage: CROWS (A, B) pairs (16 B each) and CROWS (g1..g4) rows (32 B each = 2 CROWS
    // 16-byte chunks, contiguous in the table): 1 + 2 wave-instructions (CROWS = 40: 40 + 80 lanes)
    auto issue_coef = [&](const item_t &w, int st) {
        const int l0 = w.l_begin + st * KT;
        const int l = l0 + lane;
        const double *src = (l <= lmax) ? reinterpret_cast<const double *>(coef + w.base_m + l) : zeros;
        const unsigned dst = lds_base_bytes + (unsigned)(((st % NBUF) * STAGE + KT * STRIDE) * sizeof(double));
        if (lane < CROWS) glds16(src, dst);
        const double *gsrc = polc + 4 * (size_t)(w.base_m + l0);     // the table is padded: rows past the end exist
        const unsigned gdst = dst + (unsigned)(2 * CROWS * sizeof(double));
        glds16(gsrc + 2 * lane, gdst);
        if (lane < 2 * CROWS - 64) glds16(gsrc + 2 * (64 + lane), gdst + 64 * 16);
    };
    auto issue_stage = [&](const item_t &w, int st) {
        issue_coef(w, st);
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) issue_row(w, st, rr);
    };

    int item = blockIdx.x;
    if (item >= nitems) return;
    item_t w = decode(item);
#pragma unroll
    for (int st = 0; st < NBUF - 1; st++)
        if (st < w.nstage) issue_stage(w, st);

    for (;;) {
        const int m = w.m;
        const double mval = (double)m;
        if (tid == 0) s_next = (int)(gridDim.x + atomicAdd(queue, 1u));
        d4_t awe[NT], awo[NT], axe[NT], axo[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            awe[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            awo[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            axe[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
            axo[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
        }
        if (w.nstage > 0) {
            const int ring = w.rtile * TRINGS + ri * LEG_WAVES + wave;
            double x = 0.0, r1 = 0.0, r2 = 0.0, p0 = 0.0, p1 = 0.0;
            double2 sd = make_double2(0.0, 0.0);
            int my_ls = lmax + 1;
            if (ring < npair) {
                x = z[ring];
                const double s = sth[ring];
                r1 = 1.0 / (s * s);
                r2 = x * r1;
                const long o = (long)m * npair + ring;
                my_ls = lstart[o];
                sd = seed[o];
            }
            int ls_min = my_ls;                  // wave-uniform below: the skip tests become scalar compares
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ls_min = min(ls_min, __shfl_xor(ls_min, o));
            ls_min = __builtin_amdgcn_readfirstlane(ls_min);
            int inj_l = my_ls;
            const double2 *cf = coef + w.base_m;
            {
                const int lf = w.l_begin + d;
                if (my_ls < lf) {
                    p0 = sd.x;
                    p1 = sd.y;
                    for (int l = my_ls + 1; l < lf; l++) {
                        const double2 c = (l <= lmax) ? cf[l] : make_double2(0.0, 0.0);
                        const double vv = fma(c.x * x, p1, -(c.y * p0));
                        p0 = p1;
                        p1 = vv;
                    }
                    inj_l = 0x7fffffff;
                }
            }
            int ws_maxinj = inj_l == 0x7fffffff ? -1 : inj_l;   // wave-uniform: last row at which a ring of the wave starts
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ws_maxinj = max(ws_maxinj, __shfl_xor(ws_maxinj, o));
            ws_maxinj = __builtin_amdgcn_readfirstlane(ws_maxinj);
            asm volatile("" ::"v"(x), "v"(sd.x), "v"(sd.y), "v"(my_ls), "v"(r1), "v"(r2));   // (the ring's loads are consumed here: see legendre_kernel)
            auto stage_clean = [&](int st) {
                const int ls = w.l_begin + st * KT;
                return (ls + KT - 1 <= lmax) && (ls_min <= ls) && (ws_maxinj < ls);
            };
            int st = 0;
#define LEGP_CLEAN 0
#define LEGP_MS_PRAGMA _Pragma("unroll 1")
            for (; st < w.nstage && !stage_clean(st); st++) {
#include "legpol_stage_body.inc"
            }
#undef LEGP_CLEAN
#undef LEGP_MS_PRAGMA
#define LEGP_CLEAN 1
#define LEGP_MS_PRAGMA _Pragma("unroll")
            for (; st + NBUF - 1 < w.nstage && stage_clean(st); st++) {
#include "legpol_stage_body.inc"
            }
#undef LEGP_CLEAN
#undef LEGP_MS_PRAGMA
#define LEGP_CLEAN 0
#define LEGP_MS_PRAGMA _Pragma("unroll 1")
            for (; st < w.nstage; st++) {
#include "legpol_stage_body.inc"
            }
#undef LEGP_CLEAN
#undef LEGP_MS_PRAGMA
        }

        const int cur_rtile = w.rtile, cur_cg = w.cg, cur_m = w.m, cur_nstage = w.nstage;
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_next);
        const bool have_next = item < nitems;
        if (have_next) {
            w = decode(item);
#pragma unroll
            for (int st = 0; st < NBUF - 1; st++)
                if (st < w.nstage) issue_stage(w, st);
        }
        if (cur_nstage > 0) {
            // sigma of the column inside its 8-wide cell: +1 for (Re B, Im E) columns, -1 for (Re E, Im B)
            const int c8 = ri & 7;
            const double sigma = (((c8 & 1) ^ ((c8 >> 2) & 1)) != 0) ? 1.0 : -1.0;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int col = cur_cg * TCOLS + 16 * t + ri;
                const int g = col >> 3, cv = col & 7;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double swn = awe[t][r] + awo[t][r], sws = awe[t][r] - awo[t][r];
                    const double sxn = axe[t][r] + axo[t][r], sxs = axo[t][r] - axe[t][r];
                    const double pxn = __shfl_xor(sxn, 5), pxs = __shfl_xor(sxs, 5);
                    const int ro = cur_rtile * TRINGS + (kq + 4 * r) * LEG_WAVES + wave;
                    if (ro < npair) {
                        inter[(((size_t)ro * G + g) * L + cur_m) * 8 + cv] = -(swn - sigma * pxn);
                        const int rs = nring - 1 - ro;
                        if (rs != ro) inter[(((size_t)rs * G + g) * L + cur_m) * 8 + cv] = -(sws - sigma * pxs);
                    }
                }
            }
        }
        if (!have_next) break;
        __syncthreads();
    }
}
template <int NT, int RT>
static int launch_legendre(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *alm, double *inter) {
    constexpr int STRIDE = 16 * NT + 8;
    const size_t shm = sizeof(double) * LEG_NBUF * (LEG_KT * STRIDE + 2 * (LEG_KT + 8)) + 16;
    HIP_TRY(hipFuncSetAttribute((const void *)legendre_kernel<NT, RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    const int2 *d_items = nullptr;
    int nitems = 0;
    int rc = sht_k4_items(ctx, p, RT, ncols / (16 * NT), &d_items, &nitems);
    if (rc) return rc;
    if (nitems == 0) return 0;
    // persistent: as many workgroups as fit (LDS-limited: one per CU for NT = 8)
    const int per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / shm)));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipMemsetAsync(p->d_queue, 0, 1024, ctx->stream));
    legendre_kernel<NT, RT><<<grid, 64 * LEG_WAVES, shm, ctx->stream>>>(p->lmax, p->npair, p->nring, ncols, p->d_z,
                                                                       p->d_coefmu, p->d_lstart, p->d_seed4, d_items, nitems,
                                                                       alm, p->d_zeros, inter, p->d_queue);
    LAUNCH_CHECK();
#if LEG_STAMPS
    {
        unsigned long long hs[16], zero[16] = {0};
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_leg_stamps), sizeof(hs)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_leg_stamps), zero, sizeof(zero)));
        const double tot = (double)(hs[0] + hs[1] + hs[2] + hs[3] + hs[4] + hs[5] + hs[6] + hs[11]);
        fprintf(stderr, "K4 stamps: tail stages: wait + barrier %.0f ticks per stage, %.2f macro-steps with MFMAs per stage\n",
                hs[11] / std::max<double>(1.0, (double)hs[9]), hs[12] / std::max<double>(1.0, (double)hs[9]));
        fprintf(stderr, "K4 stamps <%d,%d> wg=%u items=%llu: prologue %.3f head %.3f (%llu st) clean %.3f (%llu st) tail %.3f (%llu st) "
                "boundary %.3f epilogue %.3f barrier2 %.3f | ticks/item %.0f ticks/clean-stage %.0f ticks/head-stage %.0f ticks/tail-stage %.0f\n",
                NT, RT, grid.x, hs[10], hs[0] / tot, hs[1] / tot, hs[7], hs[2] / tot, hs[8], hs[3] / tot, hs[9], hs[4] / tot, hs[5] / tot,
                hs[6] / tot, tot / std::max<double>(1.0, (double)hs[10]), hs[2] / std::max<double>(1.0, (double)hs[8]),
                hs[1] / std::max<double>(1.0, (double)hs[7]), hs[3] / std::max<double>(1.0, (double)hs[9]));
    }
#endif
    return 0;
}
int sht_legendre(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *alm, double *inter) {
    StageTimer t(ctx, "legendre");
    const int ntile = ncols / 16;
    if (ntile % 8 == 0) return launch_legendre<8, 1>(ctx, p, ncols, alm, inter);
    if (ntile % 4 == 0) return launch_legendre<4, 2>(ctx, p, ncols, alm, inter);
    if (ntile % 2 == 0) return launch_legendre<2, 2>(ctx, p, ncols, alm, inter);
    return launch_legendre<1, 2>(ctx, p, ncols, alm, inter);
}

// (g1..g4)(l, m) of legendre_pol_kernel at alm_idx(l, m), long double on the host, once per plan
static int ensure_polc(corahip_ctx *ctx, corahip_sht_plan *p) {
    if (p->d_polc) return 0;
    const int lmax = p->lmax;
    std::vector<double> g((size_t)(p->nalm + 64) * 4, 0.0);
    for (int m = 0; m <= lmax; m++)
        for (int l = std::max(m, 2); l <= lmax; l++) {
            const long double ll = l, mm = m;
            const long double n2 = 2.0L / sqrtl((ll + 2.0L) * (ll + 1.0L) * ll * (ll - 1.0L));
            const long double al = l > m ? sqrtl((4.0L * ll * ll - 1.0L) / (ll * ll - mm * mm)) : 0.0L;
            double *o = &g[(size_t)alm_idx(l, m, lmax) * 4];
            o[0] = (double)(-n2 * (ll - mm * mm));
            o[1] = (double)(-n2 * ll * (ll - 1.0L) / 2.0L);
            o[2] = l > m ? (double)(n2 * (2.0L * ll + 1.0L) / al) : 0.0;
            o[3] = (double)(n2 * mm * (ll - 1.0L));
        }
    HIP_TRY(hipMalloc((void **)&p->d_polc, sizeof(double) * g.size()));
    HIP_TRY(hipMemcpyAsync(p->d_polc, g.data(), sizeof(double) * g.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

template <int NT>
static int launch_legendre_pol(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *alm, double *inter) {
    constexpr int STRIDE = 16 * NT + 8;
    const size_t shm = sizeof(double) * 3 * (32 * STRIDE + 6 * (32 + 8)) + 16;
    HIP_TRY(hipFuncSetAttribute((const void *)legendre_pol_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    const int ntile = (p->npair + LEG_RINGS - 1) / LEG_RINGS;
    const long nitems = (long)p->L * (ncols / (16 * NT)) * ntile;
    const int per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / shm)));
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu * per_cu));
    HIP_TRY(hipMemsetAsync(p->d_queue, 0, 64, ctx->stream));
    legendre_pol_kernel<NT><<<grid, 64 * LEG_WAVES, shm, ctx->stream>>>(p->lmax, p->npair, p->nring, ncols, p->d_z, p->d_sth,
                                                                       p->d_coef, p->d_polc, p->d_lstart, p->d_seed,
                                                                       p->d_lmin, alm, p->d_zeros, inter, p->d_queue);
    LAUNCH_CHECK();
    return 0;
}
int sht_ensure_polc(corahip_ctx *ctx, corahip_sht_plan *p) { return ensure_polc(ctx, p); }
int sht_legendre_pol(corahip_ctx *ctx, corahip_sht_plan *p, int ncols, const double *alm, double *inter) {
    int rc = ensure_polc(ctx, p);
    if (rc) return rc;
    StageTimer t(ctx, "legendre_pol");
    const int ntile = ncols / 16;
    if (ntile % 4 == 0) return launch_legendre_pol<4>(ctx, p, ncols, alm, inter);
    if (ntile % 2 == 0) return launch_legendre_pol<2>(ctx, p, ncols, alm, inter);
    return launch_legendre_pol<1>(ctx, p, ncols, alm, inter);
}
