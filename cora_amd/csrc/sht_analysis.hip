// sht_analysis.hip - K4^T: the adjoint Legendre contraction of healpy.map2alm's quadrature pass on FP64 MFMA and the
// deterministic reduction over ring tiles.  See sht_internal.h.
#include "sht_internal.h"

#ifndef ADJ_MU
#define ADJ_MU 1         // 1: scaled two-instruction recurrence (mu form), scale applied per output row; 0: (A, B) form (A/B builds)
#endif
#ifndef ADJ_ROWS_REDUCE
#define ADJ_ROWS_REDUCE 1   // 1: the tile reduction goes row by row (m) and skips what a ring tile cannot reach; 0: flat sum over zero-filled partial buffers (A/B)
#endif
#ifndef ADJ_CUNROLL
#define ADJ_CUNROLL 8    // recurrence steps whose (scalar-loaded) coefficients are fetched together; measured 8 / 16 / 32: 25.1 / 26.9 / 25.5 ms
#endif

// K4^T  legendre_adj_kernel: a_lm(col) = sum_rings lambda_lm(ring) [G_m(north) + (-1)^{l+m} G_m(south)](col)
// on FP64 MFMA with M = l, K = ring pairs, N = columns (channel re/im).  Work item = (m, 16 NCT columns,
// tile of 512 ring pairs).  A wave owns 64 ring pairs: lane = ring steps the recurrence once per l (no
// redundancy), the 32 lambda values of an l-block go through a wave-private LDS transpose ([ring][parity][16])
// into the A-operand layout (16 same-parity l x 4 rings), and the wave's G tile (64 rings x 16 NCT columns, even = N+S and
// odd = N-S combinations) stays in REGISTERS as the B operand for the whole item.  The eight waves hold
// different rings, so their [32 l x 16 NCT] partial sums are added through LDS once per l-block; the four
// ring tiles of an (m, column group) go to separate partial buffers summed by alm_reduce_kernel
// (deterministic - no atomics).
// Round 5, tried and withdrawn: ONE workgroup walking the four ring tiles of its (m, column group) and adding the later
// tiles' sums onto its own earlier stores (no partial buffers, no alm_reduce_kernel: 1.9 of 21 ms).  Correct, but the
// kernel sits at 256 VGPRs (128 of them the G tile) and the restructured loop made the allocator spill B operands into
// the MFMA loop (scratch 28 -> 450-520 bytes): 20.96 ms against 21.05, a single l block per reduction 24.5 ms.
// Round 5: the recurrence runs in the synthesis kernel's scaled two-instruction form mu_l = (alpha_l x) mu_{l-1} - mu_{l-2}
// (plan tables coefmu / seedmu); lambda_l = s_l mu_l with s_l the same for every ring, so the MFMAs contract mu and the scale
// is applied once per OUTPUT row where the eight waves' partial tiles are added: one DP multiply less per recurrence step.
template <int NCT>
__global__ void __launch_bounds__(512)
legendre_adj_kernel(int lmax, int npair, int nring, int ncols, const double *__restrict__ z,
                    const double2 *__restrict__ coef, const int32_t *__restrict__ lstart,
                    const double2 *__restrict__ seed, const int32_t *__restrict__ lmin_tab,
                    const int32_t *__restrict__ mcut, const double *__restrict__ inter, double *__restrict__ part,
                    unsigned *__restrict__ queue) {
    constexpr int TCOLS = 16 * NCT;
    constexpr int TRINGS = 64 * ADJ_WAVES;     // 512 ring pairs per workgroup
    constexpr int LB = 32;                     // l per block: 16 even + 16 odd (l - m)
    constexpr int LSTR = LB + 1;               // LDS row stride of the transpose: conflict-free both ways
    constexpr int WREG = 64 * LSTR;            // doubles per wave
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int &s_next = *reinterpret_cast<int *>(lds + ADJ_WAVES * WREG);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    const int L = lmax + 1;
    const int G = ncols >> 3;
    const long nalm = nalm_of(lmax);
    const int ntile128 = (npair + LMIN_RINGS - 1) / LMIN_RINGS;
    const int ntile = (npair + TRINGS - 1) / TRINGS;
    const int ncg = ncols / TCOLS;
    const int nitems = L * ncg * ntile;
    double *lamw = lds + wave * WREG;

    int item = blockIdx.x;
    while (item < nitems) {
        if (tid == 0) s_next = (int)(gridDim.x + atomicAdd(queue, 1u));
        const int gidx = item / ntile;
        const int rtile = item - gidx * ntile;
        const int m = gidx / ncg;
        const int cg = gidx - m * ncg;
        int lmin = lmax + 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int t128 = rtile * 4 + q;
            if (t128 < ntile128) lmin = min(lmin, lmin_tab[m * ntile128 + t128]);
        }
        const long base_m = alm_idx(0, m, lmax);
        const int lb0 = lmin <= lmax ? m + ((lmin - m) & ~(LB - 1)) : lmax + 1;
        double *pout = part + ((size_t)rtile * nalm + base_m) * ncols + (size_t)cg * TCOLS;
#if !ADJ_ROWS_REDUCE
        // multipoles this tile cannot reach contribute zero
        for (int e = tid; e < (lb0 - m) * TCOLS; e += 512) pout[(size_t)(m + e / TCOLS) * ncols + e % TCOLS] = 0.0;
#endif
        // (ADJ_ROWS_REDUCE: the rows below lb0 are neither written here nor read by alm_reduce_rows_kernel, which forms the
        //  same lb0 per (m, ring tile) - a fifth of the partial buffers' traffic was zeros)

        if (lb0 <= lmax) {
            // ---- this wave's G tile -> registers (B operand): k-step s covers rings 4s..4s+3 of the wave
            double ge[16][NCT], go[16][NCT];
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int ro = rtile * TRINGS + (4 * s + kq) * ADJ_WAVES + wave;
                const bool ok = ro < npair && m < mcut[min(ro, npair - 1)];
                const int rs = nring - 1 - ro;
#pragma unroll
                for (int t = 0; t < NCT; t++) {
                    const int col = cg * TCOLS + 16 * t + ri;
                    const size_t off = ((size_t)(col >> 3) * L + m) * 8 + (col & 7);
                    double gn = 0.0, gsv = 0.0;
                    if (ok) {
                        gn = inter[(size_t)ro * G * L * 8 + off];
                        if (rs != ro) gsv = inter[(size_t)rs * G * L * 8 + off];
                    }
                    ge[s][t] = gn + gsv;
                    go[s][t] = gn - gsv;
                }
            }
            // ---- recurrence state of this lane's ring
            const int ring = rtile * TRINGS + lane * ADJ_WAVES + wave;
            double x = 0.0, p0 = 0.0, p1 = 0.0;
            double2 sd = make_double2(0.0, 0.0);
            int my_ls = lmax + 1;
            if (ring < npair) {
                x = z[ring];
                const long o = (long)m * npair + ring;
                my_ls = lstart[o];
                sd = seed[o];
            }
            // first l of each group of 4 lanes (= one MFMA k-step): lets whole k-steps be skipped
            int ls4 = min(my_ls, __shfl_xor(my_ls, 1));
            ls4 = min(ls4, __shfl_xor(ls4, 2));
            const double2 *cf = coef + base_m;

            // one 32-l block: recurrence of this lane's ring -> transpose buffer -> MFMAs into `acc`
            auto do_block = [&](const int lb, d4_t (&acc)[2][NCT]) __attribute__((always_inline)) {
                const unsigned long long act = __ballot(ls4 <= lb + LB - 1);   // bit 4s: k-step s has a started ring
                // lambda_{lb .. lb+31} of this lane's ring -> transpose buffer [ring][l - lb]
                // (coefficients are read unconditionally - the table is padded by 32 entries - so that the scalar
                // loads of a whole unrolled group are issued together; rows past lmax are discarded below)
                // the seed injection (three selects and a compare per step) is only compiled into the blocks in
                // which some ring of the wave actually starts: 4 instead of 11 VALU instructions per step elsewhere
                if (__any(my_ls >= lb && my_ls < lb + LB)) {
#pragma unroll 8
                    for (int j = 0; j < LB; j++) {
                        const int l = lb + j;
                        const double2 c = cf[l];
#if ADJ_MU
                        double vv = fma(c.x * x, p1, -p0);
#else
                        double vv = fma(c.x * x, p1, -(c.y * p0));
#endif
                        const bool inj = (l == my_ls);
                        vv = inj ? sd.y : vv;
                        p0 = inj ? sd.x : p1;
                        p1 = vv;
                        lamw[lane * LSTR + (j & 1) * 16 + (j >> 1)] = vv;   // row = [16 even l | 16 odd l]
                    }
                } else {
#pragma unroll ADJ_CUNROLL
                    for (int j = 0; j < LB; j++) {
                        const double2 c = cf[lb + j];
#if ADJ_MU
                        const double vv = fma(c.x * x, p1, -p0);
#else
                        const double vv = fma(c.x * x, p1, -(c.y * p0));
#endif
                        p0 = p1;
                        p1 = vv;
                        lamw[lane * LSTR + (j & 1) * 16 + (j >> 1)] = vv;
                    }
                }
                // A operands are read one k-step AHEAD of the MFMAs that use them (the read of step s+1 is in flight
                // behind the four MFMAs of step s; with the read inside the skip branch every group of MFMAs waited
                // out a full LDS latency first).  Parities in separate halves of the row: each read is unit-stride over
                // the 16 lanes of a k-slot (interleaved, the pair became one ds_read2_b64 whose 16-lane groups stride
                // 4 dwords over 32 banks: 2-way conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25)
                double ae_n = lamw[kq * LSTR + ri], ao_n = lamw[kq * LSTR + 16 + ri];
                if (act == ~0ull) {
                    // every k-step has a started ring (all blocks but the first few of an item): no skip tests
#pragma unroll
                    for (int s = 0; s < 16; s++) {
                        const double ae = ae_n, ao = ao_n;
                        if (s + 1 < 16) {
                            ae_n = lamw[(4 * (s + 1) + kq) * LSTR + ri];
                            ao_n = lamw[(4 * (s + 1) + kq) * LSTR + 16 + ri];
                        }
#pragma unroll
                        for (int t = 0; t < NCT; t++) {
                            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, ge[s][t], acc[0][t], 0, 0, 0);
                            acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, go[s][t], acc[1][t], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; s++) {
                        const double ae = ae_n, ao = ao_n;
                        if (s + 1 < 16) {
                            ae_n = lamw[(4 * (s + 1) + kq) * LSTR + ri];
                            ao_n = lamw[(4 * (s + 1) + kq) * LSTR + 16 + ri];
                        }
                        if (!((act >> (4 * s)) & 1ull)) continue;
#pragma unroll
                        for (int t = 0; t < NCT; t++) {
                            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, ge[s][t], acc[0][t], 0, 0, 0);
                            acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, go[s][t], acc[1][t], 0, 0, 0);
                        }
                    }
                }
            };
            // two blocks share one cross-wave reduction (half the barriers; the partial tiles of both fit the wave's
            // transpose buffer exactly)
            for (int lb = lb0; lb <= lmax; lb += 2 * LB) {
                d4_t acc0[2][NCT], acc1[2][NCT];
#if ADJ_MU
                // the row scales s_l of this thread's outputs of the reduction below, requested now: behind the barrier
                // their latency would be exposed to the whole workgroup
                double srow[2 * NCT];
#pragma unroll
                for (int u = 0; u < 2 * NCT; u++) {
                    const int e = tid + 512 * u;
                    const int el = e & 63, r = (e >> 6) & 3, q = e >> 8, par = (q / NCT) & 1, blk = q / (2 * NCT);
                    srow[u] = cf[min(lb + blk * LB + 2 * ((el >> 4) + 4 * r) + par, lmax + 31)].y;   // (the table is padded by 32 entries; rows past lmax are discarded)
                }
#endif
#pragma unroll
                for (int par = 0; par < 2; par++)
#pragma unroll
                    for (int t = 0; t < NCT; t++) acc0[par][t] = acc1[par][t] = (d4_t){0.0, 0.0, 0.0, 0.0};
                do_block(lb, acc0);
                const bool two = lb + LB <= lmax;
                if (two) do_block(lb + LB, acc1);
                // ---- add the eight waves' partial tiles through LDS (the wave's transpose buffer is free now:
                //      LDS operations of one wave are ordered)
#pragma unroll
                for (int par = 0; par < 2; par++)
#pragma unroll
                    for (int t = 0; t < NCT; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            lamw[((par * NCT + t) * 4 + r) * 64 + lane] = acc0[par][t][r];
                            if (two) lamw[(((2 + par) * NCT + t) * 4 + r) * 64 + lane] = acc1[par][t][r];
                        }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 2 * NCT; u++) {
                    if (u >= NCT && !two) break;
                    const int e = tid + 512 * u;          // element (block, par, t, r, lane) of the reduced tiles
                    double sum = 0.0;
#pragma unroll
                    for (int w = 0; w < ADJ_WAVES; w++) sum += lds[w * WREG + e];
                    const int el = e & 63, r = (e >> 6) & 3, q = e >> 8, t = q % NCT, par = (q / NCT) & 1, blk = q / (2 * NCT);
                    const int l = lb + blk * LB + 2 * ((el >> 4) + 4 * r) + par;
#if ADJ_MU
                    if (l <= lmax) pout[(size_t)l * ncols + 16 * t + (el & 15)] = sum * srow[u];    // lambda_l = s_l mu_l: the scale is a property of the ROW
#else
                    if (l <= lmax) pout[(size_t)l * ncols + 16 * t + (el & 15)] = sum;
#endif
                }
                __syncthreads();
            }
        }
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();
    }
}

// alm_dev[idx][col] = sum over ring tiles of part[rt][idx][col]
__global__ void alm_reduce_kernel(const double *__restrict__ part, long n, int ntile, double *__restrict__ alm) {
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int t = 0; t < ntile; t++) s += part[(size_t)t * n + q];
        alm[q] = s;
    }
}
// the same sum by rows: block (chunk of 64 multipoles, m).  Ring tile t reaches row m from lb0_t(m) on (the block of
// 32 l that holds the first l any of its rings contributes to - legendre_adj_kernel's own expression); below that the
// tile's partial buffer holds nothing and is not read.
__global__ void __launch_bounds__(256)
alm_reduce_rows_kernel(const double *__restrict__ part, long nalm, int ntile, int ncols, int lmax, int ntile128,
                       const int32_t *__restrict__ lmin_tab, const double *__restrict__ zeros, double *__restrict__ alm) {
    __shared__ int s_lb0[16];
    const int m = blockIdx.y, l0 = m + 64 * (int)blockIdx.x;
    if (l0 > lmax) return;
    const int l1 = min(lmax + 1, l0 + 64), tid = threadIdx.x;
    if (tid < ntile) {
        int lmin = lmax + 1;
        for (int q = 0; q < 4; q++) {
            const int t128 = tid * 4 + q;
            if (t128 < ntile128) lmin = min(lmin, lmin_tab[m * ntile128 + t128]);
        }
        s_lb0[tid] = lmin <= lmax ? m + ((lmin - m) & ~31) : lmax + 1;
    }
    __syncthreads();
    const long base_m = alm_idx(0, m, lmax);
    const size_t n = (size_t)nalm * ncols;
    const double2 *p2 = reinterpret_cast<const double2 *>(part);
    double2 *a2 = reinterpret_cast<double2 *>(alm);
    const int half = ncols >> 1;                       // double2 per row (ncols is a multiple of 16)
    const float rcp = 1.0f / (float)half;
    const double2 *z2 = reinterpret_cast<const double2 *>(zeros);
    constexpr int UN = 4;                              // elements per thread in flight (their loads are independent)
    for (int e0 = tid; e0 < (l1 - l0) * half; e0 += 256 * UN) {
        size_t q[UN];
        int lu[UN];
        const double2 *src[UN][4];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int e = min(e0 + 256 * u, (l1 - l0) * half - 1);
            int lo = (int)((float)e * rcp);            // e / half (e < 2^23: one float multiply and a correction)
            int rem = e - lo * half;
            if (rem < 0) lo--, rem += half;
            if (rem >= half) lo++, rem -= half;
            const int l = l0 + lo;
            lu[u] = l;
            q[u] = ((size_t)(base_m + l) * ncols >> 1) + rem;
            // a tile that does not reach the row is read from a line of zeros (no branch: the loads of all tiles and of
            // the UN elements go out together)
#pragma unroll
            for (int t = 0; t < 4; t++) src[u][t] = (t < ntile && l >= s_lb0[t]) ? p2 + (size_t)t * (n >> 1) + q[u] : z2;
        }
        double2 v[UN][4];
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) v[u][t] = *src[u][t];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            double2 s = make_double2(0.0, 0.0);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                s.x += v[u][t].x;
                s.y += v[u][t].y;
            }
            for (int t = 4; t < ntile; t++)            // (more than four ring tiles: nside > 1024)
                if (lu[u] >= s_lb0[t]) {
                    const double2 w = p2[(size_t)t * (n >> 1) + q[u]];
                    s.x += w.x;
                    s.y += w.y;
                }
            if (e0 + 256 * u < (l1 - l0) * half) a2[q[u]] = s;
        }
    }
}
template <int NCT>
static int launch_legendre_adj(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *inter,
                               double *part) {
    const size_t shm = sizeof(double) * (size_t)ADJ_WAVES * 64 * 33 + 16;
    HIP_TRY(hipFuncSetAttribute((const void *)legendre_adj_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    const int ntile = (p->npair + 64 * ADJ_WAVES - 1) / (64 * ADJ_WAVES);
    const long nitems = (long)p->L * (ncols / (16 * NCT)) * ntile;
    dim3 grid((unsigned)std::min<long>(nitems, (long)ctx->num_cu));
    HIP_TRY(hipMemsetAsync(p->d_queue, 0, 64, ctx->stream));
    legendre_adj_kernel<NCT><<<grid, 512, shm, ctx->stream>>>(p->lmax, p->npair, p->nring, ncols, p->d_z,
                                                             ADJ_MU ? p->d_coefmu : p->d_coef, p->d_lstart,
                                                             ADJ_MU ? p->d_seedmu : p->d_seed, p->d_lmin, p->d_mcut, inter, part,
                                                             p->d_queue);
    LAUNCH_CHECK();
    return 0;
}
int sht_legendre_adj(corahip_ctx *ctx, const corahip_sht_plan *p, int ncols, const double *inter, double *part,
                     double *alm_dev) {
    StageTimer t(ctx, "legendre_adj");
    const int nt16 = ncols / 16;
    int rc;
    if (nt16 % 2 == 0) rc = launch_legendre_adj<2>(ctx, p, ncols, inter, part);
    else rc = launch_legendre_adj<1>(ctx, p, ncols, inter, part);
    if (rc) return rc;
    const int ntile = (p->npair + 64 * ADJ_WAVES - 1) / (64 * ADJ_WAVES);
    const long n = p->nalm * (long)ncols;
    (void)n;
#if ADJ_ROWS_REDUCE
    if (ntile <= 16) {
        const int ntile128 = (p->npair + LMIN_RINGS - 1) / LMIN_RINGS;
        dim3 grid((unsigned)((p->L + 63) / 64), (unsigned)p->L);
        alm_reduce_rows_kernel<<<grid, 256, 0, ctx->stream>>>(part, p->nalm, ntile, ncols, p->lmax, ntile128, p->d_lmin, p->d_zeros, alm_dev);
        LAUNCH_CHECK();
        return 0;
    }
    return CORAHIP_EINVAL;      // (more than 16 ring tiles: nside > 16384)
#else
    alm_reduce_kernel<<<(int)std::min<long>((n + 255) / 256, 8192), 256, 0, ctx->stream>>>(part, n, ntile, alm_dev);
    LAUNCH_CHECK();
#endif
    return 0;
}
