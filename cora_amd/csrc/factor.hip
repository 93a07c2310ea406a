// factor.hip - batched per-l matrix root of the nu x nu covariance blocks.
//
// Replaces the body of the l-loop of mkfullsky (cora/core/skysim.py:115-119):
//   cmax  = diag(C_l).max() * 1e-14;  Cm = C_l + I*cmax
//   T_l   = nputil.matrix_root_manynull(Cm, truncate=False)   (cora/util/nputil.py:51-101)
// i.e. lower Cholesky; if a pivot is not positive (scipy.linalg.cholesky raising
// LinAlgError) a symmetric eigen-decomposition with eigenvalues < max*threshold zeroed
// and root = V sqrt(Lambda).
//
// One 256-thread workgroup per matrix.  Blocked right-looking Cholesky (panel width 32)
// working in place in global memory (the 512 KB block of an F=256 matrix stays in L2):
// diagonal block factored in LDS, panel solved one row per thread against the LDS
// block, trailing update in 64x64 tiles with the two row panels staged in LDS.
// The eigen branch (rare: the all-zero l=0 block of the foreground models, or a truly
// indefinite block) is a parallel cyclic Jacobi iteration, also one workgroup per matrix.
#include "common.h"

#include <algorithm>
#include <vector>

#define CH_NB 32

// lower Cholesky of C + jitter on the diagonal -> T (upper triangle zeroed); info = 1 on failure
__global__ void __launch_bounds__(256)
chol_kernel(const double *__restrict__ C, int F, double jitter_rel, double *__restrict__ T,
            int32_t *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *D = lds;                      // [32][33] diagonal block
    double *PA = lds + CH_NB * (CH_NB + 1);  // [64][33] row panel (i tile)
    double *PB = PA + 64 * (CH_NB + 1);      // [64][33] row panel (j tile)
    double *red = PB + 64 * (CH_NB + 1);     // [256] reduction scratch
    int *flag = reinterpret_cast<int *>(red + 256);

    const int tid = threadIdx.x;
    const size_t off = (size_t)blockIdx.x * F * F;
    const double *A = C + off;
    double *Tl = T + off;

    // jitter = max(diag) * jitter_rel
    double dmax = -INFINITY;
    for (int i = tid; i < F; i += 256) dmax = fmax(dmax, A[(size_t)i * F + i]);
    red[tid] = dmax;
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    const double jit = red[0] * jitter_rel;
    __syncthreads();
    // copy lower triangle (+jitter), zero the upper
    for (long q = tid; q < (long)F * F; q += 256) {
        const int i = (int)(q / F), j = (int)(q % F);
        double v = 0.0;
        if (j <= i) v = A[q] + (i == j ? jit : 0.0);
        Tl[q] = v;
    }
    __syncthreads();

    for (int kb = 0; kb < F; kb += CH_NB) {
        const int nb = min(CH_NB, F - kb);
        // ---- 1. diagonal block into LDS and factor it
        for (int q = tid; q < nb * nb; q += 256) {
            const int i = q / nb, j = q % nb;
            D[i * (CH_NB + 1) + j] = (j <= i) ? Tl[(size_t)(kb + i) * F + kb + j] : 0.0;
        }
        __syncthreads();
        // factor it with ONE wave, row i of the block in the registers of lane i: column j is scaled, broadcast
        // with v_readlane and applied to the rows below - no LDS traffic and no barriers inside the 32 steps
        // (the workgroup-wide version needed three barriers per step: 96 per panel, ~0.8 ms per matrix)
        if (tid < 64) {
            const int lane = tid;
            double row[CH_NB];
#pragma unroll
            for (int k = 0; k < CH_NB; k++) row[k] = (lane < nb && k <= lane) ? D[lane * (CH_NB + 1) + k] : 0.0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < CH_NB; j++) {
                if (j < nb && !bad) {
                    const double d = __shfl(row[j], j);
                    if (!(d > 0.0)) {  // also catches NaN: LAPACK potrf "not positive definite"
                        bad = true;
                    } else {
                        const double lij = row[j] / sqrt(d);   // lane j: d / sqrt(d); lanes above j: unused garbage
                        row[j] = lane == j ? sqrt(d) : lij;
#pragma unroll
                        for (int k = j + 1; k < CH_NB; k++) row[k] -= lij * __shfl(lij, k);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CH_NB; k++)
                if (lane < nb && k <= lane) D[lane * (CH_NB + 1) + k] = row[k];
            if (bad && lane == 0) *flag = 1;
        }
        __syncthreads();
        if (*flag) break;
        for (int q = tid; q < nb * nb; q += 256) {
            const int i = q / nb, j = q % nb;
            if (j <= i) Tl[(size_t)(kb + i) * F + kb + j] = D[i * (CH_NB + 1) + j];
        }
        const int r0 = kb + nb;
        if (r0 >= F) break;
        // ---- 2. panel: rows r0..F-1, X = A21 * L11^{-T}; one row per thread
        for (int r = r0 + tid; r < F; r += 256) {
            double x[CH_NB];
            double *row = Tl + (size_t)r * F + kb;
#pragma unroll
            for (int j = 0; j < CH_NB; j++) x[j] = j < nb ? row[j] : 0.0;
#pragma unroll
            for (int j = 0; j < CH_NB; j++) {
                if (j < nb) {
                    double s = x[j];
#pragma unroll
                    for (int p = 0; p < CH_NB; p++)
                        if (p < j) s -= x[p] * D[j * (CH_NB + 1) + p];
                    x[j] = s / D[j * (CH_NB + 1) + j];
                }
            }
#pragma unroll
            for (int j = 0; j < CH_NB; j++)
                if (j < nb) row[j] = x[j];
        }
        __syncthreads();  // panel visible to the whole workgroup (same CU, write-through L1)
        // ---- 3. trailing update A22 -= X X^T (lower triangle), 64x64 tiles, 4x4 per thread
        // (the kernel is latency-bound: one workgroup per matrix, every tile a load -> barrier -> compute -> read-modify-
        //  write -> barrier chain against L2.  The row panel PA is loaded once per tile row, and the A22 elements the
        //  thread will update are requested BEFORE the product is formed, so their L2 latency hides behind it.)
        const int nt = (F - r0 + 63) / 64;
        for (int ti = 0; ti < nt; ti++) {
            const int i0 = r0 + ti * 64;
            for (int q = tid; q < 64 * CH_NB; q += 256) {
                const int rr = q / CH_NB, cc = q % CH_NB;
                PA[rr * (CH_NB + 1) + cc] = (i0 + rr < F && cc < nb) ? Tl[(size_t)(i0 + rr) * F + kb + cc] : 0.0;
            }
            for (int tj = 0; tj <= ti; tj++) {
                const int j0 = r0 + tj * 64;
                for (int q = tid; q < 64 * CH_NB; q += 256) {
                    const int rr = q / CH_NB, cc = q % CH_NB;
                    PB[rr * (CH_NB + 1) + cc] = (j0 + rr < F && cc < nb) ? Tl[(size_t)(j0 + rr) * F + kb + cc] : 0.0;
                }
                const int ty = tid >> 4, tx = tid & 15;  // 16 x 16 threads, each a 4x4 micro tile
                double old[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * w;
                        old[u][w] = (i < F && j <= i) ? Tl[(size_t)i * F + j] : 0.0;
                    }
                __syncthreads();
                double acc[4][4] = {};
                for (int p = 0; p < CH_NB; p++) {
                    double a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        a[u] = PA[(ty + 16 * u) * (CH_NB + 1) + p];
                        b[u] = PB[(tx + 16 * u) * (CH_NB + 1) + p];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int w = 0; w < 4; w++) acc[u][w] += a[u] * b[w];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * w;
                        if (i < F && j <= i) Tl[(size_t)i * F + j] = old[u][w] - acc[u][w];
                    }
                __syncthreads();
            }
        }
    }
    if (tid == 0) info[blockIdx.x] = *flag;
}

// ------------------------------------------------------------------------------------
// LEFT-LOOKING blocked Cholesky on FP64 MFMA (even F >= 64; chol_kernel above keeps the odd and the small sizes).
// chol_kernel is a latency chain against L2 at F = 256 (2.45 ms for 1.15e10 flop) and HBM-bound beyond: a right-looking
// update re-reads and re-writes the whole trailing matrix for every panel - F^3 / 96 doubles each way per matrix, 23 GB
// per cfg-4 step (F = 512), 92 GB for a cfg-5 rank (F = 1024) - and the matrices in flight (two per CU) do not fit any
// cache.  Left-looking, block column j (32 wide) is formed once: every 32 x 32 tile (i >= j, j) starts from the INPUT
// block (there is no copy-in pass), subtracts L_ik L_jk^T for k < j with the accumulators resident in registers
// (v_mfma_f64_16x16x4_f64, the tile in the MFMA's C layout), and is written once; the factors read are final.  HBM
// traffic: the input lower triangle + the output + F^3 / 192 doubles of factor blocks per matrix.
//  * the tiles of a block column are dealt to the four WAVES, which work on their own: a wave stages the two 32 x 32
//    factor blocks of a k-step in a wave-private piece of LDS (256-byte row loads, no workgroup barrier; the next
//    k-step's blocks are requested before the MFMAs of the current one) and reads them as MFMA operands (row stride 34
//    doubles: conflict-free for half-waves); A operand negated: acc += (-L_ik) L_jk^T;
//  * the diagonal tile goes to LDS and is factored by one wave, row i in the registers of lane i, broadcasts by
//    v_readlane, 1 / sqrt(d) by v_rsq_f64 + two Newton steps (the ds_bpermute of __shfl and sqrt + division cost 0.3 ms
//    of the 2.1 ms of a right-looking MFMA version of this kernel); 1 / L_jj is kept for the solve;
//  * the off-diagonal tiles are written as they are and solved in place, one row per thread, against the LDS block;
//  * three barriers per block column.
// Ablations of the right-looking MFMA version (F = 256, 2.11 ms): diagonal blocks 0.50, panel solve 0.35, trailing update
// 0.62, copy-in 0.5 ms.
// ------------------------------------------------------------------------------------
#define CHM_S 34    // doubles per staged block row
#ifndef CHM_STAMPS
#define CHM_STAMPS 0   // diagnostic build (make k2stamps): s_memtime per phase of chol_ll_kernel, summed over the waves
#endif
#if CHM_STAMPS
__device__ unsigned long long g_chm_stamps[8];
#define CSTAMP(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); c_acc[k] += _t - c_last; c_last = _t; }
#else
#define CSTAMP(k)
#endif
#ifndef CHM_ABLATE
#define CHM_ABLATE 0   // diagnostic builds (wrong results): 1 no diagonal-block factorisation, 2 no panel solve, 3 no MFMA updates
#endif

// TALL = true (F >= 384): the wave tiles of step A are 64 rows x 32 columns (two block rows) and the factor block
// L_jk of a k-step - the operand every tile of the block column shares - is staged ONCE per workgroup (each wave loads
// a quarter of it; double-buffered, one barrier per k-step) instead of once per tile.  At F = 512 / 1024 a matrix
// (2 / 8 MB, two per CU in flight) lives in no cache and the 32 x 32 form re-read 16 KB of factor blocks per 32 MFMAs -
// 87 MB per F = 1024 matrix, 3 TB/s for the whole launch: the kernel was bound by that traffic, not by its 2.3 ms of
// MFMA work.  The tall form moves 16 KB + 2 KB per 64 MFMAs (49 MB per matrix).  At F = 256 the matrices sit in L2,
// the per-k-step barrier costs more than the traffic: the 32 x 32 form stays.
template <bool TALL>
__global__ void __launch_bounds__(256)
chol_ll_kernel(const double *__restrict__ C, int F, double jitter_rel, double *__restrict__ T,
               int32_t *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *D = lds;                                   // [32][33] diagonal block
    double *red = lds + CH_NB * (CH_NB + 1);           // [256] reduction scratch
    int *flag = reinterpret_cast<int *>(red + 256);
    double *rdiag = red + 256 + 2;                     // [32] 1 / L_jj of the diagonal block
    double *stage = rdiag + CH_NB;                     // per wave: A block [32][CHM_S], B block [32][CHM_S]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    const size_t off = (size_t)blockIdx.x * F * F;
    const double *A = C + off;
    double *Tl = T + off;
    // 32 x 32 form: per wave an A and a B block; tall form: per wave an A block, then the shared B block twice
    double *PA = stage + (size_t)wave * (TALL ? 1 : 2) * 32 * CHM_S, *PB = PA + 32 * CHM_S;
    double *PBS = stage + 4 * 32 * CHM_S;              // (tall) [2][32][CHM_S]

    // jitter = max(diag) * jitter_rel
    double dmax = -INFINITY;
    for (int i = tid; i < F; i += 256) dmax = fmax(dmax, A[(size_t)i * F + i]);
    red[tid] = dmax;
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    const double jit = red[0] * jitter_rel;
    __syncthreads();

#if CHM_STAMPS
    unsigned long long c_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c_last;
    { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); c_last = _t; }
#endif
    const int nblk = (F + 31) / 32;
    // this lane's part of a 32 x 32 block moved as rows: half a row (16 doubles) = row (lane >> 1), columns 16 (lane & 1) ..
    const int srow = lane >> 1, scol = 16 * (lane & 1);
    // (the loads are unconditional - a row past the end reads row 0 and is zeroed by the sign when it is staged: with the
    //  zeroing at the load the compiler waited for the prefetched values right behind their request)
    auto load_block = [&](int row0, int col0, double2 (&v)[8]) {   // rows row0 .. + 31, columns col0 .. + 31 of the factor
        const double *src = Tl + (size_t)(row0 + srow < F ? row0 + srow : 0) * F + col0 + scol;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const double2 *>(src + 2 * q);
    };
    auto store_block = [&](double *P, const double2 (&v)[8], double sign, int row0) {
        const double sg = row0 + srow < F ? sign : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++)
            *reinterpret_cast<double2 *>(P + srow * CHM_S + scol + 2 * q) = make_double2(sg * v[q].x, sg * v[q].y);
    };

    for (int j = 0; j < nblk; j++) {
        const int kb = 32 * j;
        const int nb = min(CH_NB, F - kb);
        // ---- A. the tiles (i >= j, j) of the block column: input block minus the products of the factor blocks to the left.
        //         Tile (a, b), register r of the accumulators <-> row i0 + 16 a + kq + 4 r, column kb + 16 b + ri.
        if (TALL) {
            const int ntile = (nblk - j + 1) / 2;          // tiles of two block rows: (j, j+1), (j+2, j+3), ...
            const int nround = (ntile + 3) / 4;
            // this wave's quarter of a shared block: row 8 wave + (lane >> 3), columns 4 (lane & 7) .. + 3
            const int brow = 8 * wave + (lane >> 3), bcol = 4 * (lane & 7);
            auto load_bq = [&](int col0, double2 (&v)[2]) {
                const double *src = Tl + (size_t)(kb + brow < F ? kb + brow : 0) * F + col0 + bcol;
                v[0] = *reinterpret_cast<const double2 *>(src);
                v[1] = *reinterpret_cast<const double2 *>(src + 2);
            };
            auto store_bq = [&](double *P, const double2 (&v)[2]) {
                const double sg = kb + brow < F ? 1.0 : 0.0;
                *reinterpret_cast<double2 *>(P + brow * CHM_S + bcol) = make_double2(sg * v[0].x, sg * v[0].y);
                *reinterpret_cast<double2 *>(P + brow * CHM_S + bcol + 2) = make_double2(sg * v[1].x, sg * v[1].y);
            };
            for (int u = 0; u < nround; u++) {
                const int t = wave + 4 * u;
                const bool has = t < ntile;
                const int i0 = 32 * (j + 2 * t);           // first row of the tile
                d4_t acc[4][2];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = i0 + 16 * a + kq + 4 * r, col = kb + 16 * b + ri;
                            double v = 0.0;
                            if (has && row < F && col <= row) v = A[(size_t)row * F + col] + (row == col ? jit : 0.0);
                            acc[a][b][r] = v;
                        }
                if (j > 0 && CHM_ABLATE != 3) {
                    double2 va0[8], va1[8], vb[2];
                    if (has) {
                        load_block(i0, 0, va0);
                        load_block(i0 + 32, 0, va1);
                    }
                    load_bq(0, vb);
                    for (int k = 0; k < j; k++) {
                        double *PBc = PBS + (k & 1) * 32 * CHM_S;
                        store_bq(PBc, vb);
                        __syncthreads();     // block (j, k) complete; every wave is done with step k - 1 (the other buffer)
                        if (k + 1 < j) load_bq(32 * (k + 1), vb);
                        if (has) {
                            // rows 0..31 of the tile through the wave-private A buffer, then rows 32..63 through the same
                            // buffer (one wave, LDS in order: the reads of a half complete before the next half's writes)
                            store_block(PA, va0, -1.0, i0);
                            if (k + 1 < j) load_block(i0, 32 * (k + 1), va0);
#pragma unroll
                            for (int s = 0; s < 8; s++) {
                                double af[2], bf[2];
#pragma unroll
                                for (int a = 0; a < 2; a++) {
                                    af[a] = PA[(16 * a + ri) * CHM_S + 4 * s + kq];
                                    bf[a] = PBc[(16 * a + ri) * CHM_S + 4 * s + kq];
                                }
#pragma unroll
                                for (int a = 0; a < 2; a++)
#pragma unroll
                                    for (int b = 0; b < 2; b++)
                                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
                            }
                            store_block(PA, va1, -1.0, i0 + 32);
                            if (k + 1 < j) load_block(i0 + 32, 32 * (k + 1), va1);
#pragma unroll
                            for (int s = 0; s < 8; s++) {
                                double af[2], bf[2];
#pragma unroll
                                for (int a = 0; a < 2; a++) {
                                    af[a] = PA[(16 * a + ri) * CHM_S + 4 * s + kq];
                                    bf[a] = PBc[(16 * a + ri) * CHM_S + 4 * s + kq];
                                }
#pragma unroll
                                for (int a = 0; a < 2; a++)
#pragma unroll
                                    for (int b = 0; b < 2; b++)
                                        acc[2 + a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[2 + a][b], 0, 0, 0);
                            }
                        }
                    }
                    __syncthreads();         // the last step's reads of the shared buffers, before the next round refills them
                }
                if (has) {
#pragma unroll
                    for (int hb = 0; hb < 2; hb++) {       // the two block rows of the tile
                        const int r0b = i0 + 32 * hb;
                        if (r0b >= F) continue;
                        if (t == 0 && hb == 0) {            // the diagonal block goes to LDS
#pragma unroll
                            for (int a = 0; a < 2; a++)
#pragma unroll
                                for (int b = 0; b < 2; b++)
#pragma unroll
                                    for (int r = 0; r < 4; r++)
                                        D[(16 * a + kq + 4 * r) * (CH_NB + 1) + 16 * b + ri] = acc[a][b][r];
                        } else {                            // stored for the solve; its mirror block of the upper triangle zeroed
#pragma unroll
                            for (int a = 0; a < 2; a++)
#pragma unroll
                                for (int b = 0; b < 2; b++)
#pragma unroll
                                    for (int r = 0; r < 4; r++) {
                                        const int row = r0b + 16 * a + kq + 4 * r, col = kb + 16 * b + ri;
                                        if (row < F && col < F) Tl[(size_t)row * F + col] = acc[2 * hb + a][b][r];
                                    }
                            if (kb + srow < F) {
                                double *z = Tl + (size_t)(kb + srow) * F + r0b + scol;
#pragma unroll
                                for (int q = 0; q < 8; q++)
                                    if (r0b + scol + 2 * q < F) *reinterpret_cast<double2 *>(z + 2 * q) = make_double2(0.0, 0.0);
                            }
                        }
                    }
                }
            }
        } else
        for (int i = j + wave; i < nblk; i += 4) {
            const int i0 = 32 * i;
            d4_t acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = i0 + 16 * a + kq + 4 * r, col = kb + 16 * b + ri;
                        double v = 0.0;
                        if (row < F && col <= row) v = A[(size_t)row * F + col] + (row == col ? jit : 0.0);
                        acc[a][b][r] = v;
                    }
            if (j > 0 && CHM_ABLATE != 3) {
                double2 va[8], vb[8];
                load_block(i0, 0, va);
                load_block(kb, 0, vb);
                for (int k = 0; k < j; k++) {
                    store_block(PA, va, -1.0, i0);
                    store_block(PB, vb, 1.0, kb);
                    if (k + 1 < j) {          // the next k-step's blocks, in flight during the MFMAs below
                        load_block(i0, 32 * (k + 1), va);
                        load_block(kb, 32 * (k + 1), vb);
                    }
                    // (wave-private LDS: the ds_writes above complete before the reads below, and these before the next
                    //  k-step's writes - one wave, in order.  Requesting the blocks TWO k-steps ahead - two register sets,
                    //  248 VGPRs - was measured: 14.6 / 7.36 / 1.73 against 14.4 / 7.32 / 1.70 ms, nothing)
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        double af[2], bf[2];
#pragma unroll
                        for (int a = 0; a < 2; a++) {
                            af[a] = PA[(16 * a + ri) * CHM_S + 4 * s + kq];
                            bf[a] = PB[(16 * a + ri) * CHM_S + 4 * s + kq];
                        }
#pragma unroll
                        for (int a = 0; a < 2; a++)
#pragma unroll
                            for (int b = 0; b < 2; b++)
                                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
                    }
                }
            }
            if (i == j) {      // (wave 0) the diagonal tile goes to LDS
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) D[(16 * a + kq + 4 * r) * (CH_NB + 1) + 16 * b + ri] = acc[a][b][r];
            } else {           // an off-diagonal tile is stored for the solve; its mirror tile of the upper triangle is zeroed
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = i0 + 16 * a + kq + 4 * r, col = kb + 16 * b + ri;
                            if (row < F && col < F) Tl[(size_t)row * F + col] = acc[a][b][r];
                        }
                if (kb + srow < F) {
                    double *z = Tl + (size_t)(kb + srow) * F + i0 + scol;
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (i0 + scol + 2 * q < F) *reinterpret_cast<double2 *>(z + 2 * q) = make_double2(0.0, 0.0);
                }
            }
        }
        CSTAMP(0);           // A: tile initialisation, update loop, tile stores
        __syncthreads();
        CSTAMP(1);           // barrier behind A
        // ---- B. factor the diagonal block with ONE wave, row i of the block in the registers of lane i
        if (tid < 64 && CHM_ABLATE != 1) {
            auto bcast = [](double v, int src) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
                return __hiloint2double(hi, lo);
            };
            double row[CH_NB];
#pragma unroll
            for (int k = 0; k < CH_NB; k++) row[k] = (lane < nb && k <= lane) ? D[lane * (CH_NB + 1) + k] : 0.0;
            bool bad = false;
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj++) {
                if (jj < nb && !bad) {
                    const double d = bcast(row[jj], jj);
                    if (!(d > 0.0)) {  // also catches NaN: LAPACK potrf "not positive definite"
                        bad = true;
                    } else {
                        double y = __builtin_amdgcn_rsq(d);
                        const double hd = 0.5 * d;
                        y = fma(y, fma(-hd * y, y, 0.5), y);
                        y = fma(y, fma(-hd * y, y, 0.5), y);       // 1 / sqrt(d)
                        const double lij = row[jj] * y;            // lane jj: sqrt(d); lanes above jj: unused garbage
                        row[jj] = lij;
                        if (lane == jj) rdiag[jj] = y;
#pragma unroll
                        for (int k = jj + 1; k < CH_NB; k++) row[k] = fma(-lij, bcast(lij, k), row[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CH_NB; k++)
                if (lane < nb && k <= lane) D[lane * (CH_NB + 1) + k] = row[k];
            if (bad && lane == 0) *flag = 1;
        }
        CSTAMP(2);           // B (one wave; the others arrive at once)
        __syncthreads();
        CSTAMP(3);           // barrier behind B = the three idle waves' wait for the diagonal block
        if (*flag) break;
        for (int q = tid; q < nb * nb; q += 256) {      // L_jj and the zeros above its diagonal
            const int i = q / nb, jj = q % nb;
            Tl[(size_t)(kb + i) * F + kb + jj] = (jj <= i) ? D[i * (CH_NB + 1) + jj] : 0.0;
        }
        const int r0 = kb + nb;
        // ---- C. the tiles below: X = S L_jj^{-T}, one row per thread, in place (nb == 32 whenever there are rows below)
        for (int r = r0 + tid; r < F && CHM_ABLATE != 2; r += 256) {
            double x[CH_NB];
            double *row = Tl + (size_t)r * F + kb;
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj += 2) {
                const double2 v = *reinterpret_cast<const double2 *>(row + jj);
                x[jj] = v.x;
                x[jj + 1] = v.y;
            }
            // (the run-time test on nb keeps the 528 reads of D inside the row loop: without it they are hoisted in
            //  front of it as loop invariants - into 1000 registers, i.e. scratch)
            // four partial sums per dot product: the 496 multiply-adds of a row are otherwise ONE dependent chain (8 cycles
            // per link, nothing to overlap it with at one or two waves per SIMD)
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj++) {
                if (jj < nb) {
                    double s[4] = {x[jj], 0.0, 0.0, 0.0};
#pragma unroll
                    for (int p = 0; p < CH_NB; p++)
                        if (p < jj) s[p & 3] = fma(-x[p], D[jj * (CH_NB + 1) + p], s[p & 3]);
                    x[jj] = ((s[0] + s[1]) + (s[2] + s[3])) * rdiag[jj];
                }
            }
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj += 2) *reinterpret_cast<double2 *>(row + jj) = make_double2(x[jj], x[jj + 1]);
        }
        CSTAMP(4);           // C: diagonal block store + panel solve
        __syncthreads();  // the block column is final and visible to the whole workgroup (same CU, write-through L1)
        CSTAMP(5);           // barrier behind C
    }
    if (tid == 0) info[blockIdx.x] = *flag;
#if CHM_STAMPS
    if (lane == 0) {
        for (int k = 0; k < 6; k++) atomicAdd(&g_chm_stamps[k], c_acc[k]);
        atomicAdd(&g_chm_stamps[7], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------
// COOPERATIVE form of the left-looking factorisation for the last few matrices of a batch (round 4).  chol_ll_kernel
// gives a matrix one workgroup, two of them share a CU: 513 matrices of F = 1024 (a cfg-5 rank: L = 8 x 512 + 1) are two
// rounds of 512 and a third in which ONE workgroup works for 4.4 ms while 255 CUs idle.  Here G workgroups (4 G waves)
// share a matrix: the 32 x 32 tiles of a block column are dealt over all their waves (one tile each: 4 G >= F / 32),
// the diagonal tile is factored by the first wave and published through the output matrix, the rows of the panel
// solve are dealt over all threads; two device-scope barriers per block column (arrival counter in global memory, one
// spinning thread per workgroup - the grid is at most one workgroup per CU and is launched behind the batch kernel on
// the same stream, so its workgroups are co-resident).  Same arithmetic per tile as the 32 x 32 form of chol_ll_kernel.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
chol_coop_kernel(const double *__restrict__ C, int F, double jitter_rel, double *T, int32_t *__restrict__ info, int mat0, int nrem,
                 int G, unsigned *bar, int *gflag, double *rd, int *xcc_min, int *xcc_max) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *D = lds;                                   // [32][33] diagonal block
    double *red = lds + CH_NB * (CH_NB + 1);           // [256] reduction scratch
    double *rdiag = red + 256 + 2;                     // [32] 1 / L_jj of the diagonal block
    double *stage = rdiag + CH_NB;                     // per wave: A block [32][CHM_S], B block [32][CHM_S]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ri = lane & 15, kq = lane >> 4;
    // The G workgroups of a matrix are blocks with equal blockIdx % 8: under the round-robin dispatch of workgroups over
    // the eight XCDs they share ONE L2, and a barrier between them needs no L2 write-back / invalidate (an agent-scope
    // fence does both on gfx942 / gfx950 - the L2s of different XCDs are not coherent for ordinary device memory).
    // That placement is a property of the dispatcher, not a guarantee: every workgroup publishes its XCC_ID and the
    // group takes the cheap barrier only if all of them agree (otherwise agent-scope fences: slower, still correct).
    const int xslot = blockIdx.x >> 3;
    const int mi = (xslot / G) * 8 + (blockIdx.x & 7), wgl = xslot % G;   // matrix of the remainder, workgroup within its group
    if (mi >= nrem) return;
    const int gw = 4 * wgl + wave, NW = 4 * G;                  // this wave among the waves of the matrix
    const size_t off = (size_t)(mat0 + mi) * F * F;
    const double *A = C + off;
    double *Tl = T + off;
    double *PA = stage + (size_t)wave * 2 * 32 * CHM_S, *PB = PA + 32 * CHM_S;

    double dmax = -INFINITY;
    for (int i = tid; i < F; i += 256) dmax = fmax(dmax, A[(size_t)i * F + i]);
    red[tid] = dmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    const double jit = red[0] * jitter_rel;
    __syncthreads();

    unsigned nsync = 0;
    bool same_xcc = false;
    // barrier over the G workgroups of this matrix; writes before it are visible behind it.
    // same_xcc: the stores are in the shared L2 once they are acknowledged (write-through L1): no L2 write-back is needed
    // on the release side.  The acquire side invalidates the CU's vector L1 (buffer_inv sc1, a few cycles per block
    // column): a CU that wrote a tile in step A reads it again in a later block column after ANOTHER CU rewrote it in
    // place in step C - whether the L1 keeps lines it wrote through is not documented, and a stale hit would be a
    // silently wrong factor (round-4 advice; the LLVM memory model asks for the invalidate on every cross-CU acquire).
    // The words that are rewritten between barriers (1 / diag, the failure flag, the counter) are read with
    // device-scope atomic loads.
    auto group_sync = [&]() {
        if (same_xcc) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else __threadfence();
        __syncthreads();
        nsync++;
        if (tid == 0) {
            __hip_atomic_fetch_add(&bar[mi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (bounded: if a workgroup of the group never arrives - the grid was not co-resident after all: another
            //  stream or process held the CUs - the group gives up after 2^21 polls, a few seconds, and reports the
            //  matrix with info = 3: the host then factors it with the one-workgroup kernel, instead of hanging the device)
            unsigned polls = 0;
            while (__hip_atomic_load(&bar[mi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nsync * (unsigned)G) {
                __builtin_amdgcn_s_sleep(2);
                if ((++polls & 1023u) == 0) {
                    if (__hip_atomic_load(&gflag[mi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3) break;
                    if (polls >= (1u << 21)) {
                        __hip_atomic_store(&gflag[mi], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (same_xcc) asm volatile("buffer_inv sc1" ::: "memory");
        else __threadfence();
    };
    {
        const int xcc = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;     // HW_REG_XCC_ID, bits 3:0
        if (tid == 0) {
            __hip_atomic_fetch_min(&xcc_min[mi], xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&xcc_max[mi], xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        group_sync();
        same_xcc = __hip_atomic_load(&xcc_min[mi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                   __hip_atomic_load(&xcc_max[mi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(&gflag[mi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {     // (the barrier gave up)
            if (wgl == 0 && tid == 0) info[mat0 + mi] = 3;
            return;
        }
    }

    const int nblk = (F + 31) / 32;
    const int srow = lane >> 1, scol = 16 * (lane & 1);
    auto load_block = [&](int row0, int col0, double2 (&v)[8]) {
        const double *src = Tl + (size_t)(row0 + srow < F ? row0 + srow : 0) * F + col0 + scol;
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const double2 *>(src + 2 * q);
    };
    auto store_block = [&](double *P, const double2 (&v)[8], double sign, int row0) {
        const double sg = row0 + srow < F ? sign : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++)
            *reinterpret_cast<double2 *>(P + srow * CHM_S + scol + 2 * q) = make_double2(sg * v[q].x, sg * v[q].y);
    };
    int failed = 0;
    for (int j = 0; j < nblk; j++) {
        const int kb = 32 * j;
        const int nb = min(CH_NB, F - kb);
        // ---- A. tiles (i >= j, j): input block minus the products of the factor blocks to the left (one tile per wave)
        for (int i = j + gw; i < nblk; i += NW) {
            const int i0 = 32 * i;
            d4_t acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = i0 + 16 * a + kq + 4 * r, col = kb + 16 * b + ri;
                        double v = 0.0;
                        if (row < F && col <= row) v = A[(size_t)row * F + col] + (row == col ? jit : 0.0);
                        acc[a][b][r] = v;
                    }
            if (j > 0) {
                double2 va[8], vb[8];
                load_block(i0, 0, va);
                load_block(kb, 0, vb);
                for (int k = 0; k < j; k++) {
                    store_block(PA, va, -1.0, i0);
                    store_block(PB, vb, 1.0, kb);
                    if (k + 1 < j) {
                        load_block(i0, 32 * (k + 1), va);
                        load_block(kb, 32 * (k + 1), vb);
                    }
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        double af[2], bf[2];
#pragma unroll
                        for (int a = 0; a < 2; a++) {
                            af[a] = PA[(16 * a + ri) * CHM_S + 4 * s + kq];
                            bf[a] = PB[(16 * a + ri) * CHM_S + 4 * s + kq];
                        }
#pragma unroll
                        for (int a = 0; a < 2; a++)
#pragma unroll
                            for (int b = 0; b < 2; b++)
                                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
                    }
                }
            }
            if (i == j) {      // (first wave of the group) the diagonal tile goes to LDS
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) D[(16 * a + kq + 4 * r) * (CH_NB + 1) + 16 * b + ri] = acc[a][b][r];
            } else {           // stored for the solve; its mirror tile of the upper triangle is zeroed
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = i0 + 16 * a + kq + 4 * r, col = kb + 16 * b + ri;
                            if (row < F && col < F) Tl[(size_t)row * F + col] = acc[a][b][r];
                        }
                if (kb + srow < F) {
                    double *z = Tl + (size_t)(kb + srow) * F + i0 + scol;
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (i0 + scol + 2 * q < F) *reinterpret_cast<double2 *>(z + 2 * q) = make_double2(0.0, 0.0);
                }
            }
        }
        // ---- B. the first wave factors the diagonal block (row i in the registers of lane i) and publishes it
        if (gw == 0) {
            auto bcast = [](double v, int src) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
                return __hiloint2double(hi, lo);
            };
            double row[CH_NB];
#pragma unroll
            for (int k = 0; k < CH_NB; k++) row[k] = (lane < nb && k <= lane) ? D[lane * (CH_NB + 1) + k] : 0.0;
            bool bad = false;
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj++) {
                if (jj < nb && !bad) {
                    const double d = bcast(row[jj], jj);
                    if (!(d > 0.0)) {
                        bad = true;
                    } else {
                        double y = __builtin_amdgcn_rsq(d);
                        const double hd = 0.5 * d;
                        y = fma(y, fma(-hd * y, y, 0.5), y);
                        y = fma(y, fma(-hd * y, y, 0.5), y);
                        const double lij = row[jj] * y;
                        row[jj] = lij;
                        if (lane == jj) __hip_atomic_store(&rd[mi * CH_NB + jj], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int k = jj + 1; k < CH_NB; k++) row[k] = fma(-lij, bcast(lij, k), row[k]);
                    }
                }
            }
            if (lane < nb) {
                double *o = Tl + (size_t)(kb + lane) * F + kb;
#pragma unroll
                for (int k = 0; k < CH_NB; k++)
                    if (k < nb) o[k] = k <= lane ? row[k] : 0.0;
            }
            if (bad && lane == 0) __hip_atomic_store(&gflag[mi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        group_sync();
        failed = __hip_atomic_load(&gflag[mi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (failed) break;
        // ---- C. every workgroup takes L_jj and 1 / diag into its LDS and solves its share of the rows below, in place
        for (int q = tid; q < CH_NB * CH_NB; q += 256) {
            const int i = q >> 5, jj = q & 31;
            D[i * (CH_NB + 1) + jj] = (i < nb && jj <= i) ? Tl[(size_t)(kb + i) * F + kb + jj] : 0.0;
        }
        if (tid < CH_NB) rdiag[tid] = tid < nb ? __hip_atomic_load(&rd[mi * CH_NB + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
        for (int r = kb + nb + wgl * 256 + tid; r < F; r += 256 * G) {
            double x[CH_NB];
            double *row = Tl + (size_t)r * F + kb;
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj += 2) {
                const double2 v = *reinterpret_cast<const double2 *>(row + jj);
                x[jj] = v.x;
                x[jj + 1] = v.y;
            }
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj++) {
                if (jj < nb) {
                    double s[4] = {x[jj], 0.0, 0.0, 0.0};
#pragma unroll
                    for (int p = 0; p < CH_NB; p++)
                        if (p < jj) s[p & 3] = fma(-x[p], D[jj * (CH_NB + 1) + p], s[p & 3]);
                    x[jj] = ((s[0] + s[1]) + (s[2] + s[3])) * rdiag[jj];
                }
            }
#pragma unroll
            for (int jj = 0; jj < CH_NB; jj += 2) *reinterpret_cast<double2 *>(row + jj) = make_double2(x[jj], x[jj + 1]);
        }
        group_sync();       // the block column is final for every workgroup of the group
    }
    if (wgl == 0 && tid == 0) {
        info[mat0 + mi] = failed;
        xcc_max[mi] = same_xcc ? 0x100 : 0x200;       // (diagnostics: which barrier the group took - CORAHIP_K2_COOP_DEBUG)
    }
}

// ------------------------------------------------------------------------------------
// eigen branch: parallel cyclic Jacobi on Cm = C + jitter (symmetric), one workgroup per
// listed matrix.  W = working copy of Cm [F][F], V = eigenvectors [F][F] (global scratch).
// root = V * sqrt(max(lambda,0) thresholded), columns ordered by ascending eigenvalue
// (the order scipy.linalg.eigh returns, nputil.py:84-96).
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
jacobi_root_kernel(const double *__restrict__ C, const int32_t *__restrict__ list, int F, double jitter_rel,
                   double eig_thresh, double *__restrict__ Wall, double *__restrict__ Vall,
                   double *__restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *cs = lds;                 // [F/2+1][2]
    double *red = lds + 2 * (F / 2 + 1);  // [256]
    double *ev = red + 256;           // [F]
    int *perm = reinterpret_cast<int *>(ev + F);        // [F+1] round-robin player positions
    int *order = perm + (F + 2);                         // [F]
    const int tid = threadIdx.x;
    const int l = list[blockIdx.x];
    const double *A = C + (size_t)l * F * F;
    double *W = Wall + (size_t)blockIdx.x * F * F;
    double *V = Vall + (size_t)blockIdx.x * F * F;
    double *Tl = T + (size_t)l * F * F;

    double dmax = -INFINITY;
    for (int i = tid; i < F; i += 256) dmax = fmax(dmax, A[(size_t)i * F + i]);
    red[tid] = dmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    const double jit = red[0] * jitter_rel;
    __syncthreads();
    // symmetrised working copy (use the lower triangle, like LAPACK's default uplo='L')
    for (long q = tid; q < (long)F * F; q += 256) {
        const int i = (int)(q / F), j = (int)(q % F);
        const double v = (j <= i) ? A[(size_t)i * F + j] : A[(size_t)j * F + i];
        W[q] = v + (i == j ? jit : 0.0);
        V[q] = (i == j) ? 1.0 : 0.0;
    }
    const int np = (F + 1) / 2;      // pairs per round
    const int nplayers = 2 * np;     // a dummy player F when F is odd
    for (int i = tid; i < nplayers; i += 256) perm[i] = i;
    __syncthreads();

    for (int sweep = 0; sweep < 60; sweep++) {
        // convergence: off-diagonal Frobenius norm vs diagonal
        double offs = 0.0, dia = 0.0;
        for (long q = tid; q < (long)F * F; q += 256) {
            const int i = (int)(q / F), j = (int)(q % F);
            const double v = W[q];
            if (i == j) dia += v * v;
            else offs += v * v;
        }
        red[tid] = offs;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        offs = red[0];
        __syncthreads();
        red[tid] = dia;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        dia = red[0];
        __syncthreads();
        if (offs <= 1e-30 * dia || offs == 0.0) break;

        for (int round = 0; round < nplayers - 1; round++) {
            // pairs of this round: (perm[k], perm[nplayers-1-k])
            for (int k = tid; k < np; k += 256) {
                int p = perm[k], q = perm[nplayers - 1 - k];
                if (p > q) { int t = p; p = q; q = t; }
                double c = 1.0, s = 0.0;
                if (q < F) {
                    const double apq = W[(size_t)p * F + q];
                    if (apq != 0.0) {
                        const double app = W[(size_t)p * F + p], aqq = W[(size_t)q * F + q];
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                }
                cs[2 * k] = c;
                cs[2 * k + 1] = s;
            }
            __syncthreads();
            // rows: W <- J^T W
            for (int it = tid; it < np * F; it += 256) {
                const int k = it / F, j = it % F;
                int p = perm[k], q = perm[nplayers - 1 - k];
                if (p > q) { int t = p; p = q; q = t; }
                if (q >= F) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double wp = W[(size_t)p * F + j], wq = W[(size_t)q * F + j];
                W[(size_t)p * F + j] = c * wp - s * wq;
                W[(size_t)q * F + j] = s * wp + c * wq;
            }
            __syncthreads();
            // columns: W <- W J, V <- V J
            for (int it = tid; it < np * F; it += 256) {
                const int k = it / F, i = it % F;
                int p = perm[k], q = perm[nplayers - 1 - k];
                if (p > q) { int t = p; p = q; q = t; }
                if (q >= F) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double wp = W[(size_t)i * F + p], wq = W[(size_t)i * F + q];
                W[(size_t)i * F + p] = c * wp - s * wq;
                W[(size_t)i * F + q] = s * wp + c * wq;
                const double vp = V[(size_t)i * F + p], vq = V[(size_t)i * F + q];
                V[(size_t)i * F + p] = c * vp - s * vq;
                V[(size_t)i * F + q] = s * vp + c * vq;
            }
            __syncthreads();
            // rotate players 1..nplayers-1 (player at position 0 is fixed)
            if (tid == 0) {
                const int last = perm[nplayers - 1];
                for (int k = nplayers - 1; k > 1; k--) perm[k] = perm[k - 1];
                perm[1] = last;
            }
            __syncthreads();
        }
    }
    // eigenvalues, threshold, ascending order
    for (int i = tid; i < F; i += 256) ev[i] = W[(size_t)i * F + i];
    __syncthreads();
    double emax = -INFINITY;
    for (int i = tid; i < F; i += 256) emax = fmax(emax, ev[i]);
    red[tid] = emax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    emax = red[0];
    __syncthreads();
    // rank of each eigenvalue (stable) -> output column
    for (int i = tid; i < F; i += 256) {
        int rank = 0;
        const double e = ev[i];
        for (int j = 0; j < F; j++) {
            const double f = ev[j];
            if (f < e || (f == e && j < i)) rank++;
        }
        order[i] = rank;
    }
    __syncthreads();
    for (long q = tid; q < (long)F * F; q += 256) {
        const int i = (int)(q / F), j = (int)(q % F);
        double e = ev[j];
        if (e < emax * eig_thresh) e = 0.0;  // nputil.py:87 (negative / tiny eigenvalues dropped)
        Tl[(size_t)i * F + order[j]] = V[q] * sqrt(e);
    }
}

extern "C" int corahip_factor_batched(corahip_ctx *ctx, const double *C, int nl, int F, double jitter_rel,
                                      double eig_thresh, double *T, int32_t *info) {
    ARG_CHECK(ctx != nullptr && C != nullptr && T != nullptr && info != nullptr);
    ARG_CHECK(nl >= 1 && F >= 1);
    StageTimer t(ctx, "factor");
    static const bool no_mfma = getenv("CORAHIP_K2_VALU") != nullptr;   // A/B: the VALU kernel for every size
    if (F >= 64 && (F % 2) == 0 && !no_mfma) {
        // tall tiles + shared L_jk from F = 384 on (CORAHIP_K2_TALL=0 / 1 forces one form: A/B runs)
        static const char *tall_env = getenv("CORAHIP_K2_TALL");
        const bool tall = tall_env ? atoi(tall_env) != 0 : F >= 384;
        const size_t shm = sizeof(double) * (CH_NB * (CH_NB + 1) + 256 + 2 + CH_NB + (tall ? 6 : 8) * 32 * CHM_S) + 16;
        // The last nl mod (2 x CUs) matrices of a tall batch, if they are few, go to the cooperative kernel behind the batch:
        // a straggler round of the one-workgroup form costs a whole single-matrix latency (4.4 ms at F = 1024) for them.
        // CORAHIP_K2_COOP=0 switches it off (A/B).
        const char *coop_env = getenv("CORAHIP_K2_COOP");   // (read per call: the tests switch it)
        const int nblk = (F + 31) / 32;
        const int G = (nblk + 3) / 4;                      // one tile per wave in every block column
        int nrem = 0;
        // (behind the 32 x 32 form - cfg 3 is 2049 = 4 x 512 + 1 matrices of F = 256 - it was measured in round 5 and does not
        //  pay: 2048 matrices 1.50 ms, 2049 1.71 ms, 2049 with the last one here 1.75 ms: launch, memsets and barriers of
        //  the cooperative form cost what the fifth round of one workgroup costs at that size)
        if (tall && (F % 32) == 0 && !(coop_env && atoi(coop_env) == 0) && !CHM_STAMPS) {
            const int slots = 2 * ctx->num_cu;
            const int r = nl % slots;
            if (nl > slots && r > 0 && ((r + 7) / 8) * 8 * G <= ctx->num_cu && r <= 16) nrem = r;
            if (coop_env && atoi(coop_env) == 2 && ((nl + 7) / 8) * 8 * G <= ctx->num_cu) nrem = nl;     // (tests: every matrix through the cooperative kernel)
        }
        const size_t shm2 = sizeof(double) * (CH_NB * (CH_NB + 1) + 256 + 2 + CH_NB + 8 * 32 * CHM_S) + 16;
        if (nrem > 0) {
            // the cooperative grid spins on barriers between its workgroups: it must be co-resident.  The occupancy
            // calculation covers this kernel's own resources (what else holds CUs at launch time it cannot know: the
            // barrier is bounded for that case); a grid that does not fit goes through the batch kernel
            HIP_TRY(hipFuncSetAttribute((const void *)chol_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2));
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)chol_coop_kernel, 256, shm2));
            if ((long)per_cu * ctx->num_cu < (long)((nrem + 7) / 8) * 8 * G) nrem = 0;
        }
        const int nmain = nl - nrem;
        if (tall) {
            HIP_TRY(hipFuncSetAttribute((const void *)chol_ll_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            if (nmain > 0) chol_ll_kernel<true><<<nmain, 256, shm, ctx->stream>>>(C, F, jitter_rel, T, info);
            if (nrem > 0) {
                void *ws = nullptr;
                const size_t wsb = (size_t)nrem * (sizeof(unsigned) + 3 * sizeof(int) + CH_NB * sizeof(double)) + 64;
                int rcs = corahip_ctx_scratch(ctx, 8, wsb, &ws);
                if (rcs) return rcs;
                HIP_TRY(hipMemsetAsync(ws, 0, wsb, ctx->stream));
                double *rd = reinterpret_cast<double *>(ws);
                unsigned *bar = reinterpret_cast<unsigned *>(rd + (size_t)nrem * CH_NB);
                int *gflag = reinterpret_cast<int *>(bar + nrem);
                int *xmin = gflag + nrem, *xmax = xmin + nrem;
                HIP_TRY(hipMemsetAsync(xmin, 0x7f, sizeof(int) * nrem, ctx->stream));   // (min starts high; max at 0)
                chol_coop_kernel<<<((nrem + 7) / 8) * 8 * G, 256, shm2, ctx->stream>>>(C, F, jitter_rel, T, info, nmain, nrem, G, bar, gflag, rd,
                                                                                     xmin, xmax);
                if (getenv("CORAHIP_K2_COOP_DEBUG")) {
                    std::vector<int> hx(nrem);
                    HIP_TRY(hipMemcpyAsync(hx.data(), xmax, sizeof(int) * nrem, hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    int same = 0;
                    for (int v : hx) same += v == 0x100;
                    fprintf(stderr, "K2 cooperative kernel: %d matrices, %d workgroups each, %d groups on one XCC\n", nrem, G, same);
                }
            }
        } else {
            HIP_TRY(hipFuncSetAttribute((const void *)chol_ll_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            chol_ll_kernel<false><<<nl, 256, shm, ctx->stream>>>(C, F, jitter_rel, T, info);
        }
        LAUNCH_CHECK();
#if CHM_STAMPS
        {
            unsigned long long hs[8], z[8] = {0};
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_chm_stamps), sizeof(hs)));
            const double per = 1.0 / (double)std::max<unsigned long long>(hs[7], 1);
            fprintf(stderr, "K2 F=%d nl=%d tall=%d: cycles/wave  A %.0f  bar %.0f  B %.0f  bar %.0f  C %.0f  bar %.0f\n", F, nl, (int)tall,
                    hs[0] * per, hs[1] * per, hs[2] * per, hs[3] * per, hs[4] * per, hs[5] * per);
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_chm_stamps), z, sizeof(z)));
        }
#endif
    } else {
        const size_t shm = sizeof(double) * (CH_NB * (CH_NB + 1) + 2 * 64 * (CH_NB + 1) + 256) + 16;
        chol_kernel<<<nl, 256, shm, ctx->stream>>>(C, F, jitter_rel, T, info);
        LAUNCH_CHECK();
    }
    // eigen branch for the blocks whose Cholesky failed (host round trip: cold path only)
    std::vector<int32_t> hinfo(nl);
    HIP_TRY(hipMemcpyAsync(hinfo.data(), info, sizeof(int32_t) * nl, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // info = 3: a group of the cooperative kernel could not assemble - the one-workgroup kernel factors that matrix (the
    // same arithmetic: the reference's Cholesky route, not the eigen route)
    {
        bool redo = false;
        for (int l = 0; l < nl; l++)
            if (hinfo[l] == 3) {
                const size_t shm = sizeof(double) * (CH_NB * (CH_NB + 1) + 256 + 2 + CH_NB + 6 * 32 * CHM_S) + 16;
                chol_ll_kernel<true><<<1, 256, shm, ctx->stream>>>(C + (size_t)l * F * F, F, jitter_rel, T + (size_t)l * F * F, info + l);
                LAUNCH_CHECK();
                redo = true;
            }
        if (redo) {
            HIP_TRY(hipMemcpyAsync(hinfo.data(), info, sizeof(int32_t) * nl, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    std::vector<int32_t> list;
    for (int l = 0; l < nl; l++)
        if (hinfo[l] != 0) list.push_back(l);
    if (list.empty()) return 0;
    // process in batches bounded by scratch size (2 F^2 doubles per matrix)
    const size_t per = (size_t)2 * F * F * sizeof(double);
    const size_t batch = std::max<size_t>(1, std::min<size_t>(list.size(), ((size_t)1 << 31) / per));
    int32_t *dlist = nullptr;
    double *scratch = nullptr;
    HIP_TRY(hipMalloc((void **)&dlist, sizeof(int32_t) * list.size()));
    HIP_TRY(hipMalloc((void **)&scratch, per * batch));
    HIP_TRY(hipMemcpyAsync(dlist, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice, ctx->stream));
    const size_t shm = sizeof(double) * (2 * (F / 2 + 1) + 256 + F) + sizeof(int) * (2 * F + 4) + 16;
    for (size_t b0 = 0; b0 < list.size(); b0 += batch) {
        const int nb = (int)std::min(batch, list.size() - b0);
        jacobi_root_kernel<<<nb, 256, shm, ctx->stream>>>(C, dlist + b0, F, jitter_rel, eig_thresh, scratch,
                                                         scratch + (size_t)nb * F * F, T);
        LAUNCH_CHECK();
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipFree(dlist);
    (void)hipFree(scratch);
    return 0;
}
