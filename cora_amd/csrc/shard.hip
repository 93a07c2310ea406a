// shard.hip - the frequency-sharded form of the path for callers that do their own message passing (MPI through
// caput / mpi4py, the reference's distribution mechanism: cora/core/skysim.py:97-110,125-134): the shard plan and the
// two data movements around the ONE exchange of the warm path - the all-to-all of factor ROW blocks - as plain C
// entry points.  The exchange itself belongs to the caller (MPI_Alltoall on the host or device buffers, RCCL,
// files ...); cora_amd/parallel.py uses the same entry points around torch.distributed.
//
//   stage A  l-sharded: rank r holds C_l / factors T_l for its block of multipoles [n_r, F, F]   (skysim.py:97-103)
//   pack     send[q][k][i][:] = T_local[k][q nnu + i][:]   - slab q (the rows of rank q's channels), l rows k >= n_r zero
//   exchange all-to-all of the slabs [l_stride, nnu, F] (slab q -> rank q)
//   unpack   T_rows[l_off(r) + k][i][:] = recv[r][k][i][:], k < counts[r]  - the l blocks in rank order, compacted
//   stage B  nu-sharded: corahip_draw_alm_philox_rows(T_rows) + corahip_alm2map for channels [nu0, nu0 + nnu)
//
// HBM-bound row copies: F doubles (>= 2 KB at F = 256) per row, 16 bytes per lane.
#include "common.h"

#include <algorithm>

// grid: (l_stride, world); one workgroup copies the nnu rows of one (slab, l) pair
__global__ void __launch_bounds__(256)
factor_rows_pack_kernel(const double *__restrict__ T_local, int n_local, int l_stride, int F, int nnu,
                        double *__restrict__ send) {
    const int k = blockIdx.x, q = blockIdx.y;
    const size_t row2 = (size_t)F / 2;                                   // double2 per row (F even) - else scalar path
    double *dst = send + ((size_t)q * l_stride + k) * nnu * F;
    if (k >= n_local) {
        for (size_t e = threadIdx.x; e < (size_t)nnu * F; e += blockDim.x) dst[e] = 0.0;
        return;
    }
    const double *src = T_local + ((size_t)k * F + (size_t)q * nnu) * F;  // rows q nnu .. of T_k: contiguous nnu * F
    if ((F & 1) == 0) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (size_t e = threadIdx.x; e < (size_t)nnu * row2; e += blockDim.x) d2[e] = s2[e];
    } else {
        for (size_t e = threadIdx.x; e < (size_t)nnu * F; e += blockDim.x) dst[e] = src[e];
    }
}

// the l blocks of the ranks (offset in the compacted stack, length) travel as a kernel ARGUMENT: no table upload, no
// synchronisation on the cold path.  An MI355X node has 8 GPUs; 64 ranks is far beyond any single all-to-all group.
#define SHARD_MAX_WORLD 64
struct shard_blocks {
    int32_t off[SHARD_MAX_WORLD];
    int32_t cnt[SHARD_MAX_WORLD];
};
// grid: (max count, world)
__global__ void __launch_bounds__(256)
factor_rows_unpack_kernel(const double *__restrict__ recv, shard_blocks blk, int l_stride, int nnu, int F,
                          double *__restrict__ T_rows) {
    const int k = blockIdx.x, r = blockIdx.y;
    if (k >= blk.cnt[r]) return;
    const size_t n = (size_t)nnu * F;
    const double *src = recv + ((size_t)r * l_stride + k) * n;
    double *dst = T_rows + (size_t)(blk.off[r] + k) * n;
    if ((n & 1) == 0) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (size_t e = threadIdx.x; e < n / 2; e += blockDim.x) d2[e] = s2[e];
    } else {
        for (size_t e = threadIdx.x; e < n; e += blockDim.x) dst[e] = src[e];
    }
}

extern "C" {

int corahip_shard_plan(int L, int F, int rank, int world, corahip_shard *out) {
    ARG_CHECK(out != nullptr && L >= 1 && F >= 1 && world >= 1 && rank >= 0 && rank < world);
    // contiguous, balanced l and channel ranges (cost per l and per channel is uniform): cora_amd/parallel.py shard_plan
    const int l_shard = (L + world - 1) / world;
    out->l_lo = std::min(rank * l_shard, L);
    out->l_hi = std::min(out->l_lo + l_shard, L);
    out->l_shard = l_shard;
    out->l_pad = l_shard * world;
    const int base = F / world, extra = F % world;
    out->nnu = base + (rank < extra ? 1 : 0);
    out->nu0 = rank * base + std::min(rank, extra);
    out->rows_exchange = (F % world == 0) ? 1 : 0;
    out->L = L;
    return 0;
}

int corahip_factor_rows_pack(corahip_ctx *ctx, const double *T_local, int n_local, int l_stride, int F, int world,
                             double *send) {
    ARG_CHECK(ctx != nullptr && send != nullptr && (T_local != nullptr || n_local == 0));
    ARG_CHECK(F >= 1 && world >= 1 && F % world == 0 && n_local >= 0 && l_stride >= n_local && l_stride >= 1);
    dim3 grid((unsigned)l_stride, (unsigned)world);
    factor_rows_pack_kernel<<<grid, 256, 0, ctx->stream>>>(T_local, n_local, l_stride, F, F / world, send);
    LAUNCH_CHECK();
    return 0;
}

int corahip_factor_rows_unpack(corahip_ctx *ctx, const double *recv, const int32_t *host_counts, int world, int l_stride,
                               int nnu, int F, double *T_rows) {
    ARG_CHECK(ctx != nullptr && recv != nullptr && host_counts != nullptr && T_rows != nullptr);
    ARG_CHECK(world >= 1 && world <= SHARD_MAX_WORLD && l_stride >= 1 && nnu >= 1 && F >= 1);
    shard_blocks blk;
    int off = 0, cmax = 0;
    for (int r = 0; r < world; r++) {
        ARG_CHECK(host_counts[r] >= 0 && host_counts[r] <= l_stride);
        blk.off[r] = off;
        blk.cnt[r] = host_counts[r];
        off += host_counts[r];
        cmax = std::max(cmax, host_counts[r]);
    }
    if (cmax == 0) return 0;
    dim3 grid((unsigned)cmax, (unsigned)world);
    factor_rows_unpack_kernel<<<grid, 256, 0, ctx->stream>>>(recv, blk, l_stride, nnu, F, T_rows);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
