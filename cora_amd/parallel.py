"""Multi-GPU decomposition of the hot path (one process per GPU, torch.distributed / RCCL).

Mirrors what the reference does with caput.mpiarray (cora/core/skysim.py:97-134), re-cut for
a node of MI355X GPUs:

  stage A (cold)  K1 C_l integration sharded over CHANNEL PAIRS (the table profile of a pair serves
                  every l, so an l-shard would repeat it on every rank), one small all-to-all
                  (F(F+1)/2 * L / N doubles per rank) turns pair shards into l shards, K2 factors
                  the rank's contiguous l range
  exchange        all-to-all of factor ROW blocks: rank r receives T_l[nu in its channels, :] for all
                  l - 1/N of the traffic and memory of an all-gather of the [L, F, F] stack
                  (all-gather is kept for separable models and for F not divisible by N)
  stage B (warm)  nu-sharded: every rank generates the same global normal stream (counter-based,
                  so it is a function of (seed, position) only), draws a_lm for its own channels
                  and synthesises them; maps stay on the rank that made them (the reference's
                  ``MPIArray.wrap(sky, axis=0)``).

No reduction exists on this path, so there is no all-reduce.
"""
from dataclasses import dataclass


@dataclass
class ShardPlan:
    rank: int
    world: int
    l_lo: int      # this rank integrates / factors l in [l_lo, l_hi)
    l_hi: int
    l_shard: int   # padded shard length (equal on all ranks, all-gather friendly)
    l_pad: int     # l_shard * world >= L
    nu0: int       # this rank synthesises channels [nu0, nu0 + nnu)
    nnu: int
    L: int = 0     # total number of multipoles (lmax + 1)


def shard_plan(L, F, rank, world):
    """Contiguous, balanced l and channel ranges (cost per l and per channel is uniform)."""
    l_shard = (L + world - 1) // world
    l_lo = min(rank * l_shard, L)
    l_hi = min(l_lo + l_shard, L)
    base, extra = divmod(F, world)
    nnu = base + (1 if rank < extra else 0)
    nu0 = rank * base + min(rank, extra)
    return ShardPlan(rank, world, l_lo, l_hi, l_shard, l_shard * world, nu0, nnu, L)


def allgather_factors(T_local, info_local, plan):
    """All-gather the per-rank factor shards into the full [L, F, F] / [L] stacks."""
    import torch
    import torch.distributed as dist

    F = T_local.shape[1]
    pad_T = torch.zeros((plan.l_shard, F, F), dtype=T_local.dtype, device=T_local.device)
    pad_i = torch.zeros((plan.l_shard,), dtype=info_local.dtype, device=info_local.device)
    n = plan.l_hi - plan.l_lo
    pad_T[:n].copy_(T_local)
    pad_i[:n].copy_(info_local)
    T_all = torch.empty((plan.l_pad, F, F), dtype=T_local.dtype, device=T_local.device)
    i_all = torch.empty((plan.l_pad,), dtype=info_local.dtype, device=info_local.device)
    dist.all_gather_into_tensor(T_all, pad_T)
    dist.all_gather_into_tensor(i_all, pad_i)
    return T_all[: plan.L], i_all[: plan.L]


def _all_to_all(send, world):
    """send [world, ...] -> recv [world, ...] (slab q goes to rank q).  RCCL all_to_all_single; backends
    without it for device tensors (gloo, used by the single-GPU test hooks) go through an all-gather."""
    import torch
    import torch.distributed as dist

    send = send.contiguous()
    recv = torch.empty_like(send)
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(recv, send)
        return recv
    rank = dist.get_rank()
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send)
    for q in range(world):
        recv[q].copy_(parts[q][rank])
    return recv


def exchange_pair_slabs(slabs, plan):
    """K1 pair shards [world, npl, l_shard] (slab q = this rank's pairs at rank q's multipoles) ->
    [world, npl, l_shard] (slab r = rank r's pairs at THIS rank's multipoles)."""
    assert slabs.shape[0] == plan.world and slabs.shape[2] == plan.l_shard
    return _all_to_all(slabs, plan.world)


def exchange_factor_rows(T_local, info_local, plan):
    """l-sharded factors [l_hi - l_lo, F, F] -> (T_rows [L, nnu, F], info [L]): every rank ends up with the
    rows of ALL T_l that its own channels need.  Requires F % world == 0 (equal row blocks)."""
    import torch
    import torch.distributed as dist

    F = T_local.shape[1]
    W = plan.world
    assert F % W == 0 and plan.nnu == F // W
    n = plan.l_hi - plan.l_lo
    pad_T = torch.zeros((plan.l_shard, F, F), dtype=T_local.dtype, device=T_local.device)
    pad_i = torch.zeros((plan.l_shard,), dtype=info_local.dtype, device=info_local.device)
    pad_T[:n].copy_(T_local)
    pad_i[:n].copy_(info_local)
    # [l, (dst, row), k] -> [dst, l, row, k]
    send = pad_T.view(plan.l_shard, W, plan.nnu, F).permute(1, 0, 2, 3).contiguous()
    recv = _all_to_all(send, W)                     # [src, l_shard, nnu, F] = [L_pad, nnu, F]
    i_all = torch.empty((plan.l_pad,), dtype=info_local.dtype, device=info_local.device)
    dist.all_gather_into_tensor(i_all, pad_i)
    return recv.view(plan.l_pad, plan.nnu, F)[: plan.L], i_all[: plan.L]
