"""Multi-GPU decomposition of the hot path (one process per GPU, torch.distributed / RCCL).

Mirrors what the reference does with caput.mpiarray (cora/core/skysim.py:97-134), re-cut for
a node of MI355X GPUs:

  stage A (cold)  l-sharded : K1 C_l integration and K2 factors for the rank's contiguous l range
  exchange        ONE all-gather of the factor stack [L, F, F] (+ info) - nothing else ever moves
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
