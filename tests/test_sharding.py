"""Frequency / l sharding plan used for N > 1 GPUs, exercised with world_size-2 gloo on CPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_covers_everything():
    from cora_amd.parallel import shard_plan

    for L, F, world in ((2049, 256, 8), (129, 16, 2), (65, 8, 4), (10, 7, 3)):
        seen_l, seen_nu = np.zeros(L, int), np.zeros(F, int)
        for r in range(world):
            p = shard_plan(L, F, r, world)
            seen_l[p.l_lo:p.l_hi] += 1
            seen_nu[p.nu0:p.nu0 + p.nnu] += 1
            assert p.l_pad == p.l_shard * world >= L
        assert np.all(seen_l == 1) and np.all(seen_nu == 1)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from cora_amd.parallel import allgather_factors, shard_plan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, F = 11, 3
    p = shard_plan(L, F, rank, world)
    full = torch.arange(L * F * F, dtype=torch.float64).reshape(L, F, F)
    info_full = (torch.arange(L) % 2).to(torch.int32)
    T, info = allgather_factors(full[p.l_lo:p.l_hi].clone(), info_full[p.l_lo:p.l_hi].clone(), p)
    ok = bool(torch.equal(T, full) and torch.equal(info, info_full))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_allgather_factors_gloo_world2():
    """l-sharded factors -> every rank holds the full [L,F,F] stack (the single exchange step)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
