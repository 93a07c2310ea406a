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


def _worker_a2a(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from cora_amd.parallel import exchange_factor_rows, exchange_pair_slabs, shard_plan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, F = 11, 4
    p = shard_plan(L, F, rank, world)
    # factor rows: rank r must end up with T_l[nu in its channels, :] for every l
    full = torch.arange(L * F * F, dtype=torch.float64).reshape(L, F, F)
    info_full = (torch.arange(L) % 2).to(torch.int32)
    Tr, info = exchange_factor_rows(full[p.l_lo:p.l_hi].clone(), info_full[p.l_lo:p.l_hi].clone(), p)
    ok = bool(torch.equal(Tr, full[:, p.nu0:p.nu0 + p.nnu, :]) and torch.equal(info, info_full))
    # pair slabs: value encodes (source rank, destination rank, slot, l): slab q of rank r -> slab r of rank q
    npl = 5
    send = torch.empty((world, npl, p.l_shard), dtype=torch.float64)
    for dst in range(world):
        for k in range(npl):
            send[dst, k] = 1000 * rank + 100 * dst + 10 * k + torch.arange(p.l_shard, dtype=torch.float64) / 100
    recv = exchange_pair_slabs(send, p)
    for src in range(world):
        for k in range(npl):
            want = 1000 * src + 100 * rank + 10 * k + torch.arange(p.l_shard, dtype=torch.float64) / 100
            ok = ok and bool(torch.equal(recv[src, k], want))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_all_to_all_exchanges_gloo_world2():
    """pair-shard -> l-shard all-to-all of K1 and the factor row-block all-to-all (21cm multi-GPU path)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_a2a, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_chan(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from cora_amd.parallel import allgather_channels, shard_plan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for F in (8, 7, 2):                      # even split, uneven split, fewer channels than... ranks + 0 (world 3: one empty)
        full = torch.arange(F * 1 * 3 * 3, dtype=torch.float64).reshape(F, 1, 3, 3)
        p = shard_plan(5, F, rank, world)
        got = allgather_channels(full[p.nu0:p.nu0 + p.nnu].clone(), F)
        ok = ok and bool(torch.equal(got, full))
        cfull = torch.complex(full, -2.0 * full)       # the a_lm squares are complex128
        got = allgather_channels(cfull[p.nu0:p.nu0 + p.nnu].clone(), F)
        ok = ok and bool(torch.equal(got, cfull))
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_channels_gloo(world):
    """alms=True on the l-distributed path returns the full a_lm array on every rank (cora/core/skysim.py:123-125):
    channel shards of uneven length are padded, all-gathered and re-assembled."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker_chan, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
