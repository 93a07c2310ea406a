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


# ------------------------------------------------------------------ folded channel assignment (parallel.fold_chunks)
def test_folded_plan_covers_every_channel_once_and_balances_the_draw():
    """fold=True: rank r owns chunks r and 2N-1-r of 2N; every channel exactly once, every rank the same number of
    triangular terms sum(nu + 1) - the draw's cost - where the contiguous split gives the last rank (2N-1) x the first."""
    from cora_amd.parallel import fold_chunks, fold_permutation, shard_plan

    for F, world in ((256, 8), (512, 8), (1024, 8), (32, 2), (64, 4)):
        seen = np.zeros(F, int)
        work = []
        for r in range(world):
            p = shard_plan(100, F, r, world, fold=True)
            assert p.chunks == fold_chunks(F, r, world) and p.nnu == F // world and p.nu0 == p.chunks[0][0]
            ch = np.concatenate([np.arange(a, a + n) for a, n in p.chunks])
            seen[ch] += 1
            work.append(int((ch + 1).sum()))
        assert np.all(seen == 1) and len(set(work)) == 1, work
        perm = fold_permutation(F, world)
        assert sorted(perm) == list(range(F))
        contiguous = [int((np.arange(r * F // world, (r + 1) * F // world) + 1).sum()) for r in range(world)]
        assert max(contiguous) > work[0] > min(contiguous) and work[0] * world == sum(contiguous)    # (the mean)
        if world == 8:
            assert max(contiguous) > 1.8 * work[0]
    with pytest.raises(ValueError):
        fold_chunks(100, 0, 8)
    # one rank: nothing to fold
    assert shard_plan(10, 16, 0, 1, fold=True).chunks == ((0, 16),)


def _worker_fold(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from cora_amd.parallel import exchange_factor_rows, shard_plan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, F = 13, 8 * world
    p = shard_plan(L, F, rank, world, fold=True)
    full = torch.arange(L * F * F, dtype=torch.float64).reshape(L, F, F)
    info_full = (torch.arange(L) % 2).to(torch.int32)
    Tr, info = exchange_factor_rows(full[p.l_lo:p.l_hi].clone(), info_full[p.l_lo:p.l_hi].clone(), p)
    ch = torch.cat([torch.arange(a, a + n) for a, n in p.chunks])
    ok = bool(torch.equal(Tr, full[:, ch, :]) and torch.equal(info, info_full))
    # the union of the ranks' row blocks is every row exactly once
    got = [None] * world
    dist.all_gather_object(got, ch.tolist())
    ok = ok and sorted(c for g in got for c in g) == list(range(F))
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_folded_factor_row_exchange_gloo(world):
    """The factor row-block all-to-all under the folded assignment: rank r ends up with T_l[its two chunks, :] for
    every l, in the local order of its buffers; the union over the ranks is the whole stack."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker_fold, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


# ------------------------------------------------------------------ a hung exchange ends the job with the stage's name
def _worker_hang(rank, world, port):
    """Rank 0 enters an exchange; rank 1 joins the group and then never calls the collective."""
    import time

    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from cora_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CORA_DIST_TIMEOUT_S"] = "3"
    parallel.start_watchdog(rank, world)
    with parallel.exchange_stage("init_process_group (test)"):
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=parallel.dist_timeout())
    if rank == 1:
        time.sleep(30)               # the missing peer (killed by the test once rank 0 has given up)
        return
    p = parallel.shard_plan(11, 4, rank, world)
    T = torch.zeros((p.l_hi - p.l_lo, 4, 4), dtype=torch.float64)
    info = torch.zeros((p.l_hi - p.l_lo,), dtype=torch.int32)
    parallel.exchange_factor_rows(T, info, p)        # blocks: the peer never arrives
    os._exit(0)                      # (not reached)


def test_a_hung_exchange_exits_nonzero_and_names_its_stage(capfd):
    """The first real multi-GPU run must not be able to hang the driver: a collective whose peer never arrives ends the
    rank that waits - non-zero exit code, the exchange's name on stderr - through the stage guard of cora_amd.parallel
    (`exchange_stage` + watchdog, CORA_DIST_TIMEOUT_S) or through the backend's own timeout, whichever fires first."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_hang, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    procs[0].join(timeout=90)
    code = procs[0].exitcode
    procs[1].terminate()
    procs[1].join(timeout=30)
    assert code is not None and code != 0, code
    err = capfd.readouterr().err
    assert "all-to-all #2 (factor row blocks -> channel shards)" in err or "Timed out" in err or "timeout" in err.lower(), err[-2000:]


def test_exchange_stage_bookkeeping_and_watchdog_trigger(monkeypatch):
    """exchange_stage nests and restores, records the last completed stage; the watchdog calls its exit function with
    code 3 once a stage has been active for longer than CORA_DIST_TIMEOUT_S (injected exit function: same process)."""
    import time

    from cora_amd import parallel

    monkeypatch.setenv("CORA_DIST_TIMEOUT_S", "0.4")
    monkeypatch.setitem(parallel._STAGE, "thread", None)
    assert abs(parallel.dist_timeout().total_seconds() - 0.4) < 1e-9
    with parallel.exchange_stage("outer", sync=False):
        assert parallel._STAGE["name"] == "outer"
        with parallel.exchange_stage("inner", sync=False):
            assert parallel._STAGE["name"] == "inner"
        assert parallel._STAGE["name"] == "outer" and parallel._STAGE["last"] == "inner"
    assert parallel._STAGE["name"] is None and parallel._STAGE["last"] == "outer"
    fired = []
    parallel.start_watchdog(0, 1, exit_fn=fired.append)
    with parallel.exchange_stage("stuck exchange", sync=False):
        t0 = time.time()
        while not fired and time.time() - t0 < 10:
            time.sleep(0.05)
    assert fired == [3]
    monkeypatch.setitem(parallel._STAGE, "thread", None)
