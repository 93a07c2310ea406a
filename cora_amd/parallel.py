"""Multi-GPU decomposition of the hot path (one process per GPU, torch.distributed / RCCL).

Mirrors what the reference does with caput.mpiarray (cora/core/skysim.py:97-134), re-cut for
a node of MI355X GPUs:

  stage A (cold)  K1 C_l integration sharded over CHANNEL PAIRS (the table profile of a pair serves
                  every l, so an l-shard would repeat it on every rank), one small all-to-all
                  (F(F+1)/2 * L / N doubles per rank) turns pair shards into l shards, K2 factors
                  the rank's contiguous l range
  exchange        all-to-all of factor ROW blocks: rank r receives T_l[nu in its channels, :] for all
                  l - 1/N of the traffic and memory of an all-gather of the [L, F, F] stack
                  (the all-gather is kept for F not divisible by N; separable models factor their one
                  F x F block on every rank and exchange nothing); pack / unpack around the exchange are
                  the C entry points of csrc/shard.hip, the same an MPI caller of the ABI uses
  stage B (warm)  nu-sharded: every rank generates the same global normal stream (counter-based,
                  so it is a function of (seed, position) only), draws a_lm for its own channels
                  and synthesises them; maps stay on the rank that made them (the reference's
                  ``MPIArray.wrap(sky, axis=0)``).

No reduction exists on this path, so there is no all-reduce.

Channel assignment.  Default: the reference's contiguous split (rank r owns channels [r F / N, (r + 1) F / N)).  The
draw is the one stage whose cost depends on WHICH channels a rank owns (T_l is lower triangular: channel nu takes
nu + 1 terms), so the last rank of a contiguous split does (2N - 1) times the MFMA work of the first.  ``fold=True``
(opt-in: it gives up the contiguous frequency blocks ``MPIArray.wrap(sky, axis=0)`` promises) cuts the channels into
2 N chunks and gives rank r chunks r and 2 N - 1 - r: every rank then does the same work; ``SkyShard.channels`` is the
list of global channel indices of the rank's maps, in the order of its buffers.
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
    nu0: int       # this rank synthesises channels [nu0, nu0 + nnu) - folded: its first chunk starts at nu0, nnu in total
    nnu: int
    L: int = 0     # total number of multipoles (lmax + 1)
    chunks: tuple = ()   # ((first channel, count), ...): one block, or the two chunks of a folded shard


def fold_chunks(F, rank, world):
    """The two chunks (first channel, count) of rank ``rank`` under the folded assignment: chunks r and 2 N - 1 - r of
    2 N equal chunks.  Needs F divisible by 8 N (whole a_lm cells of four channels per chunk)."""
    if F % (2 * world) or (F // (2 * world)) % 4:
        raise ValueError("fold=True needs F divisible by 8 * world (F = %d, world = %d)" % (F, world))
    c = F // (2 * world)
    return ((rank * c, c), ((2 * world - 1 - rank) * c, c))


def fold_permutation(F, world):
    """Global channel index of every row of the rank-major, local-order arrangement of the folded split: rows
    [q F / N, (q + 1) F / N) are rank q's channels (its low chunk, then its high chunk)."""
    import numpy as np

    return np.concatenate([np.arange(a, a + n) for q in range(world) for a, n in fold_chunks(F, q, world)])


def shard_plan(L, F, rank, world, fold=False):
    """Contiguous, balanced l ranges and the rank's channels: the reference's contiguous block (cost per l and per
    channel is uniform except in the draw), or with ``fold`` the two chunks of :func:`fold_chunks`."""
    l_shard = (L + world - 1) // world
    l_lo = min(rank * l_shard, L)
    l_hi = min(l_lo + l_shard, L)
    if fold and world > 1:
        ch = fold_chunks(F, rank, world)
        return ShardPlan(rank, world, l_lo, l_hi, l_shard, l_shard * world, ch[0][0], 2 * ch[0][1], L, ch)
    base, extra = divmod(F, world)
    nnu = base + (1 if rank < extra else 0)
    nu0 = rank * base + min(rank, extra)
    return ShardPlan(rank, world, l_lo, l_hi, l_shard, l_shard * world, nu0, nnu, L, ((nu0, nnu),))


# ---- exchange guard: a hung collective must end the job with the stage's name, not hang it --------------------------------
# RCCL collectives are enqueued asynchronously: a peer that never arrives shows up as a host that blocks in a LATER
# synchronisation.  Two defences: (i) `init_process_group(timeout=...)` (:func:`dist_timeout`) makes torch's own
# watchdog abort a collective that does not complete; (ii) every exchange of this module runs inside :func:`exchange_stage`,
# which records its name, and a daemon thread (:func:`start_watchdog`) ends the process with exit code 3 and the name of
# the stage it is in - or was last in - when nothing has moved for the timeout.  CORA_DIST_TIMEOUT_S (default 300);
# CORA_DIST_SYNC_EXCHANGES=1 synchronises behind every exchange so that the stage named is the one that hangs.
import contextlib as _contextlib
import threading as _threading

_STAGE = {"name": None, "since": None, "last": None, "beat": None, "thread": None}


def dist_timeout():
    """The timeout (datetime.timedelta) for ``init_process_group`` and the watchdog: CORA_DIST_TIMEOUT_S, default 300 s."""
    import datetime
    import os

    return datetime.timedelta(seconds=float(os.environ.get("CORA_DIST_TIMEOUT_S", "300")))


def start_watchdog(rank=0, world=1, exit_fn=None):
    """Starts (once) the thread that ends the process when a stage entered through :func:`exchange_stage` has been
    active for longer than :func:`dist_timeout`: prints the stage, rank and world to stderr and leaves with code 3
    (``os._exit``: the main thread is blocked inside a collective and will not unwind)."""
    import os
    import sys
    import time

    if _STAGE["thread"] is not None:
        return _STAGE["thread"]
    limit = dist_timeout().total_seconds()
    leave = exit_fn or (lambda code: os._exit(code))

    def run():
        while True:
            time.sleep(min(1.0, limit / 4.0))
            name, since = _STAGE["name"], _STAGE["since"]
            if name is not None and since is not None and time.time() - since > limit:
                sys.stderr.write("cora_amd.parallel: stage '%s' has not completed after %.0f s on rank %d of %d (last completed "
                                 "stage: %s): a peer is missing or a collective hangs - aborting\n"
                                 % (name, limit, rank, world, _STAGE["last"]))
                sys.stderr.flush()
                leave(3)
                return

    t = _threading.Thread(target=run, name="cora-dist-watchdog", daemon=True)
    _STAGE["thread"] = t
    t.start()
    return t


@_contextlib.contextmanager
def exchange_stage(name, sync=None):
    """Marks the calling thread as being inside exchange ``name`` (seen by the watchdog); with ``sync`` (default: the
    environment's CORA_DIST_SYNC_EXCHANGES) the device is synchronised before the stage is left, so that a hang is
    attributed to this stage and not to a later synchronisation point."""
    import os
    import time

    prev = (_STAGE["name"], _STAGE["since"])
    _STAGE["name"], _STAGE["since"] = name, time.time()
    try:
        yield
        if os.environ.get("CORA_DIST_SYNC_EXCHANGES") if sync is None else sync:
            import torch

            if torch.cuda.is_available():
                torch.cuda.synchronize()
        _STAGE["last"] = name
    finally:
        _STAGE["name"], _STAGE["since"] = prev


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
    with exchange_stage("all-gather of the factor shards"):
        dist.all_gather_into_tensor(T_all, pad_T)
        dist.all_gather_into_tensor(i_all, pad_i)
    return T_all[: plan.L], i_all[: plan.L]


def allgather_channels(shard, F):
    """Channel shards [nnu_r, ...] of all ranks (the axis-0 split of :func:`shard_plan`, uneven when F % world != 0) ->
    the full [F, ...] tensor on every rank (the reference's ``alm_array.allgather()``, cora/core/skysim.py:123-125)."""
    import torch
    import torch.distributed as dist

    if shard.is_complex():     # (the collectives move real numbers: RCCL has no complex type)
        return torch.view_as_complex(allgather_channels(torch.view_as_real(shard).contiguous(), F))
    world = dist.get_world_size()
    nmax = (F + world - 1) // world
    pad = torch.zeros((nmax,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    pad[: shard.shape[0]].copy_(shard)
    full = torch.empty((world * nmax,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    with exchange_stage("all-gather of the channel shards"):
        dist.all_gather_into_tensor(full, pad)
    parts = []
    for r in range(world):
        sp = shard_plan(1, F, r, world)
        parts.append(full[r * nmax : r * nmax + sp.nnu])
    return torch.cat(parts, dim=0)


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
    with exchange_stage("all-to-all #1 (C_l pair slabs -> multipole shards)"):
        return _all_to_all(slabs, plan.world)


def _rows_pack(T_local, l_stride, world):
    """[n, F, F] -> [world, l_stride, F / world, F]: corahip_factor_rows_pack on the device; for the CPU tensors of
    the gloo tests of the exchange pattern (tests/test_sharding.py) the same permutation in torch."""
    import torch

    if T_local.is_cuda:
        from . import _lib

        return _lib.get_context(T_local.device.index).factor_rows_pack(T_local.contiguous(), l_stride, world)
    n, F, _ = T_local.shape
    pad = torch.zeros((l_stride, F, F), dtype=T_local.dtype)
    pad[:n].copy_(T_local)
    return pad.view(l_stride, world, F // world, F).permute(1, 0, 2, 3).contiguous()


def _rows_unpack(recv, counts):
    """[world, l_stride, nnu, F] + per-rank l counts -> [sum counts, nnu, F] (corahip_factor_rows_unpack on the device)."""
    import torch

    if recv.is_cuda and recv.shape[0] <= 64:      # (the kernel takes the l-block table as an argument: <= 64 ranks)
        from . import _lib

        return _lib.get_context(recv.device.index).factor_rows_unpack(recv, counts)
    return torch.cat([recv[r, : counts[r]] for r in range(recv.shape[0])], dim=0).contiguous()


def exchange_factor_rows(T_local, info_local, plan):
    """l-sharded factors [l_hi - l_lo, F, F] -> (T_rows [L, nnu, F], info [L]): every rank ends up with the
    rows of ALL T_l that its own channels need.  Requires F % world == 0 (equal row blocks).  The pack / unpack
    around the all-to-all are the C entry points a caput / mpi4py caller would use (include/corahip.h,
    INTEGRATION.md section 3).  A folded plan (two chunks per rank) first brings the rows into rank-major order."""
    import torch
    import torch.distributed as dist

    F = T_local.shape[1]
    W = plan.world
    assert F % W == 0 and plan.nnu == F // W
    if len(plan.chunks) > 1:
        perm = torch.as_tensor(fold_permutation(F, W), device=T_local.device)
        T_local = T_local.index_select(1, perm)
    n = plan.l_hi - plan.l_lo
    pad_i = torch.zeros((plan.l_shard,), dtype=info_local.dtype, device=info_local.device)
    pad_i[:n].copy_(info_local)
    send = _rows_pack(T_local, plan.l_shard, W)     # [dst, l, row, k]
    with exchange_stage("all-to-all #2 (factor row blocks -> channel shards)"):
        recv = _all_to_all(send, W)                     # [src, l_shard, nnu, F]
        i_all = torch.empty((plan.l_pad,), dtype=info_local.dtype, device=info_local.device)
        dist.all_gather_into_tensor(i_all, pad_i)
    counts = [max(0, min(plan.L, (r + 1) * plan.l_shard) - min(plan.L, r * plan.l_shard)) for r in range(W)]
    return _rows_unpack(recv, counts), i_all[: plan.L]


def device_memory_bytes(default=288e9):
    """The device memory the C side sizes the ring against (``prop.totalGlobalMem``, ~3.09e11 on a 288 GiB part) when a
    GPU is visible, else ``default`` (the planning figure of the no-GPU memory model)."""
    try:
        import torch

        if torch.cuda.is_available():
            return float(torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory)
    except Exception:
        pass
    return float(default)


def numpy_ring_bytes(F, lmax, device_bytes=None):
    """Bytes of the normal-stream ring ``corahip_draw_alm_numpy`` allocates by default (csrc/drawstream.hip): two slots
    in 2 GiB while the whole stream of a realisation (16 F nalm bytes) is at most 1/8 of the device memory (the measured
    optimum of the cfg-3 step: more ranges cost launches, fewer cost the kernel behind the draw its cache / TLB state),
    else in 1/16 of the memory - a slot never smaller than the normals of l = lmax, the ring never larger than the stream.
    ``device_bytes``: what the C side reads from the device (:func:`device_memory_bytes`; 288e9 without a GPU).  The
    environment's ``CORAHIP_RING_MB`` overrides the default on the C side and is honoured here as well."""
    import os

    if device_bytes is None:
        device_bytes = device_memory_bytes()
    L = lmax + 1
    if os.environ.get("CORAHIP_RING_MB"):
        ring = float(max(1, int(os.environ["CORAHIP_RING_MB"])) << 20)
        return int(min(8 * F * L * (L + 1), 2 * max(ring / 2, 16.0 * F * L)))
    stream = 8 * F * L * (L + 1)
    ring = 2.0**31 if stream <= device_bytes / 8 else max(device_bytes / 16, 2.0**31)
    slot = max(ring / 2, 16.0 * F * L)
    return int(min(stream, 2 * slot))


def rank_memory_bytes(kinds, F, nside, lmax, world, sum_mode="joint", rng="philox", fold=False, device_bytes=288e9):
    """Device bytes ONE rank of a ``world``-rank job holds at the peak of a cold step: the buffers of
    :class:`SkyShard` / :class:`SkySum` counted from their shapes (no GPU needed; ``tests/test_host.py`` sums them for
    BASELINE configs[3] and [4] at 8 ranks against the 288 GB of an MI355X, ``tests/test_gpu_parity.py`` compares the
    single-rank figure with what torch actually allocates).

    kinds : per component "table21cm" | "separable";  rng : "philox" (no normal buffer) | "numpy" (numpy's own stream,
    PCG64 + ziggurat or the legacy MT19937 + polar method, generated range of multipoles by range on EVERY rank: the
    ring of :func:`numpy_ring_bytes` plus the generator's count / scan tables, ~0.38 bytes per normal);
    fold : the folded channel assignment (one more copy of the local factor block while its rows are permuted);
    device_bytes : the memory the ring is sized against (the 288e9 planning figure by default)."""
    L = lmax + 1
    nalm = L * (L + 1) // 2
    npix = 12 * nside * nside
    npair, nring = 2 * nside, 4 * nside - 1
    nnu = -(-F // world)
    lsh = -(-L // world)
    ntab = sum(1 for k in kinds if k == "table21cm")
    joint = len(kinds) > 1 and sum_mode == "joint"
    out = {}
    out["maps"] = 8 * npix * nnu
    out["a_lm"] = 16 * nalm * (-(-nnu // 4) * 4) * (1 if (joint or len(kinds) == 1) else 2)      # separate: + the component being added
    out["synthesis workspace (F_m cells)"] = nring * (-(-nnu // 8) * 2) * L * 64
    out["plan (recurrence coefficients, polar seeds, first-l tables, ring FFT tables)"] = (
        2 * 16 * nalm + (2 * 16 + 64) * L * npair + 4 * L * npair + int(1.0e8 * (nside / 1024.0) ** 2))   # (64: the four entry states per (m, ring) of K4's lane groups)
    if ntab:
        out["21cm tables + their transposed copy"] = 2 * 3 * 500 * 32768 * 8
        pairs = F * (F + 1) // 2
        # pair shard of K1 [pairs / world, L padded], the same after all-to-all #1, the C_l block and its factor
        out["C_l: pair slabs (sent + received), block of this rank's multipoles, its factors"] = (
            (2 * 8 * (-(-pairs // world)) * (lsh * world) if world > 1 else 8 * pairs * L) + 2 * 8 * lsh * F * F)
    if joint:
        out["summed covariance block"] = 8 * lsh * F * F
    ndraw = 1 if joint else len(kinds)
    if world > 1:
        # factor row blocks of this rank's channels + the slabs of all-to-all #2 while it runs
        # (+ a folded shard's row permutation in front of the exchange copies the local [lsh, F, F] factor block once)
        out["factor rows [L, nnu, F] (+ exchange slabs)"] = (ndraw * 8 * L * nnu * F + 2 * 8 * world * lsh * nnu * F
                                                             + (8 * lsh * F * F if fold else 0))
    elif not ntab:
        out["factor rows [L, nnu, F] (+ exchange slabs)"] = ndraw * 8 * L * nnu * F
    if rng == "numpy":
        nn = 2 * F * nalm
        out["numpy normal stream: ring of l ranges + generator tables"] = numpy_ring_bytes(F, lmax, device_bytes) + int(0.38 * nn)
    out["total"] = sum(out.values())
    return out


class SkyShard:
    """One rank's share of a Gaussian sky realisation (what ``Sky3d.getsky()`` does on one GPU, cora/core/maps.py:227-237,
    cut over ``world`` GPUs as described at the top of this module).  Everything the realisation reads is put in HBM
    by the constructor; ``factors()`` is the cold part (C_l integration, exchange, factorisation, exchange),
    ``realise(seed)`` draws with the device Philox stream - identical on every rank - and synthesises this rank's
    channels ``[nu0, nu0 + nnu)``.  The maps stay on the GPU (torch tensor ``[nnu, npix]``).

    model : a ``cora_amd`` Gaussian model exposing ``_clarray_plan`` (Corr21cm, ForegroundSCK subclasses)
    freq  : all F channel centres (MHz);  zromb : Romberg order of the channel average (``oversample``)
    distributed : run the exchanges (default: world > 1);  emulate_world : measurement hook - do the work of the most
        loaded rank of an ``emulate_world``-rank job on this GPU without any communication (K1: last pair shard,
        K2: rank 0's multipoles on random SPD blocks, K3-K5: the last channel shard, factors those of a random SPD stack);
        ``emulate_rank``: the channel shard of THAT rank instead of the last one (bench.py times every rank in turn).
    fold : the folded channel assignment (module docstring): this rank owns chunks r and 2 N - 1 - r of 2 N; its maps
        are the channels ``self.channels`` (global indices, buffer order).  Ignored for one rank.
    """

    def __init__(self, model, freq, nside, lmax, zromb=3, rank=0, world=1, ctx=None, distributed=None, emulate_world=0,
                 alm_buf=None, maps_buf=None, fold=False, emulate_rank=None):
        import numpy as np

        from . import _lib
        from .core import skysim

        self.ctx = ctx if ctx is not None else _lib.get_context()
        ctx = self.ctx
        self.freq = np.asarray(freq, dtype=np.float64)
        self.F = F = self.freq.size
        self.nside, self.lmax, self.L = int(nside), int(lmax), int(lmax) + 1
        self.rank, self.world = rank, world
        self.distributed = (world > 1) if distributed is None else bool(distributed)
        self.emulate_world = int(emulate_world)
        L = self.L
        self.fold = bool(fold)
        self.plan = shard_plan(L, F, rank, world, fold=self.fold)
        if self.emulate_world > 1:
            N = self.emulate_world
            er = N - 1 if emulate_rank is None else int(emulate_rank)
            self.plan = shard_plan(L, F, er, N, fold=self.fold)
            p0 = shard_plan(L, F, 0, N)
            self.plan.l_lo, self.plan.l_hi = p0.l_lo, p0.l_hi
        sp = self.plan
        self.nu0, self.nnu = sp.nu0, sp.nnu
        self.chunks = tuple(sp.chunks)
        self.folded = len(self.chunks) > 1
        self.channels = np.concatenate([np.arange(a, a + n) for a, n in self.chunks])   # global index of every map of this rank
        self.zint = zint = 2**zromb + 1 if zromb else 1
        # channel half-width exactly as skysim.clarray takes it: from the two smallest sorted frequencies (skysim.py:41-45)
        fsort = np.sort(self.freq)
        zhalf = abs(fsort[1] - fsort[0]) / 2.0 if F > 1 else 0.0
        za = (self.freq[:, None] + np.linspace(-zhalf, zhalf, zint)[None, :]).ravel() if zromb else self.freq.copy()
        self.w = ctx.to_device(skysim.romberg_weights(zromb))
        cplan = model._clarray_plan(model.angular_powerspectrum)
        if cplan is None:
            raise NotImplementedError(
                "SkyShard needs a model whose angular_powerspectrum is the library's table (21cm) or separable "
                "(ForegroundSCK) form; %s overrides it - integrate C_l with skysim.clarray_device and use "
                "skysim.mkfullsky_device(nu_range=...) instead" % type(model).__name__)
        nshard = max(world, self.emulate_world, 1)
        self.pair_sharded = (cplan["kind"] == "table21cm" and (self.distributed or self.emulate_world > 1)
                             and F % nshard == 0)
        if self.folded and not (self.pair_sharded or cplan["kind"] != "table21cm"):
            raise ValueError("fold=True needs the row-block exchange (F divisible by the number of ranks)")
        larr = np.arange(L, dtype=np.float64)
        if cplan["kind"] == "table21cm":
            p = cplan["prepare"](ctx, za)
            lx_full = np.log10(np.where(larr == 0.0, 1e-10, larr))
            self._lx = ctx.to_device(lx_full if self.pair_sharded else lx_full[sp.l_lo:sp.l_hi])
            self._k1 = [ctx.to_device(p[k]) for k in ("chi", "pfd", "f", "b")]
            self._tabs = (p["dd"], p["dv"], p["vv"], p["kperpmin"], p["kperpmax"], p["kparmax"])
            self._sep = None
        else:
            al, bcov = cplan["prepare"](larr.copy(), za)
            self._sep = (ctx.to_device(al[sp.l_lo:sp.l_hi]), ctx.to_device(bcov), ctx.to_device(al))
        nalm = L * (L + 1) // 2
        self.npix = 12 * self.nside * self.nside
        ctx.sht_plan(self.nside, self.lmax)
        # (a SkySum hands its components one shared pair of buffers)
        self.alm_buf = alm_buf if alm_buf is not None else ctx.empty((nalm, (self.nnu + 3) // 4, 2, 4))
        self.maps_buf = maps_buf if maps_buf is not None else ctx.empty((self.nnu, self.npix))
        ctx.workspace(ctx.alm2map_workspace_bytes(ctx.sht_plan(self.nside, self.lmax), self.nnu))
        self._emulated = None
        if self.emulate_world > 1:
            import torch

            # stand-in factors of the right shape and structure (Cholesky factors of random SPD blocks), built in
            # l-chunks so that F = 1024, L = 4097 (cfg 5) never holds more than one chunk of [*, F, F] temporaries
            rows = self.pair_sharded
            Tf = ctx.empty((L, self.nnu, F) if rows else (L, F, F))
            inf = torch.empty((L,), dtype=torch.int32, device=ctx.device)
            eye = 0.1 * torch.eye(F, device=ctx.device, dtype=torch.float64)
            lc = max(1, min(L, (1 << 28) // (F * F)))
            for l0 in range(0, L, lc):
                Cc = ctx.empty((min(lc, L - l0), F, F)).normal_()
                Tc, ic = ctx.factor_batched(Cc @ Cc.transpose(1, 2) + eye)
                Tf[l0:l0 + Tc.shape[0]].copy_(Tc[:, self._chan_index(), :] if rows else Tc)
                inf[l0:l0 + Tc.shape[0]].copy_(ic)
                del Cc, Tc, ic
            self._emulated = (Tf, inf, rows)
            # stand-in input of the rank's K2 (rank 0's multipoles): random SPD blocks - the emulated K1 only fills its own
            # pair shard, and an incomplete (or zero) block would time the failure path of the factorisation, not K2
            nl0 = sp.l_hi - sp.l_lo
            self._emu_C = ctx.empty((nl0, F, F))
            for l0 in range(0, nl0, lc):
                Cc = ctx.empty((min(lc, nl0 - l0), F, F)).normal_()
                self._emu_C[l0:l0 + Cc.shape[0]].copy_(Cc @ Cc.transpose(1, 2) + eye)
                del Cc

    def _chan_index(self):
        """Device index tensor of this rank's channels (cached)."""
        import torch

        if getattr(self, "_chan_idx", None) is None:
            self._chan_idx = torch.as_tensor(self.channels, device=self.ctx.device)
        return self._chan_idx

    # -- K1 in its two shardings -----------------------------------------------------------
    def _clarray_local(self):
        ctx = self.ctx
        if self._sep is not None:
            return ctx.clarray_separable(self._sep[0], self._sep[1], self.F, self.zint, self.w)
        return ctx.clarray_table21cm(*self._tabs, *self._k1, self.F, self.zint, self.w, self._lx)

    def _separable_factors(self):
        """Separable model C_l = A_l B (ForegroundSCK, cora/foreground/gaussianfg.py:107-130): the reference factors
        every jittered block C_l + 1e-14 max(diag C_l) I = A_l (B + 1e-14 max(diag B) I) on its own (skysim.py:115-119);
        that root is sqrt(A_l) root(B + 1e-14 max(diag B) I) for every l, so ONE factorisation serves all multipoles
        (SURVEY 8 a6).  Every rank makes it for itself - an F x F Cholesky - and keeps the rows of its own channels:
        no C_l stack, no exchange.  A_l = 0 (l = 0) gives the zero root the reference's eigen branch returns.  (With
        cond(B) ~ 1e19 the factor of a scaled block differs from the scaled factor at the 1e-7 level; T T^T = C_l
        holds to rounding either way: the near-singular contract of SURVEY 8 a8.)"""
        import torch

        ctx = self.ctx
        one = torch.ones((1,), dtype=torch.float64, device=ctx.device)
        bbar = ctx.clarray_separable(one, self._sep[1], self.F, self.zint, self.w)          # channel-averaged B [1, F, F]
        tb, ib = ctx.factor_batched(bbar)
        rows = tb[0].index_select(0, self._chan_index())
        T = torch.sqrt(self._sep[2])[:, None, None] * rows[None, :, :]                       # [L, nnu, F]
        info = ib.expand(self.L).contiguous()
        return T.contiguous(), info, True

    def _clarray_pairs(self, first, step):
        return self.ctx.clarray_table21cm_pairs(*self._tabs, *self._k1, self.F, self.zint, self.w, self._lx, first, step,
                                                self.plan.l_shard, nblocks=step)

    def covariance_shard(self):
        """C_l of this rank's multipoles ``[l_hi - l_lo, F, F]`` (all of them when not distributed) - the cold
        part up to, not including, the factorisation: K1 and, when it is pair-sharded, all-to-all #1.  Under
        ``emulate_world`` the rank's K1 work is done and a stand-in stack of the right shape returned."""
        ctx, sp = self.ctx, self.plan
        if self.emulate_world > 1:
            N = self.emulate_world
            if self._sep is None and self.pair_sharded:
                slab = self._clarray_pairs(N - 1, N)
                ctx.clarray_pairs_finish(slab.new_zeros((N,) + tuple(slab.shape[1:])), self.F, sp.l_hi - sp.l_lo)
            else:
                self._clarray_local()
            return self._emu_C
        if self.distributed and self._sep is None and self.pair_sharded:
            mine = exchange_pair_slabs(self._clarray_pairs(self.rank, self.world), sp)   # all-to-all #1
            return ctx.clarray_pairs_finish(mine, self.F, sp.l_hi - sp.l_lo)
        return self._clarray_local()

    def share_factors(self, T, info):
        """l-sharded factors of this rank -> (T, info, rows) as the draw takes them: the row-block all-to-all (#2)
        when F divides over the ranks, else the all-gather of the stack; a single rank keeps what it has."""
        sp = self.plan
        if not self.distributed:
            return T, info, False
        if self.F % max(self.world, 1) == 0:
            Tr, ia = exchange_factor_rows(T, info, sp)
            return Tr, ia, True
        Ta, ia = allgather_factors(T, info, sp)
        return Ta, ia, False          # (never folded: fold_chunks needs F divisible by 8 N)

    def factors(self):
        """(T, info, rows): the factors this rank's draw needs; ``rows`` tells whether T holds only the rank's row
        blocks ``[L, nnu, F]`` (pair-sharded path) or the full ``[L, F, F]`` stack."""
        ctx, sp = self.ctx, self.plan
        if self._sep is not None:          # (also under emulate_world: this IS the rank's whole cold part)
            return self._separable_factors()
        if self.emulate_world > 1:
            N = self.emulate_world
            if self.pair_sharded:
                slab = self._clarray_pairs(N - 1, N)
                C = ctx.clarray_pairs_finish(slab.new_zeros((N,) + tuple(slab.shape[1:])), self.F, sp.l_hi - sp.l_lo)
                del C
            else:
                self._clarray_local()
            ctx.factor_batched(self._emu_C)
            return self._emulated
        if not self.distributed:
            T, info = ctx.factor_batched(self._clarray_local())
            return T, info, False
        if self.pair_sharded:
            mine = exchange_pair_slabs(self._clarray_pairs(self.rank, self.world), sp)   # all-to-all #1
            T, info = ctx.factor_batched(ctx.clarray_pairs_finish(mine, self.F, sp.l_hi - sp.l_lo))
            Tr, ia = exchange_factor_rows(T, info, sp)                                     # all-to-all #2
            return Tr, ia, True
        T, info = ctx.factor_batched(self._clarray_local())
        Ta, ia = allgather_factors(T, info, sp)
        return Ta, ia, False

    def draw(self, seed, factors=None, out=None):
        """a_lm of this rank's channels for ``seed`` (K3, device Philox stream) into ``out`` (default: the shard's
        own a_lm buffer)."""
        ctx = self.ctx
        out = self.alm_buf if out is None else out
        T, info, rows = factors if factors is not None else self.factors()
        if rows and self.folded:
            return ctx.draw_alm_philox_chunks(T, info, seed, self.lmax, self.F, self.chunks, out=out)
        if rows:
            return ctx.draw_alm_philox_rows(T, info, seed, self.lmax, self.F, self.nu0, self.nnu, out=out)
        return ctx.draw_alm_philox(T, info, seed, self.lmax, self.F, nu0=self.nu0, nnu=self.nnu, out=out)

    def realise(self, seed, factors=None):
        """Maps ``[nnu, npix]`` (device tensor, reused between calls) of this rank's channels for ``seed``."""
        self.draw(seed, factors)
        return self.ctx.alm2map(self.alm_buf, self.nside, self.lmax, self.nnu, out=self.maps_buf)

    def realise_numpy(self, rng, factors=None, prepared=None):
        """As :meth:`realise` with the REFERENCE's normal stream: ``rng`` is what cora's callers pass to
        ``mkfullsky`` - a ``numpy.random.Generator`` (on PCG64 it is continued on the device, bit for bit, and left where
        cora would leave it) or ``None`` (numpy's legacy global MT19937 state, continued on the device as well) - see
        ``skysim.draw_numpy_stream``.  Every rank consumes the whole stream (identically seeded generators) range of
        multipoles by range - no 16 F nalm byte buffer - against its own rows of the factors."""
        _, finish = self.draw_numpy(rng, factors, defer=True, prepared=prepared)
        try:     # (the synthesis is enqueued behind the draw before the generator's state is waited for)
            return self.ctx.alm2map(self.alm_buf, self.nside, self.lmax, self.nnu, out=self.maps_buf)
        finally:
            finish()

    def prepare_numpy(self, rng):
        """Starts the device generator of ``rng``'s normal stream for one realisation NOW (``skysim.prepare_numpy_stream``):
        called ahead of :meth:`factors`, the generator's own passes run beside the C_l integration and the factorisation.
        Hand the result to :meth:`realise_numpy` / :meth:`draw_numpy` as ``prepared`` (None for generators that are
        consumed on the host)."""
        from .core import skysim

        return skysim.prepare_numpy_stream(self.ctx, rng, self.lmax, self.F)

    def draw_numpy(self, rng, factors=None, out=None, defer=False, prepared=None):
        """a_lm of this rank's channels from the reference's normal stream (see :meth:`realise_numpy`); ``defer``:
        ``(alm, finish)`` as ``skysim.draw_numpy_stream``."""
        from .core import skysim

        if factors is None:
            if prepared is None:
                prepared = self.prepare_numpy(rng)          # (the generator beside K1 / K2)
            try:
                factors = self.factors()
            except BaseException:
                if prepared is not None:
                    prepared.abort()
                raise
        T, info, rows = factors
        return skysim.draw_numpy_stream(self.ctx, T, info, rng, self.lmax, self.F, nu0=self.nu0, nnu=self.nnu,
                                        out=self.alm_buf if out is None else out, rows=rows, defer=defer,
                                        chunks=self.chunks if (rows and self.folded) else None, prepared=prepared)


class SkySum:
    """Sum of independent Gaussian components on one channel grid - BASELINE configs[3]: 21cm (``Corr21cm``) +
    galactic synchrotron (``FullSkySynchrotron``, cora/foreground/galaxy.py:20-27) + unresolved point sources
    (``_UnresolvedBackground``, cora/foreground/pointsource.py:541-546).  The reference would make one
    ``getsky()`` per component and add the maps.  Two forms, both behind the ``factors()`` / ``realise(seed)`` protocol
    of :class:`SkyShard`:

    ``mode="separate"``  every component is factored and drawn on its own (its own Philox key) and the a_lm are
        ADDED before one synthesis - the transform is linear, so K4 + K5 run once per realisation; the
        per-component a_lm exist (``draw`` leaves the sum in ``alm_buf``).
    ``mode="joint"`` (default)  the sum of independent zero-mean Gaussian fields with covariances C_l^(c) IS a
        Gaussian field with covariance sum_c C_l^(c): the component C_l stacks are added - each with the
        ``1e-14 max(diag)`` jitter mkfullsky gives it (cora/core/skysim.py:115-117), so the summed covariance is
        exactly that of the reference's summed maps - and ONE factorisation and ONE draw per realisation replace
        one per component (cfg 4: K3 77 -> 26 ms, K2 9 -> 7 ms).  Same distribution, not the same numbers as
        "separate" for a given seed; the device stream is statistical mode either way (seed parity with numpy needs
        the host stream and per-component draws: ``skysim.mkfullsky``).  The foreground blocks alone have
        cond ~ 1e19; their sum with the 21cm block is better conditioned than they are.

    components : sequence of ``(model, zromb)``
    """

    _SEED_STRIDE = 0x9E3779B97F4A7C15      # component k draws with seed + k * stride (mod 2^64): disjoint Philox keys

    def __init__(self, components, freq, nside, lmax, rank=0, world=1, ctx=None, distributed=None, emulate_world=0,
                 mode="joint", fold=False, emulate_rank=None):
        from . import _lib

        if mode not in ("joint", "separate"):
            raise ValueError("mode must be 'joint' or 'separate'")
        self.mode = mode
        self.ctx = ctx if ctx is not None else _lib.get_context()
        self.shards = []
        alm_buf = maps_buf = None
        for model, zromb in components:
            sh = SkyShard(model, freq, nside, lmax, zromb=zromb, rank=rank, world=world, ctx=self.ctx,
                          distributed=distributed, emulate_world=emulate_world, alm_buf=alm_buf, maps_buf=maps_buf,
                          fold=fold, emulate_rank=emulate_rank)
            alm_buf, maps_buf = sh.alm_buf, sh.maps_buf
            self.shards.append(sh)
        s0 = self.shards[0]
        self.alm_buf, self.maps_buf = alm_buf, maps_buf
        self.nu0, self.nnu, self.F, self.nside, self.lmax, self.npix = s0.nu0, s0.nnu, s0.F, s0.nside, s0.lmax, s0.npix
        self.channels, self.chunks, self.folded = s0.channels, s0.chunks, s0.folded
        self._tmp = self.ctx.empty(tuple(alm_buf.shape)) if (len(self.shards) > 1 and mode == "separate") else None

    def covariance_shard(self):
        """sum_c (C_l^(c) + 1e-14 max(diag C_l^(c)) I) for this rank's multipoles: the covariance of the summed
        field exactly as the reference's per-component mkfullsky calls imply it."""
        total = None
        for sh in self.shards:
            C = sh.covariance_shard()
            if sh.emulate_world > 1:            # (stand-in stack shared by the components: timing only)
                total = C
                continue
            jit = C.diagonal(dim1=1, dim2=2).amax(dim=1) * 1e-14
            if total is None:
                total = C
            else:
                total.add_(C)
                del C
            total.diagonal(dim1=1, dim2=2).add_(jit[:, None])
        return total

    def factors(self):
        """Per-component factors (the cold part), in component order; one entry in joint mode."""
        if self.mode == "separate":
            return [sh.factors() for sh in self.shards]
        s0 = self.shards[0]
        C = self.covariance_shard()
        T, info = self.ctx.factor_batched(C, jitter_rel=0.0)       # (the jitters are in C already)
        del C
        if s0.emulate_world > 1:
            return [s0._emulated]
        return [s0.share_factors(T, info)]

    def draw(self, seed, factors=None):
        factors = factors if factors is not None else self.factors()
        if self.mode == "joint":
            return self.shards[0].draw(int(seed), factors[0], out=self.alm_buf)
        for k, (sh, fac) in enumerate(zip(self.shards, factors)):
            sk = (int(seed) + k * self._SEED_STRIDE) & (2**64 - 1)
            if k == 0:
                sh.draw(sk, fac, out=self.alm_buf)
            else:
                sh.draw(sk, fac, out=self._tmp)
                self.alm_buf.add_(self._tmp)
        return self.alm_buf

    def realise(self, seed, factors=None):
        """Maps ``[nnu, npix]`` of the summed sky (device tensor, reused between calls)."""
        self.draw(seed, factors)
        return self.ctx.alm2map(self.alm_buf, self.nside, self.lmax, self.nnu, out=self.maps_buf)

    def realise_numpy(self, rng, factors=None):
        """As :meth:`realise` with the REFERENCE's normal stream (:meth:`SkyShard.realise_numpy`).  ``separate``: the
        components draw one after the other from the SAME generator, in component order - the numbers of the
        reference's one ``getsky()`` per component called in that order with that generator (or, ``rng=None``, with
        numpy's global state) - and the a_lm are added before the one synthesis.  ``joint``: one draw of the summed
        covariance from the generator (the same distribution, not those numbers)."""
        factors = factors if factors is not None else self.factors()
        finish = lambda: None   # noqa: E731
        if self.mode == "joint":
            _, finish = self.shards[0].draw_numpy(rng, factors[0], out=self.alm_buf, defer=True)
        else:
            for k, (sh, fac) in enumerate(zip(self.shards, factors)):
                last = k == len(self.shards) - 1       # (a component's generator state feeds the next one's draw: only the last waits late)
                if k == 0:
                    r = sh.draw_numpy(rng, fac, out=self.alm_buf, defer=last)
                else:
                    r = sh.draw_numpy(rng, fac, out=self._tmp, defer=last)
                    self.alm_buf.add_(self._tmp)
                if last:
                    finish = r[1]
        try:
            return self.ctx.alm2map(self.alm_buf, self.nside, self.lmax, self.nnu, out=self.maps_buf)
        finally:
            finish()


def getsky_shard(sky, seed, rank=0, world=1, lmax=None, fold=False):
    """``Sky3d.getsky()`` for one rank of ``world`` GPUs: (maps [nnu, npix] device tensor, nu0) for a cora_amd
    Gaussian model instance ``sky`` (frequencies, nside, oversample taken from it; lmax default 3 nside - 1 as
    maps.py:230).  The process group must be initialised when world > 1.  ``fold``: the folded channel assignment -
    the second return value is then the array of global channel indices of the maps."""
    lmax = 3 * sky.nside - 1 if lmax is None else lmax
    zromb = getattr(sky, "oversample", 3)
    import numpy as np

    shard = SkyShard(sky, sky.nu_pixels, sky.nside, lmax, zromb=zromb if zromb is not None else 3, rank=rank, world=world,
                     fold=fold)
    maps = shard.realise(int(seed))
    freq = np.asarray(sky.nu_pixels, dtype=np.float64)[shard.channels]
    mean = shard.ctx.to_device(np.asarray(sky.mean_nu(freq), dtype=np.float64) * np.ones(shard.nnu))
    return maps + mean[:, None], (shard.channels if shard.folded else shard.nu0)


def mkfullsky_sharded(corr_local, global_shape, nside, rng=None, alms=False, ctx=None):
    """``skysim.mkfullsky`` for an l-DISTRIBUTED correlation array (cora/core/skysim.py:97-103,108-134: the reference
    takes a caput ``MPIArray`` split over axis 0, draws its local multipoles, redistributes the a_lm over
    frequency and returns ``MPIArray.wrap(sky, axis=0)``).

    corr_local   : this rank's contiguous block of multipoles ``[n_local, F, F]`` (ndarray or device tensor); blocks
                   are in rank order and cover ``global_shape[0]`` multipoles (caput's split or any other)
    global_shape : ``(lmax + 1, F, F)``
    rng          : ``DeviceRNG`` (same seed on every rank), a numpy ``Generator`` (identically seeded on every rank:
                   every rank consumes the whole stream, so the result equals the single-process realisation of
                   that seed - the reference's own multi-rank output is not a function of the seed alone, SURVEY
                   App. B), or ``None`` (rank 0 draws a seed from numpy's legacy global state and broadcasts it)

    Returns ``(out, nu0)``: the rank's channel shard - device maps ``[nnu, npix]`` or, with ``alms``,
    ``[nnu, 1, L, L]`` complex - and its first channel (caput's axis-0 split of F).  Exchange: the factor ROW
    blocks go by all-to-all when F divides evenly, else the factor stack is all-gathered; the a_lm are never
    exchanged (every rank draws from the same global normal stream for its own channels).
    """
    import numpy as np
    import torch
    import torch.distributed as dist

    from . import _lib
    from .core import skysim
    from .util.nputil import DeviceRNG

    ctx = ctx if ctx is not None else _lib.get_context()
    world, rank = dist.get_world_size(), dist.get_rank()
    L, F, F2 = (int(v) for v in global_shape)
    if F2 != F:
        raise Exception("Correlation matrix is incorrect shape.")
    if not isinstance(corr_local, torch.Tensor):
        corr_local = ctx.to_device(np.asarray(corr_local, dtype=np.float64))
    n_local = int(corr_local.shape[0])
    counts = [None] * world
    dist.all_gather_object(counts, n_local)
    if sum(counts) != L:
        raise Exception("l blocks of the ranks do not add up to global_shape[0]")
    lc = max(max(counts), 1)
    sp = shard_plan(L, F, rank, world)
    nu0, nnu = sp.nu0, sp.nnu
    T_loc, i_loc = skysim.factor_device(corr_local) if n_local else (ctx.empty((0, F, F)), torch.empty(
        (0,), dtype=torch.int32, device=ctx.device))
    pad_i = torch.zeros((lc,), dtype=torch.int32, device=ctx.device)
    pad_i[:n_local].copy_(i_loc)
    i_all = torch.empty((world * lc,), dtype=torch.int32, device=ctx.device)
    rows = F % world == 0
    with exchange_stage("mkfullsky_sharded: factor exchange (l-distributed input -> channel shards)"):
        dist.all_gather_into_tensor(i_all, pad_i)
        if rows:
            recv = _all_to_all(_rows_pack(T_loc, lc, world), world)       # [src, lc, nnu, F]
        else:
            pad_T = torch.zeros((lc, F, F), dtype=torch.float64, device=ctx.device)
            pad_T[:n_local].copy_(T_loc)
            recv = torch.empty((world * lc, F, F), dtype=torch.float64, device=ctx.device)
            dist.all_gather_into_tensor(recv, pad_T)
    if rows:
        T = _rows_unpack(recv, counts)
    else:
        recv = recv.view(world, lc, F, F)
        T = torch.cat([recv[r, : counts[r]] for r in range(world)], dim=0).contiguous()
    info = torch.cat([i_all[r * lc : r * lc + counts[r]] for r in range(world)]).contiguous()
    lmax = L - 1
    if rng is None:
        box = [int(np.random.randint(0, 2**62)) if rank == 0 else 0]
        dist.broadcast_object_list(box, src=0)
        rng = DeviceRNG(box[0])
    if isinstance(rng, DeviceRNG):
        seed = rng.next_seed()
        alm = (ctx.draw_alm_philox_rows(T, info, seed, lmax, F, nu0, nnu) if rows
               else ctx.draw_alm_philox(T, info, seed, lmax, F, nu0=nu0, nnu=nnu))
    else:
        alm = skysim.draw_numpy_stream(ctx, T, info, rng, lmax, F, nu0=nu0, nnu=nnu, rows=rows)
    if alms:
        return ctx.alm_dev_to_square(alm, lmax, nnu), nu0
    return ctx.alm2map(alm, int(nside), lmax, nnu), nu0
