#!/usr/bin/env python3
"""mkfullsky on an l-DISTRIBUTED correlation array (cora/core/skysim.py:97-110,125-134: what the reference does
with a caput MPIArray over MPI) through the C ABI alone: ctypes + numpy, no torch, no torch.distributed.  WORLD
processes (all on GPU 0 here; one per GPU in production) each factor their own block of multipoles, pack the factor
row blocks, exchange them - through FILES in a scratch directory, standing in for MPI_Alltoall / Allgather - unpack,
draw from the common Philox stream for their own channels and synthesise them.

    python tools/abi_shard_demo.py <golden key> <nside> <seed> <world> <rank> <scratch dir>     one rank
    python tools/abi_shard_demo.py <golden key> <nside> <seed> <world> launch <scratch dir>     starts the ranks + a
                                                                                                 single-process run

Every rank writes rank<r>.npz (its maps [nnu, npix], a_lm [nnu, 1, L, L], nu0); the launcher compares their
concatenation with the single-process result of the same seed (world = 1 through the same entry points) and
prints `ABI_SHARD ok <max rel map err> <max rel alm err>`; tests/test_gpu_parity.py runs it and checks the
single-process result against the oracle as well.
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
assert "torch" not in sys.modules
_lib = ctypes.CDLL(os.path.join(ROOT, "cora_amd", "libcorahip.so"))
_lib.corahip_last_error.restype = ctypes.c_char_p
P, I, D, SZ, U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t, ctypes.c_uint64


class Shard(ctypes.Structure):       # corahip_shard of include/corahip.h
    _fields_ = [(n, ctypes.c_int32) for n in ("l_lo", "l_hi", "l_shard", "l_pad", "nu0", "nnu", "rows_exchange", "L")]


def _chk(rc):
    if rc:
        raise RuntimeError(_lib.corahip_last_error().decode())


class Ctx:
    def __init__(self, dev=0):
        self.h = P()
        _chk(_lib.corahip_ctx_create(dev, ctypes.byref(self.h)))

    def alloc(self, nbytes):
        p = P()
        _chk(_lib.corahip_malloc(self.h, SZ(max(int(nbytes), 8)), ctypes.byref(p)))
        return p

    def dev(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        if a.nbytes:
            _chk(_lib.corahip_memcpy_h2d(self.h, p, a.ctypes.data_as(P), SZ(a.nbytes)))
        return p

    def host(self, p, shape, dtype=np.float64):
        out = np.empty(shape, dtype)
        if out.nbytes:
            _chk(_lib.corahip_memcpy_d2h(self.h, out.ctypes.data_as(P), p, SZ(out.nbytes)))
        return out


def _put(path, arr):
    """Atomic file drop (the 'send' of the file-based exchange)."""
    tmp = path + ".tmp.npy"
    np.save(tmp, arr)
    os.replace(tmp, path)


def _get(path, timeout=90.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError("exchange timed out waiting for " + path)
        time.sleep(0.01)
    return np.load(path)


def run_rank(key, nside, seed, world, rank, scratch):
    corr = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))[key]
    L, F, _ = corr.shape
    lmax = L - 1
    sp = Shard()
    _chk(_lib.corahip_shard_plan(L, F, rank, world, ctypes.byref(sp)))
    assert sp.rows_exchange == 1, "the demo uses the row-block all-to-all (F % world == 0)"
    # the l split of the INPUT is caput's, not the library's (the first L % world ranks hold one more multipole:
    # pack / unpack take any contiguous split): uneven blocks exercise the counts of the unpack
    base, extra = divmod(L, world)
    counts = [base + (1 if r < extra else 0) for r in range(world)]
    lo = rank * base + min(rank, extra)
    n_local, l_stride = counts[rank], max(counts)
    nnu, nu0 = sp.nnu, sp.nu0
    ctx = Ctx()
    # ---- stage A: factor this rank's multipoles (skysim.py:114-119)
    dC = ctx.dev(corr[lo:lo + n_local])
    dT, dinfo = ctx.alloc(8 * n_local * F * F), ctx.alloc(4 * n_local)
    _chk(_lib.corahip_factor_batched(ctx.h, dC, I(n_local), I(F), D(1e-14), D(1e-16), dT, dinfo))
    # ---- the exchange: pack -> all-to-all of the slabs (files) + all-gather of info -> unpack
    dsend = ctx.alloc(8 * world * l_stride * nnu * F)
    _chk(_lib.corahip_factor_rows_pack(ctx.h, dT, I(n_local), I(l_stride), I(F), I(world), dsend))
    _chk(_lib.corahip_ctx_sync(ctx.h))
    send = ctx.host(dsend, (world, l_stride, nnu, F))
    info = ctx.host(dinfo, (n_local,), np.int32)
    for q in range(world):
        _put(os.path.join(scratch, "slab_from%d_to%d.npy" % (rank, q)), send[q])
    _put(os.path.join(scratch, "info_from%d.npy" % rank), info)
    recv = np.stack([_get(os.path.join(scratch, "slab_from%d_to%d.npy" % (r, rank))) for r in range(world)])
    info_all = np.concatenate([_get(os.path.join(scratch, "info_from%d.npy" % r)) for r in range(world)]).astype(np.int32)
    drecv = ctx.dev(recv)
    dTrows = ctx.alloc(8 * L * nnu * F)
    cnt = (ctypes.c_int32 * world)(*counts)
    _chk(_lib.corahip_factor_rows_unpack(ctx.h, drecv, cnt, I(world), I(l_stride), I(nnu), I(F), dTrows))
    dinfo_all = ctx.dev(info_all)
    # ---- stage B: draw (common counter-based stream) + synthesis of this rank's channels (skysim.py:120-121,130)
    nalm = L * (L + 1) // 2
    G = (nnu + 3) // 4
    npix = 12 * nside * nside
    dalm, dmaps = ctx.alloc(8 * nalm * G * 8), ctx.alloc(8 * nnu * npix)
    _chk(_lib.corahip_draw_alm_philox_rows(ctx.h, dTrows, dinfo_all, U64(seed), I(lmax), I(F), I(nu0), I(nnu), dalm))
    plan = P()
    _chk(_lib.corahip_sht_plan_create(ctx.h, I(nside), I(lmax), ctypes.byref(plan)))
    nb = SZ()
    _chk(_lib.corahip_alm2map_workspace_bytes(plan, I(nnu), ctypes.byref(nb)))
    ws = ctx.alloc(nb.value)
    _chk(_lib.corahip_alm2map(ctx.h, plan, dalm, I(nnu), dmaps, ws, nb))
    dsq = ctx.alloc(16 * nnu * L * L)
    _chk(_lib.corahip_alm_dev_to_square(ctx.h, dalm, I(lmax), I(nnu), dsq))
    _chk(_lib.corahip_ctx_sync(ctx.h))
    np.savez(os.path.join(scratch, "rank%d_of%d.npz" % (rank, world)), maps=ctx.host(dmaps, (nnu, npix)),
             alm=ctx.host(dsq, (nnu, 1, L, L), np.complex128), nu0=nu0)
    for p in (dC, dT, dinfo, dsend, drecv, dTrows, dinfo_all, dalm, dmaps, ws, dsq):
        _chk(_lib.corahip_free(ctx.h, p))
    _chk(_lib.corahip_sht_plan_destroy(ctx.h, plan))
    _chk(_lib.corahip_ctx_destroy(ctx.h))


def launch(key, nside, seed, world, scratch):
    def start(w, r, d):
        os.makedirs(d, exist_ok=True)
        err = open(os.path.join(d, "rank%d.err" % r), "w")
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), key, str(nside), str(seed), str(w), str(r), d],
                                stderr=err), err

    def run_all(w, d):
        """Start the w ranks, wait for all; the first failure ends the others (they would only time out on its files)."""
        procs = [start(w, r, d) for r in range(w)]
        t0 = time.time()
        bad = None
        while bad is None and any(p.poll() is None for p, _ in procs):
            bad = next((r for r, (p, _) in enumerate(procs) if p.poll() not in (None, 0)), None)
            if time.time() - t0 > 300:
                bad = -1
            time.sleep(0.05)
        bad = bad if bad is not None else next((r for r, (p, _) in enumerate(procs) if p.returncode != 0), None)
        for p, e in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
            e.close()
        if bad is not None:
            for r in range(w):
                sys.stderr.write("---- rank %d of %d ----\n%s\n" % (r, w, open(os.path.join(d, "rank%d.err" % r)).read()[-2000:]))
            raise SystemExit("ABI_SHARD failed: rank %s of %d" % (bad, w))

    d1, dw = os.path.join(scratch, "w1"), os.path.join(scratch, "w%d" % world)
    run_all(1, d1)          # the single-process realisation first, then the ranks (never more than `world` on the GPU)
    run_all(world, dw)
    one = np.load(os.path.join(d1, "rank0_of1.npz"))
    parts = sorted((np.load(os.path.join(dw, "rank%d_of%d.npz" % (r, world))) for r in range(world)), key=lambda z: int(z["nu0"]))
    maps = np.concatenate([z["maps"] for z in parts])
    alm = np.concatenate([z["alm"] for z in parts])
    em = np.abs(maps - one["maps"]).max() / np.abs(one["maps"]).max()
    ea = np.abs(alm - one["alm"]).max() / np.abs(one["alm"]).max()
    np.savez(os.path.join(scratch, "single.npz"), maps=one["maps"], alm=one["alm"])
    print("ABI_SHARD ok %.3e %.3e" % (em, ea), "torch" in sys.modules)


if __name__ == "__main__":
    key, nside, seed, world, what, scratch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    if what == "launch":
        launch(key, nside, seed, world, scratch)
    else:
        run_rank(key, nside, seed, world, int(what), scratch)
