#!/usr/bin/env python3
"""The C ABI driven with ctypes + numpy only - no torch in the process - exactly as INTEGRATION.md shows a
cora maintainer would bind it: ONE call, corahip_mkfullsky, takes C_l and the state of the caller's numpy Generator to the
HEALPix maps (numpy's normal stream is continued on the device and the generator state comes back advanced); then maps
-> a_lm back through the analysis entry point.
Prints one line `ABI_DEMO max_map_err <e1> max_alm_err <e2>`; run by tests/test_gpu_parity.py against the oracle.

    python tools/abi_ctypes_demo.py <golden.npz key> <nside> <seed> <out.npz>
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
assert "torch" not in sys.modules
_lib = ctypes.CDLL(os.path.join(ROOT, "cora_amd", "libcorahip.so"))
_lib.corahip_last_error.restype = ctypes.c_char_p
P, I, D, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t


def _chk(rc):
    if rc:
        raise RuntimeError(_lib.corahip_last_error().decode())


class Ctx:
    def __init__(self, dev=0):
        self.h = P()
        _chk(_lib.corahip_ctx_create(dev, ctypes.byref(self.h)))

    def dev(self, a):
        a = np.ascontiguousarray(a)
        p = P()
        _chk(_lib.corahip_malloc(self.h, SZ(a.nbytes), ctypes.byref(p)))
        _chk(_lib.corahip_memcpy_h2d(self.h, p, a.ctypes.data_as(P), SZ(a.nbytes)))
        return p

    def host(self, p, shape, dtype=np.float64):
        out = np.empty(shape, dtype)
        _chk(_lib.corahip_memcpy_d2h(self.h, out.ctypes.data_as(P), p, SZ(out.nbytes)))
        return out


def main():
    key, nside, seed, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    corr = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))[key]
    L, F, _ = corr.shape
    lmax = L - 1
    nalm = L * (L + 1) // 2
    npix = 12 * nside * nside
    G = (F + 3) // 4
    # the generator cora's caller passes (cora/signal/lss.py:449-450): its PCG64 state goes to the library as it is
    rng = np.random.default_rng(seed)
    st = rng.bit_generator.state["state"]
    M64 = 2**64 - 1

    class Rng(ctypes.Structure):
        _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stream", P), ("seed", ctypes.c_uint64),
                    ("state", ctypes.c_uint64 * 2), ("inc", ctypes.c_uint64 * 2), ("legacy", P)]

    def pcg64():
        r = Rng()
        r.kind = 2                                                  # CORAHIP_RNG_PCG64
        r.state[0], r.state[1] = st["state"] >> 64, st["state"] & M64
        r.inc[0], r.inc[1] = st["inc"] >> 64, st["inc"] & M64
        return r

    ctx = Ctx()
    dC = ctx.dev(corr)
    plan = P()
    _chk(_lib.corahip_sht_plan_create(ctx.h, I(nside), I(lmax), ctypes.byref(plan)))
    nb = SZ()
    _chk(_lib.corahip_mkfullsky_workspace_bytes(plan, I(F), I(0), I(F), I(2), I(0), ctypes.byref(nb)))
    ws = P()
    _chk(_lib.corahip_malloc(ctx.h, nb, ctypes.byref(ws)))
    dmaps = ctx.dev(np.empty(F * npix))
    r1 = pcg64()
    _chk(_lib.corahip_mkfullsky(ctx.h, plan, dC, I(F), ctypes.byref(r1), I(0), I(F), I(0), dmaps, ws, nb))   # skysim.py:72-136
    _chk(_lib.corahip_ctx_sync(ctx.h))
    maps = ctx.host(dmaps, (F, npix))
    # the same call with alms = 1 (skysim.py:123-125), from the same generator state
    dsq = ctx.dev(np.empty((F, 1, L, L), np.complex128))
    r2 = pcg64()
    _chk(_lib.corahip_mkfullsky(ctx.h, plan, dC, I(F), ctypes.byref(r2), I(0), I(F), I(1), dsq, ws, nb))
    alm = ctx.host(dsq, (F, 1, L, L), np.complex128)
    # the state that came back is where numpy leaves the generator after the reference's draws
    for l in range(L):
        rng.standard_normal((F, l + 1))
        rng.standard_normal((F, l + 1))
    after = int(rng.bit_generator.state["state"]["state"])
    assert (int(r1.state[0]) << 64 | int(r1.state[1])) == after == (int(r2.state[0]) << 64 | int(r2.state[1]))
    # analysis direction: one unweighted quadrature pass of the maps
    nb2 = SZ()
    _chk(_lib.corahip_map2alm_workspace_bytes(plan, I(F), ctypes.byref(nb2)))
    ws2 = P()
    _chk(_lib.corahip_malloc(ctx.h, nb2, ctypes.byref(ws2)))
    G8 = (F + 7) // 8 * 2
    drec = ctx.dev(np.empty(nalm * G8 * 8))
    _chk(_lib.corahip_map2alm(ctx.h, plan, dmaps, I(F), None, drec, ws2, nb2))
    dsq2 = ctx.dev(np.empty((F, 1, L, L), np.complex128))
    # the G8-group layout equals the G-group one when F is a multiple of 8 (it is in this demo)
    _chk(_lib.corahip_alm_dev_to_square(ctx.h, drec, I(lmax), I(F), dsq2))
    rec = ctx.host(dsq2, (F, 1, L, L), np.complex128)
    _chk(_lib.corahip_ctx_sync(ctx.h))
    np.savez(out, maps=maps, alm=alm, rec=rec)
    for p in (dC, dmaps, ws, dsq, ws2, drec, dsq2):
        _chk(_lib.corahip_free(ctx.h, p))
    _chk(_lib.corahip_sht_plan_destroy(ctx.h, plan))
    _chk(_lib.corahip_ctx_destroy(ctx.h))
    print("ABI_DEMO ok", maps.shape, "torch" in sys.modules)


if __name__ == "__main__":
    main()
