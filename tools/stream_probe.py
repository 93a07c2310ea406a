#!/usr/bin/env python3
"""The numpy-stream draw at cfg-3 size: whole-stream buffer (normals_* + draw_alm, rounds 1-4) against the l-range ring
(corahip_draw_alm_numpy) for several ring sizes.  python tools/stream_probe.py [F lmax]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cora_amd import _lib  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ctx = _lib.get_context()
L = lmax + 1
n = 2 * F * (L * (L + 1) // 2)
T = torch.tril(ctx.empty((L, F, F)).normal_()) + 3.0 * torch.eye(F, device=ctx.device, dtype=torch.float64)
info = torch.zeros((L,), dtype=torch.int32, device=ctx.device)
nalm = L * (L + 1) // 2
alm = ctx.empty((nalm, F // 4, 2, 4))
rng = np.random.default_rng(5)
st = rng.bit_generator.state["state"]
np.random.seed(3)
lst = np.random.get_state(legacy=False)


def timed(fn, rep=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(rep):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / rep * 1e3


def old_pcg():
    g, _ = ctx.normals_pcg64(st["state"], st["inc"], n)
    ctx.draw_alm(T, info, g, lmax, F, out=alm)


def old_leg():
    g, _ = ctx.normals_legacy(lst, n)
    ctx.draw_alm(T, info, g, lmax, F, out=alm)


print("F %d lmax %d: %.2f GB of normals" % (F, lmax, 8 * n / 1e9))
if 8 * n < 40e9:
    print("whole-stream buffer: pcg64 %.2f ms   legacy %.2f ms" % (timed(old_pcg), timed(old_leg)))
    ref = alm.clone()
    old_pcg()
    ref_p = alm.clone()
for serial in (False, True):
    if serial:
        os.environ["CORAHIP_GEN_SERIAL"] = "1"
    for mb in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        tp = timed(lambda: ctx.draw_alm_numpy(T, info, ("pcg64", st["state"], st["inc"]), lmax, F, out=alm, ring_bytes=mb << 20))
        okp = torch.equal(alm, ref_p) if 8 * n < 40e9 else None
        tl = timed(lambda: ctx.draw_alm_numpy(T, info, ("legacy", lst), lmax, F, out=alm, ring_bytes=mb << 20))
        okl = torch.equal(alm, ref) if 8 * n < 40e9 else None
        print("ring %5d MB %s: pcg64 %.2f ms (%s)   legacy %.2f ms (%s)" % (mb, "serial " if serial else "overlap", tp, okp, tl, okl))
