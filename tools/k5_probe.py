#!/usr/bin/env python3
"""Diagnostics: per-class K5 (ring FFT) times at the cfg-3 geometry; run with CORAHIP_K5_TIMES=1 and,
for the ablation builds, CORAHIP_LIB=cora_amd/libcorahip_k5abN.so."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

nside, lmax, nnu = int(os.environ.get("NSIDE", "1024")), int(os.environ.get("LMAX", "2048")), int(os.environ.get("NNU", "256"))
ctx = _lib.get_context()
nalm = (lmax + 1) * (lmax + 2) // 2
alm = ctx.empty((nalm, nnu // 4, 2, 4)).normal_()
maps = ctx.empty((nnu, 12 * nside * nside))
ctx.alm2map(alm, nside, lmax, nnu, out=maps)
torch.cuda.synchronize()
print("---- timed pass", file=sys.stderr)
ctx.profile_reset()
ctx.profile_enable(True)
ctx.alm2map(alm, nside, lmax, nnu, out=maps)
torch.cuda.synchronize()
print("legendre %.2f ringfft %.2f" % (ctx.profile_get("legendre")[0], ctx.profile_get("ringfft")[0]), file=sys.stderr)
