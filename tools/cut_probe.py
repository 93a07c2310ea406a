#!/usr/bin/env python3
"""Diagnostics: K4 time, executed MFMAs and the change of the maps against the plan's cut exponent (terms of the Legendre
sums below 2^cut are dropped; default -80) at the cfg-3 geometry, random a_lm of unit variance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

ctx = _lib.get_context()
nside, lmax, F = 1024, 2048, 256
nalm = (lmax + 1) * (lmax + 2) // 2
g = torch.Generator(device=ctx.device)
g.manual_seed(5)
alm = torch.randn((nalm, F // 4, 2, 4), generator=g, device=ctx.device, dtype=torch.float64)
ref = None
for cut in (-900, -120, -80, -70, -60, -50, -40):
    ctx.sht_cut_exp = cut
    maps = ctx.alm2map(alm, nside, lmax, F)
    torch.cuda.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(3):
        maps = ctx.alm2map(alm, nside, lmax, F)
    torch.cuda.synchronize()
    ms = ctx.profile_get("legendre")[0] / 3
    ms5 = ctx.profile_get("ringfft")[0] / 3
    keep = maps[:2].clone()
    if ref is None:
        ref = keep
    err = (keep - ref).abs().max().item() / ref.std().item()
    print("cut 2^%d: legendre %.2f ms, ringfft %.2f ms, MFMAs %.4g, max |map - map(2^-900)| = %.2e of the rms"
          % (cut, ms, ms5, ctx.k4_mfma_count(nside, lmax, F, cut_exp=cut), err))
    del maps
