#!/usr/bin/env python3
"""Timing of the polarised (spin-2) synthesis at the cfg-3 geometry."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

nside, lmax, nfreq = 1024, 2048, int(os.environ.get("NFREQ", "32"))
nch = 2 * nfreq
ctx = _lib.get_context()
nalm = (lmax + 1) * (lmax + 2) // 2
alm = ctx.empty((nalm, nch // 4, 2, 4)).normal_()
maps = ctx.empty((nch, 12 * nside * nside))
ctx.alm2map_spin2(alm, nside, lmax, nch, out=maps)
torch.cuda.synchronize()
ctx.profile_reset()
ctx.profile_enable(True)
ctx.alm2map_spin2(alm, nside, lmax, nch, out=maps)
torch.cuda.synchronize()
t4, t5 = ctx.profile_get("legendre_pol")[0], ctx.profile_get("ringfft")[0]
# algorithmic flops: two operands (W, X) per (ring pair, l, m, column): 2 x the scalar 8 nside nalm per channel
flops = 2.0 * 8.0 * nside * nalm * nch
print(json.dumps({"nside": nside, "lmax": lmax, "fields_QU": nfreq, "legendre_pol_ms": t4, "ringfft_ms": t5,
                  "legendre_pol_TFLOPs": flops / (t4 * 1e-3) / 1e12, "QU_map_pairs_per_s": nfreq / ((t4 + t5) * 1e-3)}))
