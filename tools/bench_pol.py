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

# ---- spin-2 analysis: one (Q, U) -> (E, B) quadrature pass, composed of six scalar passes per field
nf_a = int(os.environ.get("NFIELDS_ANA", "8"))
qu = maps[: 2 * nf_a].contiguous()
ctx.map2alm_spin2(qu, nside, lmax)
torch.cuda.synchronize()
ctx.profile_reset()
ctx.profile_enable(True)
eb = ctx.map2alm_spin2(qu, nside, lmax)
torch.cuda.synchronize()
st = {k: ctx.profile_get(k)[0] for k in ("spin2_scale", "ringana", "legendre_adj", "spin2_combine")}
ctx.profile_enable(False)
tot = sum(st.values())
print(json.dumps({"nside": nside, "lmax": lmax, "analysis_fields_QU": nf_a, "stages_ms": st, "pass_ms": tot,
                  "QU_map_pairs_per_s_one_pass": nf_a / (tot * 1e-3),
                  "legendre_adj_TFLOPs_scalar_equiv": 8.0 * nside * nalm * 6 * nf_a / (st["legendre_adj"] * 1e-3) / 1e12}))
