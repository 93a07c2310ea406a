#!/usr/bin/env python3
"""K3 (Philox) at cfg 3 through the real pipeline, stage time per call: CORAHIP_K3_WIDE=1 (one 256-column group per (l, m block),
two LDS stages) against CORAHIP_K3_WIDE=0 (two 128-column groups, three stages) - the switch is read once per process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cora_amd import _lib
from cora_amd.parallel import SkyShard
from cora_amd.signal import corr21cm
ctx = _lib.get_context()
F, nside, lmax = 256, 64, 2048
freq = 400.0 + (np.arange(F) + 0.5) * (400.0 / F)
sh = SkyShard(corr21cm.Corr21cm(), freq, nside, lmax, zromb=3, ctx=ctx)
fac = sh.factors()
sh.draw(1, fac); sh.draw(2, fac)
torch.cuda.synchronize()
ctx.profile_reset(); ctx.profile_enable(True)
for i in range(6):
    alm = sh.draw(10 + i, fac)
torch.cuda.synchronize()
print("CORAHIP_K3_WIDE=%s: draw %.3f ms, checksum %.6e" % (os.environ.get("CORAHIP_K3_WIDE"), ctx.profile_get("draw")[0] / 6, float(alm.abs().sum())))
