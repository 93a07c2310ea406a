#!/usr/bin/env python3
"""K1 at cfg-3 size: stage time per call (HIP events of the library).  python tools/k1_probe.py [F lmax [VAR=value ...]] - every
further argument sets an environment switch of the library for one measurement (round 5: CORAHIP_K1_STAGGER, an experiment
that delayed the first generation of workgroups by their slot on the CU so that build and interpolation phases of co-resident
workgroups interleave: 5.31 ms with every delay, 5.31 without - the kernel was reverted, the probe kept)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cora_amd import _lib
from cora_amd.parallel import SkyShard
from cora_amd.signal import corr21cm
ctx = _lib.get_context()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
freq = 400.0 + (np.arange(F) + 0.5) * (400.0 / F)
sh = SkyShard(corr21cm.Corr21cm(), freq, 64, lmax, zromb=3, ctx=ctx)
ref = None
for tag in sys.argv[3:] or ["X=0"]:
    os.environ[tag.split("=")[0]] = tag.split("=")[1]
    for _ in range(3):
        C = sh._clarray_local()
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(10):
        C = sh._clarray_local()
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("clarray")
    same = None if ref is None else bool(torch.equal(C, ref))
    if ref is None: ref = C.clone()
    print("%s: clarray %.3f ms per call (%d calls) identical %s" % (tag, ms / n, n, same))
