#!/usr/bin/env python3
"""Overlap experiment: K1 of realisation i + 1 (44 KB of LDS, 4 waves per workgroup) on one stream
beside K2 + K3 of realisation i (K3: 106 KB, 8 waves - a K1 workgroup fits beside it on a CU) on another, against
the same kernels one after the other.  K4 / K5 are left out: nothing fits beside them."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402
from cora_amd.parallel import SkyShard  # noqa: E402
from cora_amd.signal import corr21cm  # noqa: E402

ctx = _lib.get_context()
F, nside, lmax = 256, 1024, 2048
freq = 400.0 + (np.arange(F) + 0.5) * (400.0 / F)
shard = SkyShard(corr21cm.Corr21cm(), freq, nside, lmax, zromb=3, ctx=ctx)
N = 8


def sequential():
    for i in range(N):
        C = shard._clarray_local()
        T, info = ctx.factor_batched(C)
        shard.draw(100 + i, (T, info, False))


def pipelined(sA, sB):
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    Cs = [None, None]
    with torch.cuda.stream(sA):
        ctx.use_current_stream()
        Cs[0] = shard._clarray_local()
        ready[0].record(sA)
    for i in range(N):
        if i + 1 < N:
            with torch.cuda.stream(sA):          # K1 of the next realisation: launched BEFORE K2 (whose wrapper syncs its stream)
                ctx.use_current_stream()
                Cs[(i + 1) % 2] = shard._clarray_local()
                ready[(i + 1) % 2].record(sA)
        with torch.cuda.stream(sB):
            ctx.use_current_stream()
            sB.wait_event(ready[i % 2])
            T, info = ctx.factor_batched(Cs[i % 2])
            shard.draw(100 + i, (T, info, False))
    ctx.use_current_stream()


sequential()
torch.cuda.synchronize()
t = time.time()
sequential()
torch.cuda.synchronize()
t_seq = (time.time() - t) / N * 1e3
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
pipelined(sA, sB)
torch.cuda.synchronize()
t = time.time()
pipelined(sA, sB)
torch.cuda.synchronize()
t_pipe = (time.time() - t) / N * 1e3
print("K1 + K2 + K3 one after the other: %.2f ms per realisation; K1 of the next realisation beside K2 + K3: %.2f ms" % (t_seq, t_pipe))
