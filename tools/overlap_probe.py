#!/usr/bin/env python3
"""Two-stream overlap experiment: the cold part + draw of realisation i + 1 (K1, K2, K3) on one
stream beside the synthesis (K4, K5) of realisation i on another, against the same work on one stream."""
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
alm = [shard.alm_buf, torch.empty_like(shard.alm_buf)]
N = 6


def sequential():
    for i in range(N):
        fac = shard.factors()
        shard.draw(100 + i, fac, out=alm[0])
        ctx.alm2map(alm[0], nside, lmax, F, out=shard.maps_buf)


def pipelined(sA, sB):
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    free = [torch.cuda.Event(), torch.cuda.Event()]
    for e in free:
        e.record(sB)
    for i in range(N + 1):
        if i < N:
            with torch.cuda.stream(sA):
                ctx.use_current_stream()
                sA.wait_event(free[i % 2])
                fac = shard.factors()
                shard.draw(100 + i, fac, out=alm[i % 2])
                ready[i % 2].record(sA)
        if i >= 1:
            j = i - 1
            with torch.cuda.stream(sB):
                ctx.use_current_stream()
                sB.wait_event(ready[j % 2])
                ctx.alm2map(alm[j % 2], nside, lmax, F, out=shard.maps_buf)
                free[j % 2].record(sB)
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
print("one stream: %.2f ms per realisation; K1-K3 of the next realisation on a second stream: %.2f ms" % (t_seq, t_pipe))
