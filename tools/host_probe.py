#!/usr/bin/env python3
"""Diagnostics on the GPU box's host side: (1) threads / timings of the oracle's C legs, (2) D2H bandwidth into pinned
memory with one and with several copy streams."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import healpix, sht  # noqa: E402

print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "oracle threads", sht.num_threads(), "affinity",
      len(os.sched_getaffinity(0)))
nside, lmax = 1024, 2048
n = (lmax + 1) * (lmax + 2) // 2
rng = np.random.default_rng(0)
a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
ri = healpix.ring_info(nside)
npair = 2 * nside
for rep in range(2):
    t = time.time()
    fn, fs = sht._legendre_c(lmax, ri["z"][:npair], ri["sth"][:npair], a)
    t1 = time.time() - t
    t = time.time()
    sht.synth_from_fm_c(fn, fs, nside)
    t2 = time.time() - t
    print("legendre %.3f s  rings %.3f s" % (t1, t2))

dev = torch.device("cuda", 0)
nbytes = 4 << 30
src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_()
dst = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
for nstream in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t = time.time()
        step = nbytes // nstream
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, nbytes / (time.time() - t) / 1e9)
    print("D2H pinned, %d stream(s): %.1f GB/s" % (nstream, best))
