"""GPU probe of corahip_normals_pcg64: the device stream against numpy's own Generator, and its timing.

usage: python tools/npnormal_probe.py [n ...]      (default: a ladder of sizes up to 2e8)
CORAHIP_ZIG_ONEPASS=1: the round-6 single-pass form (stage zig_chain) instead of the two-pass default; ZIG_TIME_ONLY=1: timing only.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from cora_amd import _lib


def check(ctx, seed, n, skip=0):
    rng = np.random.default_rng(seed)
    if skip:
        rng.standard_normal(skip)
    st = rng.bit_generator.state["state"]
    g, nraw = ctx.normals_pcg64(st["state"], st["inc"], n)
    torch.cuda.synchronize()
    dev = g.cpu().numpy()
    ref = rng.standard_normal(n)
    after = rng.bit_generator.state["state"]["state"]
    same = dev.view(np.uint64) == ref.view(np.uint64)
    nbad = int((~same).sum())
    ulp = 0
    if nbad:
        bad = np.flatnonzero(~same)
        ulp = int(np.abs(dev.view(np.int64)[bad] - ref.view(np.int64)[bad]).max())
        print("   first mismatches:", bad[:5], dev[bad[:5]], ref[bad[:5]])
    ok_state = _lib.pcg64_advance(st["state"], st["inc"], nraw) == after
    print("seed %d n %d skip %d: mismatches %d (max %d ulp), tails %d, n_raw/n %.5f, state after %s"
          % (seed, n, skip, nbad, ulp, int((np.abs(ref) > 3.6541528853610088).sum()), nraw / max(n, 1),
             "OK" if ok_state else "WRONG"))
    return nbad == 0 and ok_state


def main():
    ctx = _lib.get_context()
    if os.environ.get("ZIG_TIME_ONLY"):
            sizes = []
    else:
        sizes = [int(float(a)) for a in sys.argv[1:]] or [1, 2, 17, 4000, 4096, 4097, 100000, 10**7, 2 * 10**8]
    ok = True
    for i, n in enumerate(sizes):
        ok &= check(ctx, 100 + i, n, skip=(i % 3) * 1001)
    # timing at the cfg-3 size (2 F nalm = 1.075e9 normals, 8.6 GB)
    n = 2 * 256 * 2100225
    rng = np.random.default_rng(3)
    st = rng.bit_generator.state["state"]
    g = ctx.empty((n,))
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.normals_pcg64(st["state"], st["inc"], n, out=g)
        torch.cuda.synchronize()
        print("cfg-3 stream (%.3e normals): %.2f ms" % (n, 1e3 * (time.perf_counter() - t0)))
    ctx.profile_enable(True)
    ctx.profile_reset()
    reps = 5
    for _ in range(reps):
        ctx.normals_pcg64(st["state"], st["inc"], n, out=g)
    print("  per kernel (ms): " + ", ".join("%s %.3f" % (k, ctx.profile_get(k)[0] / reps)
                                          for k in ("zig_seek", "zig_count", "zig_scan", "zig_emit", "zig_chain", "normals_pcg64")))
    ctx.profile_enable(False)
    print("ALL OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
