#!/usr/bin/env python3
"""Timing of the analysis path at the BASELINE cfg-3 geometry: one weighted quadrature
pass (K5^T ringana + K4^T legendre_adj) and the healpy-equivalent map2alm(use_weights=True, iter=2)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=1024)
    ap.add_argument("--lmax", type=int, default=2048)
    ap.add_argument("--nnu", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch

    from cora_amd import _lib
    from cora_amd.util import hputil

    ctx = _lib.get_context()
    npix = 12 * a.nside**2
    maps = ctx.empty((a.nnu, npix)).normal_()
    w = ctx.to_device(hputil.ring_weights(a.nside))
    ctx.map2alm(maps, a.nside, a.lmax, w)
    torch.cuda.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.time()
    for _ in range(a.reps):
        ctx.map2alm(maps, a.nside, a.lmax, w)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.reps
    ctx.profile_enable(False)
    st = {k: ctx.profile_get(k)[0] / a.reps for k in ("ringana", "legendre_adj")}
    nalm = (a.lmax + 1) * (a.lmax + 2) // 2
    flops = 8.0 * a.nside * nalm * a.nnu
    hputil.map2alm_device(maps, a.nside, a.lmax)      # warm-up: workspace growth, allocator
    torch.cuda.synchronize()
    t1 = time.time()
    hputil.map2alm_device(maps, a.nside, a.lmax)
    torch.cuda.synchronize()
    t_iter = time.time() - t1
    print(json.dumps({"nside": a.nside, "lmax": a.lmax, "channels": a.nnu, "quadrature_pass_ms": dt * 1e3,
                      "stages_ms": st, "legendre_adj_TFLOPs": flops / (st["legendre_adj"] * 1e-3) / 1e12,
                      "map2alm_weights_iter2_ms": t_iter * 1e3, "maps_per_s_iter2": a.nnu / t_iter}))


if __name__ == "__main__":
    main()
