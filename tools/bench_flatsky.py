#!/usr/bin/env python3
"""Timing of the flat-sky Gaussian field generator (flat-sky half): one
RandomField.getfield-equivalent realisation (Philox draw x kweight -> irfftn) per cube size, with the
numpy transform of the same spectrum timed on the host at the smallest size for reference.

Algorithmic HBM traffic per realisation of an N^3 cube (nh = N/2 + 1 spectral bins on the last axis):
draw reads kweight 8 N^2 nh and writes 16 N^2 nh; the two complex passes read + write 16 N^2 nh each; the
c2r pass reads 16 N^2 nh and writes 8 N^3.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

ctx = _lib.get_context()
sizes = [int(s) for s in os.environ.get("SIZES", "256,512,1024,768").split(",")]
out = []
for n in sizes:
    nh = n // 2 + 1
    kw = ctx.empty((n, n, nh)).uniform_()
    for _ in range(2):
        fld = ctx.irfftn(ctx.randomfield_draw(kw, 1))
    torch.cuda.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    reps = 5
    t0 = time.perf_counter()
    for r in range(reps):
        fld = ctx.irfftn(ctx.randomfield_draw(kw, 2 + r))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    t_draw = ctx.profile_get("flatdraw")[0] / reps
    t_fft = ctx.profile_get("flatfft")[0] / reps
    spec_b = 16.0 * n * n * nh
    traffic = (8.0 * n * n * nh + spec_b) + 2 * (2 * spec_b) + (spec_b + 8.0 * n**3)
    passes = {k: ctx.profile_get(k)[0] / reps for k in ("fft_c2c_strided", "fft_c2c_contig", "fft_c2r")}
    rec = {"cube": [n, n, n], "draw_ms": t_draw, "irfftn_ms": t_fft, "passes_ms": passes, "wall_ms": wall,
           "algorithmic_GB": traffic / 1e9, "achieved_GBps": traffic / ((t_draw + t_fft) * 1e-3) / 1e9,
           "fields_per_s": 1e3 / wall}
    ctx.profile_enable(False)
    # round 5: the fused form RandomField.getfield(seed) takes - the spectrum generated where the first pass loads it
    spec_ws = torch.empty((n, n, nh), dtype=torch.complex128, device=ctx.device)
    for _ in range(2):
        f2 = ctx.randomfield_irfftn(kw, 1, spec=spec_ws)
    torch.cuda.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for r in range(reps):
        f2 = ctx.randomfield_irfftn(kw, 2 + r, spec=spec_ws)
    torch.cuda.synchronize()
    wall2 = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile_enable(False)
    traffic2 = (8.0 * n * n * nh + spec_b) + (2 * spec_b) + (spec_b + 8.0 * n**3)     # k-weights in, one pass less over the spectrum
    rec["fused_draw"] = {"wall_ms": wall2, "flatfft_ms": ctx.profile_get("flatfft")[0] / reps,
                         "passes_ms": {k: ctx.profile_get(k)[0] / reps for k in ("fft_c2c_draw", "fft_c2c_strided", "fft_c2r")},
                         "algorithmic_GB": traffic2 / 1e9, "achieved_GBps": traffic2 / (wall2 * 1e-3) / 1e9,
                         "fields_per_s": 1e3 / wall2, "identical_to_two_step": bool(torch.equal(f2, fld))}
    del f2, spec_ws
    if n == sizes[0]:
        spec = ctx.randomfield_draw(kw, 9).cpu().numpy()
        t0 = time.perf_counter()
        ref = np.fft.irfftn(spec)
        rec["numpy_irfftn_ms"] = (time.perf_counter() - t0) * 1e3
        rec["max_abs_diff_vs_numpy"] = float(np.abs(ctx.irfftn(ctx.to_device(spec, dtype=np.complex128)).cpu().numpy()
                                                    - ref).max())
    del fld, kw
    torch.cuda.empty_cache()
    out.append(rec)
    print(json.dumps(rec), flush=True)

# ---- Corr21cm.getfield (redshift-space cube: draw -> irfftn -> rfftn x mu^2 -> irfftn -> slice factors -> ray trace)
if os.environ.get("CUBE", "1") == "1":
    from cora_amd.signal import corr21cm

    for nu_num, npx in ((128, 128), (256, 256), (256, 512)):
        cr = corr21cm.Corr21cm()
        cr.nu_num, cr.x_num, cr.y_num, cr.x_width, cr.y_width = nu_num, npx, npx, 5.0, 5.0
        cr.nu_lower, cr.nu_upper = 600.0, 700.0
        z1, z2 = cr._band_redshifts()
        t0 = time.perf_counter()
        cr.realisation(z1, z2, 5.0, 5.0, nu_num, npx, npx, zspace=False, seed=1, device=True)   # builds k-weights (host, once)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        box = list(cr._cube_cache.keys())[0][1]
        ctx.profile_reset()
        ctx.profile_enable(True)
        reps = 5
        t0 = time.perf_counter()
        for r in range(reps):
            out_cube = cr.realisation(z1, z2, 5.0, 5.0, nu_num, npx, npx, zspace=False, seed=2 + r, device=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        st = {k: round(ctx.profile_get(k)[0] / reps, 3) for k in ("flatdraw", "flatfft", "spec_mul", "cube_affine", "raytrace")}
        st["passes"] = {k: round(ctx.profile_get(k)[0] / reps, 3) for k in ("fft_c2c_draw", "fft_c2c_strided", "fft_c2r", "fft_r2c")}
        ctx.profile_enable(False)
        print(json.dumps({"corr21cm_getfield": [nu_num, npx, npx], "comoving_box": box, "first_call_s": first,
                          "wall_ms": wall, "device_ms": st, "cubes_per_s": 1e3 / wall}), flush=True)
