#!/usr/bin/env python3
"""Do the strided passes of the flat-sky irfftn pay for the odd row pitch of the half spectrum?  (nh = n/2 + 1 = 513
complex = 8208 bytes: a tile's 64-byte segments straddle 128-byte lines on 3 of 8 rows.)  Times corahip_irfftn on
[1024, 1024, nh] spectra for nh = 513 (the cube) and nh = 512 / 520 (aligned pitches; last axis 1022 / 1038)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

ctx = _lib.get_context()
n = int(os.environ.get("N", "1024"))
for nh in (n // 2 + 1, n // 2, n // 2 + 8):
    spec = torch.empty((n, n, nh), dtype=torch.complex128, device=ctx.device)
    spec.real.normal_()
    spec.imag.normal_()
    for _ in range(2):
        ctx.irfftn(spec.clone())
    torch.cuda.synchronize()
    reps = 5
    specs = [spec.clone() for _ in range(reps)]
    ctx.profile_reset()
    ctx.profile_enable(True)
    for r in range(reps):
        ctx.irfftn(specs[r])
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    ps = {k: round(ctx.profile_get(k)[0] / reps, 3) for k in ("fft_c2c_strided", "fft_c2r", "flatfft")}
    gb = 16.0 * n * n * nh / 1e9
    print("nh = %d (pitch %d B): %s; strided passes %.2f TB/s" % (nh, 16 * nh, ps, 4 * gb / ps["fft_c2c_strided"]), flush=True)
    del spec, specs
    torch.cuda.empty_cache()
