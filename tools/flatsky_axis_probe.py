import os, sys, torch
sys.path.insert(0, os.getcwd())
from cora_amd import _lib
ctx = _lib.get_context()
x = torch.zeros((1024, 1024, 513), dtype=torch.complex128, device=ctx.device)
for axis in (0, 1, 2):
    ctx.fft_c2c(x, axis); torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3): ctx.fft_c2c(x, axis)
    torch.cuda.synchronize()
    print(axis, {k: round(ctx.profile_get(k)[0]/3, 2) for k in ("fft_c2c_strided", "fft_c2c_contig")})
    ctx.profile_enable(False)
