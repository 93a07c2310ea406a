#!/usr/bin/env python3
"""Diagnostics: time of corahip_factor_batched for NL matrices of F x F (env NL, F); CORAHIP_K2_VALU=1 forces the
right-looking VALU kernel, CORAHIP_LIB=cora_amd/libcorahip_k2abN.so the ablation builds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

nl, F = int(os.environ.get("NL", "257")), int(os.environ.get("F", "256"))
ctx = _lib.get_context()
A = ctx.empty((nl, F, F + 8)).normal_()
C = A @ A.transpose(1, 2) + 0.1 * torch.eye(F, device=ctx.device, dtype=torch.float64)
del A
for _ in range(2):
    ctx.factor_batched(C)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ctx.profile_reset()
ctx.profile_enable(True)
n = 5
for _ in range(n):
    ctx.factor_batched(C)
torch.cuda.synchronize()
print("NL %d F %d: factor %.3f ms per call" % (nl, F, ctx.profile_get("factor")[0] / n))
