#!/usr/bin/env python3
"""One irfftn of a 1024^3 cube (three line-FFT launches) for rocprofv3 counter passes (tools/pmc_flatsky.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402

n = int(os.environ.get("N", "1024"))
ctx = _lib.get_context()
spec = torch.zeros((n, n, n // 2 + 1), dtype=torch.complex128, device=ctx.device)
spec.real.normal_()
out = ctx.irfftn(spec)
torch.cuda.synchronize()
print(float(out[0, 0, 0]))
