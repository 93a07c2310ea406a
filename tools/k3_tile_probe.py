#!/usr/bin/env python3
"""Magnitude structure of the Cholesky factors T_l of the 21cm covariance: what fraction of the
16 x 16 tiles of the lower triangle that K3 stages and multiplies is negligible, max|T_ij| < 2^cut max diag(T_l)?

    python tools/k3_tile_probe.py [F lmax]        default: cfg 3 (256, 2048); cfg 5: 1024 4096
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cora_amd import _lib  # noqa: E402
from cora_amd.core import skysim  # noqa: E402
from cora_amd.signal import corr21cm  # noqa: E402

F, lmax = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 2048)
ctx = _lib.get_context()
freq = 400.0 + (np.arange(F) + 0.5) * (400.0 / F)
C = skysim.clarray_device(corr21cm.Corr21cm().angular_powerspectrum, lmax, freq, zromb=3)
T, info = ctx.factor_batched(C, jitter_rel=1e-14, eig_thresh=1e-16)
del C
L = lmax + 1
nt = F // 16
lo = torch.tril(torch.ones((nt, nt), dtype=torch.bool, device=T.device))
cuts = (-20, -30, -40, -53, -60)
rows = []
for l0 in range(0, L, 64):
    Tl = T[l0:l0 + 64].abs()
    n = Tl.shape[0]
    tmax = Tl.view(n, nt, 16, nt, 16).amax(dim=(2, 4))                     # [n, nt, nt] max of every tile
    dmax = torch.diagonal(Tl, dim1=1, dim2=2).amax(dim=1)                   # [n]
    rel = tmax / dmax[:, None, None]
    for k in range(n):
        r = rel[k][lo]
        rows.append([l0 + k] + [float((r < 2.0**c).float().mean()) for c in cuts])
rows = np.array(rows)
w = rows[:, 0] + 1.0                                                         # a multipole has l + 1 values of m: its weight in K3
print("F %d lmax %d: fraction of lower-triangle 16x16 tiles with max|T_ij| < 2^cut max diag(T_l)" % (F, lmax))
print("%12s" % "l range" + "".join("%10s" % ("2^%d" % c) for c in cuts))
for a, b in ((0, 100), (100, 500), (500, 1000), (1000, 2048), (2048, 4097)):
    m = (rows[:, 0] >= a) & (rows[:, 0] < b)
    if m.any():
        print("%12s" % ("%d-%d" % (a, b - 1)) + "".join("%10.3f" % np.average(rows[m, 1 + i], weights=w[m]) for i in range(len(cuts))))
print("%12s" % "all (m-wtd)" + "".join("%10.3f" % np.average(rows[:, 1 + i], weights=w) for i in range(len(cuts))))
# how fast does a row fall off?  |T[nu, nu - d]| / T[nu, nu] at the last channel, a few multipoles
for l in (100, 1000, lmax):
    r = T[l, F - 1].abs() / T[l, F - 1, F - 1].abs()
    d = [1, 2, 4, 8, 16, 32, 64, 128]
    print("l = %4d: |T[F-1, F-1-d]| / T[F-1, F-1] at d = %s: %s" % (l, d, ["%.1e" % float(r[F - 1 - k]) for k in d if k < F]))
