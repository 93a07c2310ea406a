import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from cora_amd import _lib
from cora_amd.signal import corrfunc
from cora_amd.util import cubicspline as cs
ctx = _lib.get_context()
r = np.concatenate([[0.0], np.logspace(-1, 4.0, 700)])
xi = np.exp(-r / 60.0) * np.cos(r / 35.0) / (1.0 + (r / 15.0) ** 2)
sps = {"sinh": cs.SinhInterpolater(np.stack([r, xi], axis=1), 1.0, 1e-4), "plain": cs.Interpolater(r, xi),
       "log": cs.LogInterpolater(np.stack([r[1:], np.abs(xi[1:]) + 1e-9], axis=1))}
lmax, F = 767, 128
xa = 1500.0 + np.arange(F) * (2500.0 / F)
for name, sp in sps.items():
    corrfunc.corr_to_clarray(sp, 64, xa[:8], xromb=1)
    torch.cuda.synchronize(); ctx.profile_reset(); ctx.profile_enable(True)
    cl = corrfunc.corr_to_clarray(sp, lmax, xa, xromb=3, q=2)
    torch.cuda.synchronize(); ctx.profile_enable(False)
    t = ctx.profile_get("xi_average")[0]
    print(name, round(t, 2), "ms", "%.3g evals/s" % (2 * lmax * F * (F + 1) / 2 * 81 / (t * 1e-3)), float(np.abs(cl).sum()))
