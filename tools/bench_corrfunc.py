#!/usr/bin/env python3
"""Timing of the xi(r) -> C_l(chi, chi') path: spline-table bin average + Legendre MFMA projection."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.special as ss
    import torch

    from cora_amd import _lib
    from cora_amd.signal import corrfunc
    from cora_amd.util import cubicspline as cs

    ctx = _lib.get_context()
    r = np.concatenate([[0.0], np.logspace(-1, 4.0, 700)])
    xi = np.exp(-r / 60.0) * np.cos(r / 35.0) / (1.0 + (r / 15.0) ** 2)
    sp = cs.SinhInterpolater(np.stack([r, xi], axis=1), 1.0, 1e-4)
    res = {}
    for name, lmax, F in (("lss_nside256", 767, 128), ("cfg3_like", 2048, 256)):
        xa = 1500.0 + np.arange(F) * (2500.0 / F)
        corrfunc.corr_to_clarray(sp, min(lmax, 64), xa[:8], xromb=1)          # warm-up (scratch, module load)
        torch.cuda.synchronize()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.time()
        cl = corrfunc.corr_to_clarray(sp, lmax, xa, xromb=3, q=2)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ctx.profile_enable(False)
        M = 2 * lmax
        t_xi, t_pr = ctx.profile_get("xi_average")[0], ctx.profile_get("legendre_project")[0]
        res[name] = {"lmax": lmax, "F": F, "mu_nodes": M, "wall_s_incl_host_copy": dt, "xi_average_ms": t_xi,
                     "spline_evals_per_s": M * F * (F + 1) / 2 * 81 / (t_xi * 1e-3),
                     "legendre_project_ms": t_pr, "gemm_TFLOPs": 2.0 * (lmax + 1) * M * F * F / (t_pr * 1e-3) / 1e12,
                     "finite": bool(np.isfinite(cl).all())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
