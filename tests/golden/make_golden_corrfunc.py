#!/usr/bin/env python3
"""Golden vectors for the xi(r) -> C_l(chi, chi') integrator: outputs of the reference's own
``corrfunc.corr_to_clarray`` and ``legendre_array`` (cora/signal/corrfunc.py:265-400), obtained by importing the
reference in this container.  Stand-ins used for absent third-party modules (import-time only unless stated):

  hankl, hankel, pyfftlog      - imported at the top of corrfunc.py, not used by the two functions
  cora.util.bilinearmap        - ditto (a Cython extension)
  caput.mpiarray               - single-process stand-in (zeros / local_offset / local_array / reshape(None, ..) /
                                 redistribute / wrap), as in make_golden.py
  caput.astro.coordinates      - ``spherical.cosine_rule(mu, x1, x2)`` IS used: restated here as
                                 r = sqrt((x1 - x2)^2 + 2 x1 x2 (1 - mu)) broadcast to [mu, x1, x2]; caput is an
                                 unpinned, absent dependency, so parity is UNPINNED for this one function.

Commits data only: tests/golden/corrfunc_vectors.npz.   python tests/golden/make_golden_corrfunc.py [/root/reference]
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def xi_model(r):
    """The test correlation function (also restated in tests/): smooth, finite at r = 0, sign-changing."""
    r = np.asarray(r, dtype=np.float64)
    return np.exp(-r / 60.0) * np.cos(r / 35.0) / (1.0 + (r / 15.0) ** 2)


def _shims():
    for name in ("hankl", "hankel", "pyfftlog"):
        sys.modules[name] = types.ModuleType(name)
    caput = types.ModuleType("caput")
    astro = types.ModuleType("caput.astro")
    coords = types.ModuleType("caput.astro.coordinates")
    sph = types.SimpleNamespace()

    def cosine_rule(mu, x1, x2):
        mu = np.asarray(mu)[:, None, None]
        a = np.asarray(x1)[None, :, None]
        b = np.asarray(x2)[None, None, :]
        return np.sqrt((a - b) ** 2 + 2.0 * a * b * (1.0 - mu))

    sph.cosine_rule = cosine_rule
    coords.spherical = sph
    astro.coordinates = coords
    caput.astro = astro
    mpa = types.ModuleType("caput.mpiarray")

    class MPIArray(np.ndarray):
        @classmethod
        def wrap(cls, a, axis=0):
            return np.asarray(a).view(cls)

        @property
        def local_array(self):
            return self.view(np.ndarray)

        @property
        def local_offset(self):
            return (0,) * self.ndim

        def redistribute(self, axis):
            return self

        def reshape(self, *shape):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
                shape = tuple(shape[0])
            shape = tuple(self.shape[i] if s is None else s for i, s in enumerate(shape))
            return np.ndarray.reshape(self.view(np.ndarray), shape).view(MPIArray)

    mpa.MPIArray = MPIArray
    mpa.zeros = lambda shape, dtype=np.float64, axis=0: np.zeros(shape, dtype=dtype).view(MPIArray)
    caput.mpiarray = mpa
    sys.modules.update({"caput": caput, "caput.astro": astro, "caput.astro.coordinates": coords, "caput.mpiarray": mpa})
    # package skeleton so that the relative imports of corrfunc.py resolve without importing cora.util's extensions
    for pkg in ("cora", "cora.signal", "cora.util"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, *pkg.split("."))]
        sys.modules[pkg] = m
    sys.modules["cora.util.bilinearmap"] = types.ModuleType("cora.util.bilinearmap")
    sys.modules["cora.util"].bilinearmap = sys.modules["cora.util.bilinearmap"]
    spec = importlib.util.spec_from_file_location("cora.util.nputil", os.path.join(REF, "cora", "util", "nputil.py"))
    npu = importlib.util.module_from_spec(spec)
    sys.modules["cora.util.nputil"] = npu
    spec.loader.exec_module(npu)


def main():
    _shims()
    spec = importlib.util.spec_from_file_location("cora.signal.corrfunc", os.path.join(REF, "cora", "signal", "corrfunc.py"))
    cf = importlib.util.module_from_spec(spec)
    sys.modules["cora.signal.corrfunc"] = cf
    spec.loader.exec_module(cf)
    g = {}
    g["legendre_l12_mu"] = np.array([-1.0, -0.3, 0.0, 0.5, 0.999, 1.0])
    g["legendre_l12"] = cf.legendre_array(12, g["legendre_l12_mu"])
    xa = np.array([1500.0, 1520.0, 1545.0, 1575.0, 1610.0, 1650.0])
    g["xarray"] = xa
    for tag, kw in (("l40_xromb2_q2", dict(lmax=40, xromb=2, q=2)),
                    ("l40_xromb0_q3", dict(lmax=40, xromb=0, q=3)),
                    ("l24_xromb1_xw10", dict(lmax=24, xromb=1, q=2, xwidth=10.0))):
        kw = dict(kw)
        lmax = kw.pop("lmax")
        g["cl_" + tag] = np.asarray(cf.corr_to_clarray(xi_model, lmax, xa, chunksize=7, **kw))
    path = os.path.join(OUT, "corrfunc_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
