#!/usr/bin/env python3
"""Golden vectors for the flat-sky row (flat-sky half): outputs of the reference's own
``RandomField`` / ``RandomFieldA2`` / ``RandomFieldA2F`` (cora/core/gaussianfield.py:9-156),
``fftutil.rfftfreqn`` (cora/util/fftutil.py:14-61) and ``ForegroundMap.getfield``
(cora/foreground/gaussianfg.py:43-84), obtained by importing the reference in this container with the
stand-ins of make_golden.py (caput constants / mpiarray, healpy stub) plus empty stand-ins for the two Cython
modules gaussianfg.py imports but does not use on this path.  Randomness: numpy's GLOBAL state seeded with
``np.random.seed`` - the reference draws from it (gaussianfield.py:115, gaussianfg.py:79).

The reference compares arrays with ``None`` by ``==`` (gaussianfield.py:35, fftutil.py:33), which current
numpy refuses to reduce to a truth value, so as written it only runs for 1-d fields.  The inputs are therefore
handed over as an ndarray subclass whose ``== None`` is ``False`` (what numpy returned when the reference was
written); no reference code is changed.

Commits data only: tests/golden/flatsky_vectors.npz.      python tests/golden/make_golden_flatsky.py
"""
import os
import sys
import types

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, OUT)
import make_golden  # noqa: E402  (shims only)


class LegacyArray(np.ndarray):
    """ndarray with the pre-1.13 scalar result of ``arr == None``."""

    def __eq__(self, other):
        if other is None:
            return False
        return np.ndarray.__eq__(self, other)

    def __ne__(self, other):
        if other is None:
            return True
        return np.ndarray.__ne__(self, other)

    __hash__ = None


def legacy(a):
    return np.array(a).view(LegacyArray)


def ps_model(karray):
    """Test spectrum (restated in tests/): finite at k = 0, falls as a power law."""
    k2 = (np.asarray(karray) ** 2).sum(axis=-1)
    return 3.0 / (1.0 + k2) ** 1.3


def _legacy_geometry(obj):
    num, wid = obj._num_array, obj._width_array
    obj._num_array = lambda: legacy(num())
    obj._width_array = lambda: legacy(wid())


def redshift_cube_vectors(gaussianfield):
    """``Corr21cm.getfield`` -> ``RedshiftCorrelation.realisation`` -> ``_realisation_dv`` (corr.py:562-760,
    corr21cm.py:241-257).  ``_realisation_dv`` builds ``RandomField(npix=n, wsize=d)`` from plain arrays, whose
    ``npix != None`` test (gaussianfield.py:30) current numpy refuses: the class is wrapped so that its arguments
    arrive as legacy arrays.  Needs the reference's compiled cubicspline (make_golden._build_cython)."""
    import tempfile

    for name in ("cora.util.cubicspline", "cora.util.bilinearmap"):
        sys.modules.pop(name, None)
    make_golden._build_cython(tempfile.mkdtemp(prefix="cora_golden_fs_"))
    for name in [m for m in sys.modules if m.startswith("cora.signal") or m.startswith("cora.foreground")]:
        sys.modules.pop(name)       # re-import against the real cubicspline
    from cora.signal import corr21cm

    base = gaussianfield.RandomField

    class LegacyArgsField(base):
        def __init__(self, npix=None, wsize=None):
            base.__init__(self, npix=legacy(npix), wsize=legacy(wsize))
            self._n, self._w = legacy(self._n), legacy(self._w)   # np.array() in __init__ drops the subclass

    gaussianfield.RandomField = LegacyArgsField
    out = {}
    try:
        for tag, kw in (("cube_a", dict(nu_num=6, x_num=8, y_num=10, x_width=3.0, y_width=4.0, nu_lower=700.0, nu_upper=720.0)),
                        ("cube_b", dict(nu_num=5, x_num=7, y_num=6, x_width=2.0, y_width=2.0, nu_lower=500.0, nu_upper=540.0))):
            cr = corr21cm.Corr21cm()
            for k, v in kw.items():
                setattr(cr, k, v)
            z1 = 1420.40575177 / cr.nu_upper - 1.0
            z2 = 1420.40575177 / cr.nu_lower - 1.0
            np.random.seed(41)
            out[tag + "__getfield"] = cr.getfield()
            np.random.seed(41)
            acube, rsf, geom = cr.realisation(z1, z2, cr.x_width, cr.y_width, cr.nu_num, cr.x_num, cr.y_num,
                                              zspace=False, report_physical=True)
            out[tag + "__acube"] = acube
            out[tag + "__rsf"] = rsf
            out[tag + "__geom"] = np.array(geom)
            np.random.seed(41)
            out[tag + "__density_only_nomean_zspace"] = cr.realisation(
                z1, z2, cr.x_width, cr.y_width, cr.nu_num, cr.x_num, cr.y_num, density_only=True, no_mean=True,
                no_evolution=True, refinement=2)
            out[tag + "__params"] = np.array([kw[k] for k in ("nu_num", "x_num", "y_num", "x_width", "y_width",
                                                              "nu_lower", "nu_upper")], dtype=np.float64)
    finally:
        gaussianfield.RandomField = base
    return out


def main():
    make_golden._install_shims()
    for name in ("cora.util.cubicspline", "cora.util.bilinearmap"):
        sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, make_golden.REF)
    import cora.util as cu

    cu.cubicspline = sys.modules["cora.util.cubicspline"]
    cu.bilinearmap = sys.modules["cora.util.bilinearmap"]
    from cora.core import gaussianfield
    from cora.foreground import gaussianfg
    from cora.util import fftutil

    g = {}
    # frequency grids, even and odd leading axes (the odd case is the reference's half-integer grid)
    g["kvec_8_6_10"] = fftutil.rfftfreqn((8, 6, 10), legacy([0.5, 0.25, 2.0]))
    g["kvec_5_7_9"] = fftutil.rfftfreqn((5, 7, 9))

    # RandomField with an explicit spectrum: 3-d power-of-two, 3-d general lengths, 2-d
    for tag, n, w, seed in (("rf_16_16_16", (16, 16, 16), (40.0, 30.0, 20.0), 11),
                            ("rf_12_10_14", (12, 10, 14), (5.0, 7.0, 9.0), 12),
                            ("rf_24_36", (24, 36), (3.0, 2.0), 13)):
        rf = gaussianfield.RandomField(npix=list(n), wsize=list(w))
        rf._n, rf._w = legacy(rf._n), legacy(rf._w)
        rf.powerspectrum = ps_model
        np.random.seed(seed)
        fld = rf.getfield()
        g[tag + "__kweight"] = rf._kweight
        g[tag + "__field"] = fld
        g[tag + "__n"] = np.array(n)
        g[tag + "__w"] = np.array(w)
        g[tag + "__seed"] = np.array(seed)

    # the Map2d / Map3d mix-ins
    a2 = gaussianfield.RandomFieldA2()
    a2.x_num, a2.y_num, a2.x_width, a2.y_width = 12, 16, 4.0, 6.0
    a2.powerspectrum = ps_model
    _legacy_geometry(a2)
    np.random.seed(21)
    g["a2__field"] = a2.getfield()
    g["a2__kweight"] = a2._kweight
    a2f = gaussianfield.RandomFieldA2F()
    a2f.x_num, a2f.y_num, a2f.nu_num = 8, 10, 6
    a2f.x_width, a2f.y_width, a2f.nu_lower, a2f.nu_upper = 3.0, 5.0, 500.0, 560.0
    a2f.powerspectrum = ps_model
    _legacy_geometry(a2f)
    np.random.seed(22)
    g["a2f__field"] = a2f.getfield()
    g["a2f__kweight"] = a2f._kweight

    # ForegroundMap.getfield: the SCK point-source model on a small patch
    syn = gaussianfg.PointSources()  # well conditioned: the Cholesky branch (the eigh branch of the reference returns a 3-d root and getfield raises)
    syn.x_num, syn.y_num, syn.nu_num = 16, 12, 5
    syn.x_width, syn.y_width, syn.nu_lower, syn.nu_upper = 6.0, 4.0, 400.0, 800.0
    syn._weight_gen = False
    # the angular field is made inside generate_weight by RandomFieldA2.like_map(self): give that class
    # the legacy-array geometry for the duration of the call
    A2 = gaussianfield.RandomFieldA2
    o_num, o_wid = A2._num_array, A2._width_array
    A2._num_array = lambda self: legacy(o_num(self))
    A2._width_array = lambda self: legacy(o_wid(self))
    try:
        np.random.seed(31)
        g["syn__field"] = syn.getfield()
    finally:
        A2._num_array, A2._width_array = o_num, o_wid
    g["syn__freq_weight"] = syn._freq_weight
    g["syn__num_corr_freq"] = np.array(syn._num_corr_freq)
    g["syn__ang_kweight"] = syn._ang_field._kweight

    g.update(redshift_cube_vectors(gaussianfield))

    path = os.path.join(OUT, "flatsky_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, len(g), "arrays", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
