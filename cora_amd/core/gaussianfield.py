"""Flat-sky Gaussian random fields on the GPU.

API counterpart of cora/core/gaussianfield.py (SURVEY 8 row a12 / "next" row n4): the same classes
and attributes, with ``getfield`` = (normals * kweight) -> ``irfftn`` running in ``csrc/flatsky.hip``.
The k-space weights come from the user's ``powerspectrum`` callable (host, once) and then stay
resident on the device.
"""
import math

import numpy as np

from .. import _lib
from ..util import constants, fftutil
from . import maps


class RandomField(object):
    """n-dimensional Gaussian field with power spectrum ``powerspectrum(karray)``.

    Parameters
    ----------
    npix : sequence of int
        Pixels along each axis.
    wsize : sequence of float, optional
        Extent of each axis (any units); ``npix`` when omitted, i.e. unit pixels.
    """

    _kweightgen = False
    _kweight_dev = None
    _n = None
    _w = None

    def __init__(self, npix=None, wsize=None):
        self._n = None if npix is None else np.array(npix)
        self._w = self._n if wsize is None else np.array(wsize)

    def _check_input(self):
        # same three conditions and messages as gaussianfield.py:34-42
        if self._n is None or self._w is None:
            raise Exception("Either self._n or self._w has not been set.")
        if len(self._n) != len(self._w):
            raise Exception("Width array must be the same length as number of pixels.")
        if not ((np.asarray(self._n) > 0).all() and (np.asarray(self._w) > 0).all()):
            raise Exception("Array elements must be positive.")

    def powerspectrum(self, karray):
        """P(k) for an array of wavevectors ``[..., ndim]`` (angular wavenumbers, 2 pi / length)."""
        raise Exception("Abstract method: need to override.")

    def generate_kweight(self, regen=False):
        """``sqrt(P(k)) prod(n) / sqrt(2 prod(w))`` on the rfftn grid, non-finite DC -> 0 (gaussianfield.py:70-100)."""
        self._check_input()
        if self._kweightgen and not regen:
            return
        n = np.asarray(self._n)
        w = np.asarray(self._w, dtype=np.float64)
        kvec = fftutil.rfftfreqn(n, (w / n) / (2 * np.pi))
        kw = self.powerspectrum(kvec) ** 0.5 * n.prod() / (2.0 * w.prod()) ** 0.5
        if not np.isfinite(kw.flat[0]):
            kw.flat[0] = 0.0
        self._kweight = kw
        self._kweight_dev = None
        self._kweightgen = True

    def _device_kweight(self):
        if self._kweight_dev is None:
            self._kweight_dev = _lib.get_context().to_device(self._kweight)
        return self._kweight_dev

    def getfield_device(self, seed=None):
        """One realisation as a device tensor.

        ``seed=None`` reproduces the reference: two ``np.random.standard_normal`` arrays from numpy's
        global state (gaussianfield.py:115), uploaded.  An integer ``seed`` (extension) draws the
        normals on the GPU from the Philox stream (element index = counter), nothing crosses PCIe.
        """
        self.generate_kweight()
        ctx = _lib.get_context()
        if seed is None:
            s = self._kweight.shape
            f = np.random.standard_normal(s) + 1.0j * np.random.standard_normal(s)
            f *= self._kweight
            spec = ctx.to_device(f, dtype=np.complex128)
        else:
            # draw + irfftn in one call: the spectrum is generated where the first transform pass loads it
            return ctx.randomfield_irfftn(self._device_kweight(), seed)
        return ctx.irfftn(spec)

    def getfield(self, seed=None):
        """One realisation, shape ``n`` with the last axis rounded down to even (gaussianfield.py:102-120)."""
        return self.getfield_device(seed=seed).cpu().numpy()


class _MapGeometryField(RandomField):
    # shared by the two map mix-ins: pixel counts and widths come from the Map2d / Map3d attributes
    def generate_kweight(self, *args):
        self._n = self._num_array()
        self._w = self._width_array()
        RandomField.generate_kweight(self, *args)


class RandomFieldA2F(_MapGeometryField, maps.Map3d):
    """Frequency x two angles; geometry from the ``Map3d`` attributes (gaussianfield.py:123-138)."""


class RandomFieldA2(_MapGeometryField, maps.Map2d):
    """Two angles; geometry from the ``Map2d`` attributes (gaussianfield.py:141-156)."""


class Cmb(RandomFieldA2):
    r"""A patch of the CMB (gaussianfield.py:159-182): ``psfile`` holds (l, l(l+1)C_l/2pi) rows as CAMB writes them
    (``cambnorm``) or (l, C_l) rows; the spectrum is a log-log spline in \|k\|.  (The reference's default file,
    ``cora/core/ps_cmb2.dat``, is not part of its tree: a ``psfile`` must be given.)"""

    def __init__(self, psfile=None, cambnorm=True):
        from ..util.cubicspline import LogInterpolater

        if psfile is None:
            from os.path import dirname, join

            psfile = join(dirname(__file__), "ps_cmb2.dat")
        if cambnorm:
            a = np.loadtxt(psfile)
            l = a[:, 0]
            tt = (2 * math.pi) * a[:, 1] / (l * (l + 1.0))
            self._powerspectrum_int = LogInterpolater(np.vstack((l, tt)).T)
        else:
            self._powerspectrum_int = LogInterpolater.fromfile(psfile)

    def powerspectrum(self, karray):
        return self._powerspectrum_int((karray**2).sum(axis=2) ** 0.5)


class TestF(RandomFieldA2F):
    """Test spectrum: a Gaussian of 250 MHz^-1 along frequency times a 1-degree Gaussian on the sky
    (gaussianfield.py:185-191)."""

    __test__ = False      # (not a pytest class)

    def powerspectrum(self, karray):
        return (np.exp(-0.5 * (karray[..., 0] / (2 * math.pi / 250.0)) ** 2)
                * np.exp(-0.5 * (karray[..., 1:3] ** 2).sum(axis=3) / (2 * math.pi / (1.0 * constants.degree)) ** 2))
