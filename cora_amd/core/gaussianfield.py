"""Counterpart of cora/core/gaussianfield.py: flat-sky Gaussian random fields.

API-surface row: ``RandomField`` keeps cora's interface so the
model classes that mix it in keep working.  The flat-sky path is NOT part of the HIP hot
path (it is the "next" row n4); it is host numpy exactly as in the reference
(``randn * kweight -> irfftn``, gaussianfield.py:70-120) and no GPU claim is made for it.
"""
import numpy as np

from ..util import constants, fftutil
from . import maps


class RandomField(object):
    r"""n-dimensional Gaussian random field with a user-supplied ``powerspectrum(karray)``.

    Parameters
    ----------
    npix : array_like
        Number of pixels along each axis.
    wsize : array_like, optional
        Physical size of each axis (defaults to ``npix``).
    """

    _kweightgen = False

    def __init__(self, npix=None, wsize=None):
        self._n = np.array(npix) if npix is not None else npix
        self._w = np.array(wsize) if wsize is not None else self._n

    def powerspectrum(self, karray):
        return (karray**2).sum(axis=3)

    def generate_kweight(self, regen=False):
        """k-space weights sqrt(P(k)) * prod(n)/sqrt(2 prod(w)), DC mode zero (gaussianfield.py:70-100)."""
        if self._kweightgen and not regen:
            return
        spacing = self._w / self._n
        kvec = fftutil.rfftfreqn(self._n, spacing / (2 * np.pi))
        self._kweight = self.powerspectrum(kvec) ** 0.5 * self._n.prod() / (2.0 * self._w.prod()) ** 0.5
        self._kweight[tuple([0] * len(self._n))] = 0.0
        self._kweightgen = True

    def getfield(self):
        """One realisation (gaussianfield.py:102-120); uses numpy's global random state."""
        self.generate_kweight()
        s = self._kweight.shape
        f = np.random.standard_normal(s) + 1.0j * np.random.standard_normal(s)
        f *= self._kweight
        return fftutil.irfftn(f)


class RandomFieldA2F(RandomField, maps.Map3d):
    """Two angular dimensions + frequency (gaussianfield.py:123-138)."""

    @classmethod
    def like_map(cls, mapobj, *args, **kwargs):
        c = super(RandomFieldA2F, cls).like_map(mapobj, *args, **kwargs)
        c._n = c._num_array()
        c._w = c._width_array()
        return c


class RandomFieldA2(RandomField, maps.Map2d):
    """Two angular dimensions (gaussianfield.py:141-156)."""

    @classmethod
    def like_map(cls, mapobj, *args, **kwargs):
        c = super(RandomFieldA2, cls).like_map(mapobj, *args, **kwargs)
        c._n = c._num_array()[-2:]
        c._w = c._width_array()[-2:]
        return c
