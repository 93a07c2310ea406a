"""Sky-map geometry classes and the ``Sky3d`` entry point of the hot path.

API counterpart of cora/core/maps.py (same class / attribute names so that code written
against the reference keeps working); the numerical work in ``Sky3d`` goes to the HIP
kernels through :mod:`cora_amd.core.skysim`.
"""
import numpy as np

from .. import _lib
from ..util import constants
from . import skysim


def _cell_centres(extent, count):
    # centres of `count` equal cells tiling [0, extent]
    return (0.5 + np.arange(count)) * (extent / count)


class Map2d(object):
    """Angular patch description (reference: cora/core/maps.py:7-74).

    ``x_width`` / ``y_width`` are in degrees, ``x_num`` / ``y_num`` are pixel counts and
    ``nside`` is the HEALPix resolution used by the full-sky subclasses.
    """

    # fields transferred by like_map(); subclasses extend the tuple
    _geometry_fields = ("x_width", "y_width", "x_num", "y_num", "_nside")

    x_width = y_width = 5.0
    x_num = y_num = 128
    _nside = 128

    @classmethod
    def like_map(cls, mapobj, *args, **kwargs):
        """New instance of ``cls`` sharing the geometry of ``mapobj``."""
        new = cls(*args, **kwargs)
        new._adopt_geometry(mapobj)
        return new

    def _adopt_geometry(self, other):
        for name in self._geometry_fields:
            setattr(self, name, getattr(other, name))

    def _width_array(self):
        deg = constants.degree
        return np.asarray((self.x_width * deg, self.y_width * deg), dtype=np.float64)

    def _num_array(self):
        return np.asarray((self.x_num, self.y_num), dtype=int)

    x_pixels = property(lambda self: _cell_centres(self.x_width, self.x_num))
    y_pixels = property(lambda self: _cell_centres(self.y_width, self.y_num))

    def _get_nside(self):
        return self._nside

    def _set_nside(self, value):
        value = int(value)
        # a positive power of two has exactly one bit set
        if value < 1 or value & (value - 1):
            raise Exception("Not a valid value of nside.")
        self._nside = value

    nside = property(_get_nside, _set_nside, doc="HEALPix resolution (a power of two).")


class Map3d(Map2d):
    """Angular patch plus a frequency axis (reference: cora/core/maps.py:77-200).

    Channels are either given explicitly through ``frequencies`` or are the ``nu_num``
    equal-width channel centres between ``nu_lower`` and ``nu_upper`` (MHz).
    """

    _geometry_fields = Map2d._geometry_fields + ("nu_upper", "nu_lower")

    nu_lower, nu_upper = 500.0, 900.0
    _nu_num = 128
    _frequencies = None

    def _adopt_geometry(self, other):
        Map2d._adopt_geometry(self, other)
        self.nu_num = other.nu_num
        self._frequencies = other._frequencies

    def _width_array(self):
        band = self.nu_upper - self.nu_lower
        return np.concatenate(([band], Map2d._width_array(self)))

    def _num_array(self):
        return np.concatenate(([self.nu_num], Map2d._num_array(self))).astype(int)

    def _get_frequencies(self):
        if self._frequencies is None:
            return self.nu_lower + _cell_centres(self.nu_upper - self.nu_lower, self._nu_num)
        return self._frequencies

    def _set_frequencies(self, freq):
        self._frequencies = freq

    frequencies = property(_get_frequencies, _set_frequencies,
                           doc="Channel centres in MHz.")
    # older spelling still used by callers of the reference
    nu_pixels = frequencies

    @classmethod
    def like_kiyo_map(cls, mapobj, *args, **kwargs):
        r"""A Map3d (or subclass) with the geometry of one of Kiyo's map objects (maps.py:175-201): ``mapobj`` offers
        ``get_axis("freq" | "ra" | "dec")`` (frequencies in Hz, angles in degrees) and ``info["dec_centre"]``."""
        c = cls(*args, **kwargs)

        freq_axis = mapobj.get_axis("freq")
        ra_axis = mapobj.get_axis("ra")
        dec_axis = mapobj.get_axis("dec")

        ra_fact = np.cos(np.pi * mapobj.info["dec_centre"] / 180.0)
        c.x_width = (max(ra_axis) - min(ra_axis)) * ra_fact
        c.y_width = max(dec_axis) - min(dec_axis)
        c.x_num, c.y_num = (len(ra_axis), len(dec_axis))

        c.nu_lower = min(freq_axis) / 1.0e6
        c.nu_upper = max(freq_axis) / 1.0e6
        c.nu_num = len(freq_axis)

        print("Map3D: %dx%d field (%fx%f deg) from nu=%f to nu=%f (%d bins)"
              % (c.x_num, c.y_num, c.x_width, c.y_width, c.nu_lower, c.nu_upper, c.nu_num))
        return c

    def _set_nu_num(self, num):
        self._nu_num = num

    nu_num = property(lambda self: len(self.frequencies), _set_nu_num,
                      doc="Number of channels.")


class Sky3d(Map3d):
    """Full-sky, multi-frequency Gaussian field (reference: cora/core/maps.py:203-252).

    Subclasses supply ``angular_powerspectrum(l, nu1, nu2)`` and optionally ``mean_nu``.
    ``oversample`` is the Romberg order of the channel-width integral of C_l; ``lmax``
    (an extension) overrides the band limit, which defaults to the reference's
    ``3 * nside - 1``.
    """

    oversample = 3
    lmax = None

    def angular_powerspectrum(self, l, nu1, nu2):
        raise Exception("Not implemented in base class.")

    def getfield(self):
        raise Exception("Not implemented in base class.")

    def mean_nu(self, freq):
        return np.zeros_like(freq)

    def _lmax(self):
        if self.lmax is None:
            return 3 * self.nside - 1
        return int(self.lmax)

    def _channels(self):
        return np.asarray(self.nu_pixels, dtype=np.float64)

    def getsky(self, rng=None):
        """Realisation of the unpolarised sky, ``[nfreq, npix]`` (maps.py:227-237).

        ``rng`` is an extension (the reference draws from numpy's global state) and is
        handed to :func:`skysim.mkfullsky_device`.
        """
        nu = self._channels()
        # the normals depend on the generator alone: its device passes (numpy's PCG64 / legacy MT19937 stream) start
        # before the C_l integration and run beside it and the factorisation
        ctx = _lib.get_context()
        prep = None
        if not isinstance(rng, skysim.DeviceRNG):
            prep = skysim.prepare_numpy_stream(ctx, rng, self._lmax(), len(nu))
        try:
            cl = skysim.clarray_device(self.angular_powerspectrum, self._lmax(), nu,
                                       zromb=self.oversample)
        except BaseException:
            if prep is not None:
                prep.abort()
            raise
        fluct = skysim.mkfullsky_device(cl, self.nside, rng=rng, prepared=prep)
        mean = np.asarray(self.mean_nu(nu), dtype=np.float64) * np.ones(nu.shape)
        if np.any(mean != 0.0):          # added on the device: a pass over 25.8 GB of host memory is seconds
            fluct += ctx.to_device(mean)[:, None]
        return ctx.to_host(fluct)        # pinned memory: PCIe-rate copy

    def getpolsky(self, rng=None):
        """``[nfreq, 4, npix]`` Stokes cube with only I populated (maps.py:239-247)."""
        stokes_i = self.getsky(rng=rng)
        nfreq, npix = stokes_i.shape
        cube = np.zeros((nfreq, 4, npix), dtype=stokes_i.dtype)
        cube[:, 0, :] = stokes_i
        return cube

    def getalms(self, lmax, rng=None):
        """Harmonic coefficients ``[nfreq, 1, lmax+1, lmax+1]`` (maps.py:249-252)."""
        cl = skysim.clarray_device(self.angular_powerspectrum, lmax, self._channels())
        return _lib.get_context().to_host(skysim.mkfullsky_device(cl, self.nside, alms=True, rng=rng))
