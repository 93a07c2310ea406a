"""Counterpart of cora/core/maps.py: map geometry / frequency bookkeeping and the
``Sky3d`` orchestration of the hot path (``getsky`` -> clarray -> mkfullsky)."""
import numpy as np

from ..util import constants
from . import skysim


class Map2d(object):
    r"""A 2-d sky map (cora/core/maps.py:7-74).

    Attributes
    ----------
    x_width, y_width : float
        Angular size along each axis (in degrees).
    x_num, y_num : int
        Number of pixels along each angular axis.
    """

    x_width = 5.0
    y_width = 5.0
    x_num = 128
    y_num = 128
    _nside = 128

    @classmethod
    def like_map(cls, mapobj, *args, **kwargs):
        c = cls(*args, **kwargs)
        for attr in ("x_width", "y_width", "x_num", "y_num", "_nside"):
            setattr(c, attr, getattr(mapobj, attr))
        return c

    def _width_array(self):
        return np.array([self.x_width, self.y_width], dtype=np.float64) * constants.degree

    def _num_array(self):
        return np.array([self.x_num, self.y_num], dtype=int)

    @property
    def x_pixels(self):
        return (np.arange(self.x_num) + 0.5) * (self.x_width / self.x_num)

    @property
    def y_pixels(self):
        return (np.arange(self.y_num) + 0.5) * (self.y_width / self.y_num)

    @property
    def nside(self):
        """The resolution of the Healpix map (must be power of 2)."""
        return self._nside

    @nside.setter
    def nside(self, value):
        ns = int(value)
        lns = np.log2(ns)
        if int(lns) != lns or lns < 0:
            raise Exception("Not a valid value of nside.")
        self._nside = ns


class Map3d(Map2d):
    r"""A 3-d sky map (cora/core/maps.py:77-200): adds the frequency axis.

    ``nu_lower``/``nu_upper`` are the band edges with ``nu_num`` channels between them,
    unless explicit ``frequencies`` are set.
    """

    nu_lower = 500.0
    nu_upper = 900.0
    _frequencies = None
    _nu_num = 128

    @classmethod
    def like_map(cls, mapobj, *args, **kwargs):
        c = cls(*args, **kwargs)
        for attr in ("x_width", "y_width", "x_num", "y_num", "_nside", "nu_upper", "nu_lower"):
            setattr(c, attr, getattr(mapobj, attr))
        c.nu_num = mapobj.nu_num
        c._frequencies = mapobj._frequencies
        return c

    def _width_array(self):
        return np.array([self.nu_upper - self.nu_lower, self.x_width * constants.degree,
                         self.y_width * constants.degree], dtype=np.float64)

    def _num_array(self):
        return np.array([self.nu_num, self.x_num, self.y_num], dtype=int)

    @property
    def nu_num(self):
        return len(self.frequencies)

    @nu_num.setter
    def nu_num(self, num):
        self._nu_num = num

    @property
    def frequencies(self):
        """List of frequencies in the map (channel centres, MHz)."""
        if self._frequencies is not None:
            return self._frequencies
        width = (self.nu_upper - self.nu_lower) / self._nu_num
        return self.nu_lower + (np.arange(self._nu_num) + 0.5) * width

    @frequencies.setter
    def frequencies(self, freq):
        self._frequencies = freq

    # Alias for frequencies for supporting old code.
    nu_pixels = frequencies


class Sky3d(Map3d):
    """Base class for full-sky multi-frequency maps (cora/core/maps.py:203-252).

    Attributes
    ----------
    oversample : int
        Romberg order of the channel-width integration of C_l.
    lmax : int or None
        Extension: band limit of the realisation; ``None`` = the reference's ``3*nside-1``.
    """

    oversample = 3
    lmax = None

    def angular_powerspectrum(self, l, nu1, nu2):
        raise Exception("Not implemented in base class.")

    def mean_nu(self, freq):
        return np.zeros_like(freq)

    def getfield(self):
        raise Exception("Not implemented in base class.")

    def _lmax(self):
        return 3 * self.nside - 1 if self.lmax is None else int(self.lmax)

    def getsky(self, rng=None):
        """Create a map of the unpolarised sky, ``[nfreq, npix]`` (maps.py:227-237).

        ``rng`` (extension; the reference always uses numpy's global state) is passed to
        :func:`skysim.mkfullsky`.
        """
        freq = np.asarray(self.nu_pixels, dtype=np.float64)
        cla = skysim.clarray_device(self.angular_powerspectrum, self._lmax(), freq, zromb=self.oversample)
        sky = skysim.mkfullsky_device(cla, self.nside, rng=rng).cpu().numpy()
        return self.mean_nu(freq)[:, np.newaxis] + sky

    def getpolsky(self, rng=None):
        """Fully polarised sky ``[nfreq, 4, npix]`` with Q = U = V = 0 (maps.py:239-247)."""
        sky_I = self.getsky(rng=rng)
        sky_IQU = np.zeros((sky_I.shape[0], 4, sky_I.shape[1]), dtype=sky_I.dtype)
        sky_IQU[:, 0] = sky_I
        return sky_IQU

    def getalms(self, lmax, rng=None):
        """a_lm ``[nfreq, 1, lmax+1, lmax+1]`` (maps.py:249-252; default Romberg order 3)."""
        freq = np.asarray(self.nu_pixels, dtype=np.float64)
        cla = skysim.clarray_device(self.angular_powerspectrum, lmax, freq)
        return skysim.mkfullsky_device(cla, self.nside, alms=True, rng=rng).cpu().numpy()
