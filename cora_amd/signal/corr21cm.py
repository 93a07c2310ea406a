"""Counterpart of cora/signal/corr21cm.py: the 21cm brightness-temperature model.

Supplies T_b(z), Pade growth factor / rate and the frequency -> redshift wrapper around the
table model of :mod:`cora_amd.signal.corr`; inherits ``Sky3d`` so ``getsky()`` runs the
whole hot path on the GPU.
"""
from os.path import dirname, join

import numpy as np

from ..core import maps
from ..util import constants
from ..util import cubicspline as cs
from . import corr


class Corr21cm(corr.RedshiftCorrelation, maps.Sky3d):
    r"""Correlation function of HI brightness temperature fluctuations (corr21cm.py:9-35)."""

    add_mean = False
    _kstar = 5.0

    def __init__(self, ps=None, redshift=0.0, sigma_v=0.0, **kwargs):
        if ps is None:
            psfile = join(dirname(__file__), "data/ps_z1.5.dat")
            redshift = 1.5
            c1 = cs.LogInterpolater.fromfile(psfile)
            ps = lambda k: np.exp(-0.5 * k**2 / self._kstar**2) * c1(k)
            # the same function in the form the device table build evaluates itself (csrc/tables21.hip)
            self._ps_plan = dict(callable=ps, spline=c1, kstar=lambda: self._kstar)
        self._sigma_v = sigma_v
        corr.RedshiftCorrelation.__init__(self, ps_vv=ps, redshift=redshift)

    def T_b(self, z):
        r"""Mean 21cm brightness temperature in K (corr21cm.py:37-62)."""
        c = self.cosmology
        return (3.9e-4 * ((c.omega_m + c.omega_l * (1 + z) ** -3) / 0.29) ** -0.5
                * ((1.0 + z) / 2.5) ** 0.5 * (self.omega_HI(z) / 1e-3))

    def mean(self, z):
        return self.T_b(z) if self.add_mean else np.zeros_like(z)

    def omega_HI(self, z):
        """Neutral hydrogen density parameter (corr21cm.py:70-87)."""
        return 6.2e-4

    def x_h(self, z):
        return 1e-3

    def prefactor(self, z):
        return self.T_b(z)

    def _pade(self, z):
        x = ((1.0 / self.cosmology.omega_m) - 1.0) / (1.0 + z) ** 3
        num = 1.0 + 1.175 * x + 0.3064 * x**2 + 0.005355 * x**3
        den = 1.0 + 1.857 * x + 1.021 * x**2 + 0.1530 * x**3
        return x, num, den

    def growth_factor(self, z):
        r"""Pade approximation of the matter growth factor (corr21cm.py:109-139)."""
        x, num, den = self._pade(z)
        return (1.0 + x) ** 0.5 / (1.0 + z) * num / den

    def growth_rate(self, z):
        r"""Growth rate from differentiating the Pade form (corr21cm.py:141-175)."""
        x, num, den = self._pade(z)
        dnum = 3.0 * x * (1.175 + 0.6127 * x + 0.01607 * x**2)
        dden = 3.0 * x * (1.857 + 2.042 * x + 0.4590 * x**2)
        return 1.0 + 1.5 * x / (1.0 + x) + dnum / num - dden / den

    def bias_z(self, z):
        return np.ones_like(z) * 1.0

    def angular_powerspectrum(self, l, nu1, nu2, redshift=False):
        """C_l(nu1, nu2); ``nu`` in MHz unless ``redshift`` is set (corr21cm.py:183-208)."""
        if not redshift:
            z1 = constants.nu21 / np.asarray(nu1, dtype=np.float64) - 1.0
            z2 = constants.nu21 / np.asarray(nu2, dtype=np.float64) - 1.0
        else:
            z1, z2 = nu1, nu2
        return corr.RedshiftCorrelation.angular_powerspectrum_fft(self, l, z1, z2)

    def _clarray_plan(self, aps):
        # skysim.clarray samples in frequency (MHz) when driven through Sky3d
        if getattr(aps, "__func__", None) is Corr21cm.angular_powerspectrum:
            return self._table_plan(lambda nu: constants.nu21 / nu - 1.0)
        return None

    def mean_nu(self, freq):
        return self.mean(constants.nu21 / freq - 1.0)

    def _band_redshifts(self):
        return constants.nu21 / self.nu_upper - 1.0, constants.nu21 / self.nu_lower - 1.0

    def getfield(self, seed=None):
        """Flat-sky realisation of the 21cm signal ``[nu_num, x_num, y_num]``, lowest frequency first
        (corr21cm.py:241-257)."""
        z1, z2 = self._band_redshifts()
        cube = self.realisation(z1, z2, self.x_width, self.y_width, self.nu_num, self.x_num, self.y_num,
                                zspace=False, seed=seed)
        return cube[::-1, :, :].copy()

    def get_kiyo_field(self, refinement=1, seed=None):
        """As :meth:`getfield` but highest frequency first and with a refined box (corr21cm.py:259-276)."""
        z1, z2 = self._band_redshifts()
        return self.realisation(z1, z2, self.x_width, self.y_width, self.nu_num, self.x_num, self.y_num,
                                refinement=refinement, zspace=False, seed=seed)


class EoR21cm(Corr21cm):
    """The 21cm model with parameters for the epoch of reionisation (corr21cm.py:333-385) - what ``cora-makesky 21cm
    --eor`` instantiates (cora/scripts/makesky.py:316-334).  Only the per-redshift quantities change (T_b, Omega_HI,
    x_h, bias 3), so the table model, the device tables and K1 are those of :class:`Corr21cm`."""

    def T_b(self, z):
        r"""Mean 21cm brightness temperature in K, Eq. (4) of Santos, Ferramacho & Silva 2009 (corr21cm.py:334-360)."""
        c = self.cosmology
        h = c.H0 / 100.0
        return (23e-3 * (c.omega_b * h**2 / 0.02) * (0.15 / (c.omega_m * h**2) * ((1.0 + z) / 10)) ** 0.5
                * (h / 0.7) ** -1)

    def omega_HI(self, z):
        return 5e-3

    def x_h(self, z):
        """Neutral hydrogen fraction: a constant (corr21cm.py:365-380)."""
        return 0.25

    def bias_z(self, z):
        """Bias 3, after the EoR estimates of Santos 2004 (corr21cm.py:382-385)."""
        return np.ones_like(z) * 3.0
