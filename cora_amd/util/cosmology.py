"""FLRW background needed by the 21cm C_l model: H(z) and comoving distance.

Counterpart of cora/util/cosmology.py:21-94,156-210,404-430 (host-side, O(nfreq) work:
chi depends only on nu, so skysim.clarray evaluates it once per call instead of once
per l-chunk as the reference does).
"""
from dataclasses import asdict, dataclass

import numpy as np
from scipy import integrate as si

from . import constants


@dataclass
class Cosmology:
    """Planck-2018 defaults (cora/util/cosmology.py:63-80); ``units`` as the reference."""

    units: str = "cosmo"
    omega_b: float = 0.04897
    omega_c: float = 0.26067
    omega_l: float = 0.69036
    omega_g: float = 0.0
    omega_n: float = 0.0
    H0: float = 67.66
    w_0: float = -1.0
    w_a: float = 0.0

    @property
    def omega_m(self):
        return self.omega_b + self.omega_c

    @property
    def omega_r(self):
        return self.omega_g + self.omega_n

    @property
    def omega_k(self):
        return 1.0 - (self.omega_l + self.omega_b + self.omega_c + self.omega_g + self.omega_n)

    def to_dict(self):
        return asdict(self)

    def H(self, z=0.0):
        """Hubble parameter in SI (1/s) (cosmology.py:156-188)."""
        zp = 1 + z
        de = self.omega_l * zp ** (3 * (1 + self.w_0 + self.w_a)) * np.exp(-3 * self.w_a * z / zp)
        e2 = self.omega_r * zp**4 + self.omega_m * zp**3 + self.omega_k * zp**2 + de
        return self.H0 * e2**0.5 * 1000.0 / constants.mega_parsec

    @property
    def _unit_distance(self):
        if self.units == "astro":
            return constants.mega_parsec
        if self.units == "cosmo":
            return constants.mega_parsec / (self.H0 / 100.0)
        if self.units == "si":
            return 1.0
        raise RuntimeError("Units not known")

    def comoving_distance(self, z):
        """chi(z) = int_0^z c/H dz' by ODE integration over the sorted redshifts
        (cosmology.py:190-210 and _intf_0_z :404-430)."""
        return _intf_0_z(lambda z1: constants.c / self.H(z1), z) / self._unit_distance


    def proper_distance(self, z):
        """Comoving transverse separation per unit angle (cosmology.py:212-241): chi, curved by omega_k."""
        x = self.comoving_distance(z)
        om_k = self.omega_k
        dhi = np.sqrt(np.fabs(om_k)) * self.H() / constants.c * self._unit_distance
        if om_k < 0.0:
            x = np.sin(x * dhi) / dhi
        elif om_k > 0.0:
            x = np.sinh(x * dhi) / dhi
        return x


def _intf_0_z(f, z):
    if not isinstance(z, np.ndarray):
        return _intf_0_z(f, np.array([z], dtype=np.float64))[0]
    x = np.zeros_like(z)
    order = np.argsort(z, axis=None)
    za = np.insert(z.ravel()[order], 0, 0)
    x.ravel()[order] = si.odeint(lambda y, t: f(t), 0.0, za)[1:, 0]
    return x
