"""Counterpart of cora/foreground/gaussianfg.py: separable Gaussian foregrounds
C_l(nu, nu') = A_l B(nu, nu') in the style of Santos, Cooray & Knox (astro-ph/0408515)."""
import numpy as np

from ..core import maps


class ForegroundMap(maps.Sky3d):
    r"""Foregrounds with separable angular and frequency covariance (gaussianfg.py:20-84)."""

    def angular_ps(self, l):
        r"""The angular function A_l (vectorised)."""
        pass

    def frequency_covariance(self, nu1, nu2):
        pass

    def angular_powerspectrum(self, l, nu1, nu2):
        return self.angular_ps(l) * self.frequency_covariance(nu1, nu2)

    def _clarray_plan(self, aps):
        """Protocol used by ``skysim.clarray``: C_l = A_l x B, one outer-product kernel."""
        if getattr(aps, "__func__", None) is not ForegroundMap.angular_powerspectrum:
            return None

        def prepare(larr, za):
            al = np.asarray(self.angular_ps(larr), dtype=np.float64)
            bcov = self.frequency_covariance(za[:, np.newaxis], za[np.newaxis, :])
            return al, np.ascontiguousarray(bcov, dtype=np.float64)

        return dict(kind="separable", prepare=prepare)

    def getfield(self):
        raise NotImplementedError("flat-sky foreground cubes (gaussianfg.py:43-84) are out of scope of cora_amd")


class ForegroundSCK(ForegroundMap):
    r"""SCK-style foregrounds; needs ``A``, ``alpha``, ``beta``, ``zeta`` (gaussianfg.py:87-130).

    C_l = A (l/l_0)^-beta (nu1 nu2/nu_0^2)^-alpha exp(-(ln(nu1/nu2))^2 / (2 zeta^2)), C_0 = 0.
    """

    nu_0 = 130.0
    l_0 = 1000.0

    def angular_ps(self, larray):
        if isinstance(larray, np.ndarray):
            mask0 = np.where(larray == 0)
            larray[mask0] = 1.0  # in place, as the reference does (gaussianfg.py:108-110)
        psarray = self.A * (larray / self.l_0) ** (-self.beta)
        if isinstance(larray, np.ndarray):
            psarray[mask0] = 0.0
        return psarray

    def frequency_covariance(self, nu1, nu2):
        return (self.frequency_variance(nu1) * self.frequency_variance(nu2)) ** 0.5 * self.frequency_correlation(
            nu1, nu2)

    def frequency_variance(self, nu):
        r"""Variance on a single frequency slice."""
        return (nu / self.nu_0) ** (-2 * self.alpha)

    def frequency_correlation(self, nu1, nu2):
        r"""Correlation between two frequency slices."""
        return np.exp(-0.5 * (np.log(nu1 / nu2) / self.zeta) ** 2)

    def frequency_correlation_dlog(self, dlognu):
        return np.exp(-(dlognu**2) / (2 * self.zeta**2))


class Synchrotron(ForegroundSCK):
    A = 7.00e-4
    alpha = 2.80
    beta = 2.4
    zeta = 4.0


class ExtraGalacticFreeFree(ForegroundSCK):
    A = 1.40e-8
    alpha = 2.10
    beta = 1.0
    zeta = 35.0


class GalacticFreeFree(ForegroundSCK):
    A = 8.80e-8
    alpha = 2.15
    beta = 3.0
    zeta = 35.0


class PointSources(ForegroundSCK):
    A = 5.70e-5
    alpha = 2.07
    beta = 1.1
    zeta = 1.0
