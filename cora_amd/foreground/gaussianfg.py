"""Separable Gaussian foreground models, C_l(nu, nu') = A_l * B(nu, nu').

API counterpart of cora/foreground/gaussianfg.py (model family of Santos, Cooray & Knox,
astro-ph/0408515).  The split into an angular factor and a frequency covariance is what
lets K1 evaluate these models as a single outer product on the device.
"""
import numpy as np

from .. import _lib
from ..core import gaussianfield, maps
from ..util import nputil


class ForegroundMap(maps.Sky3d):
    """Base of the separable models (reference: gaussianfg.py:20-84).

    A subclass provides ``angular_ps(l)`` (vectorised A_l) and
    ``frequency_covariance(nu1, nu2)``; their product is the ``angular_powerspectrum``.
    """

    def angular_ps(self, l):
        pass

    def frequency_covariance(self, nu1, nu2):
        pass

    def angular_powerspectrum(self, l, nu1, nu2):
        return self.angular_ps(l) * self.frequency_covariance(nu1, nu2)

    def _clarray_plan(self, aps):
        # skysim.clarray asks the model whether it can describe `aps` in closed form.
        # Only the un-overridden product above qualifies.
        if getattr(aps, "__func__", None) is not ForegroundMap.angular_powerspectrum:
            return None

        def prepare(larr, za):
            a_l = np.asarray(self.angular_ps(larr), dtype=np.float64)
            b = self.frequency_covariance(za[:, np.newaxis], za[np.newaxis, :])
            return a_l, np.ascontiguousarray(b, dtype=np.float64)

        return {"kind": "separable", "prepare": prepare}

    _weight_gen = False

    def generate_weight(self, regen=False):
        """Frequency root and angular field generator of the flat-sky cube (gaussianfg.py:43-70)."""
        if self._weight_gen and not regen:
            return
        nu = np.asarray(self.nu_pixels, dtype=np.float64)
        cov = self.frequency_covariance(nu[np.newaxis, :], nu[:, np.newaxis])
        root, self._num_corr_freq = nputil.matrix_root_manynull(cov)
        # the eigen-decomposition branch of the reference hands back a [1, F, k] root, with which its own
        # getfield raises in tensordot (gaussianfg.py:80); use the [F, k] matrix it means
        self._freq_weight = root[0] if root.ndim == 3 else root
        patch = gaussianfield.RandomFieldA2.like_map(self)
        patch.powerspectrum = lambda karray: self.angular_ps((karray**2).sum(axis=2) ** 0.5)
        self._ang_field = patch
        self._weight_gen = True

    def getfield_device(self):
        """Flat-sky cube ``[nfreq, x_num, y_num]`` as a device tensor (gaussianfg.py:72-84).

        One angular realisation is transformed back to k-space, multiplied by frequency-correlated
        normals (numpy global state, as the reference) and inverse transformed over the two angles -
        rfftn, mixing and the batched irfft2 all run in csrc/flatsky.hip.
        """
        self.generate_weight()
        ctx = _lib.get_context()
        aff = ctx.rfftn(self._ang_field.getfield_device())
        normals = np.random.standard_normal((self._num_corr_freq,) + tuple(aff.shape))
        mixed = ctx.fg_mix(ctx.to_device(self._freq_weight), ctx.to_device(normals), aff)
        return ctx.irfftn(mixed, naxes=2)

    def getfield(self):
        return self.getfield_device().cpu().numpy()


class ForegroundSCK(ForegroundMap):
    """Power laws in l and nu with a log-normal frequency coherence (gaussianfg.py:87-130).

    C_l = A (l/l_0)^-beta (nu1 nu2 / nu_0^2)^-alpha exp(-ln(nu1/nu2)^2 / (2 zeta^2)),
    with the monopole forced to zero.  Concrete models set ``A, alpha, beta, zeta``.
    """

    l_0 = 1000.0
    nu_0 = 130.0

    def angular_ps(self, larray):
        if not isinstance(larray, np.ndarray):
            return self.A * (larray / self.l_0) ** (-self.beta)
        # The reference overwrites l = 0 entries of the caller's array with 1
        # (gaussianfg.py:108-110); callers observe that, so it is kept.
        monopole = np.where(larray == 0)
        larray[monopole] = 1.0
        a_l = self.A * (larray / self.l_0) ** (-self.beta)
        a_l[monopole] = 0.0
        return a_l

    def frequency_variance(self, nu):
        """B(nu, nu)."""
        return (nu / self.nu_0) ** (-2 * self.alpha)

    def frequency_correlation(self, nu1, nu2):
        """B(nu1, nu2) / sqrt(B(nu1, nu1) B(nu2, nu2))."""
        return np.exp(-0.5 * (np.log(nu1 / nu2) / self.zeta) ** 2)

    def frequency_correlation_dlog(self, dlognu):
        return np.exp(-(dlognu**2) / (2 * self.zeta**2))

    def frequency_covariance(self, nu1, nu2):
        sigma2 = (self.frequency_variance(nu1) * self.frequency_variance(nu2)) ** 0.5
        return sigma2 * self.frequency_correlation(nu1, nu2)


def _sck_model(name, A, alpha, beta, zeta):
    return type(name, (ForegroundSCK,),
                {"A": A, "alpha": alpha, "beta": beta, "zeta": zeta, "__module__": __name__})


# Table 1 of Santos, Cooray & Knox as used by the reference (gaussianfg.py:133-158)
Synchrotron = _sck_model("Synchrotron", 7.00e-4, 2.80, 2.4, 4.0)
ExtraGalacticFreeFree = _sck_model("ExtraGalacticFreeFree", 1.40e-8, 2.10, 1.0, 35.0)
GalacticFreeFree = _sck_model("GalacticFreeFree", 8.80e-8, 2.15, 3.0, 35.0)
PointSources = _sck_model("PointSources", 5.70e-5, 2.07, 1.1, 1.0)
