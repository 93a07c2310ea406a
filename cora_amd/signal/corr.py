"""Counterpart of the hot-path part of cora/signal/corr.py: the flat-sky FFT table model
of C_l(z, z') (``RedshiftCorrelation.angular_powerspectrum_fft``, corr.py:891-982).

Setup (host, one-off per instance, exactly the reference's recipe corr.py:909-942):
P(k) on a [500 log k_perp] x [32768 linear k_par] grid, times {1, mu^2, mu^4}, DCT-I along
k_par -> three [500, 32768] tables, uploaded once to HBM (393 MB) and cached.
Evaluation (device): K1 in csrc/clarray.hip, either fused with the Romberg channel
average (through ``skysim.clarray``) or at arbitrary broadcast points (calling
``angular_powerspectrum`` directly).

Everything else in the reference's corr.py (flat-sky cubes, multipole correlation
functions, ``angular_powerspectrum_full``) is outside this package's scope.
"""
import math

import numpy as np
import scipy.fftpack

from .. import _lib
from ..util.cosmology import Cosmology


class RedshiftCorrelation(object):
    r"""Redshift-space correlations in the linear regime: the C_l(z, z') table model.

    Parameters
    ----------
    ps_vv : function
        Velocity power spectrum P(k) [k in h/Mpc] at ``redshift``.
    ps_dd, ps_dv : function, optional
        Accepted for interface compatibility; the table model uses ``ps_vv`` with ``bias``.
    redshift : scalar, optional
        Redshift at which the power spectra are calculated.
    bias : scalar, optional
        Bias between the observable and the velocities.
    """

    ps_vv = None
    ps_dd = None
    ps_dv = None
    ps_2d = False
    ps_redshift = 0.0
    bias = 1.0
    _vv_only = False

    kperpmin = 1e-4
    kperpmax = 40.0
    nkperp = 500
    kparmax = 20.0
    nkpar = 32768
    _freq_window = 0.0
    _aps_cache = False

    def __init__(self, ps_vv=None, ps_dd=None, ps_dv=None, redshift=0.0, bias=1.0):
        self.ps_vv = ps_vv
        self.ps_dd = ps_dd
        self.ps_dv = ps_dv
        self.ps_redshift = redshift
        self.bias = bias
        self._vv_only = False if ps_dd and ps_dv else True
        self.cosmology = Cosmology()
        self._dev_tables = {}

    # ---- model hooks (corr.py:448-530) --------------------------------------------------
    def bias_z(self, z):
        return self.bias * np.ones_like(z)

    def growth_factor(self, z):
        return 1.0 / (1.0 + z)

    def growth_rate(self, z):
        return 1.0 * np.ones_like(z)

    def prefactor(self, z):
        return 1.0 * np.ones_like(z)

    def mean(self, z):
        return np.ones_like(z) * 0.0

    # ---- table build / cache (corr.py:870-887, 909-942) ---------------------------------
    def _build_tables(self):
        kperp = np.logspace(np.log10(self.kperpmin), np.log10(self.kperpmax), self.nkperp)[:, np.newaxis]
        kpar = np.linspace(0, self.kparmax, self.nkpar)[np.newaxis, :]
        k = (kpar**2 + kperp**2) ** 0.5
        mu2 = kpar**2 / k**2
        window = np.sinc(kpar * self._freq_window / (2 * np.pi)) ** 2
        dd = (self.ps_vv(k, kpar / k) if self.ps_2d else self.ps_vv(k)) * window
        norm = self.kparmax / (2 * self.nkpar)
        self._aps_dd = scipy.fftpack.dct(dd, type=1) * norm
        self._aps_dv = scipy.fftpack.dct(dd * mu2, type=1) * norm
        self._aps_vv = scipy.fftpack.dct(dd * mu2**2, type=1) * norm
        self._aps_cache = True
        self._dev_tables = {}

    def save_fft_cache(self, fname):
        """Save the three lookup tables (corr.py:870-877)."""
        if not self._aps_cache:
            self._build_tables()
        np.savez(fname, dd=self._aps_dd, dv=self._aps_dv, vv=self._aps_vv)

    def load_fft_cache(self, fname):
        """Load lookup tables saved by :meth:`save_fft_cache` (corr.py:879-887)."""
        a = np.load(fname)
        self._aps_dd, self._aps_dv, self._aps_vv = a["dd"], a["dv"], a["vv"]
        self.nkperp, self.nkpar = self._aps_dd.shape
        self._aps_cache = True
        self._dev_tables = {}

    def _tables_on(self, ctx):
        if not self._aps_cache:
            self._build_tables()
        key = ctx.device.index
        if key not in self._dev_tables:
            self._dev_tables[key] = tuple(ctx.to_device(t) for t in (self._aps_dd, self._aps_dv, self._aps_vv))
        return self._dev_tables[key]

    # ---- per-redshift quantities (corr.py:944-951) --------------------------------------
    def _z_quantities(self, za):
        za = np.asarray(za, dtype=np.float64)
        chi = self.cosmology.comoving_distance(za)
        pfd = self.prefactor(za) * self.growth_factor(za) / self.growth_factor(self.ps_redshift)
        return chi, pfd, self.growth_rate(za) * np.ones_like(za), self.bias_z(za) * np.ones_like(za)

    def _table_plan(self, za_to_z):
        def prepare(ctx, za):
            dd, dv, vv = self._tables_on(ctx)
            chi, pfd, f, b = self._z_quantities(za_to_z(za))
            return dict(dd=dd, dv=dv, vv=vv, kperpmin=self.kperpmin, kperpmax=self.kperpmax,
                        kparmax=self.kparmax, chi=chi, pfd=pfd, f=f, b=b)

        return dict(kind="table21cm", prepare=prepare)

    def _clarray_plan(self, aps):
        """Protocol used by ``skysim.clarray`` to recognise the table model."""
        return self._table_plan(lambda z: z)

    # ---- the aps callable ------------------------------------------------------------------
    def angular_powerspectrum_fft(self, la, za1, za2):
        """C_l(z1, z2) in the flat-sky limit, at broadcast points (corr.py:891-982)."""
        import torch

        ctx = _lib.get_context()
        dd, dv, vv = self._tables_on(ctx)
        la, za1, za2 = (np.asarray(v, dtype=np.float64) for v in (la, za1, za2))
        shape = np.broadcast(la, za1, za2).shape
        # per-redshift work on the unique redshifts only (chi is an ODE solve)
        zu, inv = np.unique(np.concatenate([za1.ravel(), za2.ravel()]), return_inverse=True)
        chi, pfd, f, b = self._z_quantities(zu)
        i1 = inv[: za1.size].reshape(za1.shape)
        i2 = inv[za1.size :].reshape(za2.shape)
        la = np.where(la == 0.0, 1e-10, la)

        def full(a):
            return np.array(np.broadcast_to(a, shape), dtype=np.float64, order="C").ravel()

        P = pfd[i1] * pfd[i2]
        args = [np.log10(la), chi[i1], chi[i2], b[i1] * b[i2] * P, (f[i1] * b[i2] + f[i2] * b[i1]) * P,
                f[i1] * f[i2] * P]
        dev = [torch.from_numpy(full(a)).to(ctx.device) for a in args]
        out = ctx.aps_table21cm_points(dd, dv, vv, self.kperpmin, self.kperpmax, self.kparmax, *dev)
        res = out.cpu().numpy().reshape(shape)
        return res if res.ndim else float(res)

    angular_powerspectrum = angular_powerspectrum_fft
