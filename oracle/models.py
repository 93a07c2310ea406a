"""C_l(nu,nu') models: numpy restatement (oracle; test infrastructure only).

Follows, relative to /root/reference:
  * cora/util/cosmology.py:63-94,156-210,404-430   (Cosmology.H, comoving_distance)
  * cora/util/cubicspline.pyx:126-231,254-288       (natural cubic spline, log variant)
  * cora/util/bilinearmap.pyx:8-59                  (clipped bilinear table lookup)
  * cora/signal/corr.py:891-982                     (flat-sky FFT C_l table method)
  * cora/signal/corr21cm.py:19-35,37-62,106-208     (21cm T_b, Pade growth, nu->z)
  * cora/foreground/gaussianfg.py:40-41,107-130     (separable SCK foregrounds)
  * cora/foreground/galaxy.py:20-27, cora/foreground/pointsource.py:541-546 (parameters)
"""
import math
import os

import numpy as np
import scipy.fftpack
from scipy import integrate as si

# caput.astro.constants values used on the path (caput itself is absent here)
C_LIGHT = 299792458.0
NU21 = 1420.40575177
MEGA_PARSEC = 3.08568025e22

_PS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cora_amd", "signal", "data",
                        "ps_z1.5.dat")


# ---------------------------------------------------------------- cosmology
class Cosmology:
    """cora/util/cosmology.py:21-94 defaults (Planck 2018), units='cosmo'."""

    def __init__(self, omega_b=0.04897, omega_c=0.26067, omega_l=0.69036, H0=67.66,
                 omega_g=0.0, omega_n=0.0, w_0=-1.0, w_a=0.0):
        self.omega_b, self.omega_c, self.omega_l = omega_b, omega_c, omega_l
        self.omega_g, self.omega_n, self.H0, self.w_0, self.w_a = omega_g, omega_n, H0, w_0, w_a

    @property
    def omega_m(self):
        return self.omega_b + self.omega_c

    @property
    def omega_r(self):
        return self.omega_g + self.omega_n

    @property
    def omega_k(self):
        return 1.0 - (self.omega_l + self.omega_b + self.omega_c + self.omega_g + self.omega_n)

    def H(self, z=0.0):
        """cosmology.py:156-188 (SI units, 1/s)."""
        H = self.H0 * (
            self.omega_r * (1 + z) ** 4
            + self.omega_m * (1 + z) ** 3
            + self.omega_k * (1 + z) ** 2
            + self.omega_l * (1 + z) ** (3 * (1 + self.w_0 + self.w_a)) * np.exp(-3 * self.w_a * z / (1 + z))
        ) ** 0.5
        return H * 1000.0 / MEGA_PARSEC

    def comoving_distance(self, z):
        """cosmology.py:190-210 + _intf_0_z :404-430 (odeint over sorted z), Mpc/h."""
        z = np.asarray(z, dtype=np.float64)
        scalar = z.ndim == 0
        zz = np.atleast_1d(z)

        def f(z1):
            return C_LIGHT / self.H(z1)

        x = np.zeros_like(zz)
        sort_ind = np.argsort(zz, axis=None)
        za = np.insert(zz.ravel()[sort_ind], 0, 0)
        x.ravel()[sort_ind] = si.odeint(lambda y, t: f(t), 0.0, za)[1:, 0]
        x = x / (MEGA_PARSEC / (self.H0 / 100.0))
        return x[0] if scalar else x


# ---------------------------------------------------------------- cubic spline
class Interpolater:
    """Natural cubic spline, cubicspline.pyx:38-231 (linear extrapolation outside)."""

    def __init__(self, x, y):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        n = len(self.x)
        length = n - 2
        x_, y_ = self.x, self.y
        al, bt, gm, f = (np.zeros(length) for _ in range(4))
        for i in range(length):
            f[i] = (y_[i + 2] - y_[i + 1]) / (x_[i + 2] - x_[i + 1]) - (y_[i + 1] - y_[i]) / (x_[i + 1] - x_[i])
            al[i] = (x_[i + 2] - x_[i]) / 3
            if i != 0:
                bt[i] = (x_[i + 1] - x_[i]) / 6
            if i != n - 3:
                gm[i] = (x_[i + 2] - x_[i + 1]) / 6
        l, m, zz = np.zeros(length), np.zeros(length), np.zeros(length)
        l[0] = al[0]
        m[0] = gm[0] / al[0]
        for i in range(1, length):
            l[i] = al[i] - bt[i] * m[i - 1]
            m[i] = gm[i] / l[i]
        zz[0] = f[0] / l[0]
        for i in range(1, length):
            zz[i] = (f[i] - bt[i] * zz[i - 1]) / l[i]
        for i in range(n - 4, -1, -1):
            zz[i] = zz[i] - m[i] * zz[i + 1]
        self.y2 = np.zeros(n)
        self.y2[1 : length + 1] = zz

    def __call__(self, xv):
        xv = np.asarray(xv, dtype=np.float64)
        x, y, y2 = self.x, self.y, self.y2
        n = len(x)
        out = np.empty_like(xv)
        lo = xv < x[0]
        hi = xv >= x[n - 1]
        mid = ~(lo | hi)
        # below range: cubicspline.pyx:144-148
        h = x[1] - x[0]
        a = (y[1] - y[0]) / h
        out[lo] = (a - h * y2[1] / 6) * (xv[lo] - x[0]) + y[0]
        # above range: cubicspline.pyx:150-155
        h = x[n - 1] - x[n - 2]
        a = (y[n - 1] - y[n - 2]) / h
        out[hi] = (a + h * y2[n - 2] / 6) * (xv[hi] - x[n - 1]) + y[n - 1]
        # interior: bisection -> kl with x[kl] <= xv < x[kh]
        xm = xv[mid]
        kh = np.searchsorted(x, xm, side="right")
        kl = kh - 1
        h = x[kh] - x[kl]
        a = (x[kh] - xm) / h
        b = (xm - x[kl]) / h
        c = (a**3 - a) * h**2 / 6
        d = (b**3 - b) * h**2 / 6
        out[mid] = a * y[kl] + b * y[kh] + c * y2[kl] + d * y2[kh]
        return out


class LogInterpolater(Interpolater):
    """cubicspline.pyx:254-288: spline of log(data), exp(spline(log x))."""

    def __init__(self, x, y):
        super().__init__(np.log(x), np.log(y))

    def __call__(self, xv):
        return np.exp(super().__call__(np.log(xv)))


# ---------------------------------------------------------------- bilinear
def bilinear_interp_numpy(arr, x, y):
    """bilinearmap.pyx:14-59 in numpy."""
    nx, ny = arr.shape
    xx = np.clip(x, 0.0, nx - 1e-5)
    yy = np.clip(y, 0.0, ny - 1e-5)
    x0 = xx.astype(np.int64)
    y0 = yy.astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    wa = (x1 - xx) * (y1 - yy)
    wb = (x1 - xx) * (yy - y0)
    wc = (xx - x0) * (y1 - yy)
    wd = (xx - x0) * (yy - y0)
    return wa * arr[x0, y0] + wb * arr[x0, y1] + wc * arr[x1, y0] + wd * arr[x1, y1]


def bilinear_interp(arr, x, y):
    """bilinearmap.pyx:14-59 in C/OpenMP over points (as the reference's prange), falling back to numpy for
    non-contiguous tables.  Identical arithmetic: the two agree bit for bit (tests/test_oracle.py)."""
    import ctypes

    from . import sht

    if not (isinstance(arr, np.ndarray) and arr.flags["C_CONTIGUOUS"] and arr.dtype == np.float64):
        return bilinear_interp_numpy(arr, x, y)
    x, y = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
    xr, yr = np.ascontiguousarray(x).ravel(), np.ascontiguousarray(y).ravel()
    out = np.empty(xr.size)
    lib = sht._load()
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_bilinear_interp(arr.ctypes.data_as(dp), ctypes.c_long(arr.shape[0]), ctypes.c_long(arr.shape[1]),
                               xr.ctypes.data_as(dp), yr.ctypes.data_as(dp), ctypes.c_long(xr.size),
                               out.ctypes.data_as(dp))
    return out.reshape(x.shape)


# ---------------------------------------------------------------- 21cm model
KPERPMIN, KPERPMAX, NKPERP, KPARMAX, NKPAR = 1e-4, 40.0, 500, 20.0, 32768


class Corr21cm:
    """The 21cm angular power spectrum (flat-sky FFT table method)."""

    _kstar = 5.0
    ps_redshift = 1.5
    bias = 1.0

    def __init__(self, cosmology=None, nkpar=NKPAR, nkperp=NKPERP):
        self.cosmology = cosmology if cosmology is not None else Cosmology()
        d = np.loadtxt(_PS_FILE)
        self._c1 = LogInterpolater(d[:, 0], d[:, 1])
        self.nkpar, self.nkperp = nkpar, nkperp
        self._tables = None

    def ps_vv(self, k):
        """corr21cm.py:24-29."""
        k = np.asarray(k, dtype=np.float64)
        return np.exp(-0.5 * k**2 / self._kstar**2) * self._c1(k)

    def T_b(self, z):
        """corr21cm.py:37-62, omega_HI :70-87."""
        c = self.cosmology
        return (3.9e-4 * ((c.omega_m + c.omega_l * (1 + z) ** -3) / 0.29) ** -0.5
                * ((1.0 + z) / 2.5) ** 0.5 * (6.2e-4 / 1e-3))

    def growth_factor(self, z):
        """corr21cm.py:109-139."""
        x = ((1.0 / self.cosmology.omega_m) - 1.0) / (1.0 + z) ** 3
        num = 1.0 + 1.175 * x + 0.3064 * x**2 + 0.005355 * x**3
        den = 1.0 + 1.857 * x + 1.021 * x**2 + 0.1530 * x**3
        return (1.0 + x) ** 0.5 / (1.0 + z) * num / den

    def growth_rate(self, z):
        """corr21cm.py:141-175."""
        x = ((1.0 / self.cosmology.omega_m) - 1.0) / (1.0 + z) ** 3
        dnum = 3.0 * x * (1.175 + 0.6127 * x + 0.01607 * x**2)
        dden = 3.0 * x * (1.857 + 2.042 * x + 0.4590 * x**2)
        num = 1.0 + 1.175 * x + 0.3064 * x**2 + 0.005355 * x**3
        den = 1.0 + 1.857 * x + 1.021 * x**2 + 0.1530 * x**3
        return 1.0 + 1.5 * x / (1.0 + x) + dnum / num - dden / den

    def tables(self):
        """corr.py:909-942: dd, dv, vv tables = DCT-I along k_par."""
        if self._tables is None:
            kperp = np.logspace(np.log10(KPERPMIN), np.log10(KPERPMAX), self.nkperp)[:, np.newaxis]
            kpar = np.linspace(0, KPARMAX, self.nkpar)[np.newaxis, :]
            k = (kpar**2 + kperp**2) ** 0.5
            mu2 = kpar**2 / k**2
            dd = self.ps_vv(k) * np.sinc(kpar * 0.0 / (2 * np.pi)) ** 2
            dv = dd * mu2
            vv = dd * mu2**2
            s = KPARMAX / (2 * self.nkpar)
            self._tables = tuple(scipy.fftpack.dct(t, type=1) * s for t in (dd, dv, vv))
        return self._tables

    def aps_z(self, la, za1, za2):
        """corr.py:944-982."""
        dd, dv, vv = self.tables()
        xa1 = self.cosmology.comoving_distance(za1)
        xa2 = self.cosmology.comoving_distance(za2)
        b1, b2 = self.bias * np.ones_like(za1), self.bias * np.ones_like(za2)
        f1, f2 = self.growth_rate(za1), self.growth_rate(za2)
        pf1, pf2 = self.T_b(za1), self.T_b(za2)
        D1 = self.growth_factor(za1) / self.growth_factor(self.ps_redshift)
        D2 = self.growth_factor(za2) / self.growth_factor(self.ps_redshift)
        xc = 0.5 * (xa1 + xa2)
        rpar = np.abs(xa2 - xa1)
        la = np.where(la == 0.0, 1e-10, la)
        x = (np.log10(la) - np.log10(xc * KPERPMIN)) / np.log10(KPERPMAX / KPERPMIN) * (self.nkperp - 1)
        y = rpar / (math.pi / KPARMAX)
        x, y = np.broadcast_arrays(x, y)
        psdd = bilinear_interp(dd, x, y)
        psdv = bilinear_interp(dv, x, y)
        psvv = bilinear_interp(vv, x, y)
        return (D1 * D2 * pf1 * pf2 / (xc**2 * np.pi)) * (
            (b1 * b2) * psdd + (f1 * b2 + f2 * b1) * psdv + (f1 * f2) * psvv)

    def angular_powerspectrum(self, l, nu1, nu2):
        """corr21cm.py:183-208 (frequencies in MHz)."""
        l, nu1, nu2 = (np.asarray(v, dtype=np.float64) for v in (l, nu1, nu2))
        return self.aps_z(l, NU21 / nu1 - 1.0, NU21 / nu2 - 1.0)


class EoR21cm(Corr21cm):
    """corr21cm.py:333-385: the 21cm model with reionisation-epoch parameters (T_b of Santos et al. 2009, bias 3).
    ``share`` = a Corr21cm whose lookup tables are reused (they depend on P(k) only)."""

    bias = 3.0

    def __init__(self, cosmology=None, share=None, **kw):
        Corr21cm.__init__(self, cosmology=cosmology, **kw)
        if share is not None:
            self._tables = share.tables()

    def T_b(self, z):
        """corr21cm.py:334-360."""
        c = self.cosmology
        return (23e-3 * (c.omega_b * (c.H0 / 100.0) ** 2 / 0.02)
                * (0.15 / (c.omega_m * (c.H0 / 100.0) ** 2) * ((1.0 + z) / 10)) ** 0.5
                * ((c.H0 / 100.0) / 0.7) ** -1)


# ---------------------------------------------------------------- foregrounds
class ForegroundSCK:
    """gaussianfg.py:87-130.  C_l = A (l/l0)^-beta (nu1 nu2/nu0^2)^-alpha exp(-(ln(nu1/nu2)/zeta)^2/2), C_0=0."""

    A, alpha, beta, zeta, nu_0, l_0 = 1.0, 0.0, 0.0, 1.0, 130.0, 1000.0

    def angular_ps(self, larray):
        larray = np.array(larray, dtype=np.float64)  # copy (reference mutates in place, :108-110)
        mask0 = larray == 0
        larray[mask0] = 1.0
        ps = self.A * (larray / self.l_0) ** (-self.beta)
        ps[mask0] = 0.0
        return ps

    def frequency_covariance(self, nu1, nu2):
        var = lambda nu: (nu / self.nu_0) ** (-2 * self.alpha)
        return (var(nu1) * var(nu2)) ** 0.5 * np.exp(-0.5 * (np.log(nu1 / nu2) / self.zeta) ** 2)

    def angular_powerspectrum(self, l, nu1, nu2):
        nu1, nu2 = np.asarray(nu1, dtype=np.float64), np.asarray(nu2, dtype=np.float64)
        return self.angular_ps(l) * self.frequency_covariance(nu1, nu2)


class FullSkySynchrotron(ForegroundSCK):
    """galaxy.py:20-27 over gaussianfg.py:188-193."""

    A, alpha, beta, zeta, nu_0, l_0 = 6.6e-3, 2.80, 2.8, 4.0, 408.0, 100.0


class UnresolvedBackground(ForegroundSCK):
    """pointsource.py:541-546 over gaussianfg.py:209-213."""

    A, alpha, beta, zeta, nu_0, l_0 = 3.55e-5, 2.07, 1.1, 1.0, 408.0, 100.0
