"""Natural cubic spline (counterpart of cora/util/cubicspline.pyx:38-288).

Setup-stage code: it tabulates P(k) once per 21cm model instance.  Same conventions as
the reference's Cython class: natural end conditions, bisection lookup, LINEAR
extrapolation with the end slope outside the knots (:144-155), log variant (:254-288).
"""
import numpy as np


class InterpolationException(Exception):
    """Exceptions in the Interpolation module."""


class Interpolater:
    @classmethod
    def fromfile(cls, file, colspec=None):
        if colspec is None:
            colspec = (0, 1)
        if len(colspec) != 2:
            raise InterpolationException("Can only use two columns of a file.")
        return cls(np.loadtxt(file, usecols=colspec))

    def __init__(self, data1, data2=None):
        if data2 is None:
            data = np.asarray(data1)
        else:
            try:
                data = np.dstack((data1, data2))[0]
            except ValueError as e:
                raise InterpolationException("Failure stacking x and y data.") from e
        s = data.shape
        if len(s) != 2:
            raise InterpolationException("Array must be 2d.")
        if s[1] != 2:
            raise InterpolationException("Array must consist of X-Y pairs.")
        if s[0] < 4:
            raise InterpolationException("Cubic spline interpolation requires at least 4 points.")
        if np.isinf(data).any() or np.isnan(data).any():
            raise InterpolationException("Some values invalid.")
        self._data = np.ascontiguousarray(data).astype(np.float64, copy=False)
        self._gen_spline()

    def _gen_spline(self):
        """Second derivatives from the tridiagonal system (cubicspline.pyx:179-231)."""
        x, y = self._data[:, 0], self._data[:, 1]
        n = len(x)
        dx = np.diff(x)
        slope = np.diff(y) / dx
        rhs = slope[1:] - slope[:-1]          # length n-2
        diag = (x[2:] - x[:-2]) / 3
        lower = dx[1:-1] / 6                  # couples row i to i-1, i = 1..n-3
        upper = dx[1:-1] / 6                  # couples row i to i+1, i = 0..n-4
        m = n - 2
        lfac = np.empty(m)
        mu = np.zeros(m)
        zz = np.empty(m)
        lfac[0] = diag[0]
        mu[0] = (upper[0] / diag[0]) if m > 1 else 0.0
        zz[0] = rhs[0] / lfac[0]
        for i in range(1, m):
            lfac[i] = diag[i] - lower[i - 1] * mu[i - 1]
            mu[i] = (upper[i] / lfac[i]) if i < m - 1 else 0.0
            zz[i] = (rhs[i] - lower[i - 1] * zz[i - 1]) / lfac[i]
        for i in range(m - 2, -1, -1):
            zz[i] -= mu[i] * zz[i + 1]
        self._y2 = np.zeros(n)
        self._y2[1:-1] = zz

    def data(self):
        return (self._data, self._y2)

    def value(self, x):
        scalar = not isinstance(x, np.ndarray)
        r = self._eval(np.atleast_1d(np.asarray(x, dtype=np.float64)))
        return float(r[0]) if scalar else r

    __call__ = value

    def value_array(self, x):
        """Spline at every element of an array, same shape (cubicspline.pyx:107-123)."""
        x = np.asarray(x, dtype=np.float64)
        return self._eval(np.ravel(x, order="C")).reshape(x.shape)

    def _eval(self, xv):
        xs, ys, y2 = self._data[:, 0], self._data[:, 1], self._y2
        n = len(xs)
        out = np.empty_like(xv)
        below = xv < xs[0]
        above = xv >= xs[-1]
        inside = ~(below | above)
        h0 = xs[1] - xs[0]
        out[below] = ((ys[1] - ys[0]) / h0 - h0 * y2[1] / 6) * (xv[below] - xs[0]) + ys[0]
        h1 = xs[-1] - xs[-2]
        out[above] = ((ys[-1] - ys[-2]) / h1 + h1 * y2[-2] / 6) * (xv[above] - xs[-1]) + ys[-1]
        xi = xv[inside]
        hi = np.searchsorted(xs, xi, side="right")
        lo = hi - 1
        h = xs[hi] - xs[lo]
        a = (xs[hi] - xi) / h
        b = (xi - xs[lo]) / h
        out[inside] = (a * ys[lo] + b * ys[hi] + (a**3 - a) * h**2 / 6 * y2[lo] + (b**3 - b) * h**2 / 6 * y2[hi])
        return out

    def _device_spline(self):
        """(kind, knots x, y, y'', x_t, f_t) for the device evaluation of this interpolater
        (``corahip_xi_table_average``; kind 0 = plain, 1 = log-log, 2 = sinh)."""
        return 0, self._data[:, 0].copy(), self._data[:, 1].copy(), self._y2.copy(), 1.0, 1.0

    def test(self, min, max, samp):
        h = 1.0 * (max - min) / samp
        xs = min + h * np.arange(samp)
        return np.stack([xs, self._eval(xs)], axis=1)


class LogInterpolater(Interpolater):
    """Cubic spline in log-log space (cubicspline.pyx:254-288)."""

    def __init__(self, data):
        data = np.asarray(data)
        if np.any(data <= 0):
            raise InterpolationException("Data must be non-negative.")
        Interpolater.__init__(self, np.log(data))

    def value(self, x):
        scalar = not isinstance(x, np.ndarray)
        r = np.exp(self._eval(np.log(np.atleast_1d(np.asarray(x, dtype=np.float64)))))
        return float(r[0]) if scalar else r

    __call__ = value

    def value_log_array(self, x):
        """exp(spline(log x)) at every element of an array (cubicspline.pyx:273-288)."""
        x = np.asarray(x, dtype=np.float64)
        return np.exp(self._eval(np.log(np.ravel(x, order="C")))).reshape(x.shape)

    def _device_spline(self):
        return 1, self._data[:, 0].copy(), self._data[:, 1].copy(), self._y2.copy(), 1.0, 1.0


class SinhInterpolater(Interpolater):
    """Cubic spline in asinh-scaled space, ``f_t sinh(spline(asinh(x / x_t)))`` (cubicspline.pyx:290-345): log-like
    for |values| above the thresholds, linear below, so zero and negative values are allowed."""

    def __init__(self, data, x_t, f_t):
        self.x_t = float(x_t)
        self.f_t = float(f_t)
        Interpolater.__init__(self, np.arcsinh(np.asarray(data, dtype=np.float64) / np.array([x_t, f_t], dtype=np.float64)))

    def value(self, x):
        scalar = not isinstance(x, np.ndarray)
        r = self.f_t * np.sinh(self._eval(np.arcsinh(np.atleast_1d(np.asarray(x, dtype=np.float64)) / self.x_t)))
        return float(r[0]) if scalar else r

    __call__ = value

    def value_sinh_array(self, x):
        x = np.asarray(x, dtype=np.float64)
        return (self.f_t * np.sinh(self._eval(np.arcsinh(np.ravel(x, order="C") / self.x_t)))).reshape(x.shape)

    def _device_spline(self):
        return 2, self._data[:, 0].copy(), self._data[:, 1].copy(), self._y2.copy(), self.x_t, self.f_t
