"""Oracle (test infrastructure only): numpy restatement of the xi(r) -> C_l(chi, chi') integrator
(cora/signal/corrfunc.py:265-400), pinned by outputs of the reference itself
(tests/golden/make_golden_corrfunc.py -> corrfunc_vectors.npz).

    C_l(x_i, x_j) = 2 pi  int_-1^1 dmu  P_l(mu)  < xi( r(mu, x, x') ) >_bins ,   r^2 = x^2 + x'^2 - 2 x x' mu

with Gauss-Legendre quadrature in mu (M = q lmax nodes) and, for xromb > 0, Gauss-Legendre averaging of xi over the
radial bins (2^xromb + 1 nodes per bin, half-widths from neighbouring distances or xwidth / 2).
`cosine_rule` is a dependency (caput) that is absent: restated, parity unpinned for that one expression.
"""
import numpy as np
import scipy.special as ss


def cosine_rule(mu, x1, x2):
    """r(mu, x1, x2) broadcast to [mu, x1, x2] (what corrfunc.py:372 takes from caput.astro.coordinates)."""
    mu = np.asarray(mu)[:, None, None]
    a = np.asarray(x1)[None, :, None]
    b = np.asarray(x2)[None, None, :]
    return np.sqrt((a - b) ** 2 + 2.0 * a * b * (1.0 - mu))


def legendre_array(lmax, mu):
    """P_l(mu), l = 0..lmax: [lmax+1, len(mu)] (corrfunc.py:265-287) by the three-term recurrence."""
    mu = np.asarray(mu, dtype=np.float64)
    out = np.empty((lmax + 1, mu.size))
    out[0] = 1.0
    if lmax >= 1:
        out[1] = mu
    for l in range(2, lmax + 1):
        out[l] = ((2 * l - 1) * mu * out[l - 1] - (l - 1) * out[l - 2]) / l
    return out


def radial_nodes(xarray, xromb, xwidth=None):
    """Sub-sample distances [F * xint] and normalised weights [xint] of the radial-bin average (corrfunc.py:337-361)."""
    xarray = np.asarray(xarray, dtype=np.float64)
    if xromb <= 0:
        return xarray.copy(), np.ones(1), 1
    if xwidth is None:
        xhalf = np.empty_like(xarray)
        xhalf[0] = abs(xarray[1] - xarray[0]) / 2.0
        xhalf[1:] = np.abs(xarray[1:] - xarray[:-1]) / 2.0
    else:
        xhalf = np.full(xarray.shape, xwidth / 2.0)
    xint = 2**xromb + 1
    nodes, wts, wsum = ss.roots_legendre(xint, mu=True)
    return (xarray[:, None] + xhalf[:, None] * nodes).ravel(), wts / wsum, xint


def corr_to_clarray(corr, lmax, xarray, xromb=3, xwidth=None, q=2):
    """[lmax+1, F, F] (corrfunc.py:290-400; the chunking over mu there only bounds memory)."""
    xarray = np.asarray(xarray, dtype=np.float64)
    F = xarray.size
    M = q * lmax
    mu, w, wsum = ss.roots_legendre(M, mu=True)
    xa, xw, xint = radial_nodes(xarray, xromb, xwidth)
    xi = corr(cosine_rule(mu, xa, xa))                                  # [M, F xint, F xint]
    if xromb > 0:
        xi = xi.reshape(M, F, xint, F, xint)
        xi = np.einsum("mialb,a,b->mil", xi, xw, xw)                     # bin average over both distances
    lm = legendre_array(lmax, mu) * (w[None, :] * 4.0 * np.pi / wsum)
    return np.dot(lm, xi.reshape(M, F * F)).reshape(lmax + 1, F, F)
