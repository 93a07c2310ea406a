"""Spherical-harmonic synthesis oracle (test infrastructure only).

Restates what the reference obtains from ``healpy.alm2map(almp, nside)``
(cora/util/hputil.py:369-391) following the published HEALPix definition
(SURVEY.md Appendix A).  PARITY UNPINNED against healpy itself (absent here);
pinned against brute-force ``scipy.special.sph_harm_y`` sums in tests/.

Three implementations, slowest to fastest:
  * ``alm2map_bruteforce``  - independent: sum a_lm Y_lm via scipy (tiny sizes)
  * ``alm2map_numpy``       - scaled Legendre recurrence in numpy
  * ``alm2map``             - same recurrence in C/OpenMP (oracle/sht_ref.c)
"""
import ctypes
import os
import subprocess

import numpy as np

from . import healpix

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.oracle_legendre_synth.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp]
        _lib.oracle_legendre_synth.restype = None
        _lib.oracle_legendre_synth_blocked.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp]
        _lib.oracle_legendre_synth_blocked.restype = None
        _lib.oracle_legendre_anal.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp]
        _lib.oracle_legendre_anal.restype = None
        _lib.oracle_lambda_lm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp]
        _lib.oracle_lambda_lm.restype = None
        _lib.oracle_legendre_synth_spin2.argtypes = [ctypes.c_int, ctypes.c_int] + [dp] * 8
        _lib.oracle_legendre_synth_spin2.restype = None
        _lib.oracle_bilinear_interp.argtypes = [dp, ctypes.c_long, ctypes.c_long, dp, dp, ctypes.c_long, dp]
        _lib.oracle_bilinear_interp.restype = None
        _lib.oracle_ring_synth.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ctypes.POINTER(ctypes.c_long),
                                           ctypes.POINTER(ctypes.c_int), dp, dp]
        _lib.oracle_ring_synth.restype = None
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return int(_load().oracle_num_threads())


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def alm_index(l, m, lmax):
    """healpy packed index (cora/util/hputil.py:124-152 ordering)."""
    return m * (2 * lmax + 1 - m) // 2 + l


def lambda_lm(lmax, m, x):
    """lambda_lm(x) for l = m..lmax via the C recurrence."""
    out = np.zeros(lmax - m + 1)
    sth = np.sqrt((1.0 - x) * (1.0 + x))
    _load().oracle_lambda_lm(lmax, m, float(x), float(sth), _dp(out))
    return out


def _legendre_numpy(lmax, z, sth, alm):
    """F_m on north rings / mirrored south rings; numpy restatement."""
    L = lmax + 1
    npair = len(z)
    fn = np.zeros((npair, L), dtype=np.complex128)
    fs = np.zeros((npair, L), dtype=np.complex128)
    lp = np.empty(L)
    lp[0] = -0.5 * np.log2(4.0 * np.pi)
    for m in range(1, L):
        lp[m] = lp[m - 1] + 0.5 * np.log2((2.0 * m + 1.0) / (2.0 * m))
    l2s = np.log2(sth)
    for m in range(L):
        L2 = lp[m] + m * l2s
        sc = np.floor(L2).astype(np.int64)
        lam = np.exp2(L2 - sc) * (-1.0 if (m & 1) else 1.0)
        lam_prev = np.zeros(npair)
        inv_alpha_prev = 0.0
        fe = np.zeros(npair, dtype=np.complex128)
        fo = np.zeros(npair, dtype=np.complex128)
        for l in range(m, L):
            v = np.where(sc > -1000, np.ldexp(lam, np.maximum(sc, -1000)), 0.0)
            a = alm[alm_index(l, m, lmax)]
            if ((l - m) & 1) == 0:
                fe += a * v
            else:
                fo += a * v
            lp1 = l + 1.0
            alpha = np.sqrt((4.0 * lp1 * lp1 - 1.0) / (lp1 * lp1 - float(m) * m))
            nxt = alpha * (z * lam - lam_prev * inv_alpha_prev)
            lam_prev, lam = lam, nxt
            inv_alpha_prev = 1.0 / alpha
            big = np.abs(lam) > 2.0**300
            if big.any():
                lam = np.where(big, lam * 2.0**-300, lam)
                lam_prev = np.where(big, lam_prev * 2.0**-300, lam_prev)
                sc = np.where(big, sc + 300, sc)
        fn[:, m] = fe + fo
        fs[:, m] = fe - fo
    return fn, fs


def _legendre_c(lmax, z, sth, alm, blocked=False):
    L = lmax + 1
    npair = len(z)
    a = np.ascontiguousarray(alm, dtype=np.complex128).view(np.float64)
    fn = np.zeros((npair, L), dtype=np.complex128)
    fs = np.zeros((npair, L), dtype=np.complex128)
    zz = np.ascontiguousarray(z, dtype=np.float64)
    ss = np.ascontiguousarray(sth, dtype=np.float64)
    fun = _load().oracle_legendre_synth_blocked if blocked else _load().oracle_legendre_synth
    fun(lmax, npair, _dp(zz), _dp(ss), _dp(a), _dp(fn.view(np.float64)), _dp(fs.view(np.float64)))
    return fn, fs


def ring_synthesis(fm, nphi, phi0):
    """T_j (j < nphi) on one ring from F_m, m = 0..lmax: phase, alias fold, c2r FFT.

    T_j = Re(c_0) + 2 sum_{m>=1} Re(c_m e^{2 pi i j m/nphi}),  c_m = F_m e^{i m phi0}.
    Each m >= 1 adds c_m to bin (m mod nphi) and conj(c_m) to bin (-m mod nphi) of a
    Hermitian length-nphi spectrum X; T = nphi * irfft(X[:nphi/2+1]).
    """
    L = len(fm)
    m = np.arange(L)
    c = fm * np.exp(1j * m * phi0)
    X = np.zeros(nphi, dtype=np.complex128)
    X[0] = c[0].real
    if L > 1:
        np.add.at(X, m[1:] % nphi, c[1:])
        np.add.at(X, (-m[1:]) % nphi, np.conj(c[1:]))
    return np.fft.irfft(X[: nphi // 2 + 1], n=nphi) * nphi


def synth_from_fm_c(fn, fs, nside):
    """synth_from_fm in C/OpenMP (oracle_ring_synth in sht_ref.c: phase, alias fold, radix-2 / Bluestein inverse DFT per
    ring, rings in parallel).  Same arithmetic; used where the oracle is TIMED (bench.py's cpu_baseline) so that the
    baseline runs on every host core instead of a Python loop over 4 nside rings."""
    ri = healpix.ring_info(nside)
    lmax = fn.shape[1] - 1
    out = np.empty(healpix.nside2npix(nside))
    a = np.ascontiguousarray(fn, dtype=np.complex128).view(np.float64)
    b = np.ascontiguousarray(fs, dtype=np.complex128).view(np.float64)
    start = np.ascontiguousarray(ri["start"], dtype=np.int64)
    nphi = np.ascontiguousarray(ri["nphi"], dtype=np.int32)
    phi0 = np.ascontiguousarray(ri["phi0"], dtype=np.float64)
    _load().oracle_ring_synth(nside, lmax, _dp(a), _dp(b), start.ctypes.data_as(ctypes.POINTER(ctypes.c_long)),
                              nphi.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(phi0), _dp(out))
    return out


def synth_from_fm(fn, fs, nside):
    """Assemble the RING map from north/south F_m arrays ([2 nside][lmax+1])."""
    ri = healpix.ring_info(nside)
    npix = healpix.nside2npix(nside)
    out = np.empty(npix)
    nring = 4 * nside - 1
    npair = 2 * nside
    for r in range(npair):
        n, s, p0 = int(ri["nphi"][r]), int(ri["start"][r]), float(ri["phi0"][r])
        out[s : s + n] = ring_synthesis(fn[r], n, p0)
        rs = nring - 1 - r
        if rs != r:
            n, s, p0 = int(ri["nphi"][rs]), int(ri["start"][rs]), float(ri["phi0"][rs])
            out[s : s + n] = ring_synthesis(fs[r], n, p0)
    return out


def alm2map(alm, nside, lmax=None, impl="c", rings_c=False):
    """Packed (healpy-ordered) alm -> RING map, float64.  ``rings_c``: the ring stage in C/OpenMP too
    (synth_from_fm_c) instead of the ring-by-ring numpy loop."""
    alm = np.asarray(alm, dtype=np.complex128)
    if lmax is None:
        # nalm = (lmax+1)(lmax+2)/2
        lmax = int(round((-3 + np.sqrt(1 + 8 * alm.size)) / 2))
    assert alm.size == (lmax + 1) * (lmax + 2) // 2
    ri = healpix.ring_info(nside)
    npair = 2 * nside
    z, sth = ri["z"][:npair], ri["sth"][:npair]
    if rings_c:      # the timed configuration: ring-blocked, vectorised Legendre + C ring stage
        fn, fs = _legendre_c(lmax, z, sth, alm, blocked=True)
        return synth_from_fm_c(fn, fs, nside)
    fn, fs = (_legendre_c if impl == "c" else _legendre_numpy)(lmax, z, sth, alm)
    return synth_from_fm(fn, fs, nside)


def alm2map_bruteforce(alm, nside, lmax):
    """Independent definition-level synthesis with scipy.special.sph_harm_y."""
    from scipy.special import sph_harm_y

    theta, phi = healpix.pix2ang_ring(nside)
    out = np.zeros(theta.size)
    for m in range(lmax + 1):
        cm = 1.0 if m == 0 else 2.0
        for l in range(m, lmax + 1):
            a = alm[alm_index(l, m, lmax)]
            y = sph_harm_y(l, m, theta, phi)
            if m == 0:
                out += a.real * y.real
            else:
                out += cm * (a * y).real
    return out


# ------------------------------------------------------------------------------------
# analysis (healpy.map2alm as the reference calls it: use_weights=True, iter=2;
# cora/util/hputil.py:46-47,195-234) - "next" row n1 of SURVEY 8(f)
# ------------------------------------------------------------------------------------
def ring_weights(nside, lmax_exact=None):
    """Ring quadrature weights w_r (north rings, equator included; the south mirrors them).

    healpy's use_weights=True multiplies every ring by (1 + w) read from the HEALPix data file
    weight_ring_n<nside>.fits - DATA that is absent here (healpy is not installed), so parity with
    those files is UNPINNED.  The defining property of ring weights is restated instead: the
    minimum-norm correction to uniform weights that integrates the zonal Legendre polynomials
    P_l(z), even l <= lmax_exact (default 3 nside, rounded down to even), exactly:
        sum_rings w_r n_r (4 pi / npix) P_l(z_r) = 4 pi delta_l0.
    """
    from numpy.polynomial import legendre as npleg

    ri = healpix.ring_info(nside)
    npair = 2 * nside
    z = ri["z"][:npair]
    cnt = ri["nphi"][:npair].astype(np.float64) * 2.0
    cnt[npair - 1] = ri["nphi"][npair - 1]        # the equator ring has no mirror
    if lmax_exact is None:
        lmax_exact = 3 * nside
    ls = np.arange(0, lmax_exact + 1, 2)
    # P_l(z_r) for the even l by the three-term recurrence
    P = np.empty((lmax_exact + 1, npair))
    P[0] = 1.0
    if lmax_exact >= 1:
        P[1] = z
    for l in range(2, lmax_exact + 1):
        P[l] = ((2 * l - 1) * z * P[l - 1] - (l - 1) * P[l - 2]) / l
    M = P[ls] * (cnt * 4.0 * np.pi / healpix.nside2npix(nside))[None, :]
    rhs = np.zeros(len(ls))
    rhs[0] = 4.0 * np.pi
    dw = np.linalg.lstsq(M, rhs - M @ np.ones(npair), rcond=None)[0]   # minimum-norm solution
    del npleg
    return 1.0 + dw


def ring_analysis(x, nphi, phi0, L):
    """G_m = sum_j x_j e^{-i m phi_j}, m = 0..L-1, phi_j = phi0 + 2 pi j / nphi (adjoint of ring_synthesis)."""
    X = np.fft.fft(x)
    m = np.arange(L)
    return X[m % nphi] * np.exp(-1j * m * phi0)


def anal_to_gm(hpmap, nside, lmax, ring_w=None):
    """Weighted ring spectra (gn, gs) [2 nside][lmax+1] of a RING map."""
    ri = healpix.ring_info(nside)
    nring = 4 * nside - 1
    npair = 2 * nside
    L = lmax + 1
    area = 4.0 * np.pi / healpix.nside2npix(nside)
    gn = np.zeros((npair, L), dtype=np.complex128)
    gs = np.zeros((npair, L), dtype=np.complex128)
    for r in range(npair):
        wr = area * (1.0 if ring_w is None else ring_w[r])
        n, s, p0 = int(ri["nphi"][r]), int(ri["start"][r]), float(ri["phi0"][r])
        gn[r] = wr * ring_analysis(hpmap[s : s + n], n, p0, L)
        rs = nring - 1 - r
        if rs != r:
            n, s, p0 = int(ri["nphi"][rs]), int(ri["start"][rs]), float(ri["phi0"][rs])
            gs[r] = wr * ring_analysis(hpmap[s : s + n], n, p0, L)
    return gn, gs


def map2alm_adjoint(hpmap, nside, lmax, ring_w=None):
    """One quadrature pass: a_lm = sum_pix w_ring(pix) (4 pi / npix) map(pix) conj(Y_lm(pix)), packed order."""
    ri = healpix.ring_info(nside)
    npair = 2 * nside
    gn, gs = anal_to_gm(np.asarray(hpmap, dtype=np.float64), nside, lmax, ring_w)
    alm = np.zeros((lmax + 1) * (lmax + 2) // 2, dtype=np.complex128)
    z = np.ascontiguousarray(ri["z"][:npair])
    sth = np.ascontiguousarray(ri["sth"][:npair])
    _load().oracle_legendre_anal(lmax, npair, _dp(z), _dp(sth), _dp(np.ascontiguousarray(gn).view(np.float64)),
                                 _dp(np.ascontiguousarray(gs).view(np.float64)), _dp(alm.view(np.float64)))
    return alm


def map2alm(hpmap, nside, lmax, use_weights=True, niter=2):
    """healpy.map2alm(map, lmax=lmax, use_weights=..., iter=niter): quadrature + Jacobi refinement
    alm <- alm + A(map - S alm)."""
    w = ring_weights(nside) if use_weights else None
    alm = map2alm_adjoint(hpmap, nside, lmax, w)
    for _ in range(niter):
        alm = alm + map2alm_adjoint(hpmap - alm2map(alm, nside, lmax), nside, lmax, w)
    return alm


def map2alm_bruteforce(hpmap, nside, lmax, ring_w=None):
    """Independent definition-level quadrature with scipy.special.sph_harm_y (tiny sizes)."""
    from scipy.special import sph_harm_y

    theta, phi = healpix.pix2ang_ring(nside)
    ri = healpix.ring_info(nside)
    nring = 4 * nside - 1
    wpix = np.empty(theta.size)
    for r in range(nring):
        n, s = int(ri["nphi"][r]), int(ri["start"][r])
        rr = min(r, nring - 1 - r)
        wpix[s : s + n] = 1.0 if ring_w is None else ring_w[rr]
    wpix *= 4.0 * np.pi / theta.size
    alm = np.zeros((lmax + 1) * (lmax + 2) // 2, dtype=np.complex128)
    for m in range(lmax + 1):
        for l in range(m, lmax + 1):
            alm[alm_index(l, m, lmax)] = np.sum(wpix * hpmap * np.conj(sph_harm_y(l, m, theta, phi)))
    return alm


# ------------------------------------------------------------------------------------
# polarisation (spin-2) synthesis: healpy.alm2map([T, E, B], nside) as hputil.sphtrans_inv_real_pol uses it
# (cora/util/hputil.py:394-432) - "next" row n4 of SURVEY 8(f).  Convention and formulas: oracle/sht_ref.c.
# ------------------------------------------------------------------------------------
def alm2map_spin2(alm_e, alm_b, nside, lmax):
    """Packed E and B coefficients -> (Q, U) RING maps."""
    ri = healpix.ring_info(nside)
    npair = 2 * nside
    L = lmax + 1
    z = np.ascontiguousarray(ri["z"][:npair])
    sth = np.ascontiguousarray(ri["sth"][:npair])
    ae = np.ascontiguousarray(alm_e, dtype=np.complex128).view(np.float64)
    ab = np.ascontiguousarray(alm_b, dtype=np.complex128).view(np.float64)
    outs = [np.zeros((npair, L), dtype=np.complex128) for _ in range(4)]
    _load().oracle_legendre_synth_spin2(lmax, npair, _dp(z), _dp(sth), _dp(ae), _dp(ab),
                                        *[_dp(o.view(np.float64)) for o in outs])
    qn, qs, un, us = outs
    return synth_from_fm(qn, qs, nside), synth_from_fm(un, us, nside)


def spin2_wx(l, m, theta):
    """(W_lm, X_lm)(theta) from the scalar lambda_lm via scipy (independent of the C recurrence)."""
    from scipy.special import sph_harm_y

    x, s2 = np.cos(theta), np.sin(theta) ** 2
    lam = sph_harm_y(l, m, theta, 0.0).real
    lam1 = sph_harm_y(l - 1, m, theta, 0.0).real if l - 1 >= m else 0.0 * lam
    if l < 2:
        return 0.0 * lam, 0.0 * lam
    N2 = 2.0 / np.sqrt((l + 2.0) * (l + 1.0) * l * (l - 1.0))
    c = np.sqrt((2.0 * l + 1.0) / (2.0 * l - 1.0) * (l - m) / (l + m)) if l + m > 0 else 0.0
    W = N2 * (-((l - m * m) / s2 + l * (l - 1.0) / 2.0) * lam + (l + m) * x / s2 * c * lam1)
    X = N2 * m / s2 * ((l - 1.0) * x * lam - (l + m) * c * lam1)
    return W, X


def alm2map_spin2_bruteforce(alm_e, alm_b, nside, lmax):
    """Definition-level (Q, U): Q +- iU = - sum (E +- iB) (W -+ X) e^{i m phi} over all m (negative m by the
    reality conditions a_{l,-m} = (-1)^m conj(a_lm), W_{l,-m} = (-1)^m W_lm, X_{l,-m} = -(-1)^m X_lm)."""
    theta, phi = healpix.pix2ang_ring(nside)
    qpu = np.zeros(theta.size, dtype=np.complex128)      # Q + iU
    for m in range(0, lmax + 1):
        for l in range(max(m, 2), lmax + 1):
            W, X = spin2_wx(l, m, theta)
            e, b = alm_e[alm_index(l, m, lmax)], alm_b[alm_index(l, m, lmax)]
            ph = np.exp(1j * m * phi)
            qpu += -(e + 1j * b) * (W - X) * ph
            if m > 0:      # the -m term: (E + iB)_{l,-m} = (-1)^m (conj E + i conj B), (W - X)_{l,-m} = (-1)^m (W + X)
                qpu += -(np.conj(e) + 1j * np.conj(b)) * (W + X) * np.conj(ph)
    return qpu.real, qpu.imag


# ------------------------------------------------------------------------------------
# polarisation (spin-2) ANALYSIS: healpy.map2alm([T, Q, U], ...) as hputil.sphtrans_real_pol uses it
# (cora/util/hputil.py:274-323).  (E, B)_lm = - sum_pix w (4 pi / npix) [(W Q - i X U), (W U + i X Q)] e^{-i m phi}:
# half the sum / difference of the quadratures of -(Q +- iU) (W -+ X) e^{-i m phi}, the spin +-2 harmonics being
# (W -+ X) e^{i m phi} in the convention of the synthesis above (Zaldarriaga & Seljak 1997).
# ------------------------------------------------------------------------------------
def _wx_from_lambda(lmax, m, z):
    """W_lm, X_lm for l = m..lmax at cos(theta) = z from this oracle's own lambda recurrence."""
    lam = lambda_lm(lmax, m, z)
    l = np.arange(m, lmax + 1, dtype=np.float64)
    lam1 = np.concatenate([[0.0], lam[:-1]])          # lambda_{l-1,m}, zero below l = m
    s2 = (1.0 - z) * (1.0 + z)
    W = np.zeros_like(lam)
    X = np.zeros_like(lam)
    ok = l >= 2
    lo = l[ok]
    N2 = 2.0 / np.sqrt((lo + 2.0) * (lo + 1.0) * lo * (lo - 1.0))
    c = np.where(lo + m > 0, np.sqrt((2.0 * lo + 1.0) / (2.0 * lo - 1.0) * (lo - m) / np.maximum(lo + m, 1.0)), 0.0)
    W[ok] = N2 * (-((lo - m * m) / s2 + lo * (lo - 1.0) / 2.0) * lam[ok] + (lo + m) * z / s2 * c * lam1[ok])
    X[ok] = N2 * m / s2 * ((lo - 1.0) * z * lam[ok] - (lo + m) * c * lam1[ok])
    return W, X


def map2alm_spin2_adjoint(q, u, nside, lmax, ring_w=None):
    """One quadrature pass (Q, U) RING maps -> packed (E, B)."""
    ri = healpix.ring_info(nside)
    npair = 2 * nside
    nring = 4 * nside - 1
    qn, qs = anal_to_gm(np.asarray(q, dtype=np.float64), nside, lmax, ring_w)
    un, us = anal_to_gm(np.asarray(u, dtype=np.float64), nside, lmax, ring_w)
    nalm = (lmax + 1) * (lmax + 2) // 2
    e = np.zeros(nalm, dtype=np.complex128)
    b = np.zeros(nalm, dtype=np.complex128)
    for m in range(lmax + 1):
        i0 = alm_index(m, m, lmax)
        for r in range(npair):
            z = float(ri["z"][r])
            W, X = _wx_from_lambda(lmax, m, z)
            l = np.arange(m, lmax + 1)
            sgn = np.where((l + m) % 2 == 0, 1.0, -1.0)      # W(-z) = sgn W(z), X(-z) = -sgn X(z)
            south = (nring - 1 - r) != r
            wq = W * qn[r, m] + (sgn * W * qs[r, m] if south else 0.0)
            wu = W * un[r, m] + (sgn * W * us[r, m] if south else 0.0)
            xq = X * qn[r, m] + (-sgn * X * qs[r, m] if south else 0.0)
            xu = X * un[r, m] + (-sgn * X * us[r, m] if south else 0.0)
            e[i0 : i0 + l.size] += -(wq - 1j * xu)
            b[i0 : i0 + l.size] += -(wu + 1j * xq)
    return e, b


def map2alm_spin2(q, u, nside, lmax, use_weights=True, niter=2):
    """healpy.map2alm([T, Q, U], lmax, use_weights, iter) restricted to (Q, U): quadrature + Jacobi refinement."""
    w = ring_weights(nside) if use_weights else None
    e, b = map2alm_spin2_adjoint(q, u, nside, lmax, w)
    for _ in range(niter):
        q1, u1 = alm2map_spin2(e, b, nside, lmax)
        de, db = map2alm_spin2_adjoint(q - q1, u - u1, nside, lmax, w)
        e, b = e + de, b + db
    return e, b


def map2alm_spin2_bruteforce(q, u, nside, lmax, ring_w=None):
    """Definition-level pixel sum with the scipy-based W / X (independent of the recurrence); tiny sizes."""
    theta, phi = healpix.pix2ang_ring(nside)
    ri = healpix.ring_info(nside)
    nring = 4 * nside - 1
    wpix = np.empty(theta.size)
    for r in range(nring):
        n, s = int(ri["nphi"][r]), int(ri["start"][r])
        wpix[s : s + n] = 1.0 if ring_w is None else ring_w[min(r, nring - 1 - r)]
    wpix *= 4.0 * np.pi / theta.size
    nalm = (lmax + 1) * (lmax + 2) // 2
    e = np.zeros(nalm, dtype=np.complex128)
    b = np.zeros(nalm, dtype=np.complex128)
    for m in range(lmax + 1):
        ph = np.exp(-1j * m * phi)
        for l in range(max(m, 2), lmax + 1):
            W, X = spin2_wx(l, m, theta)
            e[alm_index(l, m, lmax)] = -np.sum(wpix * (W * q - 1j * X * u) * ph)
            b[alm_index(l, m, lmax)] = -np.sum(wpix * (W * u + 1j * X * q) * ph)
    return e, b
