"""clarray / mkfullsky / matrix root / normal draw: numpy restatement (oracle).

TEST INFRASTRUCTURE ONLY.  Follows, relative to /root/reference:
  * cora/core/skysim.py:10-69     clarray
  * cora/core/skysim.py:72-136    mkfullsky
  * cora/util/nputil.py:51-101    matrix_root_manynull
  * cora/util/nputil.py:104-125   complex_std_normal
  * cora/util/hputil.py:93-152    unpack_alm / pack_alm
  * cora/util/hputil.py:369-391,500-531  sphtrans_inv_real / sphtrans_inv_sky
"""
import numpy as np
import scipy.integrate as si
import scipy.linalg as la

from . import healpix, sht


def clarray(aps, lmax, zarray, zromb=3, zwidth=None, rows=None):
    """skysim.py:10-69.  ``rows`` (checker convenience, not in the reference): evaluate only these multipoles
    (each l is independent in the reference's loop) and return ``[len(rows), F, F]``, the first channel axis
    processed in blocks so that F = 1024, zromb = 3 fits in memory."""
    zarray = np.asarray(zarray, dtype=np.float64)
    if rows is not None:
        return _clarray_rows(aps, np.asarray(rows), zarray, zromb, zwidth)
    if zromb == 0:
        return aps(np.arange(lmax + 1)[:, np.newaxis, np.newaxis], zarray[np.newaxis, :, np.newaxis],
                   zarray[np.newaxis, np.newaxis, :])
    zsort = np.sort(zarray)
    zhalf = np.abs(zsort[1] - zsort[0]) / 2.0 if zwidth is None else zwidth / 2.0
    zlen = zarray.size
    zint = 2**zromb + 1
    zspace = 2.0 * zhalf / 2**zromb
    za = (zarray[:, np.newaxis] + np.linspace(-zhalf, zhalf, zint)[np.newaxis, :]).flatten()
    lsections = np.array_split(np.arange(lmax + 1), lmax // 5)
    cla = np.zeros((lmax + 1, zlen, zlen), dtype=np.float64)
    for lsec in lsections:
        clt = aps(lsec[:, np.newaxis, np.newaxis], za[np.newaxis, :, np.newaxis], za[np.newaxis, np.newaxis, :])
        clt = clt.reshape(-1, zlen, zint, zlen, zint)
        clt = si.romb(clt, dx=zspace, axis=4)
        clt = si.romb(clt, dx=zspace, axis=2)
        cla[lsec] = clt / (2 * zhalf) ** 2
    return cla


def _clarray_rows(aps, rows, zarray, zromb, zwidth, block=32):
    """The body of clarray's l-section loop (skysim.py:56-67) for the multipoles ``rows`` only."""
    lcol = rows.astype(np.int64)[:, np.newaxis, np.newaxis]
    if zromb == 0:
        return aps(lcol, zarray[np.newaxis, :, np.newaxis], zarray[np.newaxis, np.newaxis, :])
    zsort = np.sort(zarray)
    zhalf = np.abs(zsort[1] - zsort[0]) / 2.0 if zwidth is None else zwidth / 2.0
    zlen = zarray.size
    zint = 2**zromb + 1
    zspace = 2.0 * zhalf / 2**zromb
    za = (zarray[:, np.newaxis] + np.linspace(-zhalf, zhalf, zint)[np.newaxis, :]).flatten()
    cla = np.zeros((len(rows), zlen, zlen), dtype=np.float64)
    for i0 in range(0, zlen, block):
        i1 = min(zlen, i0 + block)
        zb = za[i0 * zint : i1 * zint]
        clt = aps(lcol, zb[np.newaxis, :, np.newaxis], za[np.newaxis, np.newaxis, :])
        clt = clt.reshape(-1, i1 - i0, zint, zlen, zint)
        clt = si.romb(clt, dx=zspace, axis=4)
        clt = si.romb(clt, dx=zspace, axis=2)
        cla[:, i0:i1] = clt / (2 * zhalf) ** 2
    return cla


def matrix_root_manynull(mat, threshold=1e-16, truncate=True):
    """nputil.py:51-101."""
    try:
        root = la.cholesky(mat, lower=True)
        num_pos = mat.shape[0]
    except la.LinAlgError:
        evals, evecs = la.eigh(mat)
        evals[np.where(evals < evals.max() * threshold)] = 0.0
        num_pos = len(np.flatnonzero(evals))
        if truncate:
            evals = evals[np.newaxis, -num_pos:]
            evecs = evecs[:, -num_pos:]
        root = evecs * evals[np.newaxis, :] ** 0.5
    if truncate:
        return root, num_pos
    return root


def complex_std_normal(shape, rng=None):
    """nputil.py:104-125 (real block first, then imag block)."""
    if rng is None:
        return (np.random.standard_normal(shape) + 1.0j * np.random.standard_normal(shape)) / 2**0.5
    return (rng.standard_normal(shape) + 1.0j * rng.standard_normal(shape)) / 2**0.5


def pack_alm(almarray, lmax=None):
    """hputil.py:124-152 (half-m input only)."""
    if not lmax:
        lmax = almarray.shape[0] - 1
    return (almarray.T)[np.triu_indices(lmax + 1)]


def unpack_alm(alm, lmax):
    """hputil.py:93-121."""
    almarray = np.zeros((lmax + 1, lmax + 1), dtype=alm.dtype)
    (almarray.T)[np.triu_indices(lmax + 1)] = alm
    return almarray


def factors(corr):
    """Per-l jittered roots, skysim.py:114-119."""
    numz = corr.shape[1]
    out = np.empty_like(corr)
    for l in range(corr.shape[0]):
        cmax = corr[l].diagonal().max() * 1e-14
        corrm = corr[l] + np.identity(numz) * cmax
        out[l] = matrix_root_manynull(corrm, truncate=False)
    return out


def mkfullsky(corr, nside, alms=False, rng=None, normals=None):
    """skysim.py:72-136 (single process).  `normals`, if given, is a list of the
    per-l complex (numz, l+1) draws to use instead of consuming `rng`."""
    numz = corr.shape[1]
    maxl = corr.shape[0] - 1
    if corr.shape[2] != numz:
        raise Exception("Correlation matrix is incorrect shape.")
    alm_array = np.zeros((numz, 1, maxl + 1, maxl + 1), dtype=np.complex128)
    for l in range(maxl + 1):
        cmax = corr[l].diagonal().max() * 1e-14
        corrm = corr[l] + np.identity(numz) * cmax
        trans = matrix_root_manynull(corrm, truncate=False)
        gaussvars = normals[l] if normals is not None else complex_std_normal((numz, l + 1), rng=rng)
        alm_array[:, 0, l, : (l + 1)] = np.dot(trans, gaussvars)
    if alms:
        return alm_array
    return sphtrans_inv_sky(alm_array, nside)[:, 0]


def sphtrans_inv_real(alm, nside):
    """hputil.py:369-391 with the oracle SHT in place of healpy.alm2map."""
    if alm.shape[1] != alm.shape[0]:
        raise Exception("a_lm array wrong shape.")
    return sht.alm2map(pack_alm(alm), nside)


def sphtrans_inv_sky(alm, nside):
    """hputil.py:500-531, unpolarised branch."""
    nfreq = alm.shape[0]
    sky = np.empty((nfreq, alm.shape[1], healpix.nside2npix(nside)), dtype=np.float64)
    for i in range(nfreq):
        sky[i, 0] = sphtrans_inv_real(alm[i, 0], nside)
    return sky


def sph_ps_from_alm(alm2d):
    """hputil.py:613-617 estimator applied to a [l,m] alm array."""
    prod = alm2d * alm2d.conj()
    s = prod[:, 0] + 2 * prod[:, 1:].sum(axis=1).real
    return (s / (2.0 * np.arange(alm2d.shape[0]) + 1.0)).real


def mkconstrained(corr, constraints, nside):
    """cora/core/skysim.py:139-205 with the oracle transforms standing in for healpy
    (map2alm with healpy's defaults iter=3 / no weights; alm2map)."""
    numz = corr.shape[1]
    maxl = corr.shape[0] - 1
    larr = np.concatenate([np.arange(m, maxl + 1) for m in range(maxl + 1)])     # healpy.Alm.getlm order
    nmodes = len(constraints)
    f_ind = [c[0] for c in constraints]
    if corr.shape[2] != numz:
        raise Exception("Correlation matrix is incorrect shape.")
    trans = np.zeros((corr.shape[0], nmodes, corr.shape[2]))
    tmat = np.zeros((corr.shape[0], nmodes, nmodes))
    cmap = np.zeros(larr.shape + (nmodes,), dtype=np.complex128)
    cv = np.zeros((numz,) + larr.shape, dtype=np.complex128)
    for i in range(maxl + 1):
        trans[i] = la.eigh(corr[i])[1][:, -nmodes:].T
        tmat[i] = trans[i][:, f_ind]
    for i, cons in enumerate(constraints):
        ns_c = int(round(np.sqrt(np.asarray(cons[1]).size / 12.0)))
        cmap[:, i] = sht.map2alm(np.asarray(cons[1], dtype=np.float64), ns_c, maxl, use_weights=False, niter=3)
    for i, l in enumerate(larr):
        if l == 0:
            cv[:, i] = 0.0
        else:
            cv[:, i] = np.dot(trans[l].T, la.solve(tmat[l].T, cmap[i]))
    hpmaps = np.empty((numz, healpix.nside2npix(nside)))
    for i in range(numz):
        hpmaps[i] = sht.alm2map(cv[i], nside, maxl)
    return hpmaps
