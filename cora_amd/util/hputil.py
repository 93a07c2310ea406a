"""Counterpart of the part of cora/util/hputil.py on the hot path.

``sphtrans_inv_real`` / ``sphtrans_inv_sky`` call the library's HEALPix synthesis
(K4 Legendre MFMA contraction + K5 ring FFT) where the reference calls
``healpy.alm2map`` (cora/util/hputil.py:388-391); all channels go through in one batch.
``sphtrans_real`` / ``sphtrans_sky`` / ``sph_ps`` call the adjoint kernels where the reference
calls ``healpy.map2alm(use_weights=True, iter=2)`` (cora/util/hputil.py:46-47,195-234,460-497,607-619).
"""
import numpy as np

from .. import _lib


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def ang2pix(nside, theta, phi, lonlat=False):
    """RING pixel index of a direction (what the reference takes from ``healpy.ang2pix`` in
    scripts/makesky.py:412-420): colatitude/longitude in radians, or (lon, lat) in degrees with
    ``lonlat=True``.  Standard HEALPix geometry (Gorski et al. 2005, eqs. 2-9)."""
    nside = int(nside)
    theta = np.asarray(theta, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    if lonlat:
        theta, phi = np.pi / 2.0 - np.radians(phi), np.radians(theta)
    z = np.cos(theta)
    za = np.abs(z)
    tt = np.mod(phi, 2.0 * np.pi) / (np.pi / 2.0)          # [0, 4)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    # equatorial belt
    t1 = nside * (0.5 + tt)
    t2 = nside * z * 0.75
    jp = np.floor(t1 - t2).astype(np.int64)
    jm = np.floor(t1 + t2).astype(np.int64)
    ir = nside + 1 + jp - jm
    kshift = 1 - (ir & 1)
    ip = np.mod((jp + jm - nside + kshift + 1) // 2, 4 * nside)
    belt = ncap + (ir - 1) * 4 * nside + ip
    # polar caps
    tp = tt - np.floor(tt)
    tmp = nside * np.sqrt(3.0 * (1.0 - za))
    jp = np.floor(tp * tmp).astype(np.int64)
    jm = np.floor((1.0 - tp) * tmp).astype(np.int64)
    irc = jp + jm + 1
    ipc = np.mod(np.floor(tt * irc).astype(np.int64), 4 * irc)
    cap = np.where(z > 0, 2 * irc * (irc - 1) + ipc, npix - 2 * irc * (irc + 1) + ipc)
    out = np.where(za <= 2.0 / 3.0, belt, cap)
    return out if out.ndim else int(out)


def pix2ang(nside, ipix):
    """(theta, phi) of RING pixel centres (what the reference takes from ``healpy.pix2ang``): caps
    z = 1 - i^2/(3 nside^2), phi = (j + 1/2) pi/(2 i); belt z = 4/3 - 2 i/(3 nside), phi = (j + s/2) pi/(2 nside)."""
    nside = int(nside)
    ipix = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    p = np.where(ipix >= npix - ncap, npix - 1 - ipix, ipix)             # mirror the south cap onto the north
    # north cap: ring i (1-based) holds pixels 2 i (i - 1) .. 2 i (i + 1) - 1
    i_cap = ((1 + np.sqrt(1 + 2 * np.minimum(p, max(ncap - 1, 0)).astype(np.float64))) / 2).astype(np.int64)
    i_cap = np.where(2 * i_cap * (i_cap - 1) > p, i_cap - 1, i_cap)
    i_cap = np.where(2 * i_cap * (i_cap + 1) <= p, i_cap + 1, i_cap)
    i_cap = np.maximum(i_cap, 1)
    j_cap = p - 2 * i_cap * (i_cap - 1)
    z_cap = 1.0 - i_cap.astype(np.float64) ** 2 / (3.0 * nside * nside)
    phi_cap = (j_cap + 0.5) * np.pi / (2.0 * i_cap)
    # equatorial belt
    pb = ipix - ncap
    i_b = pb // (4 * nside) + nside
    j_b = pb % (4 * nside)
    s_b = (i_b - nside + 1) & 1
    z_b = 4.0 / 3.0 - 2.0 * i_b / (3.0 * nside)
    phi_b = (j_b + 0.5 * s_b) * np.pi / (2.0 * nside)
    in_n = ipix < ncap
    in_s = ipix >= npix - ncap
    z = np.where(in_n, z_cap, np.where(in_s, -z_cap, z_b))
    # a mirrored south-cap pixel runs backwards in phi on its ring
    phi = np.where(in_n, phi_cap, np.where(in_s, 2.0 * np.pi - phi_cap, phi_b))
    return np.arccos(z), phi


def ang_positions(nside):
    """Angular position [theta, phi] of every pixel: [npix, 2] (cora/util/hputil.py:53-73)."""
    npix = nside2npix(int(nside))
    angpos = np.empty([npix, 2], dtype=np.float64)
    angpos[:, 0], angpos[:, 1] = pix2ang(nside, np.arange(npix))
    return angpos


def nside_for_lmax(lmax, accuracy_boost=1):
    """cora/util/hputil.py:76-90."""
    return int(2 ** (accuracy_boost + np.ceil(np.log((lmax + 1) / 3.0) / np.log(2.0))))


def _make_full_alm(alm_half, centered=False):
    """a_lm for m >= 0 -> both signs of m, using a_{l,-m} = (-1)^m conj(a_lm) (cora/util/hputil.py:155-174).
    Output [..., l, 2 mmax - 1]: FFT order (m >= 0 first, then m = -(mmax-1)..-1) or, ``centered``, m ascending."""
    lmax, mmax = alm_half.shape[-2:]
    alm = np.zeros(alm_half.shape[:-2] + (lmax, 2 * mmax - 1), dtype=alm_half.dtype)
    neg = ((-1) ** np.arange(mmax)[:0:-1]) * alm_half[..., :, :0:-1].conj()
    if not centered:
        alm[..., :mmax] = alm_half
        alm[..., mmax:] = neg
    else:
        alm[..., (mmax - 1):] = alm_half
        alm[..., : (mmax - 1)] = neg
    return alm


def _make_half_alm(alm_full):
    """[l, 2 lside - 1] (FFT order in m) -> the m >= 0 coefficients of its REAL part: the projection
    (a_lm + (-1)^m conj(a_{l,-m})) / 2 (cora/util/hputil.py:177-192)."""
    lside = alm_full.shape[-2]
    alm = np.zeros(alm_full.shape[:-2] + (lside, lside), dtype=alm_full.dtype)
    alm[..., 0] = alm_full[..., :, 0]
    for mi in range(1, lside):
        alm[..., mi] = 0.5 * (alm_full[..., mi] + (-1) ** mi * alm_full[..., -mi].conj())
    return alm


def unpack_alm(alm, lmax, fullm=False):
    """Healpix-packed a_lm -> 2D [l, m] (cora/util/hputil.py:93-121)."""
    almarray = np.zeros((lmax + 1, lmax + 1), dtype=alm.dtype)
    (almarray.T)[np.triu_indices(lmax + 1)] = alm
    if fullm:
        full = np.zeros((lmax + 1, 2 * lmax + 1), dtype=alm.dtype)
        full[:, : lmax + 1] = almarray
        mm = np.arange(1, lmax + 1)
        full[:, -mm] = ((-1.0) ** mm) * almarray[:, mm].conj()
        almarray = full
    return almarray


def pack_alm(almarray, lmax=None):
    """2D [l, m] a_lm -> Healpix packing, idx(l,m) = m(2 lmax+1-m)/2 + l (hputil.py:124-152)."""
    if (2 * almarray.shape[1] - 1) == almarray.shape[0]:
        almarray = _make_half_alm(almarray)
    if not lmax:
        lmax = almarray.shape[0] - 1
    return (almarray.T)[np.triu_indices(lmax + 1)]


def _synth(alm_list, nside):
    """alm_list: [n, L, L] complex -> [n, npix] maps on the GPU."""
    n, L, _ = alm_list.shape
    lmax = L - 1
    packed = np.stack([pack_alm(a) for a in alm_list]).astype(np.complex128)
    ctx = _lib.get_context()
    import torch

    alm_dev = ctx.alm_packed_to_dev(torch.from_numpy(np.ascontiguousarray(packed)).to(ctx.device), lmax)
    maps = ctx.alm2map(alm_dev, int(nside), lmax, n)
    return _lib.get_context().to_host(maps)


def sphtrans_inv_real(alm, nside):
    """Inverse SHT onto a real field (cora/util/hputil.py:369-391)."""
    if alm.shape[1] != alm.shape[0]:
        raise Exception("a_lm array wrong shape.")
    return _synth(np.asarray(alm)[np.newaxis], nside)[0]


def _synth_pol(alm_e, alm_b, nside):
    """alm_e, alm_b: [n, L, L] complex -> (Q, U) maps [n, npix] each (spin-2 synthesis on the GPU)."""
    import torch

    n, L, _ = alm_e.shape
    lmax = L - 1
    packed = np.empty((2 * n, L * (L + 1) // 2), dtype=np.complex128)
    for i in range(n):
        packed[2 * i] = pack_alm(alm_e[i])
        packed[2 * i + 1] = pack_alm(alm_b[i])
    ctx = _lib.get_context()
    out = np.empty((2 * n, nside2npix(nside)))
    # the spin-2 entry point wants its channel count in whole groups of 8 (or 5..7 mod 8): chunks of 4 fields
    for c0 in range(0, 2 * n, 8):
        c1 = min(c0 + 8, 2 * n)
        blk = packed[c0:c1]
        if (c1 - c0) % 8 not in (0, 6):          # pad with zero fields to 8 channels
            blk = np.concatenate([blk, np.zeros((8 - (c1 - c0), packed.shape[1]), dtype=np.complex128)])
        dev = ctx.alm_packed_to_dev(torch.from_numpy(np.ascontiguousarray(blk)).to(ctx.device), lmax)
        maps = ctx.alm2map_spin2(dev, int(nside), lmax, blk.shape[0])
        out[c0:c1] = maps[: c1 - c0].cpu().numpy()
    return out[0::2], out[1::2]


def sphtrans_inv_real_pol(alm, nside):
    """Inverse transform onto a real polarised field: alm [npol, L, L] for T, E, B (and V) -> T, Q, U (and V) maps
    [npol, npix] (cora/util/hputil.py:394-432; healpy.alm2map of the three packed arrays).  Q and U come from the
    spin-2 synthesis kernel (convention: Zaldarriaga & Seljak 1997, the one HEALPix documents), T and V from the
    scalar one."""
    alm = np.asarray(alm)
    npol = alm.shape[0]
    if alm.shape[1] != alm.shape[2] or not (npol == 3 or npol == 4):
        raise Exception("a_lm array wrong shape.")
    maps = np.zeros((npol, nside2npix(nside)), dtype=np.float64)
    scal = _synth(alm[[0] + ([3] if npol == 4 else [])], nside)
    maps[0] = scal[0]
    q, u = _synth_pol(alm[1:2], alm[2:3], nside)
    maps[1], maps[2] = q[0], u[0]
    if npol == 4:
        maps[3] = scal[1]
    return maps


def sphtrans_inv_sky(alm, nside):
    """[freq, pol, l, m] a_lm -> [freq, pol, npix] sky (cora/util/hputil.py:500-531): the polarised transform when
    the pol axis has 3 or 4 entries (T, E, B[, V] -> T, Q, U[, V]), else the scalar one; all frequencies in batches."""
    alm = np.asarray(alm)
    nfreq, npol = alm.shape[0], alm.shape[1]
    if alm.shape[3] != alm.shape[2]:
        raise Exception("a_lm array wrong shape.")
    sky = np.empty((nfreq, npol, nside2npix(nside)), dtype=np.float64)
    if npol >= 3:
        if npol > 4:
            raise Exception("a_lm array wrong shape.")
        sky[:, 0] = _synth(alm[:, 0], nside)
        sky[:, 1], sky[:, 2] = _synth_pol(alm[:, 1], alm[:, 2], nside)
        if npol == 4:
            sky[:, 3] = _synth(alm[:, 3], nside)
        return sky
    for p in range(npol):
        sky[:, p] = _synth(np.asarray(alm[:, p]), nside)
    return sky


# ------------------------------------------------------------------------------------
# analysis: healpy.map2alm(use_weights=_weight, iter=_iter) as the reference configures it
# ------------------------------------------------------------------------------------
_weight = True   # cora/util/hputil.py:46
_iter = 2        # cora/util/hputil.py:47
_ring_weight_cache = {}


def ring_weights(nside, lmax_exact=None):
    """Quadrature weights of the 2 nside north rings (equator included).

    healpy reads these from the HEALPix data file weight_ring_n<nside>.fits; that file is data of a
    dependency which is not available to this package, so the defining property is used: the
    minimum-norm correction to uniform weights that integrates the zonal P_l(z), even l <= lmax_exact
    (default 3 nside), exactly.  Host numpy, once per nside (cached)."""
    key = (int(nside), lmax_exact)
    if key not in _ring_weight_cache:
        nside = int(nside)
        npair = 2 * nside
        npix = nside2npix(nside)
        i = np.arange(1, npair + 1, dtype=np.float64)
        cap = i < nside
        z = np.where(cap, 1.0 - i * i / (3.0 * nside * nside), 4.0 / 3.0 - 2.0 * i / (3.0 * nside))
        cnt = np.where(cap, 4.0 * i, 4.0 * nside) * 2.0
        cnt[-1] = 4.0 * nside                       # the equator ring has no mirror
        lx = 3 * nside if lmax_exact is None else int(lmax_exact)
        P = np.empty((lx + 1, npair))
        P[0] = 1.0
        if lx >= 1:
            P[1] = z
        for l in range(2, lx + 1):
            P[l] = ((2 * l - 1) * z * P[l - 1] - (l - 1) * P[l - 2]) / l
        M = P[0::2] * (cnt * 4.0 * np.pi / npix)[None, :]
        rhs = np.zeros(M.shape[0])
        rhs[0] = 4.0 * np.pi
        _ring_weight_cache[key] = 1.0 + np.linalg.lstsq(M, rhs - M.sum(axis=1), rcond=None)[0]
    return _ring_weight_cache[key]


def map2alm_device(maps, nside, lmax, use_weights=None, niter=None):
    """Device maps [n, npix] -> alm_dev: quadrature pass + `niter` Jacobi refinements
    alm <- alm + A(map - S alm), i.e. healpy.map2alm(..., use_weights, iter) for all maps at once."""
    ctx = _lib.get_context()
    use_weights = _weight if use_weights is None else use_weights
    niter = _iter if niter is None else niter
    w = ctx.to_device(ring_weights(nside)) if use_weights else None
    n = maps.shape[0]
    alm = ctx.map2alm(maps, int(nside), int(lmax), w)
    for _ in range(niter):
        resid = maps - ctx.alm2map(alm, int(nside), int(lmax), n)
        alm = alm + ctx.map2alm(resid, int(nside), int(lmax), w)
        del resid
    return alm


def map2alm_pol_device(maps_qu, nside, lmax, use_weights=None, niter=None):
    """Device maps [2 n, npix], (Q_f, U_f) interleaved -> alm_dev with (E_f, B_f) interleaved: the (Q, U) half of
    healpy.map2alm([T, Q, U], use_weights, iter) - a quadrature pass composed of scalar passes over ring-scaled
    maps (csrc/sht_polana.hip) plus `niter` refinements alm <- alm + A(map - S alm) with the spin-2 synthesis."""
    ctx = _lib.get_context()
    use_weights = _weight if use_weights is None else use_weights
    niter = _iter if niter is None else niter
    w = ctx.to_device(ring_weights(nside)) if use_weights else None
    n2 = maps_qu.shape[0]
    alm = ctx.map2alm_spin2(maps_qu, int(nside), int(lmax), w)
    nnu_pad = 4 * alm.shape[1]
    for _ in range(niter):
        back = ctx.alm2map_spin2(alm, int(nside), int(lmax), nnu_pad)[:n2]
        alm = alm + ctx.map2alm_spin2(maps_qu - back, int(nside), int(lmax), w)
        del back
    return alm


def _analyse_pol(q, u, lmax):
    """Host (Q, U) maps [n, npix] each -> (E, B) alm[l, m] arrays [n, lmax+1, lmax+1] each."""
    import torch

    q = np.ascontiguousarray(q, dtype=np.float64)
    n, npix = q.shape
    nside = int(round(np.sqrt(npix / 12.0)))
    if 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")   # healpy.npix2nside
    qu = np.empty((2 * n, npix))
    qu[0::2], qu[1::2] = q, u
    ctx = _lib.get_context()
    alm = map2alm_pol_device(torch.from_numpy(qu).to(ctx.device), nside, lmax)
    sq = ctx.alm_dev_to_square(alm, lmax, 4 * alm.shape[1]).cpu().numpy()[: 2 * n, 0]   # (padding channels dropped)
    return sq[0::2], sq[1::2]


def _analyse(hpmaps, lmax):
    """[n, npix] host maps -> [n, lmax+1, lmax+1] complex alm[l, m] (m > l entries zero)."""
    import torch

    hpmaps = np.ascontiguousarray(hpmaps, dtype=np.float64)
    n, npix = hpmaps.shape
    nside = int(round(np.sqrt(npix / 12.0)))
    if 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")   # healpy.npix2nside
    ctx = _lib.get_context()
    alm = map2alm_device(torch.from_numpy(hpmaps).to(ctx.device), nside, lmax)
    return ctx.alm_dev_to_square(alm, lmax, n).cpu().numpy()[:, 0]


def sphtrans_real(hpmap, lmax=None, lside=None):
    """Spherical harmonic transform of a real map -> alm[l, m], m >= 0 (cora/util/hputil.py:195-234)."""
    hpmap = np.asarray(hpmap)
    if lmax is None:
        lmax = 3 * int(round(np.sqrt(hpmap.size / 12.0))) - 1
    if lside is None or lside < lmax:
        lside = lmax
    alm = np.zeros([lside + 1, lside + 1], dtype=np.complex128)
    alm[: lmax + 1, : lmax + 1] = _analyse(hpmap.reshape(1, -1), lmax)[0]
    return alm


def sphtrans_real_pol(hpmaps, lmax=None, lside=None):
    """T, Q, U (and V) maps [npol, npix] -> a^T, a^E, a^B (and a^V) as alm[pol, l, m], m >= 0
    (cora/util/hputil.py:274-323: healpy.map2alm of the T, Q, U triple, V on its own)."""
    hpmaps = np.ascontiguousarray(hpmaps, dtype=np.float64)
    npol = len(hpmaps)
    if lmax is None:
        lmax = 3 * int(round(np.sqrt(hpmaps[0].size / 12.0))) - 1
    if lside is None or lside < lmax:
        lside = lmax
    alms = np.zeros([npol, lside + 1, lside + 1], dtype=np.complex128)
    scal = _analyse(hpmaps[[0] + ([3] if npol == 4 else [])], lmax)
    alms[0, : lmax + 1, : lmax + 1] = scal[0]
    e, b = _analyse_pol(hpmaps[1:2], hpmaps[2:3], lmax)
    alms[1, : lmax + 1, : lmax + 1], alms[2, : lmax + 1, : lmax + 1] = e[0], b[0]
    if npol == 4:
        alms[3, : lmax + 1, : lmax + 1] = scal[1]
    return alms


def sphtrans_complex_pol(hpmaps, lmax=None, centered=False, lside=None):
    """Complex T, Q, U (and V) maps -> a_lm for both signs of m (cora/util/hputil.py:326-366)."""
    hpmaps = np.asarray(hpmaps)
    if lmax is None:
        lmax = 3 * int(round(np.sqrt(hpmaps[0].size / 12.0))) - 1
    alm = _make_full_alm(sphtrans_real_pol(hpmaps.real, lmax=lmax, lside=lside), centered=centered)
    alm += 1.0j * _make_full_alm(sphtrans_real_pol(hpmaps.imag, lmax=lmax, lside=lside), centered=centered)
    return alm


def sphtrans_sky(skymap, lmax=None):
    """[freq, npix] (or [freq, pol, npix]) sky -> alm [freq, (pol,) l, m] (cora/util/hputil.py:460-497).
    All frequency slices go through the GPU in one batch; with 3 or 4 polarisation components the
    (Q, U) planes take the spin-2 analysis, T (and V) the scalar one."""
    skymap = np.asarray(skymap)
    if skymap.ndim == 3 and skymap.shape[1] >= 3:
        if skymap.shape[1] > 4:
            raise Exception("Wrong number of polarisation components.")
        if lmax is None:
            lmax = 3 * int(round(np.sqrt(skymap.shape[-1] / 12.0))) - 1
        nfreq, npol = skymap.shape[:2]
        alm = np.zeros((nfreq, npol, lmax + 1, lmax + 1), dtype=np.complex128)
        alm[:, 0] = _analyse(skymap[:, 0].astype(np.float64), lmax)
        alm[:, 1], alm[:, 2] = _analyse_pol(skymap[:, 1], skymap[:, 2], lmax)
        if npol == 4:
            alm[:, 3] = _analyse(skymap[:, 3].astype(np.float64), lmax)
        return alm
    if lmax is None:
        lmax = 3 * int(round(np.sqrt(skymap.shape[-1] / 12.0))) - 1
    flat = skymap.reshape(-1, skymap.shape[-1]).astype(np.float64)
    alm = _analyse(flat, lmax)
    return alm.reshape(skymap.shape[:-1] + (lmax + 1, lmax + 1))


def sphtrans_complex(hpmap, lmax=None, centered=False, lside=None):
    """Spherical harmonic transform of a complex map: a_lm for both signs of m (cora/util/hputil.py:237-263)."""
    hpmap = np.asarray(hpmap)
    if lmax is None:
        lmax = 3 * int(round(np.sqrt(hpmap.size / 12.0))) - 1
    both = _analyse(np.stack([hpmap.real, hpmap.imag]), lmax)
    if lside is not None and lside > lmax:
        pad = np.zeros((2, lside + 1, lside + 1), dtype=np.complex128)
        pad[:, : lmax + 1, : lmax + 1] = both
        both = pad
    return _make_full_alm(both[0], centered=centered) + 1.0j * _make_full_alm(both[1], centered=centered)


def sphtrans_inv_complex(alm, nside):
    """Inverse transform onto a complex field from a_lm with both signs of m (cora/util/hputil.py:435-457).

    Mirrors the reference formula exactly, including its sign: the imaginary part is built from
    ``1j * (a - a_real)`` = minus the coefficients of Im f, so the result is the complex CONJUGATE of the field
    whose :func:`sphtrans_complex` is ``alm``; and all of a_l0 goes to the real part (m = 0 modes of Im f drop)."""
    alm = np.asarray(alm)
    if alm.shape[1] != (2 * alm.shape[0] - 1):
        raise Exception("a_lm array wrong shape: " + repr(alm.shape))
    almr = _make_half_alm(alm)
    almi = 1.0j * (alm[:, : almr.shape[1]] - almr)
    both = _synth(np.stack([almr, almi]), nside)
    return both[0] + 1.0j * both[1]


def sph_ps(map1, map2=None, lmax=None):
    """Angular (cross) power spectrum of maps (cora/util/hputil.py:607-619).

    The reference's test ``if map is not None`` looks at the builtin ``map`` and is always true, so its
    auto-spectrum branch never runs and ``map2=None`` fails inside healpy; here ``map2=None`` means the
    auto spectrum, which is what the signature documents."""
    map1 = np.asarray(map1)
    lmax = lmax if lmax is not None else (3 * int(round(np.sqrt(map1.size / 12.0))) - 1)
    if map2 is None:
        alm1 = alm2 = sphtrans_real(map1, lmax)
    else:
        both = _analyse(np.stack([map1, np.asarray(map2)]), lmax)
        alm1, alm2 = both[0], both[1]
    prod = alm1 * alm2.conj()
    s = prod[:, 0] + 2 * prod[:, 1:].sum(axis=1).real
    return s / (2.0 * np.arange(lmax + 1) + 1.0)
