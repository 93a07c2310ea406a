"""Counterpart of the part of cora/util/hputil.py on the hot path.

``sphtrans_inv_real`` / ``sphtrans_inv_sky`` call the library's HEALPix synthesis
(K4 Legendre MFMA contraction + K5 ring FFT) where the reference calls
``healpy.alm2map`` (cora/util/hputil.py:388-391); all channels go through in one batch.
"""
import numpy as np

from .. import _lib


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def nside_for_lmax(lmax, accuracy_boost=1):
    """cora/util/hputil.py:76-90."""
    return int(2 ** (accuracy_boost + np.ceil(np.log((lmax + 1) / 3.0) / np.log(2.0))))


def _make_half_alm(alm_full):
    """[l, 2 lmax+1] (negative m in the second half) -> [l, m >= 0] (hputil.py:181-192)."""
    lside = alm_full.shape[-2]
    return alm_full[..., :lside]


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
    return maps.cpu().numpy()


def sphtrans_inv_real(alm, nside):
    """Inverse SHT onto a real field (cora/util/hputil.py:369-391)."""
    if alm.shape[1] != alm.shape[0]:
        raise Exception("a_lm array wrong shape.")
    return _synth(np.asarray(alm)[np.newaxis], nside)[0]


def sphtrans_inv_sky(alm, nside):
    """[freq, pol, l, m] a_lm -> [freq, pol, npix] sky (cora/util/hputil.py:500-531).

    Only the unpolarised branch is on the hot path; 3- or 4-component polarised input
    (spin-2 synthesis) is outside this package's scope.
    """
    nfreq, npol = alm.shape[0], alm.shape[1]
    if npol >= 3:
        raise NotImplementedError("polarised synthesis (hputil.py:394-432) is out of scope of cora_amd")
    if alm.shape[3] != alm.shape[2]:
        raise Exception("a_lm array wrong shape.")
    sky = np.empty((nfreq, npol, nside2npix(nside)), dtype=np.float64)
    for p in range(npol):
        sky[:, p] = _synth(np.asarray(alm[:, p]), nside)
    return sky
