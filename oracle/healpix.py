"""HEALPix RING geometry (oracle; test infrastructure only).

Stands in for the parts of healpy the reference uses implicitly through
``healpy.alm2map`` (cora/util/hputil.py:388-391) and ``healpy.nside2npix``
(cora/util/hputil.py:521-523).  Conventions are the HEALPix *software* ones
(``pix2ang_ring``), see SURVEY.md Appendix A.
"""
import numpy as np


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def ring_info(nside):
    """Per-ring geometry for rings i = 1 .. 4*nside-1 (north to south).

    Returns dict of arrays (index r = i-1): nphi, start (first RING pixel),
    z = cos(theta), sth = sin(theta), phi0 (azimuth of first pixel).
    """
    nside = int(nside)
    nring = 4 * nside - 1
    i = np.arange(1, nring + 1)
    npix = nside2npix(nside)
    nphi = np.empty(nring, dtype=np.int64)
    start = np.empty(nring, dtype=np.int64)
    z = np.empty(nring)
    sth = np.empty(nring)
    phi0 = np.empty(nring)

    fact2 = 4.0 / npix  # 1/(3 nside^2)
    fact1 = 2.0 * nside * fact2  # 2/(3 nside)

    north = i < nside
    belt = (i >= nside) & (i <= 3 * nside)
    south = i > 3 * nside

    # north cap
    ic = i[north].astype(np.float64)
    tmp = ic * ic * fact2
    z[north] = 1.0 - tmp
    sth[north] = np.sqrt(tmp * (2.0 - tmp))
    nphi[north] = 4 * i[north]
    phi0[north] = np.pi / (4.0 * ic)
    start[north] = 2 * i[north] * (i[north] - 1)

    # belt
    ib = i[belt]
    zb = (2 * nside - ib) * fact1
    z[belt] = zb
    sth[belt] = np.sqrt((1.0 - zb) * (1.0 + zb))
    nphi[belt] = 4 * nside
    shifted = ((ib - nside) & 1) == 0
    phi0[belt] = np.where(shifted, np.pi / (4.0 * nside), 0.0)
    start[belt] = 2 * nside * (nside - 1) + (ib - nside) * 4 * nside

    # south cap: mirror of ring i' = 4 nside - i
    ip = (4 * nside - i[south])
    ipf = ip.astype(np.float64)
    tmp = ipf * ipf * fact2
    z[south] = -(1.0 - tmp)
    sth[south] = np.sqrt(tmp * (2.0 - tmp))
    nphi[south] = 4 * ip
    phi0[south] = np.pi / (4.0 * ipf)
    start[south] = npix - 2 * ip * (ip + 1)

    return dict(nphi=nphi, start=start, z=z, sth=sth, phi0=phi0)


def pix2ang_ring(nside):
    """(theta, phi) of every RING-ordered pixel centre."""
    ri = ring_info(nside)
    npix = nside2npix(nside)
    theta = np.empty(npix)
    phi = np.empty(npix)
    for r in range(len(ri["nphi"])):
        n, s = int(ri["nphi"][r]), int(ri["start"][r])
        theta[s : s + n] = np.arctan2(ri["sth"][r], ri["z"][r])
        phi[s : s + n] = ri["phi0"][r] + 2.0 * np.pi * np.arange(n) / n
    return theta, phi
