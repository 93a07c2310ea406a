"""The CPU oracle against (a) outputs of the reference itself (tests/golden/reference_vectors.npz,
made by tests/golden/make_golden.py), (b) the reference's own known-answer tests
(tests/test_corr.py), (c) definition-level checks of the synthesis that stands in for healpy."""
import os

import numpy as np
import pytest

from oracle import healpix, models, sht
from oracle import skysim as osk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


# ------------------------------------------------------------------ reference KATs
def test_reference_kat_values_are_in_golden(golden):
    """tests/test_corr.py:15-31,44-57 constants, and what the reference computes for them here."""
    kat = golden["kat_test_corr"]
    assert kat[0] == 1.5963772205823096e-09 and kat[3] == 75.47681191093129
    # foreground KATs reproduce bit-exactly with the reference code in this container
    assert np.allclose(golden["fg_kat"], kat[3:], rtol=1e-12)
    # 21cm KATs reproduce with the Planck-2013 cosmology they were computed with
    assert np.allclose(golden["sig_kat_planck13"], kat[:3], rtol=1e-7)


def test_foreground_kat():
    cr = models.FullSkySynchrotron()
    aps1 = cr.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    assert len(aps1) == 1000
    assert np.allclose(aps1.sum(), 75.47681191093129, rtol=1e-7)
    fa = np.linspace(400.0, 800.0, 64)
    aps2 = cr.angular_powerspectrum(np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :])
    assert aps2.shape == (1000, 64, 64)
    assert np.allclose(aps2[400, 40, 40], 9.690708728692975e-06, rtol=1e-7)
    assert np.allclose(aps2[200, 10, 40], 0.00017630767166797886, rtol=1e-7)


def test_signal_kat(model21, golden):
    aps1 = model21.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    assert np.allclose(aps1.sum(), golden["sig_kat_default"][0], rtol=1e-12)
    assert np.allclose(aps1[[0, 1, 2, 10, 100, 500, 999]], golden["sig_aps_800_800"], rtol=1e-12)
    c13 = models.Cosmology(omega_b=0.0483, omega_c=0.2589, omega_l=0.6914, H0=67.77)
    m13 = models.Corr21cm(cosmology=c13)
    m13._tables = model21._tables
    a1 = m13.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    fa = np.linspace(400.0, 800.0, 64)
    a2 = m13.angular_powerspectrum(np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :])
    assert np.allclose(a1.sum(), 1.5963772205823096e-09, rtol=1e-7)
    assert np.allclose(a2[400, 40, 40], 8.986790805379046e-13, rtol=1e-7)
    assert np.allclose(a2[200, 10, 40], 1.1939298801340165e-18, rtol=1e-7)


# ------------------------------------------------------------------ golden vectors
def test_cosmology_golden(golden):
    c = models.Cosmology()
    assert _rel(c.comoving_distance(golden["cosmo_z"]), golden["cosmo_chi"]) < 1e-14
    assert _rel(c.H(golden["cosmo_z"]), golden["cosmo_H"]) < 1e-15
    c13 = models.Cosmology(omega_b=0.0483, omega_c=0.2589, omega_l=0.6914, H0=67.77)
    assert _rel(c13.comoving_distance(golden["cosmo_z"]), golden["cosmo13_chi"]) < 1e-14


def test_spline_golden(golden):
    sp = models.Interpolater(golden["spl_xk"], golden["spl_yk"])
    assert np.abs(sp(golden["spl_xe"]) - golden["spl_ye"]).max() < 1e-14
    assert np.abs(sp.y2 - golden["spl_y2"]).max() < 1e-14
    lsp = models.LogInterpolater(golden["spl_xk"] + 0.5, np.exp(golden["spl_yk"]))
    assert _rel(lsp(np.abs(golden["spl_xe"]) + 0.25), golden["lspl_ye"]) < 1e-13


def test_21cm_tables_and_ps_golden(model21, golden):
    assert _rel(model21.ps_vv(golden["ps_k"]), golden["ps_vv"]) < 1e-14
    ix = np.ix_(golden["tab_rows"], golden["tab_cols"])
    for t, nm in zip(model21.tables(), ("dd", "dv", "vv")):
        assert _rel(t[ix], golden["tab_" + nm]) < 1e-14
    z = golden["m21_z"]
    assert _rel(model21.T_b(z), golden["m21_Tb"]) < 1e-15
    assert _rel(model21.growth_factor(z), golden["m21_D"]) < 1e-15
    assert _rel(model21.growth_rate(z), golden["m21_f"]) < 1e-15


@pytest.mark.parametrize("key,lmax,fkey,zromb,zwidth", [("cla_21cm_F8_l64_zromb0", 64, "f8", 0, None),
                                                        ("cla_21cm_F8_l64_zromb1", 64, "f8", 1, None),
                                                        ("cla_21cm_F8_l64_zromb3", 64, "f8", 3, None),
                                                        ("cla_21cm_F6n_l96_zromb3", 96, "f6", 3, None),
                                                        ("cla_21cm_F6n_l96_zromb2_zw", 96, "f6", 2, 1.0),
                                                        ("cla_21cm_F4_l16_zromb1", 16, "f4", 1, None)])
def test_clarray_21cm_golden(model21, golden, key, lmax, fkey, zromb, zwidth):
    cla = osk.clarray(model21.angular_powerspectrum, lmax, golden[fkey], zromb=zromb, zwidth=zwidth)
    assert _rel(cla, golden[key]) < 1e-14


@pytest.mark.parametrize("name,zromb", [("syn", 0), ("syn", 3), ("ups", 0), ("ups", 3)])
def test_clarray_foreground_golden(golden, name, zromb):
    m = models.FullSkySynchrotron() if name == "syn" else models.UnresolvedBackground()
    cla = osk.clarray(m.angular_powerspectrum, 64, golden["f8"], zromb=zromb)
    assert _rel(cla, golden["cla_%s_F8_l64_zromb%d" % (name, zromb)]) < 1e-15


def test_matrix_root_golden(golden):
    assert _rel(osk.matrix_root_manynull(golden["root_well_in"], truncate=False), golden["root_well_out"]) < 1e-15
    assert _rel(osk.matrix_root_manynull(golden["root_sing_in"], truncate=False), golden["root_sing_out"]) < 1e-15
    r, npos = osk.matrix_root_manynull(golden["root_rank3_in"])
    # quirk of the reference (nputil.py:92-96): on the eigen branch with truncate=True the root
    # comes back with a leading axis of length 1
    assert npos == 3 and r.shape == (1, 6, 3)
    assert np.abs(r[0] @ r[0].T - golden["root_rank3_in"]).max() < 1e-13
    assert np.array_equal(osk.matrix_root_manynull(np.zeros((5, 5)), truncate=False), golden["root_zero_out"])


def test_complex_std_normal_order(golden):
    """Real block first, then imaginary block (cora/util/nputil.py:125)."""
    v = osk.complex_std_normal((3, 5), rng=np.random.default_rng(7))
    assert np.array_equal(v, golden["csn_3x5_seed7"])
    r = np.random.default_rng(7)
    re, im = r.standard_normal((3, 5)), r.standard_normal((3, 5))
    assert np.array_equal(v, (re + 1j * im) / 2**0.5)


@pytest.mark.parametrize("key,cl,seed", [("alm_21cm_F4_l16_seed3", "cla_21cm_F4_l16_zromb1", 3),
                                         ("alm_21cm_F8_l64_seed4", "cla_21cm_F8_l64_zromb3", 4),
                                         ("alm_syn_F8_l64_seed5", "cla_syn_F8_l64_zromb0", 5)])
def test_mkfullsky_alms_golden(golden, key, cl, seed):
    a = osk.mkfullsky(golden[cl], 8, alms=True, rng=np.random.default_rng(seed))
    assert a.shape == golden[key].shape
    assert _rel(a, golden[key]) < 1e-15


def test_mkfullsky_legacy_rng_golden(golden):
    np.random.seed(1234)
    a = osk.mkfullsky(golden["cla_21cm_F4_l16_zromb1"], 8, alms=True)
    assert _rel(a, golden["alm_21cm_F4_l16_legacy1234"]) < 1e-15


def test_mkfullsky_shape_error():
    with pytest.raises(Exception, match="Correlation matrix is incorrect shape."):
        osk.mkfullsky(np.zeros((4, 3, 2)), 4)


def test_pack_alm_golden(golden):
    assert np.array_equal(osk.pack_alm(golden["pack_in"]), golden["pack_out"])
    assert np.array_equal(osk.unpack_alm(golden["pack_out"], 5), golden["unpack_out"])
    lmax = 5
    for l in range(lmax + 1):
        for m in range(l + 1):
            assert golden["pack_out"][sht.alm_index(l, m, lmax)] == golden["pack_in"][l, m]


# ------------------------------------------------------------------ HEALPix geometry known answers
def test_healpix_geometry():
    for nside in (1, 2, 4, 64):
        ri = healpix.ring_info(nside)
        assert ri["nphi"].sum() == 12 * nside * nside
        assert np.array_equal(ri["start"], np.concatenate([[0], np.cumsum(ri["nphi"])[:-1]]))
        assert np.allclose(ri["z"], -ri["z"][::-1], atol=1e-15)
        assert np.allclose(ri["z"] ** 2 + ri["sth"] ** 2, 1.0, atol=1e-15)
    theta, phi = healpix.pix2ang_ring(1)
    # nside = 1: pixel 0 at (acos 2/3, pi/4), pixel 4 at (pi/2, 0)  (pix2ang_ring convention)
    assert np.isclose(theta[0], np.arccos(2.0 / 3.0)) and np.isclose(phi[0], np.pi / 4)
    assert np.isclose(theta[4], np.pi / 2) and np.isclose(phi[4], 0.0)
    ri = healpix.ring_info(4)
    i = np.arange(1, 4)
    assert np.allclose(ri["z"][:3], 1 - i**2 / (3.0 * 16))
    assert np.allclose(ri["z"][3:12], 4.0 / 3 - 2 * np.arange(4, 13) / (3.0 * 4))
    assert np.allclose(ri["phi0"][3:7], [np.pi / 16, 0, np.pi / 16, 0])


# ------------------------------------------------------------------ synthesis (healpy stand-in)
@pytest.mark.parametrize("nside,lmax", [(1, 2), (2, 5), (4, 11), (8, 16), (8, 23)])
def test_alm2map_vs_bruteforce(nside, lmax):
    rng = np.random.default_rng(nside * 100 + lmax)
    n = (lmax + 1) * (lmax + 2) // 2
    alm = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = sht.alm2map_bruteforce(alm, nside, lmax)
    assert np.abs(sht.alm2map(alm, nside, lmax) - ref).max() < 1e-12 * ref.std()
    assert np.abs(sht.alm2map(alm, nside, lmax, impl="numpy") - ref).max() < 1e-12 * ref.std()


def test_alm2map_c_vs_numpy_medium():
    nside, lmax = 32, 95
    rng = np.random.default_rng(3)
    n = (lmax + 1) * (lmax + 2) // 2
    alm = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a, b = sht.alm2map(alm, nside, lmax), sht.alm2map(alm, nside, lmax, impl="numpy")
    assert np.abs(a - b).max() < 1e-12 * a.std()


def test_alm2map_single_modes():
    nside, lmax = 8, 4
    theta, phi = healpix.pix2ang_ring(nside)
    n = (lmax + 1) * (lmax + 2) // 2
    a = np.zeros(n, dtype=np.complex128)
    a[0] = 2.5 + 7j
    assert np.allclose(sht.alm2map(a, nside, lmax), 2.5 / np.sqrt(4 * np.pi), atol=1e-14)
    a[:] = 0
    a[1] = -1.25
    assert np.allclose(sht.alm2map(a, nside, lmax), -1.25 * np.sqrt(3 / (4 * np.pi)) * np.cos(theta), atol=1e-14)
    a[:] = 0
    a[lmax + 1] = 0.5 - 0.75j
    ref = -np.sqrt(3 / (8 * np.pi)) * 2 * ((0.5 - 0.75j) * np.exp(1j * phi)).real * np.sin(theta)
    assert np.allclose(sht.alm2map(a, nside, lmax), ref, atol=1e-14)


def _lambda_mp(mp, lmax, m, x, ls):
    """lambda_lm(x) in 50-digit arithmetic: closed-form lambda_mm, then the three-term recurrence."""
    x = mp.mpf(x)
    sth = mp.sqrt((1 - x) * (1 + x))
    lam = mp.sqrt(mp.mpf(1) / (4 * mp.pi))
    for k in range(1, m + 1):
        lam *= mp.sqrt(mp.mpf(2 * k + 1) / (2 * k)) * sth
    if m & 1:
        lam = -lam
    prev, out = mp.mpf(0), {}
    alpha_prev = None
    for l in range(m, lmax + 1):
        if l in ls:
            out[l] = lam
        l1 = l + 1
        alpha = mp.sqrt(mp.mpf(4 * l1 * l1 - 1) / (l1 * l1 - m * m))
        nxt = alpha * (x * lam - (prev / alpha_prev if alpha_prev is not None else 0))
        prev, lam, alpha_prev = lam, nxt, alpha
    return out


def test_lambda_lm_high_l_vs_mpmath():
    """Spot values of the normalised Legendre functions at l, m ~ 2048 on polar, mid and equatorial
    rings of nside = 1024 against 50-digit arithmetic (the double-precision scaled recurrence must
    neither under/overflow nor lose accuracy), plus direct mpmath.legenp values at small l."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    ri = healpix.ring_info(1024)
    for m, pair, ls in ((2048, 2047, [2048]), (1000, 2047, [1000, 1500, 2048]), (1000, 700, [1500, 2048]),
                        (0, 0, [10, 2048]), (2, 5, [2, 100, 2048]), (300, 100, [2000]), (2047, 1000, [2047, 2048])):
        x = float(ri["z"][pair])
        lam = sht.lambda_lm(2048, m, x)
        ref = _lambda_mp(mp, 2048, m, x, set(ls))
        for l in ls:
            r = float(ref[l])
            # the fp64 three-term recurrence loses ~l^2 eps near the poles (x -> 1: second-difference
            # cancellation), 1e-11..1e-10 at l = 2048 on the first ring; libsharp/healpy share this
            assert abs(lam[l - m] - r) <= 1e-9 * max(abs(r), 1e-300), (l, m, pair, lam[l - m], r)
    # independent definition (hypergeometric P_l^m) where mpmath's legenp converges
    for m, l, x in ((0, 7, 0.3), (3, 9, -0.62), (20, 40, 0.11), (5, 60, 0.9)):
        norm = mp.sqrt(mp.mpf(2 * l + 1) / (4 * mp.pi) * mp.factorial(l - m) / mp.factorial(l + m))
        r = float(norm * mp.legenp(l, m, mp.mpf(x), type=2))
        v = sht.lambda_lm(l, m, x)[l - m]
        assert abs(v - r) <= 1e-13 * abs(r), (l, m, x, v, r)


def test_alm2map_linearity_and_m0_imag():
    nside, lmax = 16, 40
    rng = np.random.default_rng(9)
    n = (lmax + 1) * (lmax + 2) // 2
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ma, mb, mab = sht.alm2map(a, nside, lmax), sht.alm2map(b, nside, lmax), sht.alm2map(2 * a - 3 * b, nside, lmax)
    assert np.abs(mab - (2 * ma - 3 * mb)).max() < 1e-12 * mab.std()
    a2 = a.copy()
    a2[: lmax + 1] = a2[: lmax + 1].real  # imaginary part of a_l0 never contributes
    assert np.abs(sht.alm2map(a2, nside, lmax) - ma).max() < 1e-13 * ma.std()


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32 with 10 rounds (Salmon et al. SC'11) pin the device-stream oracle."""
    from oracle import philox

    kats = [
        ((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), "d16cfe09 94fdcceb 5001e420 24126ea1"),
    ]
    for ctr, key, want in kats:
        r = philox.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert " ".join("%08x" % int(v[0]) for v in r) == want


def test_device_stream_oracle_layout_and_moments():
    from oracle import philox

    F, lmax = 6, 30
    g = philox.device_normals(12345, lmax, F)
    assert g.size == 2 * F * (lmax + 1) * (lmax + 2) // 2
    # element (l, c, nu', m) sits at F l (l+1) + c F (l+1) + nu' (l+1) + m
    l, c, nu, m = 17, 1, 4, 9
    a, b = philox.normal_pairs(12345, l, F, nu, m)
    assert g[F * l * (l + 1) + c * F * (l + 1) + nu * (l + 1) + m] == (b if c else a)
    assert abs(g.mean()) < 5 / np.sqrt(g.size) and abs(g.var() - 1) < 5 * np.sqrt(2 / g.size)


def _device_boxmuller_model(r0, r1, r2, r3):
    """numpy model of rng_boxmuller_bits (cora_amd/csrc/rng_dev.h): the same bit constructions, table look-ups
    (tables parsed from rng_tab.inc) and polynomials, plain double arithmetic (no fma)."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "cora_amd", "csrc", "rng_tab.inc")).read()
    pairs = re.findall(r"\{(-?0x[0-9a-f.]+p[+-]\d+), (-?0x[0-9a-f.]+p[+-]\d+)\}", txt)
    tab = np.array([[float.fromhex(a), float.fromhex(b)] for a, b in pairs])
    assert tab.shape == (257 + 256, 2)
    lg, sc = tab[:257], tab[257:]
    U = np.uint64
    r0, r1, r2, r3 = (v.astype(U) for v in (r0, r1, r2, r3))
    k = (r0 << U(20)) | (r1 >> U(12))
    u1 = ((U(0x3FF) << U(52)) | k).view(np.float64) - (1.0 - 2.0**-53)
    ub = u1.view(U)
    uh = ub >> U(32)
    mh = uh & U(0xFFFFF)
    idx = ((mh + U(0x800)) >> U(12)).astype(np.int64)
    m = (((mh | U(0x3FF00000)) << U(32)) | (ub & U(0xFFFFFFFF))).view(np.float64)
    e = (uh >> U(20)).astype(np.int64) - 1023 + (idx > 106)
    rr = m * lg[idx, 0] - 1.0
    q = rr * (-2 / 5) + 0.5
    q = q * rr - 2 / 3
    q = q * rr + 1.0
    t = e * (-2 * 0.69314718055994530942) + ((rr * rr) * q + (rr * -2.0 + lg[idx, 1]))
    rad = np.sqrt(t)
    j = (r2 >> U(24)).astype(np.int64)
    w = ((r2 & U(0xFFFFFF)) << U(28)) | (r3 >> U(4))
    tw = ((U(0x3FF) << U(52)) | w).view(np.float64) - 1.5
    c = 2 * np.pi / 256
    x = tw * c + 2.0**-53 * c
    z = x * x
    sx = (x * z) * (z * (1 / 120) - 1 / 6) + x
    cx = z * ((z * (-1 / 720) + 1 / 24) * z - 0.5) + 1.0
    a, b = rad * sc[j, 0], rad * sc[j, 1]
    return a * cx - b * sx, a * sx + b * cx


def test_device_boxmuller_algorithm_matches_stream_spec(monkeypatch):
    """The bit-level Box-Muller of the kernels (tables of rng_tab.inc, split of the mantissa at ~sqrt2, sector
    rotation) evaluates the stream that oracle/philox.py specifies: random words plus the corner cases of the
    mantissa split, u1 -> 0, u1 -> 1 and the table's last entries."""
    from oracle import philox

    rng = np.random.default_rng(1)
    n = 400000
    r = [rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32) for _ in range(4)]
    r[0][:10] = [0, 0, 0xFFFFFFFF, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x6A09E667, 0x6A09E668, 0x6A000000, 0x6AFFFFFF]
    r[1][:10] = [0, 0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0xFFFFFFFF, 0xF3BCC908, 0, 0, 0xFFFFFFFF]
    r[2][:4] = [0, 0xFFFFFFFF, 0x00FFFFFF, 0x01000000]
    r[3][:4] = [0, 0xFFFFFFFF, 0xFFFFFFFF, 0]
    a, b = _device_boxmuller_model(*r)
    monkeypatch.setattr(philox, "philox4x32_10", lambda *args: tuple(r))
    oa, ob = philox.boxmuller_counter(0, 0, 0)
    # without fma the model's r = m / c - 1 carries 1e-16 absolute, i.e. 1e-16 / radius in the normal: 4e-15 here
    assert np.abs(a - oa).max() < 4e-15 and np.abs(b - ob).max() < 4e-15, (np.abs(a - oa).max(), np.abs(b - ob).max())
    # the radius is accurate RELATIVE to itself where u1 -> 1 (words 2, 3 of the corner cases)
    for i in (2, 3):
        assert abs(np.hypot(a[i], b[i]) / np.hypot(oa[i], ob[i]) - 1) < 1e-14


def test_map2alm_oracle_against_bruteforce_and_adjointness():
    """The analysis oracle (n1: healpy.map2alm restated) vs an independent scipy Y_lm quadrature,
    <S a, x> = <a, A x> adjointness with the synthesis oracle, and convergence of the Jacobi iterations."""
    nside, lmax = 8, 16
    rng = np.random.default_rng(11)
    n = (lmax + 1) * (lmax + 2) // 2
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a[: lmax + 1] = a[: lmax + 1].real
    m = sht.alm2map(a, nside, lmax)
    w = sht.ring_weights(nside)
    # weights integrate the monopole exactly and are symmetric small corrections
    ri = healpix.ring_info(nside)
    cnt = ri["nphi"][: 2 * nside].astype(float) * 2
    cnt[-1] /= 2
    assert abs((w * cnt).sum() * 4 * np.pi / (12 * nside**2) - 4 * np.pi) < 1e-12
    assert np.abs(w - 1).max() < 0.2
    b = sht.map2alm_adjoint(m, nside, lmax, w)
    assert np.abs(b - sht.map2alm_bruteforce(m, nside, lmax, w)).max() < 1e-13
    x = rng.standard_normal(m.size)
    Ax = sht.map2alm_adjoint(x, nside, lmax, None)
    mm = np.concatenate([np.full(lmax + 1 - k, k) for k in range(lmax + 1)])
    wgt = np.where(mm == 0, 1.0, 2.0)
    assert abs(np.dot(m, x) * 4 * np.pi / m.size - np.sum(wgt * (a.conj() * Ax).real)) < 1e-12 * np.abs(m).sum()
    errs = [np.abs(sht.map2alm(m, nside, lmax, True, it) - a).max() for it in (0, 1, 2, 3)]
    assert all(e1 < 0.2 * e0 for e0, e1 in zip(errs, errs[1:])) and errs[2] < 1e-3
    # band limit well inside the grid: two iterations are already at the 1e-9 level
    lmax2 = 8
    n2 = (lmax2 + 1) * (lmax2 + 2) // 2
    a2 = rng.standard_normal(n2) + 1j * rng.standard_normal(n2)
    a2[: lmax2 + 1] = a2[: lmax2 + 1].real
    assert np.abs(sht.map2alm(sht.alm2map(a2, nside, lmax2), nside, lmax2) - a2).max() < 1e-8


def test_bilinear_interp_c_equals_numpy():
    """The C/OpenMP restatement of bilinearmap.interp equals the numpy one bit for bit (incl. the clips)."""
    from oracle import models

    rng = np.random.default_rng(2)
    tab = rng.standard_normal((37, 53))
    x = rng.uniform(-3, 40, size=(5, 1, 7))
    y = rng.uniform(-3, 56, size=(1, 4, 7))
    a = models.bilinear_interp(tab, x, y)
    xb, yb = np.broadcast_arrays(x, y)
    # stay away from the last 1e-5 of the table, where the reference itself would read out of bounds
    ok = (xb < 36 - 1e-5) & (yb < 52 - 1e-5)
    b = models.bilinear_interp_numpy(tab, np.where(ok, xb, 0.0), np.where(ok, yb, 0.0))
    assert a.shape == (5, 4, 7) and np.array_equal(a[ok], b[ok])


def _xi_model(r):
    r = np.asarray(r, dtype=np.float64)
    return np.exp(-r / 60.0) * np.cos(r / 35.0) / (1.0 + (r / 15.0) ** 2)


def test_corr_to_clarray_oracle_matches_reference_vectors():
    """oracle/corrfunc.py against outputs of the reference's own corr_to_clarray / legendre_array
    (tests/golden/make_golden_corrfunc.py): Gauss-Legendre nodes, bin half-widths, xwidth, normalisation."""
    from oracle import corrfunc as ocf

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corrfunc_vectors.npz"))
    assert np.abs(ocf.legendre_array(12, g["legendre_l12_mu"]) - g["legendre_l12"]).max() < 1e-14
    xa = g["xarray"]
    for tag, lmax, kw in (("l40_xromb2_q2", 40, dict(xromb=2, q=2)), ("l40_xromb0_q3", 40, dict(xromb=0, q=3)),
                          ("l24_xromb1_xw10", 24, dict(xromb=1, q=2, xwidth=10.0))):
        got = ocf.corr_to_clarray(_xi_model, lmax, xa, **kw)
        ref = g["cl_" + tag]
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), tag


def test_spin2_harmonics_against_eth_operator_definition():
    """W_lm / X_lm of the polarisation oracle against the definition of the spin-weighted harmonics
    (+-2)Y_lm = sqrt((l-2)!/(l+2)!) eth^2 Y_lm / ethbar^2 Y_lm (Zaldarriaga & Seljak 1997), with eth applied by
    high-precision numerical differentiation of mpmath's scalar Y_lm:  (2Y + -2Y)/2 = W e^{im phi}, (2Y - -2Y)/2 = -X e^{im phi}."""
    import mpmath as mp

    mp.mp.dps = 25

    def eth(f, s, sign):
        def g(th, ph):
            h = lambda t, p: mp.sin(t) ** (-sign * s) * f(t, p)   # noqa: E731
            dth = mp.diff(lambda t: h(t, ph), th)
            dph = mp.diff(lambda p: h(th, p), ph)
            return -mp.sin(th) ** (sign * s) * (dth + sign * 1j / mp.sin(th) * dph)
        return g

    th, ph = mp.mpf("0.9"), mp.mpf("0.4")
    for l, m in ((2, 1), (5, 3), (4, 0)):
        N = mp.sqrt(mp.factorial(l - 2) / mp.factorial(l + 2))
        f0 = lambda t, p: mp.spherharm(l, m, t, p)   # noqa: E731
        yp = N * eth(eth(f0, 0, +1), 1, +1)(th, ph)              # eth eth Y (spins 0 -> 1 -> 2)
        ym = N * eth(eth(f0, 0, -1), -1, -1)(th, ph)             # ethbar ethbar Y (spins 0 -> -1 -> -2)
        e = mp.e ** (1j * m * ph)
        W, X = sht.spin2_wx(l, m, np.array([0.9]))
        assert abs(complex((yp + ym) / 2 / e) - W[0]) < 1e-10 and abs(complex((yp - ym) / 2 / e) + X[0]) < 1e-10, (l, m)


def test_spin2_synthesis_oracle_vs_bruteforce():
    """The C spin-2 Legendre part + ring FFTs == the definition-level sum over pixels; E-only input gives
    Q/U maps whose pure-B projection vanishes is not asserted here (no analysis for spin 2) - linearity is."""
    nside, lmax = 4, 9
    rng = np.random.default_rng(3)
    n = (lmax + 1) * (lmax + 2) // 2
    e = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    e[: lmax + 1] = e[: lmax + 1].real
    b[: lmax + 1] = b[: lmax + 1].real
    q, u = sht.alm2map_spin2(e, b, nside, lmax)
    qb, ub = sht.alm2map_spin2_bruteforce(e, b, nside, lmax)
    assert np.abs(q - qb).max() < 1e-13 * np.abs(qb).max() and np.abs(u - ub).max() < 1e-13 * np.abs(ub).max()
    # l < 2 carries no polarisation; swapping E -> B, B -> -E rotates (Q, U) -> (U, -Q)... by 45 degrees: (Q,U) -> (-U, Q)
    e2, b2 = e.copy(), b.copy()
    for m in range(2):
        for l in range(m, 2):
            e2[sht.alm_index(l, m, lmax)] = 7.0
            b2[sht.alm_index(l, m, lmax)] = -3.0
    q2, u2 = sht.alm2map_spin2(e2, b2, nside, lmax)
    assert np.array_equal(q2, q) and np.array_equal(u2, u)
    qr, ur = sht.alm2map_spin2(-b, e, nside, lmax)         # (E, B) -> (-B, E) is a rotation of the polarisation by 45 deg
    assert np.abs(qr + u).max() < 1e-13 * np.abs(u).max() and np.abs(ur - q).max() < 1e-13 * np.abs(q).max()


def test_spin2_analysis_oracle_definition_and_round_trip():
    """oracle.sht.map2alm_spin2*: the ring-sum quadrature equals the definition-level pixel sum with scipy's
    Y_lm-based W / X, and with ring weights + Jacobi refinements it inverts the spin-2 synthesis oracle."""
    from oracle import sht

    nside, lmax = 8, 12
    rng = np.random.default_rng(1)
    nalm = (lmax + 1) * (lmax + 2) // 2

    def rnd():
        a = rng.standard_normal(nalm) + 1j * rng.standard_normal(nalm)
        for m in range(lmax + 1):
            for l in range(m, min(lmax, 1) + 1):
                a[sht.alm_index(l, m, lmax)] = 0.0
        a[: lmax + 1] = a[: lmax + 1].real
        return a

    e, b = rnd(), rnd()
    q, u = sht.alm2map_spin2(e, b, nside, lmax)
    w = sht.ring_weights(nside)
    e1, b1 = sht.map2alm_spin2_adjoint(q, u, nside, lmax, w)
    e2, b2 = sht.map2alm_spin2_bruteforce(q, u, nside, lmax, w)
    assert np.abs(e1 - e2).max() < 1e-13 and np.abs(b1 - b2).max() < 1e-13
    e3, b3 = sht.map2alm_spin2(q, u, nside, lmax, True, 3)
    assert np.abs(e3 - e).max() < 1e-6 and np.abs(b3 - b).max() < 1e-6
    # a pure-E sky analysed as (Q, U) -> (-U, Q) (45 degree rotation) comes back as pure B
    e4, b4 = sht.map2alm_spin2(-u, q, nside, lmax, True, 3)
    assert np.abs(e4 + b).max() < 1e-6 and np.abs(b4 - e).max() < 1e-6


@pytest.mark.parametrize("nside,lmax", [(1, 2), (2, 5), (4, 11), (8, 23), (16, 32), (32, 95)])
def test_c_ring_stage_equals_the_numpy_definition(nside, lmax):
    """oracle_ring_synth (C/OpenMP: the ring stage of the timed CPU baseline) == oracle.sht.ring_synthesis ring by
    ring - power-of-two rings (radix-2) and every other length (Bluestein), aliased (lmax >= nphi) and not."""
    from oracle import sht

    rng = np.random.default_rng(nside + lmax)
    n = (lmax + 1) * (lmax + 2) // 2
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = sht.alm2map(a, nside, lmax)
    got = sht.alm2map(a, nside, lmax, rings_c=True)
    assert np.abs(got - ref).max() <= 1e-13 * ref.std()


# ------------------------------------------------------------------ numpy's seeded normal stream (oracle/npnormal.py)
def test_zig_tables_are_numpys_own():
    """cora_amd/csrc/zig_tab.inc (what the kernels and the oracle use) equals the tables inside numpy's shipped library."""
    import os
    import sys

    from oracle import npnormal

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_zig_tabs

    try:
        ki, wi, fi = gen_zig_tabs.tables()
    except (OSError, KeyError) as e:
        pytest.skip("numpy's libnpyrandom.a not readable here: %s" % e)
    gen_zig_tabs.check_construction(ki, wi, fi)
    oki, owi, ofi = npnormal.tables()
    assert [int(v) for v in ki] == oki
    assert np.array_equal(np.array(owi), wi) and np.array_equal(np.array(ofi), fi)


@pytest.mark.parametrize("seed", [0, 3, 12345])
def test_npnormal_oracle_is_numpys_stream(seed):
    """The restatement of PCG64 + ziggurat equals numpy's Generator value by value - fast path, wedge and tail samples -
    and consumes the same number of raw draws (the generator's own state afterwards); advance = bit_generator.advance."""
    from oracle import npnormal as o

    rng = np.random.default_rng(seed)
    s, inc = o.state_of(rng)
    n = 150_000
    stats = {}
    v, nraw = o.standard_normal(s, inc, n, stats)
    ref = rng.standard_normal(n)
    assert np.array_equal(v.view(np.uint64), ref.view(np.uint64))
    assert stats["wedge_accept"] > 500 and stats["wedge_reject"] > 500 and stats["tail"] > 10
    assert o.advance(s, inc, nraw) == o.state_of(rng)[0]
    raw, s5 = o.raw_stream(s, inc, 5)
    assert np.array_equal(raw, np.random.PCG64(seed).random_raw(5))
    for d in (0, 1, 4097, 10**12 + 7, 2**64 - 3):
        bg = np.random.PCG64(seed)
        bg.advance(d)
        assert o.advance(s, inc, d) == o.state_of(bg)[0]


def test_glibc_log1p_restatement_matches_libm():
    """The operation-by-operation log1p the kernels run for tail samples is libm's: every range of the argument."""
    import math
    import random

    from oracle import npnormal as o

    random.seed(1)
    for i in range(60_000):
        k = random.getrandbits(53)
        if i % 3 == 0:
            k >>= random.randrange(0, 50)
        if i % 7 == 0:
            k = (1 << 53) - 1 - (k >> random.randrange(0, 50))
        u = k * (1.0 / 9007199254740992.0)
        assert o.glibc_log1p(-u) == math.log1p(-u), (k, u)


@pytest.mark.parametrize("seed,n,R,T,margin", [(0, 30000, 16, 8, 1.0225), (5, 30000, 7, 3, 1.0225), (2, 20000, 30, 2, 1.0225),
                                               (9, 20000, 16, 4, 1.0), (4, 40000, 64, 16, 1.0225)])
def test_npnormal_parallel_model_is_numpys_stream(seed, n, R, T, margin):
    """The block / scan / emit decomposition of the device (run scan by integer add, boundary state = positions consumed,
    repair of blocks entered with k >= 2, several rounds when the raw range was short) gives numpy's values and count."""
    from oracle import npnormal as o
    from oracle import npnormal_model as m

    rng = np.random.default_rng(seed)
    s, inc = o.state_of(rng)
    stats = {}
    v, nraw = m.parallel_normals(s, inc, n, R=R, T=T, margin=margin, stats=stats)
    ref = rng.standard_normal(n)
    assert np.array_equal(v.view(np.uint64), ref.view(np.uint64))
    assert o.advance(s, inc, nraw) == o.state_of(rng)[0]
    if margin < 1.01:
        assert stats["rounds"] >= 2


def test_pcg64_advance_abi_is_numpys_advance():
    """corahip_pcg64_advance (host arithmetic of the library, no GPU) against numpy's bit_generator.advance."""
    from cora_amd import _lib
    from oracle import npnormal as o

    for seed in (0, 7):
        s, inc = o.state_of(np.random.PCG64(seed))
        for d in (0, 1, 5, 4096, 10**9 + 7, 2**63 + 12345):
            bg = np.random.PCG64(seed)
            bg.advance(d)
            assert _lib.pcg64_advance(s, inc, d) == o.state_of(bg)[0] == o.advance(s, inc, d)


# ------------------------------------------------------------------ numpy's LEGACY normal stream (oracle/mtlegacy.py)
@pytest.mark.parametrize("seed", [1234, 7])
def test_legacy_oracle_is_numpys_stream(seed):
    """MT19937 + tempering + the polar method with its cached value, restated, equals np.random.standard_normal bit
    for bit and leaves the same global state (key, pos, has_gauss, gauss)."""
    from oracle import mtlegacy as m

    np.random.seed(seed)
    ls = m.LegacyStream()
    ref = np.random.standard_normal(3001)
    got = ls.standard_normal(3001)
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))
    st = np.random.get_state(legacy=False)
    assert list(st["state"]["key"]) == ls.key and st["state"]["pos"] == ls.pos
    assert st["has_gauss"] == ls.has_gauss == 1 and st["gauss"] == ls.gauss


def test_mt19937_jump_polynomials():
    """The minimal polynomial of MT19937 (Berlekamp-Massey on numpy's output: degree 19937, 135 terms) annihilates the
    output bits of another seed, x^J mod phi applied to a window equals plain stepping, and the first entries of the
    generated table the kernels use (cora_amd/csrc/mt_jump.inc) are x^(S 2^k - 1) mod phi."""
    import os
    import re

    from oracle import mtlegacy as m

    phi = m.minimal_polynomial()
    assert phi.bit_length() - 1 == m.DEG and bin(phi).count("1") == 135
    w = np.random.RandomState(99).randint(0, 2**32, size=2 * m.DEG + 300, dtype=np.uint64)
    bits = 0
    for i, v in enumerate(w):
        bits |= ((int(v) >> 7) & 1) << i
    for t0 in (0, 17, m.DEG + 100):
        assert bin((bits >> t0) & phi).count("1") % 2 == 0
    np.random.seed(1234)
    key = [int(v) for v in np.random.get_state(legacy=False)["state"]["key"]]
    J = 70000
    g1 = m.x_pow_mod(J - 1, phi)
    x = m.extend(key, m.DEG + 1)
    out = [0] * m.N
    i = 0
    while g1:
        if g1 & 1:
            for k in range(m.N):
                out[k] ^= x[i + 1 + k]
        g1 >>= 1
        i += 1
    assert out == m.step_window(key, J)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "cora_amd", "csrc", "mt_jump.inc")).read()
    seg = int(re.search(r"#define MT_SEG_LOG2 (\d+)", txt).group(1))
    words = [int(t, 16) for t in re.findall(r"0x([0-9a-f]{8})u,", txt)]
    npoly = int(re.search(r"#define MT_NPOLY (\d+)", txt).group(1))
    assert len(words) == npoly * 624
    # radix-8 table (round 6): entry 7 j + q - 1 = x^(q 8^j S - 1); spot checks on both levels by square-and-multiply
    assert npoly == 7 * int(re.search(r"#define MT_NLEV (\d+)", txt).group(1))
    for k, e in ((0, 1), (1, 2), (2, 3), (6, 7), (7, 8), (11, 5 * 8)):
        got = sum(v << (32 * j) for j, v in enumerate(words[624 * k:624 * (k + 1)]))
        assert got == m.x_pow_mod(e * (1 << seg) - 1, phi), k


def test_glibc_log_restatement_is_the_hosts_log():
    """oracle.mtlegacy.glibc_log_fma - glibc's table-driven log in the evaluation order of its FMA build, the sequence
    cora_amd/csrc/mtlegacy.hip runs for the legacy normals - against math.log (the libm numpy calls) on this host:
    both branches (table + degree-5 polynomial; degree-11 polynomial around 1), r2-like arguments down to 2^-104.
    Skipped where the host's libm is another routine (no FMA, another libc): the device values are then only within
    4 ulp of numpy's (tests/test_gpu_npnormal.py takes the same switch)."""
    import math
    import random

    from oracle import mtlegacy

    probe = [0.9375, 0.99999, 0.5, 0.1234567, 3e-7, 2.0**-104, 0.7071, 0.96, 0.30103]
    if not all(mtlegacy.glibc_log_fma(x) == math.log(x) for x in probe):
        pytest.skip("this host's log is not glibc's FMA build")
    random.seed(11)
    xs = [random.random() for _ in range(6000)] + [1.0 - random.random() * 2.0**-4 for _ in range(2000)]
    xs += [random.random() * 10.0 ** (-random.randint(1, 30)) for _ in range(1000)] + [1.0 - 2.0**-53, 0.9375, 2.0**-104]
    bad = [x for x in xs if x > 0.0 and mtlegacy.glibc_log_fma(x) != math.log(x)]
    assert not bad, bad[:5]
    ln2hi, ln2lo, A, B, T = mtlegacy.glibc_log_tables()
    assert ln2hi + ln2lo == math.log(2.0) and abs(A[0] + 0.5) < 1e-15 and B[0] == -0.5
    for i in range(128):
        assert abs(math.log(1.0 / T[2 * i]) - T[2 * i + 1]) < 1e-9


def test_glibc_exp_restatement_is_the_hosts_exp():
    """oracle.npnormal.glibc_exp_fma - glibc's table-driven exp in the evaluation order of its FMA build, the sequence
    cora_amd/csrc/npnormal.hip runs in the wedge test of numpy's ziggurat (numpy/random/src/distributions/distributions.c
    ``random_standard_normal``, reached from cora/util/nputil.py:125) - against math.exp (the libm numpy calls) on this
    host: the wedge test's whole argument range -x^2 / 2 in [-6.7, 0), every table index, tiny arguments.  Skipped where
    the host's libm is another routine (no FMA, another libc)."""
    import math
    import random

    from oracle import npnormal

    probe = [-0.5, -6.6, -1e-3, -3.21, -0.6931471805599453, -2.0**-30, -5.0]
    if not all(npnormal.glibc_exp_fma(x) == math.exp(x) for x in probe):
        pytest.skip("this host's exp is not glibc's FMA build")
    random.seed(12)
    xs = [-6.7 * random.random() for _ in range(8000)] + [-random.random() * 2.0 ** (-random.randint(1, 60)) for _ in range(1500)]
    xs += [-0.5 * (3.6541528853610088 * random.random()) ** 2 for _ in range(2000)] + [-(k + 0.5) * math.log(2.0) / 128.0 for k in range(1300)]
    xs += [-2.0**-54, -2.0**-55, -0.0, 0.0]
    bad = [x for x in xs if npnormal.glibc_exp_fma(x) != math.exp(x)]
    assert not bad, bad[:5]
    invln2n, shift, neghi, neglo, C, T = npnormal.glibc_exp_tables()
    assert abs(invln2n * math.log(2.0) - 128.0) < 1e-12 and shift == 1.5 * 2.0**52 and abs(C[0] - 0.5) < 1e-12
    assert T[0] == 0 and T[1] == 0x3FF0000000000000


# ------------------------------------------------------------------ EoR21cm (cora/signal/corr21cm.py:333-385)
@pytest.fixture(scope="module")
def eor_golden():
    """Outputs of the reference's own EoR21cm / Cmb / TestF (tests/golden/make_golden_eor.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "eor_vectors.npz"))


def test_eor21cm_oracle_vs_reference_outputs(model21, eor_golden):
    """The oracle's EoR21cm (T_b of Santos et al. 2009, bias 3) against the reference's own class: T_b, the aps on the
    150-200 MHz band, and clarray as Sky3d.getsky() drives it (zromb 0 / 1 / 3, and a 100-200 MHz band with zwidth)."""
    g = eor_golden
    eor = models.EoR21cm(share=model21)
    assert _rel(eor.T_b(g["z"]), g["T_b"]) < 1e-15
    fa = g["fa"]
    aps1 = eor.angular_powerspectrum(np.arange(1000.0), 180.0, 180.0)
    assert _rel(aps1, g["aps_180_180"]) < 1e-13
    row = eor.angular_powerspectrum(np.full((1, 1), 200.0), fa[:, None], fa[None, :])
    assert _rel(row, g["aps2_l200"]) < 1e-13
    for zr in (0, 1, 3):
        cla = osk.clarray(eor.angular_powerspectrum, 64, g["f8"], zromb=zr)
        assert _rel(cla, g["cla_eor_F8_l64_zromb%d" % zr]) < 1e-13, zr
    cla = osk.clarray(eor.angular_powerspectrum, 40, g["f6"], zromb=2, zwidth=1.0)
    assert _rel(cla, g["cla_eor_F6_l40_zromb2_zw1"]) < 1e-13
