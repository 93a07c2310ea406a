"""Host-side logic of cora_amd (no GPU): API surface mirrored from cora, error behaviour,
the C ABI library loads and exports every symbol of include/corahip.h."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from cora_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "corahip.h")).read()
    declared = set(re.findall(r"\b(corahip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.corahip_abi_version() == 1
    assert isinstance(lib.corahip_last_error(), bytes)


def test_c_shard_plan_matches_python_plan():
    """corahip_shard_plan (what a caput / mpi4py caller of the C ABI uses, cora/core/skysim.py:97-110) and
    cora_amd.parallel.shard_plan (what the torch.distributed path uses) are the same split: host arithmetic only."""
    from cora_amd import _lib
    from cora_amd.parallel import shard_plan

    class Shard(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in ("l_lo", "l_hi", "l_shard", "l_pad", "nu0", "nnu", "rows_exchange", "L")]

    lib = _lib.load()
    for L, F, world in ((2049, 256, 8), (4097, 1024, 8), (129, 16, 2), (65, 8, 3), (10, 7, 3), (3, 2, 4)):
        for r in range(world):
            c = Shard()
            assert lib.corahip_shard_plan(L, F, r, world, ctypes.byref(c)) == 0
            p = shard_plan(L, F, r, world)
            assert (c.l_lo, c.l_hi, c.l_shard, c.l_pad, c.nu0, c.nnu, c.L) == (p.l_lo, p.l_hi, p.l_shard, p.l_pad, p.nu0, p.nnu, L)
            assert c.rows_exchange == (1 if F % world == 0 else 0)
    assert lib.corahip_shard_plan(10, 4, 2, 2, ctypes.byref(Shard())) == -1        # rank >= world: CORAHIP_EINVAL
    assert b"invalid argument" in lib.corahip_last_error()


def test_no_gpu_fails_loudly():
    """Without a GPU the compute entry points raise; they never fall back to a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cora_amd import CoraHipError
    from cora_amd.core import skysim
    from cora_amd.util import hputil, nputil

    with pytest.raises(CoraHipError):
        skysim.mkfullsky(np.ones((3, 2, 2)), 4)
    with pytest.raises(CoraHipError):
        skysim.clarray(lambda l, a, b: l + a + b, 8, np.array([1.0, 2.0]), zromb=0)
    with pytest.raises(CoraHipError):
        nputil.matrix_root_manynull(np.eye(3))
    with pytest.raises(CoraHipError):
        hputil.sphtrans_inv_real(np.zeros((3, 3), dtype=complex), 2)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: no module under cora_amd/ may reference it."""
    for dp, _, files in os.walk(os.path.join(ROOT, "cora_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), os.path.join(dp, f)
                assert "liboracle" not in txt, os.path.join(dp, f)


# ------------------------------------------------------------------ maps / frequencies
def test_map3d_frequencies(golden):
    from cora_amd.core import maps

    m = maps.Map3d()
    assert np.array_equal(m.frequencies, golden["map3d_freq_default"])
    m.nu_lower, m.nu_upper, m.nu_num = 400.0, 800.0, 32
    assert np.array_equal(m.frequencies, golden["map3d_freq_400_800_32"])
    assert m.nu_num == 32 and np.array_equal(m.nu_pixels, m.frequencies)
    m.frequencies = np.array([500.0, 600.0])
    assert m.nu_num == 2
    c = maps.Map3d.like_map(m)
    assert np.array_equal(c.frequencies, m.frequencies) and c.nu_lower == 400.0


def test_nside_setter():
    from cora_amd.core import maps

    m = maps.Map2d()
    m.nside = 64
    assert m.nside == 64
    with pytest.raises(Exception, match="Not a valid value of nside."):
        m.nside = 48
    s = maps.Sky3d()
    with pytest.raises(Exception, match="Not implemented in base class."):
        s.angular_powerspectrum(1, 2, 3)
    assert s.oversample == 3 and s._lmax() == 3 * s.nside - 1


# ------------------------------------------------------------------ hputil / nputil host parts
def test_pack_unpack_alm(golden):
    from cora_amd.util import hputil

    assert np.array_equal(hputil.pack_alm(golden["pack_in"]), golden["pack_out"])
    assert np.array_equal(hputil.unpack_alm(golden["pack_out"], 5), golden["unpack_out"])
    full = hputil.unpack_alm(golden["pack_out"], 5, fullm=True)
    assert full.shape == (6, 11)
    assert np.array_equal(hputil.pack_alm(full), golden["pack_out"])
    for lmax, nside in golden["nside_for_lmax"]:
        assert hputil.nside_for_lmax(int(lmax)) == nside
    with pytest.raises(Exception, match="a_lm array wrong shape."):
        hputil.sphtrans_inv_real(np.zeros((3, 4), dtype=complex), 2)


def test_complex_std_normal(golden):
    from cora_amd.util import nputil

    assert np.array_equal(nputil.complex_std_normal((3, 5), rng=np.random.default_rng(7)), golden["csn_3x5_seed7"])
    np.random.seed(3)
    a = nputil.complex_std_normal((2, 2))
    np.random.seed(3)
    re, im = np.random.standard_normal((2, 2)), np.random.standard_normal((2, 2))
    assert np.array_equal(a, (re + 1j * im) / 2**0.5)


def test_host_normal_stream_order():
    """The uploaded stream is exactly the reference's per-l draw sequence."""
    from cora_amd.core import skysim
    from oracle import skysim as osk

    F, lmax = 3, 4
    g = skysim._host_normals(F, lmax, np.random.default_rng(11))
    rng = np.random.default_rng(11)
    o = 0
    for l in range(lmax + 1):
        v = osk.complex_std_normal((F, l + 1), rng=rng) * 2**0.5
        n = F * (l + 1)
        assert np.allclose(g[o : o + n], v.real.ravel(), rtol=4e-16, atol=0)  # (x/sqrt2)*sqrt2 round trip
        assert np.allclose(g[o + n : o + 2 * n], v.imag.ravel(), rtol=4e-16, atol=0)
        o += 2 * n
    assert o == g.size


def test_romberg_weights():
    import scipy.integrate as si

    from cora_amd.core import skysim

    assert np.allclose(skysim.romberg_weights(3) * 2835 * 8, [868, 4096, 1408, 4096, 1744, 4096, 1408, 4096, 868])
    assert np.allclose(skysim.romberg_weights(2) * 45 * 4, [14, 64, 24, 64, 14])
    assert np.allclose(skysim.romberg_weights(1) * 3 * 2, [1, 4, 1])
    assert np.array_equal(skysim.romberg_weights(0), [1.0])
    y = np.random.default_rng(0).standard_normal(9)
    assert np.isclose(si.romb(y, dx=0.25) / 2.0, skysim.romberg_weights(3) @ y)


# ------------------------------------------------------------------ model classes (host part)
def test_foreground_models_kat(golden):
    from cora_amd.foreground import galaxy, gaussianfg, pointsource

    cr = galaxy.FullSkySynchrotron()
    aps1 = cr.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    assert np.allclose(aps1.sum(), 75.47681191093129, rtol=1e-7)
    fa = np.linspace(400.0, 800.0, 64)
    aps2 = cr.angular_powerspectrum(np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :])
    assert np.allclose(aps2[400, 40, 40], 9.690708728692975e-06, rtol=1e-7)
    assert np.allclose(aps2[200, 10, 40], 0.00017630767166797886, rtol=1e-7)
    ub = pointsource.CombinedPointSources._UnresolvedBackground()
    assert (ub.A, ub.alpha, ub.beta, ub.zeta, ub.nu_0, ub.l_0, ub.oversample) == (3.55e-5, 2.07, 1.1, 1.0, 408.0, 100.0, 0)
    assert gaussianfg.Synchrotron.A == 7.00e-4 and galaxy.FullSkyPolarisedSynchrotron.zeta == 0.04
    plan = cr._clarray_plan(cr.angular_powerspectrum)
    al, bcov = plan["prepare"](np.arange(5.0), np.array([400.0, 500.0]))
    assert al[0] == 0.0 and bcov.shape == (2, 2) and np.allclose(bcov, bcov.T)


def test_corr21cm_host_quantities(golden):
    from cora_amd.signal import corr21cm
    from cora_amd.util import cosmology, cubicspline

    cr = corr21cm.Corr21cm()
    z = golden["m21_z"]
    assert np.allclose(cr.T_b(z), golden["m21_Tb"], rtol=1e-15)
    assert np.allclose(cr.growth_factor(z), golden["m21_D"], rtol=1e-15)
    assert np.allclose(cr.growth_rate(z), golden["m21_f"], rtol=1e-15)
    assert np.allclose(cr.ps_vv(golden["ps_k"]), golden["ps_vv"], rtol=1e-13)
    assert np.allclose(cr.mean_nu(np.array([600.0])), 0.0)
    c = cosmology.Cosmology()
    assert np.allclose(c.comoving_distance(golden["cosmo_z"]), golden["cosmo_chi"], rtol=1e-14)
    assert np.allclose(c.H(golden["cosmo_z"]), golden["cosmo_H"], rtol=1e-15)
    assert np.isclose(c.comoving_distance(1.0), c.comoving_distance(np.array([1.0]))[0])
    plan = cr._clarray_plan(cr.angular_powerspectrum)
    assert plan["kind"] == "table21cm"
    assert cr._clarray_plan(lambda l, a, b: 0) is None


# ------------------------------------------------------------------ cubic spline (mirrors tests/test_cubicspline.py)
def test_cubicspline_usage_errors():
    from cora_amd.util import cubicspline as cs

    with pytest.raises(cs.InterpolationException):
        cs.Interpolater(np.zeros((5, 3)))
    with pytest.raises(cs.InterpolationException):
        cs.Interpolater(np.zeros((3, 2)))
    with pytest.raises(cs.InterpolationException):
        cs.Interpolater(np.array([[0, 1.0], [1, np.nan], [2, 1], [3, 1]]))
    with pytest.raises(cs.InterpolationException):
        cs.LogInterpolater(np.array([[0, 1.0], [1, 2], [2, 1], [3, 1]]))


def test_cubicspline_accuracy(golden):
    from cora_amd.util import cubicspline as cs

    x = np.linspace(0, 10, 50)
    assert np.allclose(cs.Interpolater(x, np.full(50, 3.0))(np.linspace(-2, 12, 31)), 3.0, atol=1e-12)
    lin = cs.Interpolater(x, 2 * x + 1)
    assert np.allclose(lin(np.linspace(-2, 12, 31)), 2 * np.linspace(-2, 12, 31) + 1, atol=1e-10)
    xf = np.linspace(0, 10, 2000)
    sm = cs.Interpolater(xf, np.sin(xf))
    xe = np.linspace(0.1, 9.9, 333)
    assert np.abs(sm(xe) - np.sin(xe)).max() < 1e-7
    assert np.isclose(sm(2.5), np.sin(2.5), atol=1e-7) and isinstance(sm(2.5), float)
    sp = cs.Interpolater(golden["spl_xk"], golden["spl_yk"])
    assert np.abs(sp(golden["spl_xe"]) - golden["spl_ye"]).max() < 1e-13
    assert np.abs(sp.data()[1] - golden["spl_y2"]).max() < 1e-13
    lsp = cs.LogInterpolater(np.dstack((golden["spl_xk"] + 0.5, np.exp(golden["spl_yk"])))[0])
    assert np.abs(lsp(np.abs(golden["spl_xe"]) + 0.25) / golden["lspl_ye"] - 1).max() < 1e-12


def test_makesky_freqstate_matches_reference_vectors():
    """FreqState (cora/scripts/makesky.py:44-92) against outputs of the reference's own class
    (tests/golden/make_golden_makesky.py) in every channelisation mode."""
    import json

    from cora_amd.scripts import makesky

    g = np.load(os.path.join(ROOT, "tests", "golden", "makesky_vectors.npz"))
    cases = json.loads(str(g["cases_json"]))
    assert len(cases) >= 7
    for name, kw in cases.items():
        fs = makesky.FreqState()
        for k, v in kw.items():
            setattr(fs, k, tuple(v) if k in ("freq", "channel_range") else v)
        assert np.array_equal(fs.frequencies, g[name + "__frequencies"]), name
        assert np.array_equal(np.asarray(fs.freq_width), g[name + "__freq_width"]), name


def test_write_map_hdf5_branch_layout_with_a_recording_h5py(tmp_path, monkeypatch):
    """The HDF5 branch of write_map (h5py is not installed in this image, so the branch never ran before round 6): an
    in-test stand-in module records what is created, and the record is compared with the layout of the reference's
    writer (cora/scripts/makesky.py:412-450): file attribute ``__memh5_distributed_file``; dataset ``map`` [freq, pol,
    pixel] f64 with ``axis`` = (freq, pol, pixel) as variable-length strings and ``__memh5_distributed_dset`` True;
    ``index_map/freq`` (centre, width) / ``index_map/pol`` (vlen str) / ``index_map/pixel``, each with
    ``__memh5_distributed_dset`` False - in that order."""
    import sys
    import types

    from cora_amd.scripts import makesky

    log = []

    class Attrs(dict):
        def __init__(self, owner):
            super().__init__()
            self.owner = owner

        def __setitem__(self, k, v):
            log.append(("attr", self.owner, k))
            super().__setitem__(k, v)

    class Dataset:
        def __init__(self, name, data):
            self.name, self.data, self.attrs = name, np.asarray(data), Attrs(name)

    class File:
        opened = []

        def __init__(self, filename, mode):
            assert mode == "w"
            self.filename, self.attrs, self.datasets = filename, Attrs("/"), {}
            File.opened.append(self)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self.closed = True
            return False

        def create_dataset(self, name, data=None):
            log.append(("dataset", name))
            self.datasets[name] = Dataset(name, data)
            return self.datasets[name]

    vlen = np.dtype(object, metadata={"vlen": str})           # what h5py.special_dtype(vlen=str) is: object dtype + metadata
    fake = types.ModuleType("h5py")
    fake.File = File
    fake.special_dtype = lambda vlen=None: np.dtype(object, metadata={"vlen": vlen})
    monkeypatch.setitem(sys.modules, "h5py", fake)

    rs = np.random.default_rng(0)
    data = rs.standard_normal((3, 48))
    freq = np.array([400.0, 500.0, 600.0])
    out = str(tmp_path / "map.h5")
    assert makesky.write_map(out, data, freq, fwidth=25.0, include_pol=True) == out
    (f,) = File.opened
    assert f.filename == out and f.closed and f.attrs["__memh5_distributed_file"] is True
    assert [e[1] for e in log if e[0] == "dataset"] == ["map", "index_map/freq", "index_map/pol", "index_map/pixel"]
    m = f.datasets["map"]
    assert m.data.shape == (3, 4, 48) and m.data.dtype == np.float64
    assert np.array_equal(m.data[:, 0], data) and not m.data[:, 1:].any()
    assert list(m.attrs["axis"]) == ["freq", "pol", "pixel"] and m.attrs["axis"].dtype == vlen
    assert m.attrs["__memh5_distributed_dset"] is True
    fq = f.datasets["index_map/freq"].data
    assert fq.dtype.names == ("centre", "width") and np.array_equal(fq["centre"], freq) and np.all(fq["width"] == 25.0)
    pol = f.datasets["index_map/pol"].data
    assert list(pol) == ["I", "Q", "U", "V"] and pol.dtype == vlen
    assert np.array_equal(f.datasets["index_map/pixel"].data, np.arange(48))
    for k in ("index_map/freq", "index_map/pol", "index_map/pixel"):
        assert f.datasets[k].attrs["__memh5_distributed_dset"] is False
    # unpolarised container, default width = the channel spacing; a 3-D input keeps its planes
    File.opened.clear()
    makesky.write_map(out, data, freq, include_pol=False)
    f = File.opened[-1]
    assert f.datasets["map"].data.shape == (3, 1, 48) and list(f.datasets["index_map/pol"].data) == ["I"]
    assert np.all(f.datasets["index_map/freq"].data["width"] == 100.0)
    makesky.write_map(out, np.ones((3, 4, 48)), freq)
    assert File.opened[-1].datasets["map"].data.shape == (3, 4, 48) and list(File.opened[-1].datasets["index_map/pol"].data) == ["I", "Q", "U", "V"]


def test_makesky_cli_surface_and_map_container(tmp_path):
    """Option parsing, the single-source map (no GPU needed) and the container layout of write_map."""
    from click.testing import CliRunner

    from cora_amd.scripts import makesky
    from cora_amd.util import hputil
    from oracle import healpix

    assert set(makesky.cli.commands) == {"foreground", "galaxy", "pointsource", "21cm", "gaussianfg", "singlesource"}
    out = str(tmp_path / "src.h5")
    r = CliRunner().invoke(makesky.cli, ["singlesource", "--nside", "8", "--freq", "400", "800", "4", "--freq-mode", "edge",
                                         "--pol", "none", "--ra", "33.0", "--dec", "-12.5", "--channel-list", "[0, 2]",
                                         "--filename", out])
    assert r.exit_code == 0, r.output
    path = out if os.path.exists(out) else out + ".npz"
    assert path.endswith(".npz")          # h5py is not installed in this image
    f = np.load(path)
    assert f["map"].shape == (2, 1, 768) and f["map"].sum() == 2.0
    pix = int(np.argmax(f["map"][0, 0]))
    th, ph = healpix.pix2ang_ring(8)
    assert pix == hputil.ang2pix(8, 33.0, -12.5, lonlat=True)
    assert abs(np.degrees(ph[pix]) - 33.0) < 8 and abs(90 - np.degrees(th[pix]) + 12.5) < 8
    assert np.array_equal(f["index_map__freq"]["centre"], [450.0, 650.0]) and np.all(f["index_map__freq"]["width"] == 100.0)
    # a 3-D input always gets the four-entry pol index, even with one Stokes plane (scripts/makesky.py:419-420)
    assert list(f["index_map__pol"]) == ["I", "Q", "U", "V"] and np.array_equal(f["index_map__pixel"], np.arange(768))
    assert list(makesky.map_container(np.ones((3, 48)), np.array([1.0, 2.0, 3.0]), include_pol=False)["index_map/pol"]) == ["I"]
    c = makesky.map_container(np.ones((3, 48)), np.array([1.0, 2.0, 3.0]))
    assert c["map"].shape == (3, 4, 48) and np.all(c["map"][:, 1:] == 0) and list(c["index_map/pol"]) == ["I", "Q", "U", "V"]
    assert np.all(c["index_map/freq"]["width"] == 1.0)
    r = CliRunner().invoke(makesky.cli, ["galaxy", "--nside", "8"])
    assert r.exit_code != 0 and "not part of cora_amd" in r.output
    r = CliRunner().invoke(makesky.cli, ["singlesource", "--channel-list", "[0, 'a']"])
    assert r.exit_code != 0


def test_ang2pix_inverts_pix2ang():
    from cora_amd.util import hputil
    from oracle import healpix

    for ns in (1, 2, 8, 32):
        th, ph = healpix.pix2ang_ring(ns)
        assert np.array_equal(hputil.ang2pix(ns, th, ph), np.arange(12 * ns * ns))
        # the package's own pix2ang / ang_positions (hputil.py:53-73) against the oracle geometry, and round trip
        ap = hputil.ang_positions(ns)
        assert ap.shape == (12 * ns * ns, 2) and np.abs(ap[:, 0] - th).max() < 1e-14 and np.abs(ap[:, 1] - ph).max() < 1e-14
        assert np.array_equal(hputil.ang2pix(ns, ap[:, 0], ap[:, 1]), np.arange(12 * ns * ns))


def test_cubicspline_array_entry_points():
    """value_array / value_log_array / __call__ agree and extrapolate a constant exactly
    (the cases of the reference's tests/test_cubicspline.py::test_constant)."""
    from cora_amd.util import cubicspline as cs

    x = np.arange(1, 8)
    data = np.dstack((x, np.ones(7)))[0]
    pts = np.asarray([0.025, 1, 2.5, 4, 5.55, 7.01, 19])
    p = cs.Interpolater(data)
    assert (p(pts) == 1).all() and (p.value_array(np.asarray([-0.025, 1, 2.5, 4, 5.55, 7.01, 19])) == 1).all()
    q = cs.LogInterpolater(data)
    assert (q(pts) == 1).all() and (q.value_log_array(np.asarray([0.0125, 1, 2.5, 4, 5.55, 7.01, 19])) == 1).all()
    grid = np.linspace(1.2, 6.8, 12).reshape(3, 4)
    assert p.value_array(grid).shape == (3, 4) and np.array_equal(p.value_array(grid), p(grid.ravel()).reshape(3, 4))


def test_sinh_interpolater_matches_reference(golden):
    """cubicspline.SinhInterpolater (cubicspline.pyx:290-345) against the reference's compiled class, incl. the
    extrapolated points on both sides and zero / negative data."""
    from cora_amd.util import cubicspline as cs

    sp = cs.SinhInterpolater(np.dstack((golden["spl_xk"], golden["spl_yk"]))[0], 0.7, 0.05)
    got = sp(golden["spl_xe"])
    assert np.abs(got - golden["sspl_ye"]).max() <= 1e-13 * np.abs(golden["sspl_ye"]).max()
    assert abs(sp(3.3) - float(sp(np.array([3.3]))[0])) < 1e-15
    kind, kx, ky, ky2, x_t, f_t = sp._device_spline()
    assert kind == 2 and x_t == 0.7 and f_t == 0.05 and kx.shape == ky.shape == ky2.shape == (12,)


# ------------------------------------------------------------------ plan protocol / launcher (no GPU needed)
def test_clarray_plan_only_for_unoverridden_methods():
    """skysim.clarray takes the table / separable fast path only for the library's own aps methods: a subclass that
    overrides angular_powerspectrum (or any other bound method) must fall through to the generic host-callable path
    instead of being silently evaluated as the un-overridden model."""
    from cora_amd.core import skysim
    from cora_amd.foreground import galaxy
    from cora_amd.signal import corr, corr21cm

    class MyCorr(corr.RedshiftCorrelation):
        def angular_powerspectrum(self, l, z1, z2):
            return 2.0 * corr.RedshiftCorrelation.angular_powerspectrum_fft(self, l, z1, z2)

    base = corr.RedshiftCorrelation.__new__(corr.RedshiftCorrelation)
    sub = MyCorr.__new__(MyCorr)
    assert skysim._plan_of(base.angular_powerspectrum)["kind"] == "table21cm"
    assert skysim._plan_of(base.angular_powerspectrum_fft)["kind"] == "table21cm"
    assert skysim._plan_of(sub.angular_powerspectrum) is None
    assert skysim._plan_of(sub.angular_powerspectrum_fft)["kind"] == "table21cm"

    class My21(corr21cm.Corr21cm):
        def angular_powerspectrum(self, l, nu1, nu2):
            return 0.0 * l

    assert skysim._plan_of(My21.__new__(My21).angular_powerspectrum) is None

    class MySync(galaxy.FullSkySynchrotron):
        def angular_powerspectrum(self, l, nu1, nu2):
            return 1.0 + 0.0 * l

    assert skysim._plan_of(MySync().angular_powerspectrum) is None
    assert skysim._plan_of(galaxy.FullSkySynchrotron().angular_powerspectrum)["kind"] == "separable"
    assert skysim._plan_of(lambda l, a, b: l) is None


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N from a plain shell starts torch.distributed.run as a child (never an exec), hands the
    arguments through and relays exactly the JSON line."""
    import subprocess

    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    class R:
        returncode = 0
        stdout = b"RCCL version banner\n{\"metric\": \"x\"}\n"

    def fake(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return R()

    monkeypatch.setattr(subprocess, "run", fake)
    assert bench.launch_ranks(4, ["--gpus", "4", "--steps", "2"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    for name in ("cfg3", "cfg4", "cfg5"):
        assert name in bench.WORKLOADS


# ------------------------------------------------------------------ bench line bookkeeping (no GPU)
@pytest.mark.parametrize("line,workload,nranks,nu0,nnu,mode", [
    ("bench_cfg3_r03_final.json", "cfg3", 1, 0, 256, "joint"),
    ("bench_cfg3_emulated_shard8_r03.json", "cfg3", 8, 224, 32, "joint"),
    ("bench_cfg3_emulated_shard2_r03.json", "cfg3", 2, 128, 128, "joint"),
    ("bench_cfg4_r03_joint.json", "cfg4", 1, 0, 512, "joint"),
    ("bench_cfg4_r03_separate.json", "cfg4", 1, 0, 512, "separate"),
    ("bench_cfg5_emulated_shard_r03.json", "cfg5", 8, 896, 128, "joint")])
def test_stage_rooflines_are_fractions_for_sharded_and_multicomponent_lines(line, workload, nranks, nu0, nnu, mode):
    """bench.stage_rooflines prices every stage with the work of THE RANK that ran it: recorded stage times of the
    sharded / emulated / multi-component lines give fractions in (0, 1] (round 3 printed 13.9 for K1 of the cfg-5 share:
    the full F^2 over one eighth of the time)."""
    import json

    import bench

    d = json.load(open(os.path.join(ROOT, "profiles", line)))
    comps, F, _, _, nside, lmax = bench.WORKLOADS[workload]
    stages = {k: {"ms_per_step": v} for k, v in d["stages_ms"].items()}
    r = bench.stage_rooflines(stages, comps, F, nside, lmax, nnu, nu0, nranks, mode,
                              d["roofline"]["executed_flops_per_launch"])
    assert set(r) == set(stages)
    for k, e in r.items():
        assert 0.0 < e["frac"] <= 1.0, (line, k, e)
        # nothing in a roofline record above its roof: no rate over the FP64 peak either (K1's nominal flop count - 1.09 of
        # the peak in the round-5 line - is kept under its own name and K1 is priced against the L2 traffic of its tables)
        assert e.get("TFLOPs", 0.0) <= bench.FP64_MFMA_PEAK_TFLOPS and e["GBs"] <= (e.get("peak_GBs") or bench.HBM_PEAK_GBS * 5), (line, k, e)
    if "clarray" in r and workload != "cfg2":
        assert r["clarray"]["bound"] == "l2" and "TFLOPs" not in r["clarray"] and r["clarray"]["nominal_TFLOPs"] > 0


def test_rank_memory_of_the_8_gpu_configs_fits_an_mi355x():
    """BASELINE configs[3] (three components, 512 channels) and configs[4] (1024 channels, nside 2048, lmax 4096) at 8
    ranks: maps, a_lm, F_m workspace, factor rows, C_l shard and plan of ONE rank stay under 0.9 x 288 GB; cfg 5 on
    one GPU does not (that is why the bench emulates one rank of it)."""
    from cora_amd.parallel import rank_memory_bytes

    hbm = 288e9
    c4 = rank_memory_bytes(["table21cm", "separable", "separable"], 512, 1024, 2048, 8)
    c4s = rank_memory_bytes(["table21cm", "separable", "separable"], 512, 1024, 2048, 8, sum_mode="separate")
    c5 = rank_memory_bytes(["table21cm"], 1024, 2048, 4096, 8)
    for m in (c4, c4s, c5):
        assert m["total"] == sum(v for k, v in m.items() if k != "total")
        assert m["total"] < 0.9 * hbm, m
    assert 120e9 < c5["total"] < 200e9
    assert rank_memory_bytes(["table21cm"], 1024, 2048, 4096, 1)["total"] > hbm
    # the numpy-seeded mode (the reference's own call): since round 5 the stream is generated range of multipoles by
    # range - a ring of 2 GiB at cfg 3 / 4 (8.6 / 17.2 GB streams), of 18 GB instead of the 137 GB stream at cfg 5
    from cora_amd.parallel import numpy_ring_bytes

    assert numpy_ring_bytes(256, 2048) == 2**31 and numpy_ring_bytes(8, 64) == 16 * 8 * (65 * 66 // 2)
    assert numpy_ring_bytes(1024, 4096, 288e9) == int(288e9 / 16)
    # the C side sizes the ring against prop.totalGlobalMem (~3.09e11 on a 288 GiB part), the model follows when told;
    # a folded shard holds one more copy of its factor block while the rows are permuted
    assert numpy_ring_bytes(1024, 4096, 3.09e11) == int(3.09e11 / 16)
    c5f = rank_memory_bytes(["table21cm"], 1024, 2048, 4096, 8, fold=True, rng="numpy", device_bytes=3.09e11)
    assert c5f["total"] - rank_memory_bytes(["table21cm"], 1024, 2048, 4096, 8, rng="numpy", device_bytes=3.09e11)["total"] == 8 * 513 * 1024 * 1024
    assert c5f["total"] < 0.9 * hbm
    assert rank_memory_bytes(["table21cm"], 256, 1024, 2048, 1, rng="numpy")["total"] < 0.9 * hbm
    c5n = rank_memory_bytes(["table21cm"], 1024, 2048, 4096, 8, rng="numpy")
    assert c5n["total"] < 0.9 * hbm and c5n["total"] - c5["total"] < 30e9, c5n
    c4n = rank_memory_bytes(["table21cm", "separable", "separable"], 512, 1024, 2048, 8, sum_mode="separate", rng="numpy")
    assert c4n["total"] < 0.9 * hbm


# ------------------------------------------------------------------ EoR21cm, Cmb, TestF, like_kiyo_map (no GPU)
def test_eor21cm_host_quantities_match_the_reference():
    """EoR21cm (cora/signal/corr21cm.py:333-385): T_b, bias, Omega_HI, x_h, prefactor against outputs of the reference's
    own class (tests/golden/make_golden_eor.py) - the per-redshift inputs K1 is fed with."""
    from cora_amd.signal import corr21cm

    g = np.load(os.path.join(ROOT, "tests", "golden", "eor_vectors.npz"))
    eor = corr21cm.EoR21cm()
    assert isinstance(eor, corr21cm.Corr21cm)
    z = g["z"]
    assert np.array_equal(eor.T_b(z), g["T_b"]) and np.array_equal(eor.prefactor(z), g["prefactor"])
    assert np.array_equal(eor.bias_z(z), g["bias_z"])
    assert eor.omega_HI(z) == float(g["omega_HI"]) and eor.x_h(z) == float(g["x_h"])
    # the table plan of the device path is the parent's: only the per-redshift quantities differ
    assert eor._clarray_plan(eor.angular_powerspectrum)["kind"] == "table21cm"


def test_cmb_and_testf_spectra_match_the_reference(tmp_path):
    """gaussianfield.Cmb / TestF (cora/core/gaussianfield.py:159-191) against the reference's own classes."""
    from cora_amd.core import gaussianfield

    g = np.load(os.path.join(ROOT, "tests", "golden", "eor_vectors.npz"))
    psfile = str(tmp_path / "ps.dat")
    np.savetxt(psfile, g["cmb_table"])
    for camb, key in ((True, "cmb_ps_cambnorm"), (False, "cmb_ps_plain")):
        ps = gaussianfield.Cmb(psfile=psfile, cambnorm=camb).powerspectrum(g["cmb_karray"])
        assert ps.shape == g[key].shape and np.abs(ps / g[key] - 1.0).max() < 1e-13
    assert issubclass(gaussianfield.Cmb, gaussianfield.RandomFieldA2)
    tf = gaussianfield.TestF.__new__(gaussianfield.TestF)
    assert np.array_equal(tf.powerspectrum(g["testf_karray"]), g["testf_ps"])
    assert issubclass(gaussianfield.TestF, gaussianfield.RandomFieldA2F)


def test_map3d_like_kiyo_map(capsys):
    """Map3d.like_kiyo_map (cora/core/maps.py:175-201): geometry from a map object with freq / ra / dec axes."""
    from cora_amd.core import maps

    class Kiyo:
        info = {"dec_centre": 60.0}

        def get_axis(self, name):
            return {"freq": np.array([7.0e8, 7.5e8, 8.0e8]), "ra": np.array([10.0, 11.0, 12.0, 14.0]),
                    "dec": np.array([58.0, 62.0])}[name]

    m = maps.Map3d.like_kiyo_map(Kiyo())
    assert (m.x_num, m.y_num, m.nu_num) == (4, 2, 3)
    assert m.x_width == 4.0 * np.cos(np.pi * 60.0 / 180.0) and m.y_width == 4.0
    assert (m.nu_lower, m.nu_upper) == (700.0, 800.0)
    assert "Map3D: 4x2 field" in capsys.readouterr().out
