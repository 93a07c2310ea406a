"""Flat-sky Gaussian fields (flat-sky half): oracle vs the reference's own outputs (CPU),
and the HIP line-FFT engine / RandomField / ForegroundMap.getfield vs oracle and goldens (-m gpu)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RF_CASES = ("rf_16_16_16", "rf_12_10_14", "rf_24_36")


@pytest.fixture(scope="module")
def fsg():
    """Outputs of the reference's gaussianfield / fftutil / gaussianfg (tests/golden/make_golden_flatsky.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "flatsky_vectors.npz"))


def ps_model(karray):
    k2 = (np.asarray(karray) ** 2).sum(axis=-1)
    return 3.0 / (1.0 + k2) ** 1.3


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()


# ------------------------------------------------------------------ CPU: oracle and host logic vs the reference
def test_oracle_flatsky_matches_reference_vectors(fsg):
    from oracle import flatsky as ofs

    assert np.array_equal(ofs.rfftfreqn((8, 6, 10), np.array([0.5, 0.25, 2.0])), fsg["kvec_8_6_10"])
    assert np.array_equal(ofs.rfftfreqn((5, 7, 9)), fsg["kvec_5_7_9"])   # half-integer grid of the odd axes
    for tag in RF_CASES:
        kw = ofs.kweight(ps_model, fsg[tag + "__n"], fsg[tag + "__w"])
        assert _rel(kw, fsg[tag + "__kweight"]) < 1e-15
        np.random.seed(int(fsg[tag + "__seed"]))
        a = np.random.standard_normal(kw.shape)
        b = np.random.standard_normal(kw.shape)
        assert _rel(ofs.getfield(kw, a, b), fsg[tag + "__field"]) < 1e-14
    # ForegroundMap.getfield: the angular field, then ncorr blocks of real normals, all from seed 31
    kwa = fsg["syn__ang_kweight"]
    np.random.seed(31)
    ang = ofs.getfield(kwa, np.random.standard_normal(kwa.shape), np.random.standard_normal(kwa.shape))
    ncorr = int(fsg["syn__num_corr_freq"])
    nrm = np.random.standard_normal((ncorr,) + kwa.shape)
    assert _rel(ofs.foreground_getfield(fsg["syn__freq_weight"], ang, nrm), fsg["syn__field"]) < 1e-13


def test_host_kweight_and_geometry_match_reference(fsg):
    """The host half of RandomField (frequency grid, k-weights, Map2d/Map3d geometry) - no GPU involved."""
    from cora_amd.core import gaussianfield
    from cora_amd.util import fftutil

    d = np.array([0.5, 0.25, 2.0])
    assert np.array_equal(fftutil.rfftfreqn((8, 6, 10), d), fsg["kvec_8_6_10"])
    assert np.array_equal(d, [0.5, 0.25, 2.0])          # caller's spacing untouched
    assert np.array_equal(fftutil.rfftfreqn((5, 7, 9)), fsg["kvec_5_7_9"])
    with pytest.raises(Exception, match="wrong length"):
        fftutil.rfftfreqn((4, 4), [1.0])
    for tag in RF_CASES:
        rf = gaussianfield.RandomField(npix=list(fsg[tag + "__n"]), wsize=list(fsg[tag + "__w"]))
        rf.powerspectrum = ps_model
        rf.generate_kweight()
        assert _rel(rf._kweight, fsg[tag + "__kweight"]) < 1e-15
    a2 = gaussianfield.RandomFieldA2()
    a2.x_num, a2.y_num, a2.x_width, a2.y_width = 12, 16, 4.0, 6.0
    a2.powerspectrum = ps_model
    a2.generate_kweight()
    assert list(a2._n) == [12, 16] and _rel(a2._kweight, fsg["a2__kweight"]) < 1e-15
    a2f = gaussianfield.RandomFieldA2F()
    a2f.x_num, a2f.y_num, a2f.nu_num = 8, 10, 6
    a2f.x_width, a2f.y_width, a2f.nu_lower, a2f.nu_upper = 3.0, 5.0, 500.0, 560.0
    a2f.powerspectrum = ps_model
    a2f.generate_kweight()
    assert list(a2f._n) == [6, 8, 10] and _rel(a2f._kweight, fsg["a2f__kweight"]) < 1e-15
    # error behaviour of gaussianfield.py:34-42 and :68
    with pytest.raises(Exception, match="has not been set"):
        gaussianfield.RandomField().generate_kweight()
    with pytest.raises(Exception, match="same length"):
        gaussianfield.RandomField(npix=[4, 4], wsize=[1.0]).generate_kweight()
    with pytest.raises(Exception, match="positive"):
        gaussianfield.RandomField(npix=[4, 0]).generate_kweight()
    with pytest.raises(Exception, match="Abstract method"):
        gaussianfield.RandomField(npix=[4, 4]).generate_kweight()


# ------------------------------------------------------------------ GPU: line-FFT engine
LENGTHS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 64, 100, 128, 256, 257, 384, 512, 768, 1000, 1024, 1265, 1700, 2000, 2048, 3000, 4095,
           4096]


@pytest.mark.gpu
@pytest.mark.parametrize("n", LENGTHS)
def test_fft_c2c_every_axis_position_vs_numpy(ctx, n):
    import torch

    rng = np.random.default_rng(n)
    for shape, axis in (((3, n), 1), ((n, 5), 0), ((2, n, 37), 1)):
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        for inverse, ref in ((False, np.fft.fft), (True, np.fft.ifft)):
            d = torch.from_numpy(x.copy()).to(ctx.device)
            ctx.fft_c2c(d, axis, inverse=inverse)
            assert _rel(d.cpu().numpy(), ref(x, axis=axis)) < 1e-13, (shape, axis, inverse)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 6, 10), (16, 16, 16), (30, 20, 14), (5, 7, 9), (128, 128), (1, 4), (100,),
                                   (4096,), (3, 250), (64, 64, 64), (17, 33, 50), (2, 3, 4, 6), (1, 1, 2),
                                   # last axes 512 / 1024 / 2048 / 4096: the compile-time c2r pass (16 / 16 / 8 / 4 lines per
                                   # item; line counts that are not multiples of it, and a single line)
                                   (5, 7, 512), (3, 11, 1024), (21, 2048), (9, 4096), (1, 1024), (33, 512),
                                   # 3 * 2^k: radix 12 in the first pass
                                   (5, 7, 384), (19, 768), (9, 1536), (5, 3072), (384, 40, 6),
                                   # arbitrary lengths on strided axes: Bluestein on the compile-time passes (P = 640, 640, 2048, 320)
                                   (261, 316, 24), (628, 9, 10), (130, 20),
                                   # even last axes 2 h with arbitrary h: the contiguous passes through the same convolution
                                   # (P = 320, 640, 1024, 2048, 768, 512), and one line more than the lines per item
                                   (7, 316), (5, 9, 628), (4, 1000), (6, 1800), (5, 700), (17, 500), (2500,),
                                   # convolution lengths 2560 .. 4096 (two lines per item)
                                   (3, 2600), (5, 4000), (1265, 12), (1500, 3, 6), (3, 3300)])
def test_rfftn_irfftn_vs_numpy(ctx, shape):
    import torch

    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape)
    ref = np.fft.rfftn(x)
    assert _rel(ctx.rfftn(torch.from_numpy(x).to(ctx.device)).cpu().numpy(), ref) < 1e-13
    # a spectrum that is NOT Hermitian-consistent: numpy ignores Im of the DC / Nyquist bins of the last axis
    sp = ref + 0.3j * rng.standard_normal(ref.shape)
    want = np.fft.irfftn(sp, s=shape, axes=list(range(len(shape))))
    got = ctx.irfftn(torch.from_numpy(sp.copy()).to(ctx.device), last=shape[-1]).cpu().numpy()
    assert got.shape == tuple(shape) and _rel(got, want) < 1e-13
    # round trip
    back = ctx.irfftn(ctx.rfftn(torch.from_numpy(x).to(ctx.device)), last=shape[-1]).cpu().numpy()
    assert np.abs(back - x).max() < 1e-13


@pytest.mark.gpu
def test_partial_axes_and_argument_errors(ctx):
    import torch

    from cora_amd._lib import CoraHipError

    rng = np.random.default_rng(5)
    x = rng.standard_normal((5, 12, 9)) + 1j * rng.standard_normal((5, 12, 9))
    want = np.fft.irfft(np.fft.ifft(x, axis=1), axis=2)
    got = ctx.irfftn(torch.from_numpy(x.copy()).to(ctx.device), naxes=2).cpu().numpy()
    assert _rel(got, want) < 1e-14
    r = rng.standard_normal((4, 6, 10))
    assert _rel(ctx.rfftn(torch.from_numpy(r).to(ctx.device), naxes=1).cpu().numpy(), np.fft.rfft(r, axis=2)) < 1e-14
    with pytest.raises(CoraHipError):     # a transformed axis longer than the LDS line limit
        ctx.fft_c2c(torch.zeros((2, 5000), dtype=torch.complex128, device=ctx.device), 1)
    with pytest.raises(CoraHipError):
        ctx.irfftn(torch.zeros((4, 3), dtype=torch.complex128, device=ctx.device), last=7)
    # an untransformed leading axis may be arbitrarily long
    big = rng.standard_normal((5000, 8))
    assert _rel(ctx.rfftn(torch.from_numpy(big).to(ctx.device), naxes=1).cpu().numpy(), np.fft.rfft(big, axis=1)) < 1e-14


# ------------------------------------------------------------------ GPU: RandomField / ForegroundMap.getfield
@pytest.mark.gpu
def test_randomfield_getfield_matches_reference_vectors(fsg):
    """Same global-numpy-state normals as the reference (np.random.seed), field from the HIP irfftn."""
    from cora_amd.core import gaussianfield

    for tag in RF_CASES:
        rf = gaussianfield.RandomField(npix=list(fsg[tag + "__n"]), wsize=list(fsg[tag + "__w"]))
        rf.powerspectrum = ps_model
        np.random.seed(int(fsg[tag + "__seed"]))
        f = rf.getfield()
        assert f.shape == fsg[tag + "__field"].shape and _rel(f, fsg[tag + "__field"]) < 1e-13, tag
    a2 = gaussianfield.RandomFieldA2()
    a2.x_num, a2.y_num, a2.x_width, a2.y_width = 12, 16, 4.0, 6.0
    a2.powerspectrum = ps_model
    np.random.seed(21)
    assert _rel(a2.getfield(), fsg["a2__field"]) < 1e-13
    a2f = gaussianfield.RandomFieldA2F()
    a2f.x_num, a2f.y_num, a2f.nu_num = 8, 10, 6
    a2f.x_width, a2f.y_width, a2f.nu_lower, a2f.nu_upper = 3.0, 5.0, 500.0, 560.0
    a2f.powerspectrum = ps_model
    np.random.seed(22)
    assert _rel(a2f.getfield(), fsg["a2f__field"]) < 1e-13


@pytest.mark.gpu
def test_foreground_getfield_matches_reference_vectors(fsg):
    from cora_amd.foreground import gaussianfg

    fg = gaussianfg.PointSources()
    fg.x_num, fg.y_num, fg.nu_num = 16, 12, 5
    fg.x_width, fg.y_width, fg.nu_lower, fg.nu_upper = 6.0, 4.0, 400.0, 800.0
    np.random.seed(31)
    cube = fg.getfield()
    # the 5x5 frequency covariance has condition number ~1e7: the Cholesky factor is reproducible to ~1e-13 * few
    assert _rel(fg._freq_weight, fsg["syn__freq_weight"]) < 1e-11
    assert fg._num_corr_freq == int(fsg["syn__num_corr_freq"])
    assert cube.shape == fsg["syn__field"].shape and _rel(cube, fsg["syn__field"]) < 1e-10
    # near-singular frequency covariance (eigen branch): the reference raises here, the build returns a cube
    syn = gaussianfg.Synchrotron()
    syn.x_num, syn.y_num, syn.nu_num = 16, 12, 8
    syn.nu_lower, syn.nu_upper = 600.0, 700.0
    np.random.seed(32)
    c2 = syn.getfield()
    assert c2.shape == (8, 16, 12) and np.isfinite(c2).all() and c2.std() > 0


@pytest.mark.gpu
def test_device_seeded_field_vs_oracle_stream_and_spectrum(ctx):
    """Throughput mode: Philox normals drawn on the device; equals the oracle's restatement of the stream,
    is reproducible, and (size-independent property at 256^3) its binned power follows P(k)."""
    from cora_amd.core import gaussianfield
    from cora_amd.util import fftutil
    from oracle import flatsky as ofs

    rf = gaussianfield.RandomField(npix=[12, 10, 14], wsize=[5.0, 7.0, 9.0])
    rf.powerspectrum = ps_model
    f = rf.getfield(seed=77)
    want = np.fft.irfftn(ofs.device_spec(rf._kweight, 77))
    assert _rel(f, want) < 1e-13
    assert np.array_equal(f, rf.getfield(seed=77)) and not np.array_equal(f, rf.getfield(seed=78))

    # getfield(seed) is ONE library call since round 5 (the spectrum generated inside the first transform pass): the same
    # field, bit for bit, as the draw followed by irfftn - 3-d, 2-d, non-power-of-two and odd sizes, and 1-d (two-step inside)
    import torch

    # (first axes 256 / 512 / 1024: the generating pass with the compile-time FFT passes, whole and partial tiles)
    for shape in ([12, 10, 14], [16, 64], [9, 20], [33, 7, 11], [40], [256, 6, 12], [512, 64], [1024, 5, 8], [256, 256, 30],
                  [384, 20, 12], [768, 3, 30], [261, 30, 18], [100, 17, 8]):
        kw = ctx.empty(tuple(shape[:-1]) + (shape[-1] // 2 + 1,)).uniform_()
        two = ctx.irfftn(ctx.randomfield_draw(kw, 123), last=shape[-1])
        one = ctx.randomfield_irfftn(kw, 123, last=shape[-1])
        assert torch.equal(one, two), shape

    n, w = 256, 400.0
    big = gaussianfield.RandomField(npix=[n, n, n], wsize=[w, w, w])
    big.powerspectrum = ps_model
    fld = big.getfield_device(seed=5)
    assert tuple(fld.shape) == (n, n, n)
    spec = ctx.rfftn(fld).cpu().numpy()
    # <|F_k|^2> = 2 kweight^2 for the interior bins; compare in 12 log bins of |k|
    kmag = np.sqrt((fftutil.rfftfreqn([n, n, n], np.full(3, w / n / (2 * np.pi))) ** 2).sum(axis=-1))
    interior = np.ones(spec.shape, dtype=bool)
    interior[..., 0] = False
    interior[..., -1] = False
    ratio = (np.abs(spec) ** 2)[interior] / (2.0 * big._kweight[interior] ** 2)
    bins = np.digitize(np.log(kmag[interior]), np.linspace(np.log(kmag[interior].min()), np.log(kmag.max()), 13))
    for b in range(1, 13):
        sel = bins == b
        if sel.sum() > 2000:
            assert abs(ratio[sel].mean() - 1.0) < 6.0 / np.sqrt(sel.sum()), (b, ratio[sel].mean(), sel.sum())
    # Parseval through the HIP transforms: sum f^2 = (1/N) (sum over the full spectrum)
    full = 2.0 * (np.abs(spec) ** 2).sum() - (np.abs(spec[..., 0]) ** 2).sum() - (np.abs(spec[..., -1]) ** 2).sum()
    assert abs(float((fld**2).sum().item()) * n**3 / full - 1.0) < 1e-12


# ------------------------------------------------------------------ redshift-space cube (Corr21cm.getfield)
CUBES = ("cube_a", "cube_b")


def _cube_args(fsg, tag):
    nu_num, x_num, y_num, xw, yw, nlo, nhi = fsg[tag + "__params"]
    return int(nu_num), int(x_num), int(y_num), float(xw), float(yw), float(nlo), float(nhi)


def test_oracle_redshift_cube_matches_reference_vectors(fsg):
    """oracle/flatsky.py::realisation vs outputs of the reference's own Corr21cm.getfield / realisation
    (same global-numpy-state normals), and its scipy call vs the definition-level trilinear restatement."""
    import scipy.ndimage

    from oracle import flatsky as ofs
    from oracle import models

    m = models.Corr21cm()
    for tag in CUBES:
        nu_num, x_num, y_num, xw, yw, nlo, nhi = _cube_args(fsg, tag)
        z1, z2 = ofs.NU21 / nhi - 1.0, ofs.NU21 / nlo - 1.0
        np.random.seed(41)
        ac, rsf, geom = ofs.realisation(m, z1, z2, xw, yw, nu_num, x_num, y_num, np.random.standard_normal, zspace=False)
        assert _rel(ac, fsg[tag + "__acube"]) < 1e-14 and _rel(rsf, fsg[tag + "__rsf"]) < 1e-14
        assert np.allclose(geom, fsg[tag + "__geom"], rtol=1e-15)
        assert _rel(ac[::-1], fsg[tag + "__getfield"]) < 1e-14
        np.random.seed(41)
        ac2 = ofs.realisation(m, z1, z2, xw, yw, nu_num, x_num, y_num, np.random.standard_normal, density_only=True,
                              no_mean=True, no_evolution=True, refinement=2)[0]
        assert _rel(ac2, fsg[tag + "__density_only_nomean_zspace"]) < 1e-14
    rng = np.random.default_rng(0)
    co = rng.uniform(-1.0, max(rsf.shape) + 0.5, size=(3, 300))
    co[:, :5] = 0.0
    co[0, 5], co[1, 6], co[2, 7] = rsf.shape[0] - 1, rsf.shape[1] - 1, rsf.shape[2] - 1
    assert np.abs(ofs.trilinear_constant(rsf, co) - scipy.ndimage.map_coordinates(rsf, co, order=1)).max() < 1e-18


@pytest.mark.gpu
def test_corr21cm_getfield_matches_reference_vectors(fsg):
    """The HIP cube builder (field draw -> rfftn x mu^2 -> irfftn -> slice factors -> ray tracing) with the
    reference's normals vs the reference's own outputs."""
    from cora_amd.signal import corr21cm

    for tag in CUBES:
        nu_num, x_num, y_num, xw, yw, nlo, nhi = _cube_args(fsg, tag)
        cr = corr21cm.Corr21cm()
        cr.nu_num, cr.x_num, cr.y_num, cr.x_width, cr.y_width, cr.nu_lower, cr.nu_upper = nu_num, x_num, y_num, xw, yw, nlo, nhi
        z1, z2 = cr._band_redshifts()
        np.random.seed(41)
        cube = cr.getfield()
        assert cube.shape == fsg[tag + "__getfield"].shape and _rel(cube, fsg[tag + "__getfield"]) < 1e-12, tag
        np.random.seed(41)
        ac, rsf, geom = cr.realisation(z1, z2, xw, yw, nu_num, x_num, y_num, zspace=False, report_physical=True)
        assert _rel(rsf, fsg[tag + "__rsf"]) < 1e-12 and _rel(ac, fsg[tag + "__acube"]) < 1e-12
        assert np.allclose(geom, fsg[tag + "__geom"], rtol=1e-14)
        np.random.seed(41)
        ac2 = cr.realisation(z1, z2, xw, yw, nu_num, x_num, y_num, density_only=True, no_mean=True, no_evolution=True,
                             refinement=2)
        assert _rel(ac2, fsg[tag + "__density_only_nomean_zspace"]) < 1e-12
        np.random.seed(41)
        df, vf = cr._realisation_dv(np.array([50.0, 40.0, 30.0]), np.array([6, 8, 10]))
        assert df.shape == vf.shape == (6, 8, 10)


@pytest.mark.gpu
def test_raytrace_kernel_edges_and_affine(ctx):
    """raytrace_slices vs the definition-level trilinear restatement, with lines of sight that leave the box
    (exact 0 outside [0, n-1], exact sample on the last plane), and cube_affine with and without velocities."""
    from oracle import flatsky as ofs

    rng = np.random.default_rng(3)
    cube = rng.standard_normal((7, 9, 11))
    zc = np.array([0.0, 0.3, 5.999, 6.0, 6.0000001, -1e-12, 3.5])
    scale = np.array([1.0, 0.7, 1.3, 2.0, 1.0, 1.0, 2.5])      # 2.0 / 2.5 push the outer pixels out of the box
    tx = np.linspace(-0.5, 0.5, 6)
    ty = np.linspace(-0.5, 0.5, 5)
    wx, wy = 1.0, 1.0
    got = ctx.raytrace_slices(ctx.to_device(cube), ctx.to_device(zc), ctx.to_device(scale), ctx.to_device(tx),
                              ctx.to_device(ty), wx, wy).cpu().numpy()
    tgy, tgx = np.meshgrid(ty, tx)
    want = np.zeros_like(got)
    for i in range(len(zc)):
        co = np.stack([np.full_like(tgx, zc[i]), (tgx * scale[i]) / wx * 8.0 + 4.0, (tgy * scale[i]) / wy * 10.0 + 5.0])
        want[i] = ofs.trilinear_constant(cube, co)
    assert np.abs(got - want).max() < 1e-14
    assert (got[4] == 0).all() and (got[5] == 0).all() and (got[3, 0, 0] == 0) and np.abs(got[3]).max() > 0
    a, b, c = rng.standard_normal(7), rng.standard_normal(7), rng.standard_normal(7)
    vf = rng.standard_normal(cube.shape)
    dev = [ctx.to_device(x) for x in (cube, vf, a, b, c)]
    full = ctx.cube_affine(dev[0], dev[1], dev[2], dev[3], dev[4]).cpu().numpy()
    assert np.abs(full - (cube * a[:, None, None] + vf * b[:, None, None] + c[:, None, None])).max() < 1e-14
    dens = ctx.cube_affine(dev[0], None, dev[2], dev[3], dev[4]).cpu().numpy()
    assert np.abs(dens - (cube * a[:, None, None] + c[:, None, None])).max() < 1e-15


@pytest.mark.gpu
def test_corr21cm_getfield_device_seed_vs_oracle():
    """Throughput mode of the cube: normals from the device Philox stream == oracle fed with its restatement."""
    from cora_amd.signal import corr21cm
    from oracle import flatsky as ofs
    from oracle import models, philox

    cr = corr21cm.Corr21cm()
    cr.nu_num, cr.x_num, cr.y_num, cr.x_width, cr.y_width, cr.nu_lower, cr.nu_upper = 6, 12, 10, 3.0, 2.5, 600.0, 640.0
    cube = cr.getfield(seed=9)

    def stream(shape, state={"first": True}):
        e = np.arange(int(np.prod(shape)), dtype=np.uint64)
        a, b = philox.boxmuller_counter(9, e & np.uint64(0xFFFFFFFF), e >> np.uint64(32))
        first = state["first"]
        state["first"] = False
        return (a if first else b).reshape(shape)

    z1, z2 = cr._band_redshifts()
    want = ofs.realisation(models.Corr21cm(), z1, z2, 3.0, 2.5, 6, 12, 10, stream, zspace=False)[0][::-1]
    assert cube.shape == (6, 12, 10) and _rel(cube, want) < 1e-11
    assert np.array_equal(cube, cr.getfield(seed=9))


_AB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from cora_amd import _lib
ctx = _lib.get_context()
out = {}
for shape in [(256, 96, 64), (64, 384, 100), (37, 261, 316), (20, 130, 628)]:
    kw = ctx.empty(tuple(shape[:-1]) + (shape[-1] // 2 + 1,)).uniform_(generator=torch.Generator(device=ctx.device).manual_seed(5))
    f = ctx.randomfield_irfftn(kw, 11, last=shape[-1])
    out["f_%d_%d_%d" % shape] = f.cpu().numpy()
    out["s_%d_%d_%d" % shape] = ctx.rfftn(f).cpu().numpy()
np.savez(sys.argv[2], **out)
"""


@pytest.mark.gpu
def test_compile_time_passes_agree_with_the_generic_line_kernel(ctx, tmp_path):
    """The transforms of flatsky_ct.hip (scheduled lengths, Bluestein on the compile-time passes, generated first pass)
    against the generic line kernel of flatsky.hip (CORAHIP_FLAT_GENERIC=1, read once per process: a child process) on
    the same seeded fields: field and its rfftn to rounding."""
    import subprocess
    import sys

    script = tmp_path / "ab.py"
    script.write_text(_AB_SCRIPT)
    res = {}
    for tag, extra in (("ct", {}), ("generic", {"CORAHIP_FLAT_GENERIC": "1"})):
        env = dict(os.environ)
        env.pop("CORAHIP_FLAT_GENERIC", None)
        env.update(extra)
        out = tmp_path / (tag + ".npz")
        p = subprocess.run([sys.executable, str(script), ROOT, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = np.load(out)
    for k in res["ct"].files:
        a, b = res["ct"][k], res["generic"][k]
        assert a.shape == b.shape and _rel(a, b) < 1e-13, (k, _rel(a, b))
