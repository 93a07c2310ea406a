"""Pixel- and element-level parity of the HIP path against the CPU oracle AT BASELINE SIZES (cfg 3: nside 1024,
lmax 2048, 256 channels; cfg 5: nside 2048, lmax 4096, F = 1024).  The small-size tests of test_gpu_parity.py never
reach the transform classes that carry these configurations (ring-FFT Bluestein lengths 2048 / 4096 / 8192, two
channels or one channel per workgroup, the F = 256 / 1024 tiles of K1 - K3); the tests here compare them
directly with the oracle on a few channels / multipoles of a full-size launch.  Run with -m gpu.

Reference code the compared quantities come from: cora/util/hputil.py:369-391,500-531 (alm2map),
cora/util/hputil.py:195-234 (map2alm), cora/core/skysim.py:41-67 (clarray), :114-121 (factor + draw).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ring_classes(nside, lmax):
    """K5 transform class of every ring (0-based, north to south) AS THE PLAN LAUNCHES IT
    (corahip_sht_plan_ring_classes): 0 = direct transform (belt and the cap rings with 4i a power of two), else the
    Bluestein length of the kernel that takes the ring - a power of two >= 2 h - 1 (h = 2 i) or, for the rings whose
    2 h - 1 fits it, 3 * 2^k (ringfft_blu_ct<3072> / <1536>) or 5 * 2^9 / 7 * 2^9."""
    from cora_amd import _lib

    cls = _lib.get_context().sht_ring_classes(nside, lmax).astype(np.int64)
    # the plan's classes against the rule they implement
    for r in range(4 * nside - 1):
        i = r + 1
        icap = i if i < nside else (4 * nside - i if i > 3 * nside else 0)
        h = 2 * icap
        if icap == 0 or h & (h - 1) == 0:
            assert cls[r] == 0, (r, cls[r])
        else:
            # a Bluestein length that holds the ring: 2^k, 3 * 2^k, or 5 * 2^9 / 7 * 2^9 (ringfft_blu_ct<2560> / <3584>)
            odd = int(cls[r])
            while odd % 2 == 0:
                odd //= 2
            assert cls[r] >= 2 * h - 1 and (odd in (1, 3) or cls[r] in (2560, 3584)), (r, cls[r])
    return cls


def _per_class_error(dev_map, ref, nside, lmax, skip_polar=0):
    """max |dev - ref| / rms(ref) per ring-FFT class -> dict {class: err}."""
    from oracle import healpix

    ri = healpix.ring_info(nside)
    cls = _ring_classes(nside, lmax)
    rms = ref.std()
    d = np.abs(dev_map - ref)
    out = {}
    start, nphi = ri["start"], ri["nphi"]
    ring_max = np.maximum.reduceat(d, start.astype(np.int64))
    assert len(ring_max) == len(cls) and int(start[-1] + nphi[-1]) == d.size
    if skip_polar:
        # rings i <= skip_polar of both caps are reported under the key "polar" instead of their class
        nring = len(cls)
        i_cap = np.minimum(np.arange(nring) + 1, nring - np.arange(nring))
        polar = i_cap <= skip_polar
        out["polar"] = float(ring_max[polar].max() / rms)
        cls = np.where(polar, -1, cls)
    for c in np.unique(cls):
        if c >= 0:
            out[int(c)] = float(ring_max[cls == c].max() / rms)
    return out


def _packed_of(alm_dev, f):
    """Channel f of an alm_dev tensor [nalm, G, 2, 4] as a packed complex128 numpy vector."""
    g, v = divmod(f, 4)
    return alm_dev[:, g, 0, v].cpu().numpy() + 1j * alm_dev[:, g, 1, v].cpu().numpy()


def _red_alm(ctx, nalm_shape, lmax, seed):
    import torch

    L = lmax + 1
    gen = torch.Generator(device=ctx.device).manual_seed(seed)
    idx_l = torch.cat([torch.arange(m, L, device=ctx.device) for m in range(L)])
    amp = 1.0 / (1.0 + idx_l.double())
    a = torch.randn(nalm_shape, generator=gen, device=ctx.device, dtype=torch.float64)
    return a * amp[:, None, None, None]


# ------------------------------------------------------------------ a10: synthesis, cfg 3
def test_cfg3_alm2map_pixel_parity_256_channel_launch(ctx):
    """configs[2] geometry, ONE launch of all 256 channels; four channels (first, last, and lanes 1 / 2 of an
    interior cell: both halves of the two-channel workgroups of the P = 4096 class) against the C/OpenMP oracle,
    pixel by pixel, with the error reported per ring-FFT class."""
    import torch
    from oracle import sht

    nside, lmax, F = 1024, 2048, 256
    nalm = (lmax + 1) * (lmax + 2) // 2
    alm = _red_alm(ctx, (nalm, F // 4, 2, 4), lmax, 31)
    maps = ctx.alm2map(alm, nside, lmax, F)
    worst = {}
    for f in (0, 129, 130, 255):
        ref = sht.alm2map(_packed_of(alm, f), nside, lmax)
        err = _per_class_error(maps[f].cpu().numpy(), ref, nside, lmax)
        for c, e in err.items():
            worst[c] = max(worst.get(c, 0.0), e)
    del maps, alm
    torch.cuda.empty_cache()
    print("cfg3 alm2map max|err|/rms per class (0 = direct, else Bluestein P):", worst)
    # every kernel class of the cfg-3 step must have been compared: belt + power-of-two caps (0), ringfft_blu_ct of
    # the lengths 4096, 3072, 2048, 1536, 1024 and the run-time kernel's short classes
    assert set(worst) >= {0, 1024, 1536, 2048, 2560, 3072, 3584, 4096}, worst
    assert max(worst.values()) <= 1e-11, worst


def test_cfg5_alm2map_pixel_parity(ctx):
    """configs[4] geometry (nside 2048, lmax 4096): an 8-channel launch, channels 0 and 5 against the oracle pixel
    by pixel; classes here: belt h = 4096 (two channels per workgroup), Bluestein P = 4096 / 3072 (two) and the
    one-channel compile-time kernels of P = 6144 (24 x 16 x 16) and 8192 (32 x 16 x 16) for the rings 1025 .. 2047; the
    same launch through the generic run-time kernels must agree."""
    import torch
    from oracle import sht

    nside, lmax, nnu = 2048, 4096, 8
    nalm = (lmax + 1) * (lmax + 2) // 2
    alm = _red_alm(ctx, (nalm, 2, 2, 4), lmax, 32)
    maps = ctx.alm2map(alm, nside, lmax, nnu)
    worst = {}
    for f in (0, 5):
        ref = sht.alm2map(_packed_of(alm, f), nside, lmax)
        for c, e in _per_class_error(maps[f].cpu().numpy(), ref, nside, lmax).items():
            worst[c] = max(worst.get(c, 0.0), e)
    del maps, alm
    torch.cuda.empty_cache()
    print("cfg5 alm2map max|err|/rms per class:", worst)
    assert set(worst) >= {0, 3072, 4096, 6144, 8192}, worst
    # l^2 eps growth of the fp64 three-term recurrence: 4x the cfg-3 bound at lmax = 4096
    assert max(worst.values()) <= 4e-11, worst


# ------------------------------------------------------------------ n1: analysis, cfg-3 geometry
def test_cfg3_map2alm_quadrature_pass_vs_oracle(ctx):
    """One unweighted quadrature pass (K5^T + K4^T) of 8 white-noise maps at nside 1024 / lmax 2048; channels 0
    and 5 against oracle.sht.map2alm_adjoint element by element."""
    import torch
    from cora_amd.util import hputil
    from oracle import sht

    nside, lmax, nnu = 1024, 2048, 8
    npix = 12 * nside * nside
    gen = torch.Generator(device=ctx.device).manual_seed(33)
    x = torch.randn((nnu, npix), generator=gen, device=ctx.device, dtype=torch.float64)
    alm = ctx.map2alm(x, nside, lmax, None)
    got = ctx.alm_dev_to_square(alm, lmax, nnu)
    for k in (0, 5):
        ref = hputil.unpack_alm(sht.map2alm_adjoint(x[k].cpu().numpy(), nside, lmax, None), lmax)
        err = np.abs(got[k, 0].cpu().numpy() - ref).max() / np.abs(ref).max()
        print("cfg3 map2alm channel", k, "max|err|/max|ref| =", err)
        assert err <= 1e-12, (k, err)
    del x, alm, got
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ n4: spin-2 synthesis, cfg-3 geometry
def test_cfg3_alm2map_spin2_pixel_parity(ctx):
    """(E, B) -> (Q, U) at nside 1024 / lmax 2048, four frequencies (8 interleaved channels); the first and last
    (Q, U) pair against the oracle pixel by pixel, per ring-FFT class."""
    import torch
    from oracle import sht

    nside, lmax, nf = 1024, 2048, 4
    L = lmax + 1
    nalm = L * (L + 1) // 2
    alm = _red_alm(ctx, (nalm, 2, 2, 4), lmax, 34)
    alm[:L, :, 1, :] = 0.0                      # Im a_l0 = 0
    l_of = torch.cat([torch.arange(m, L, device=ctx.device) for m in range(L)])
    alm[l_of < 2] = 0.0                         # spin-2: l >= 2
    maps = ctx.alm2map_spin2(alm, nside, lmax, 2 * nf)
    worst = {}
    for f in (0, nf - 1):
        q, u = sht.alm2map_spin2(_packed_of(alm, 2 * f), _packed_of(alm, 2 * f + 1), nside, lmax)
        for name, ref, dev in (("Q", q, maps[2 * f]), ("U", u, maps[2 * f + 1])):
            for c, e in _per_class_error(dev.cpu().numpy(), ref, nside, lmax, skip_polar=8).items():
                worst[c] = max(worst.get(c, 0.0), e)
    del maps, alm
    torch.cuda.empty_cache()
    print("cfg3 spin-2 max|err|/rms per class:", worst)
    # the spin-2 operands carry r1 = 1/sin^2(theta), r2 = cos/sin^2 (1.5e6 on the first ring of nside 1024): W and X
    # are differences of terms that large, so the rounding of the scalar recurrence (l^2 eps) is amplified on the
    # few rings next to the poles - in the oracle as much as on the device, with a different operation order
    polar = worst.pop("polar")
    assert polar <= 1e-9, polar
    assert max(worst.values()) <= 1e-11, worst


# ------------------------------------------------------------------ a1/a2/a3: C_l rows at F = 256 and F = 1024
def _freqs(F):
    return 400.0 + (np.arange(F) + 0.5) * (400.0 / F)


def test_cfg3_clarray_rows_vs_oracle(model21):
    """K1 at configs[2] size (F = 256, zromb 3, lmax 2048: the F = 256 tile mirroring, the l = 2048 table rows):
    multipoles 0, 1, 2, 700, 2048 of the full launch against oracle.skysim.clarray, element by element."""
    import torch
    from cora_amd.core import skysim
    from cora_amd.signal import corr21cm
    from oracle import skysim as osk

    F, lmax = 256, 2048
    rows = [0, 1, 2, 700, 2048]
    freq = _freqs(F)
    C = skysim.clarray_device(corr21cm.Corr21cm().angular_powerspectrum, lmax, freq, zromb=3)
    got = C[rows].cpu().numpy()
    sym = (C - C.transpose(1, 2)).abs().max().item()
    del C
    torch.cuda.empty_cache()
    ref = osk.clarray(model21.angular_powerspectrum, lmax, freq, zromb=3, rows=rows)
    for k, l in enumerate(rows):
        err = np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max()
        print("cfg3 C_l row", l, "max|err|/max =", err)
        assert err <= 1e-11, (l, err)
    assert sym == 0.0


def test_eor_band_clarray_rows_vs_oracle(model21):
    """K1 at configs[2] size on the EoR band (EoR21cm, F = 256 channels 100-200 MHz, zromb 3, lmax 2048): table rows
    x ~ 300-345 at l = 2048 and y up to ~19000 columns - twice the |chi - chi'| of the 400-800 MHz band -, multipoles
    0, 1, 2, 700, 2048 of the full launch against the oracle's EoR21cm, element by element."""
    import torch
    from cora_amd.core import skysim
    from cora_amd.signal import corr21cm
    from oracle import models
    from oracle import skysim as osk

    F, lmax = 256, 2048
    rows = [0, 1, 2, 700, 2048]
    freq = 100.0 + (np.arange(F) + 0.5) * (100.0 / F)
    C = skysim.clarray_device(corr21cm.EoR21cm().angular_powerspectrum, lmax, freq, zromb=3)
    got = C[rows].cpu().numpy()
    sym = (C - C.transpose(1, 2)).abs().max().item()
    del C
    torch.cuda.empty_cache()
    ref = osk.clarray(models.EoR21cm(share=model21).angular_powerspectrum, lmax, freq, zromb=3, rows=rows)
    for k, l in enumerate(rows):
        err = np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max()
        print("EoR band C_l row", l, "max|err|/max =", err)
        assert err <= 1e-11, (l, err)
    assert sym == 0.0


@pytest.fixture(scope="module")
def cfg5_cl(ctx):
    """C_l of configs[4] (F = 1024, lmax 4096, zromb 3; 34.4 GB) integrated once on the device."""
    import torch
    from cora_amd.core import skysim
    from cora_amd.signal import corr21cm

    F, lmax = 1024, 4096
    C = skysim.clarray_device(corr21cm.Corr21cm().angular_powerspectrum, lmax, _freqs(F), zromb=3)
    yield C
    del C
    torch.cuda.empty_cache()


def test_cfg5_clarray_rows_vs_oracle(cfg5_cl, model21):
    """K1 at configs[4] size: multipoles 1 and 4096 of the F = 1024 launch against the oracle."""
    from oracle import skysim as osk

    rows = [1, 4096]
    got = cfg5_cl[rows].cpu().numpy()
    ref = osk.clarray(model21.angular_powerspectrum, 4096, _freqs(1024), zromb=3, rows=rows)
    for k, l in enumerate(rows):
        err = np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max()
        print("cfg5 C_l row", l, "max|err|/max =", err)
        assert err <= 1e-11, (l, err)
        assert np.array_equal(got[k], got[k].T)


def test_cfg5_factor_and_draw_F1024_vs_numpy(ctx, cfg5_cl):
    """K2 + K3 at F = 1024 (1024 x 1024 Cholesky, 8 column groups in the draw): eight covariance blocks of the
    configs[4] C_l are factored and drawn from as the l = 0..7 blocks of an lmax = 7 problem and compared with
    numpy (scipy Cholesky of the jittered block, T g / sqrt 2 with the reference's normal order)."""
    import scipy.linalg as la
    import torch

    F = 1024
    ls = [1, 2, 100, 1000, 2048, 3000, 4000, 4096]
    Csub = cfg5_cl[ls].contiguous()
    T, info = ctx.factor_batched(Csub)                       # jitter 1e-14 max(diag) as skysim.py:116-117
    assert int(info.abs().sum().item()) == 0
    Th, Ch = T.cpu().numpy(), Csub.cpu().numpy()
    lmax = len(ls) - 1
    for k in range(len(ls)):
        Cj = Ch[k] + np.eye(F) * Ch[k].diagonal().max() * 1e-14
        ref = la.cholesky(Cj, lower=True)
        err = np.abs(Th[k] - ref).max() / np.abs(ref).max()
        res = np.abs(Th[k] @ Th[k].T - Cj).max() / np.abs(Cj).max()
        print("F=1024 block l=%d: |T - chol|/max = %.2e, |T T^T - C|/max = %.2e" % (ls[k], err, res))
        assert np.all(np.triu(Th[k], 1) == 0)
        assert res <= 1e-13, (ls[k], res)
        assert err <= 1e-9, (ls[k], err)                      # cond(C) ~ 1e3..1e5 at 0.39 MHz channels
    rng = np.random.default_rng(1024)
    nalm = (lmax + 1) * (lmax + 2) // 2
    g = np.empty(2 * F * nalm)
    o = 0
    blocks = []
    for l in range(lmax + 1):
        re = rng.standard_normal((F, l + 1))
        im = rng.standard_normal((F, l + 1))
        blocks.append((re + 1j * im) / 2**0.5)
        n = F * (l + 1)
        g[o : o + n] = re.ravel()
        g[o + n : o + 2 * n] = im.ravel()
        o += 2 * n
    alm = ctx.draw_alm(T, info, ctx.to_device(g), lmax, F)
    sq = ctx.alm_dev_to_square(alm, lmax, F).cpu().numpy()
    for l in range(lmax + 1):
        ref = Th[l] @ blocks[l]
        err = np.abs(sq[:, 0, l, : l + 1] - ref).max() / np.abs(ref).max()
        assert err <= 1e-12, (l, err)
    # the fused-RNG kernel on the same factors: equal to the materialised Philox stream
    gp = ctx.normals_philox(77, lmax, F)
    a = ctx.alm_dev_to_square(ctx.draw_alm(T, info, gp, lmax, F), lmax, F)
    b = ctx.alm_dev_to_square(ctx.draw_alm_philox(T, info, 77, lmax, F), lmax, F)
    assert (a - b).abs().max().item() <= 1e-13 * a.abs().max().item()
    # and a frequency shard of it (the 128 channels one of eight ranks owns)
    c = ctx.alm_dev_to_square(ctx.draw_alm_philox(T, info, 77, lmax, F, nu0=640, nnu=128), lmax, 128)
    assert torch.equal(c, b[640:768])


def test_default_lmax_aliased_rings_pixel_parity(ctx):
    """The reference's own default, lmax = 3 nside - 1 (cora/core/maps.py:230), at nside 1024: lmax = 3071 exceeds
    half the ring length on EVERY ring, so the fold of the ring FFT aliases everywhere (belt included) and the cell
    rows are longer than the prefetch window of the compile-time kernels - paths the lmax = 2 nside configurations
    never take.  Channels 0 and 6 of an 8-channel launch against the oracle, per ring-FFT class."""
    import torch
    from oracle import sht

    nside, lmax, nnu = 1024, 3071, 8
    nalm = (lmax + 1) * (lmax + 2) // 2
    alm = _red_alm(ctx, (nalm, 2, 2, 4), lmax, 35)
    maps = ctx.alm2map(alm, nside, lmax, nnu)
    worst = {}
    for f in (0, 6):
        ref = sht.alm2map(_packed_of(alm, f), nside, lmax)
        for c, e in _per_class_error(maps[f].cpu().numpy(), ref, nside, lmax).items():
            worst[c] = max(worst.get(c, 0.0), e)
    del maps, alm
    torch.cuda.empty_cache()
    print("nside 1024 / lmax 3071 alm2map max|err|/rms per class:", worst)
    assert max(worst.values()) <= 2e-11, worst


@pytest.mark.parametrize("nside,lmax,nnu", [(1024, 700, 5), (1024, 1500, 8), (512, 1024, 12), (2048, 2048, 3),
                                            (1024, 2048, 32), (1024, 2048, 16)])
def test_compile_time_ring_kernels_other_shapes(ctx, nside, lmax, nnu):
    """The compile-time ring-FFT kernels away from lmax = 2 nside and from whole channel groups: short cell rows (the
    register prefetch window reaches past the row), ragged channel counts (3, 5, 12: padding lanes of the 4- and
    2-channel workgroups), nside 512 (Bluestein classes 2048 / 1536 / 1024 compile-time, belt run-time) and nside
    2048 at lmax = nside.  (1024, 2048, 32) and (1024, 2048, 16) are what one rank of an 8- / 16-way frequency shard
    of cfg 3 launches: legendre_kernel<4, 2> / <2, 2> (64 / 32 columns, two recurrences per lane) and the ring-FFT
    grids of 8 / 4 channel groups.  First and last channel against the oracle, per class."""
    import torch
    from oracle import sht

    nalm = (lmax + 1) * (lmax + 2) // 2
    G = (nnu + 3) // 4
    alm = _red_alm(ctx, (nalm, G, 2, 4), lmax, 36 + nnu)
    maps = ctx.alm2map(alm, nside, lmax, nnu)
    worst = {}
    for f in (0, nnu - 1):
        ref = sht.alm2map(_packed_of(alm, f), nside, lmax)
        for c, e in _per_class_error(maps[f].cpu().numpy(), ref, nside, lmax).items():
            worst[c] = max(worst.get(c, 0.0), e)
    del maps, alm
    torch.cuda.empty_cache()
    print("nside %d lmax %d nnu %d: max|err|/rms per class: %s" % (nside, lmax, nnu, worst))
    assert max(worst.values()) <= 2e-11, worst


# ------------------------------------------------------------------ (d): the executed-flop count behind roofline.frac
def test_k4_mfma_count_matches_pmc_profile(ctx):
    """bench.py prices K4's roofline on the FP64 MFMA instructions the launch ISSUES, counted at run time from the
    plan's first-contributing-l tables (corahip_sht_plan_k4_mfma_count).  That count must be what the hardware
    counter read for the same launch: SQ_INSTS_VALU_MFMA_F64 of legendre_kernel<8, 1> in the committed rocprofv3
    --pmc profile of `bench.py` (cfg 3: nside 1024, lmax 2048, 256 channels), within 1 %."""
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pmc = None
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
        f = os.path.join(root, "profiles", name)
        if os.path.exists(f):
            k = json.load(open(f))["kernels"]
            pmc = next(v["SQ_INSTS_VALU_MFMA_F64_per_launch"] for n, v in k.items() if n.startswith("legendre_kernel<8, 1>"))
            break
    assert pmc, "no committed PMC profile with the K4 MFMA counter"
    n = ctx.k4_mfma_count(1024, 2048, 256)
    alg = 8.0 * 1024 * (2049 * 2050 // 2) * 256 / 2048.0       # SURVEY 8(d) count in MFMA instructions
    print("K4 MFMA instructions per cfg-3 launch: plan %d, PMC %.0f (ratio %.4f); algorithmic %.4g -> executed share %.3f"
          % (n, pmc, n / pmc, alg, n / alg))
    assert abs(n / pmc - 1.0) <= 0.01, (n, pmc)
    assert n < alg
    # shard shapes: per-channel count is the same for the wide shape at any multiple of 64 channels, and the
    # 64 / 32-column shapes (<4, 2>, <2, 2>: 256-ring tiles) execute a little more per channel
    assert ctx.k4_mfma_count(1024, 2048, 64) * 4 == n
    n32 = ctx.k4_mfma_count(1024, 2048, 32)
    assert n / 8 <= n32 <= 1.1 * n / 8, (n32, n / 8)
    # a plan without truncation executes (nearly) the algorithmic count
    n900 = ctx.k4_mfma_count(1024, 2048, 256, cut_exp=-900)
    print("cut 2^-900: %d MFMA instructions (%.3f of algorithmic)" % (n900, n900 / alg))
    assert n < n900 <= 1.02 * alg


def test_legendre_cut_is_a_plan_parameter(ctx):
    """corahip_sht_plan_create_ex: the truncation exponent of the Legendre sums is chosen per plan.  With the cut
    the oracle uses (2^-900) the device maps equal the oracle's to the digits the default plan (2^-70) gives - the
    default drops nothing that matters for O(1) coefficients - and coefficients of extreme dynamic range, where
    the documented bound sum |a_lm| 2^cut says the default is NOT enough, are reproduced only by the lower cut."""
    from oracle import sht

    nside, lmax, nnu = 256, 512, 8
    nalm = (lmax + 1) * (lmax + 2) // 2
    alm = _red_alm(ctx, (nalm, 2, 2, 4), lmax, 77)
    ref = sht.alm2map(_packed_of(alm, 3), nside, lmax)
    e80 = np.abs(ctx.alm2map(alm, nside, lmax, nnu)[3].cpu().numpy() - ref).max() / ref.std()
    e900 = np.abs(ctx.alm2map(alm, nside, lmax, nnu, cut_exp=-900)[3].cpu().numpy() - ref).max() / ref.std()
    print("nside 256 / lmax 512: max|err|/rms with the default cut: %.2e, with 2^-900: %.2e" % (e80, e900))
    assert e80 <= 1e-11 and e900 <= 1e-11
    # one huge coefficient at high m next to an O(1) sky: on the polar rings lambda_lm of that mode is below the cut
    # but 1e40 x it is not negligible - the bound sum |a_lm| 2^cut tells the caller so
    m0 = 500
    idx = m0 * (2 * lmax + 1 - m0) // 2 + lmax
    alm2 = alm.clone()
    alm2[idx, 0, 0, 3] = 1e40
    ref2 = sht.alm2map(_packed_of(alm2, 3), nside, lmax)
    from oracle import healpix

    start = healpix.ring_info(nside)["start"].astype(np.int64)
    ring_of = lambda x: np.maximum.reduceat(x, start)            # noqa: E731  (per-ring maximum)
    rmax = ring_of(np.abs(ref2))
    d80 = np.abs(ctx.alm2map(alm2, nside, lmax, nnu)[3].cpu().numpy() - ref2)
    d900 = np.abs(ctx.alm2map(alm2, nside, lmax, nnu, cut_exp=-900)[3].cpu().numpy() - ref2)
    rel80, rel900 = (ring_of(d80) / rmax).max(), (ring_of(d900) / rmax).max()
    cut = ctypes_int()
    assert ctx.lib.corahip_sht_plan_cut_exp(ctx.sht_plan(nside, lmax), cut) == 0 and cut.value == -70      # the default
    bound = 2.0 * 1e40 * 2.0**cut.value                          # 2 sum |a_lm| 2^cut (c_m = 2 for m > 0)
    print("1e40 coefficient: worst ring error / ring max with the default cut: %.2e, with 2^-900: %.2e; max abs error of the "
          "default plan on the rings it truncates %.3e <= bound %.3e" % (rel80, rel900, ring_of(d80)[ring_of(d80) > 1e-6 * rmax].max(), bound))
    assert rel900 <= 1e-7, rel900                    # (l^2 eps of the recurrence relative to the ring's own scale)
    assert rel80 >= 1e-3, rel80                      # the default cut visibly drops the mode where it is below it ...
    assert np.all(ring_of(d80) <= bound * 1.01 + 1e-7 * rmax)     # ... and never more than the documented bound
    assert ctx.lib.corahip_sht_plan_cut_exp(ctx.sht_plan(nside, lmax, -900), cut) == 0 and cut.value == -900


def ctypes_int():
    import ctypes

    return ctypes.c_int()


# ------------------------------------------------------------------ a7 + a9: the reference's seeded call at cfg-3 size
def test_cfg3_seeded_mkfullsky_channels_vs_oracle(ctx):
    """``mkfullsky(C[2049, 256, 256], 1024, rng=default_rng(s))`` - the reference's own call (cora/core/skysim.py:72-136)
    at the headline configuration, with numpy's stream continued on the DEVICE - against the oracle chain for the same
    seed on channels 0 and 255: numpy's normals (host), scipy Cholesky of the jittered blocks, T_l g_l, the C/OpenMP
    synthesis.  Tolerance: 1e-10 x rms (21cm blocks are well conditioned)."""
    import scipy.linalg as la
    import torch
    from cora_amd.core import skysim
    from cora_amd.signal import corr21cm
    from oracle import sht as osht

    F, lmax, nside, seed = 256, 2048, 1024, 31
    L = lmax + 1
    Cd = skysim.clarray_device(corr21cm.Corr21cm().angular_powerspectrum, lmax, _freqs(F), zromb=3)
    rng = np.random.default_rng(seed)
    maps = skysim.mkfullsky_device(Cd, nside, rng=rng)
    got = {f: maps[f].cpu().numpy() for f in (0, F - 1)}
    del maps
    C = Cd.cpu().numpy()
    del Cd
    torch.cuda.empty_cache()
    twin = np.random.default_rng(seed)
    nalm = L * (L + 1) // 2
    packed = {0: np.zeros(nalm, dtype=np.complex128), F - 1: np.zeros(nalm, dtype=np.complex128)}
    m_all = np.arange(L)
    for l in range(L):
        cm = C[l] + np.identity(F) * C[l].diagonal().max() * 1e-14
        T = la.cholesky(cm, lower=True)
        re = twin.standard_normal((F, l + 1))
        im = twin.standard_normal((F, l + 1))
        idx = m_all[:l + 1] * (2 * lmax + 1 - m_all[:l + 1]) // 2 + l
        for f in packed:
            packed[f][idx] = (T[f] @ re + 1j * (T[f] @ im)) / np.sqrt(2.0)
    # the caller's generator is where numpy leaves it after the same draws
    assert rng.bit_generator.state == twin.bit_generator.state
    for f in packed:
        ref = osht.alm2map(packed[f], nside, lmax, rings_c=True)
        err = np.abs(got[f] - ref).max() / ref.std()
        print("cfg3 seeded mkfullsky channel", f, "max|err|/rms =", err)
        assert err <= 1e-10, (f, err)


def test_rank_memory_model_against_torch_at_cfg3(ctx):
    """parallel.rank_memory_bytes (the figure tests/test_host.py holds against 288 GB for the 8-GPU configurations) is
    what a rank really allocates: one cold cfg-3 step on one GPU, torch's peak allocation within 25 % of the model."""
    import torch
    from cora_amd.parallel import SkyShard, rank_memory_bytes
    from cora_amd.signal import corr21cm

    F, nside, lmax = 256, 1024, 2048
    torch.cuda.synchronize()
    ctx._workspace = None
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    shard = SkyShard(corr21cm.Corr21cm(), _freqs(F), nside, lmax, zromb=3, ctx=ctx)
    shard.realise(5)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    model = rank_memory_bytes(["table21cm"], F, nside, lmax, 1)["total"]
    print("cfg3 peak allocation %.1f GB, model %.1f GB" % (peak / 1e9, model / 1e9))
    assert 0.75 * model <= peak <= 1.25 * model, (peak, model)
    del shard


def test_k2_cooperative_kernel_matches_one_workgroup_form(ctx, monkeypatch):
    """The cooperative Cholesky (G workgroups per matrix: the straggler matrices of a tall batch, csrc/factor.hip
    chol_coop_kernel) against the one-workgroup form on the same matrices, F = 1024 and 512: equal to rounding (same
    tile arithmetic; the panel solve and the tile sums run in the same order), info flags equal - including a matrix
    that is not positive definite, whose eigen-route root must come out the same - and T T^T = C."""
    import torch

    for F, nl, bad in ((1024, 3, False), (512, 11, False), (384, 4, True)):     # (11 matrices: two rows of XCD groups)
        A = ctx.empty((nl, F, F + 8)).normal_()
        C = A @ A.transpose(1, 2) + 0.1 * torch.eye(F, device=ctx.device, dtype=torch.float64)
        del A
        if bad:
            C[1, 300, 300] = -1.0                          # block 1: Cholesky fails in block column 300 / 32 (eigen route)
        monkeypatch.setenv("CORAHIP_K2_COOP", "0")
        T0, i0 = ctx.factor_batched(C)
        T0, i0 = T0.clone(), i0.clone()
        monkeypatch.setenv("CORAHIP_K2_COOP", "2")         # every matrix through the cooperative kernel
        T1, i1 = ctx.factor_batched(C)
        monkeypatch.delenv("CORAHIP_K2_COOP")
        assert torch.equal(i0, i1), (i0, i1)
        assert (int(i0[1].item()) != 0) == bad and int(i0[0].item()) == 0
        scale = T0.abs().amax().item()
        err = (T0 - T1).abs().amax().item() / scale
        print("F=%d: cooperative vs one-workgroup Cholesky, max |dT| / max |T| = %.2e" % (F, err))
        assert err <= 1e-13, (F, err)
        for k in (0, nl - 1):
            jit = C[k].diagonal().max() * 1e-14
            res = (T1[k] @ T1[k].T - C[k] - jit * torch.eye(F, device=ctx.device, dtype=torch.float64)).abs().amax().item()
            assert res <= 1e-12 * C[k].abs().amax().item(), (F, k, res)
            assert torch.all(torch.triu(T1[k], 1) == 0)


@pytest.mark.parametrize("nrem", [1, 7, 16])
def test_k2_cooperative_kernel_repeats_bit_identically(ctx, monkeypatch, nrem):
    """Stress of the cooperative Cholesky's same-XCD barrier (round-4 advice: a CU re-reads tiles another CU rewrote in
    place; since round 5 the acquire side invalidates the vector L1): F = 1024, the straggler counts 1 / 7 / 16 of a
    cfg-5 rank (L mod 512), every matrix through the cooperative kernel, 25 repetitions - every repetition bit-identical
    to the first and to the one-workgroup kernel's factor within rounding; a stale line would show as a changed factor."""
    import torch

    F = 1024
    A = ctx.empty((nrem, F, F + 8)).normal_()
    C = A @ A.transpose(1, 2) + 0.1 * torch.eye(F, device=ctx.device, dtype=torch.float64)
    del A
    monkeypatch.setenv("CORAHIP_K2_COOP", "0")
    T0, i0 = ctx.factor_batched(C)
    T0 = T0.clone()
    monkeypatch.setenv("CORAHIP_K2_COOP", "2")
    first = None
    for rep in range(25):
        T1, i1 = ctx.factor_batched(C)
        assert int(i1.abs().max().item()) == 0
        if first is None:
            first = T1.clone()
            assert (first - T0).abs().amax().item() <= 1e-13 * T0.abs().amax().item()
        else:
            assert torch.equal(T1, first), rep
    monkeypatch.delenv("CORAHIP_K2_COOP")
