"""The reference's own seeded normal stream on the device (corahip_normals_pcg64) - run with -m gpu.

cora draws every normal of mkfullsky from the caller's numpy Generator (cora/util/nputil.py:121-125 called from
cora/core/skysim.py:120; ``default_rng(seed)`` in cora/signal/lss.py:449-450).  numpy is on the GPU box, so the device
stream is compared with numpy ITSELF, bit for bit: fast-path, wedge and tail samples, the raw draws consumed (= the
state the generator is left in), block / tile boundaries of the kernels' decomposition, and the layouts and callers on
top (stream order of complex_std_normal, the reference's golden a_lm, K3 reading the stream buffer).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ZIG_R = 3.6541528853610088


def _device_stream(ctx, rng, n):
    from cora_amd import _lib

    st = rng.bit_generator.state["state"]
    g, nraw = ctx.normals_pcg64(st["state"], st["inc"], n)
    return g.cpu().numpy(), _lib.pcg64_advance(st["state"], st["inc"], nraw)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 4096, 100003, 524288 * 2 + 77])
def test_device_stream_is_numpys_small_and_boundary_sizes(ctx, n):
    """Sizes around the kernels' row (64), block (1024) and tile (512 blocks) boundaries, from a generator that has
    already been used (a state that is not a seed state)."""
    for seed in (0, 1, 2):
        rng = np.random.default_rng(1000 * n + seed)
        rng.standard_normal(seed * 37)
        dev, state_after = _device_stream(ctx, rng, n)
        ref = rng.standard_normal(n)
        assert np.array_equal(dev.view(np.uint64), ref.view(np.uint64)), (n, seed)
        assert state_after == int(rng.bit_generator.state["state"]["state"]), (n, seed)


def test_device_stream_is_numpys_over_1e8_samples(ctx):
    """1.2e8 normals: every value equal to numpy's bit for bit - 3e4 tail samples (glibc's log1p restated on the device)
    and 9e5 accepted wedge samples among them - and the same number of raw draws consumed."""
    n = 120_000_000
    rng = np.random.default_rng(20240)
    dev, state_after = _device_stream(ctx, rng, n)
    ref = rng.standard_normal(n)
    same = dev.view(np.uint64) == ref.view(np.uint64)
    ntail = int((np.abs(ref) > ZIG_R).sum())
    assert ntail > 25_000
    if not same.all():
        bad = np.flatnonzero(~same)
        ulp = np.abs(dev.view(np.int64)[bad] - ref.view(np.int64)[bad])
        # (a libm other than glibc may round its log1p differently: tail samples only, one ulp at most)
        assert np.all(np.abs(ref[bad]) > ZIG_R) and ulp.max() <= 1, (bad[:5], ulp.max())
    assert state_after == int(rng.bit_generator.state["state"]["state"])
    assert abs(dev.mean()) < 5e-4 and abs(dev.std() - 1.0) < 5e-4


def test_device_exp_of_the_wedge_test_is_the_hosts_exp(ctx):
    """The wedge test of numpy's ziggurat compares against libm's exp(-x^2 / 2) (random_standard_normal, reached from
    cora/util/nputil.py:125); a last-bit difference at the threshold would flip an accept and shift every later sample.
    The device runs glibc's routine operation by operation (glibc_exp_fma): 1.2e7 arguments - the wedge test's own
    arguments -x^2 / 2 for x across the ziggurat, a dense sweep of [-6.7, 0), neighbours of every reduction boundary
    k ln2 / 128, tiny arguments - against numpy's exp on the host, bit for bit.  Skipped where the host's libm is not
    glibc's FMA build (tests/test_oracle.py pins the same restatement on the CPU)."""
    import math

    from oracle import npnormal

    probe = [-0.5, -6.6, -1e-3, -3.21, -0.6931471805599453, -2.0**-30, -5.0]
    if not all(npnormal.glibc_exp_fma(x) == math.exp(x) for x in probe):
        pytest.skip("this host's exp is not glibc's FMA build")
    rs = np.random.default_rng(5)
    xw = ZIG_R * rs.random(5_000_000)
    parts = [-0.5 * xw * xw, -6.7 * rs.random(5_000_000), -rs.random(500_000) * 2.0 ** (-rs.integers(1, 60, 500_000).astype(np.float64))]
    kb = -(np.arange(1300) + 0.5) * (math.log(2.0) / 128.0)
    parts.append(np.concatenate([np.nextafter(kb, 0.0), kb, np.nextafter(kb, -10.0)] * 400) + 0.0)
    parts.append(np.array([-2.0**-54, -2.0**-55, -0.0, 0.0, -6.7, -2.0**-53]))
    x = np.concatenate(parts)
    assert x.size > 12_000_000
    y = ctx.glibc_exp(ctx.to_device(x)).cpu().numpy()
    ref = np.exp(x)
    if not np.array_equal(np.array([math.exp(v) for v in x[:2000]]), ref[:2000]):
        ref = np.array([math.exp(v) for v in x])          # (numpy's vectorised exp is not libm's on this host: take libm's)
    assert np.array_equal(y.view(np.uint64), ref.view(np.uint64)), int((y != ref).sum())


def test_serial_walk_of_the_top_scan_is_numpys_too(ctx, monkeypatch):
    """The top level of the two-pass generator's scan composes its 1024 thread functions by one wave (round 6); a stream in
    which a thread hands k >= 2 to its successor (a tail sample across the boundary: ~15 % of the cfg-3 streams) falls back to
    the serial walk - forced here (CORAHIP_ZIG_TOP_SERIAL=1), so that the fall-back is exercised whatever the seeds do."""
    monkeypatch.setenv("CORAHIP_ZIG_TOP_SERIAL", "1")
    for n in (5, 100003, 20_000_003):
        rng = np.random.default_rng(900 + n)
        dev, state_after = _device_stream(ctx, rng, n)
        ref = rng.standard_normal(n)
        same = dev.view(np.uint64) == ref.view(np.uint64)
        if not same.all():
            bad = np.flatnonzero(~same)
            ulp = np.abs(dev.view(np.int64)[bad] - ref.view(np.int64)[bad])
            assert np.all(np.abs(ref[bad]) > ZIG_R) and ulp.max() <= 1, (bad[:5], ulp.max())
        assert state_after == int(rng.bit_generator.state["state"]["state"])


@pytest.mark.parametrize("n", [1, 65, 1024, 4097, 100003, 3 * 4096 * 64 + 5, 30_000_011])
def test_single_pass_form_is_numpys_too(ctx, monkeypatch, n):
    """The round-6 single pass (classify + chained scan with decoupled look-back + emit in ONE launch,
    CORAHIP_ZIG_ONEPASS=1: slower than the two-pass default, kept as the measured alternative) produces the same
    stream: sizes around its row / block / chunk (4 blocks) / look-back window (64 chunks) boundaries and 3e7 samples
    (4e3 blocks handing k >= 2 across a boundary: the request path inside a chunk and the wait across chunks)."""
    monkeypatch.setenv("CORAHIP_ZIG_ONEPASS", "1")
    rng = np.random.default_rng(77 + n)
    rng.standard_normal(n % 41)
    dev, state_after = _device_stream(ctx, rng, n)
    ref = rng.standard_normal(n)
    same = dev.view(np.uint64) == ref.view(np.uint64)
    if not same.all():      # (a libm other than glibc: tail samples only, one ulp at most - as above)
        bad = np.flatnonzero(~same)
        ulp = np.abs(dev.view(np.int64)[bad] - ref.view(np.int64)[bad])
        assert np.all(np.abs(ref[bad]) > ZIG_R) and ulp.max() <= 1, (bad[:5], ulp.max())
    assert state_after == int(rng.bit_generator.state["state"]["state"])


@pytest.mark.parametrize("F,lmax", [(5, 33), (8, 64), (40, 200)])
def test_stream_normals_order_and_generator_state(ctx, F, lmax):
    """skysim.stream_normals = the reference's draw order (per l: F (l + 1) reals, then the imaginaries; nputil.py:121-125)
    and leaves the caller's Generator exactly where numpy would: the next draws agree, a buffered 32-bit half included."""
    from cora_amd.core import skysim

    a, b = np.random.default_rng(77), np.random.default_rng(77)
    for r in (a, b):
        r.integers(0, 2**32, dtype=np.uint32)              # leaves has_uint32 = 1 in the bit generator's state
    g_dev = skysim.stream_normals(ctx, F, lmax, a).cpu().numpy()
    g_ref = skysim._host_normals(F, lmax, b)
    assert np.array_equal(g_dev.view(np.uint64), g_ref.view(np.uint64))
    assert a.bit_generator.state == b.bit_generator.state
    assert np.array_equal(a.standard_normal(100), b.standard_normal(100))
    assert a.integers(0, 2**32, dtype=np.uint32) == b.integers(0, 2**32, dtype=np.uint32)


def test_other_generators_are_consumed_on_the_host(ctx, monkeypatch):
    """Generator(PCG64) and the legacy MT19937 state (rng=None, RandomState) are continued on the device; a Generator on
    any other bit generator keeps the host path (same values as the reference draws)."""
    from cora_amd.core import skysim

    def boom(*a, **k):
        raise AssertionError("device PCG64 stream used for a generator that is not PCG64")

    monkeypatch.setattr(type(ctx), "normals_pcg64", boom)
    F, lmax = 6, 20
    for make in (lambda: np.random.Generator(np.random.MT19937(5)), lambda: np.random.Generator(np.random.PCG64DXSM(5)),
                 lambda: np.random.Generator(np.random.Philox(5))):
        assert np.array_equal(skysim.stream_normals(ctx, F, lmax, make()).cpu().numpy(), skysim._host_normals(F, lmax, make()))


@pytest.mark.parametrize("key,cl,seed,nside", [("alm_21cm_F4_l16_seed3", "cla_21cm_F4_l16_zromb1", 3, 8),
                                               ("alm_21cm_F8_l64_seed4", "cla_21cm_F8_l64_zromb3", 4, 32)])
def test_reference_golden_alm_with_the_device_stream(ctx, golden, monkeypatch, key, cl, seed, nside):
    """The reference's own mkfullsky(alms=True, rng=default_rng(seed)) output reproduced with NO normal generated on the
    host, and the generator left where the reference leaves it."""
    from cora_amd.core import skysim

    def boom(*a, **k):
        raise AssertionError("host normal stream used for a PCG64 Generator")

    monkeypatch.setattr(skysim, "_upload_host_normals", boom)
    monkeypatch.setattr(skysim, "_host_normals", boom)
    rng = np.random.default_rng(seed)
    a = skysim.mkfullsky(golden[cl], nside, alms=True, rng=rng)
    ref = golden[key]
    assert a.shape == ref.shape and np.abs(a - ref).max() <= 1e-12 * np.abs(ref).max()
    twin = np.random.default_rng(seed)
    F, L = golden[cl].shape[1], golden[cl].shape[0]
    for l in range(L):
        twin.standard_normal((F, l + 1))
        twin.standard_normal((F, l + 1))
    assert rng.bit_generator.state == twin.bit_generator.state


@pytest.mark.parametrize("F,lmax,nu0,nnu", [(8, 40, 0, 8), (24, 70, 0, 24), (72, 150, 0, 72), (136, 260, 0, 136),
                                            (256, 300, 64, 64), (256, 300, 192, 64), (40, 90, 8, 16), (264, 140, 0, 264),
                                            (256, 300, 112, 16), (256, 300, 240, 16)])
def test_draw_from_stream_buffer_matches_numpy(ctx, F, lmax, nu0, nnu, monkeypatch):
    """K3 with its normals read from a stream-order buffer (the persistent MFMA kernel's FROMG mode; every tile width, a
    frequency shard, lower-triangular and dense factors, row-block factors) against numpy's T_l g_l, and against the
    generic kernel of round 1.  (The 16-channel shards past channel 64: three or more chunks of nu' with the 16-column
    shape, where half of the waves stage nothing - their counted wait let an operand load through until round 5.)"""
    import torch

    rs = np.random.default_rng(F * 1000 + lmax)
    L = lmax + 1
    T = np.tril(rs.standard_normal((L, F, F))) + 3.0 * np.eye(F)
    T[3] = rs.standard_normal((F, F))                          # a dense root (eigen branch)
    info = np.zeros(L, dtype=np.int32)
    info[3] = 1
    g = skysim_host_normals(F, lmax, rs)
    Td, infod, gd = ctx.to_device(T), torch.from_numpy(info).to(ctx.device), ctx.to_device(g)
    alm = ctx.draw_alm(Td, infod, gd, lmax, F, nu0=nu0, nnu=nnu)
    sq = ctx.alm_dev_to_square(alm, lmax, nnu).cpu().numpy()[:, 0]          # [nnu, l, m]
    off = 0
    worst = 0.0
    for l in range(L):
        n = F * (l + 1)
        gl = (g[off:off + n].reshape(F, l + 1) + 1j * g[off + n:off + 2 * n].reshape(F, l + 1)) / np.sqrt(2.0)
        ref = T[l, nu0:nu0 + nnu] @ gl
        worst = max(worst, np.abs(sq[:, l, :l + 1] - ref).max() / np.abs(ref).max())
        off += 2 * n
    assert worst < 1e-13, worst
    rows = ctx.draw_alm_rows(Td[:, nu0:nu0 + nnu, :].contiguous(), infod, gd, lmax, F, nu0, nnu)
    assert torch.equal(rows, alm)
    for _ in range(3):         # (run to run: an operand prefetch that is not covered by its wait shows as a bit difference)
        assert torch.equal(ctx.draw_alm(Td, infod, gd, lmax, F, nu0=nu0, nnu=nnu), alm)
    monkeypatch.setenv("CORAHIP_DRAW_GENERIC", "1")
    old = ctx.draw_alm(Td, infod, gd, lmax, F, nu0=nu0, nnu=nnu)
    assert (old - alm).abs().max().item() <= 1e-12 * alm.abs().max().item()


def skysim_host_normals(F, lmax, rng):
    from cora_amd.core import skysim

    return skysim._host_normals(F, lmax, rng)


# ------------------------------------------------------------------ numpy's LEGACY stream (rng=None) on the device
def _ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def _host_log_is_glibc_fma():
    """True when this host's libm ``log`` is the routine the device restates (glibc >= 2.28, FMA build): the oracle's
    restatement of it equals math.log on a spread of arguments.  Then the device values must be numpy's bit for bit;
    on any other libm the 4-ulp bound of the legacy tests applies."""
    import math

    from oracle import mtlegacy

    xs = [0.9375, 0.99999, 0.5, 0.1234567, 3e-7, 2.0**-104, 0.7071, 0.96, 0.30103]
    try:
        return all(mtlegacy.glibc_log_fma(x) == math.log(x) for x in xs)
    except Exception:
        return False


_LEGACY_ULP = 0 if _host_log_is_glibc_fma() else 4


@pytest.mark.parametrize("n,skip", [(1, 0), (2, 3), (3, 1001), (311, 1), (312, 2), (313, 77), (262143, 0), (524289, 5),
                                    (2_100_000, 11)])
def test_legacy_device_stream_is_numpys(ctx, n, skip):
    """corahip_normals_mt19937_legacy against numpy's RandomState from a state with an arbitrary position and a cached
    value: every value BIT-IDENTICAL on a host whose libm is glibc's FMA build (the device runs glibc's log operation by
    operation; sqrt, the division and the products are correctly rounded on both sides) - within 4 ulp on any other
    libm -, and the generator state afterwards equivalent to numpy's - the next uniforms AND normals agree.
    Sizes around the 156-attempt block and the 262144-attempt segment (jump-ahead polynomial) boundaries."""
    rs = np.random.RandomState(9000 + n % 1000)
    rs.standard_normal(skip)
    st = rs.get_state(legacy=False)
    g, new = ctx.normals_legacy(st, n)
    dev = g.cpu().numpy()
    ref = rs.standard_normal(n)
    d = _ulps(dev, ref)
    assert d.max() <= _LEGACY_ULP, (n, skip, int(d.max()))
    twin = np.random.RandomState(0)
    twin.set_state(new)
    assert np.array_equal(twin.random_sample(1500), rs.random_sample(1500))
    assert _ulps(twin.standard_normal(7), rs.standard_normal(7)).max() == 0


def test_legacy_device_stream_3e7(ctx):
    """3e7 normals = 73 segments (the doubling tree of jump polynomials, seven levels): numpy's values (bit for bit on
    glibc's FMA log, see above), the same accepted attempts, the state equivalent afterwards."""
    rs = np.random.RandomState(31)
    rs.standard_normal(1)                                  # leaves a cached value
    st = rs.get_state(legacy=False)
    n = 30_000_001
    g, new = ctx.normals_legacy(st, n)
    ref = rs.standard_normal(n)
    d = _ulps(g.cpu().numpy(), ref)
    assert d.max() <= _LEGACY_ULP, int(d.max())
    twin = np.random.RandomState(0)
    twin.set_state(new)
    assert twin.get_state(legacy=False)["has_gauss"] == rs.get_state(legacy=False)["has_gauss"]
    assert np.array_equal(twin.random_sample(3000), rs.random_sample(3000))


def test_mkfullsky_without_a_generator_draws_on_the_device(ctx, golden, monkeypatch):
    """rng=None - what Sky3d.getsky() does (cora/core/maps.py:235-237) - reproduces the reference's own legacy-seeded
    a_lm golden with NO normal generated on the host, and np.random is left where the reference leaves it."""
    from cora_amd.core import skysim

    def boom(*a, **k):
        raise AssertionError("host normal stream used for the legacy global state")

    monkeypatch.setattr(skysim, "_upload_host_normals", boom)
    monkeypatch.setattr(skysim, "_host_normals", boom)
    C = golden["cla_21cm_F4_l16_zromb1"]
    np.random.seed(1234)
    a = skysim.mkfullsky(C, 8, alms=True)
    ref = golden["alm_21cm_F4_l16_legacy1234"]
    assert np.abs(a - ref).max() <= 1e-12 * np.abs(ref).max()
    after = np.random.random_sample(100)
    np.random.seed(1234)
    for l in range(C.shape[0]):
        np.random.standard_normal((C.shape[1], l + 1))
        np.random.standard_normal((C.shape[1], l + 1))
    assert np.array_equal(after, np.random.random_sample(100))
    # a RandomState instance is the same legacy generator
    rs, twin = np.random.RandomState(5), np.random.RandomState(5)
    g = skysim.stream_normals(ctx, 6, 20, rs).cpu().numpy()
    monkeypatch.undo()
    assert _ulps(g, skysim._host_normals(6, 20, twin)).max() <= _LEGACY_ULP
    assert np.array_equal(rs.random_sample(50), twin.random_sample(50))


# ------------------------------------------------------------------ the stream in l ranges (corahip_draw_alm_numpy)
def _factors(ctx, F, lmax, seed):
    import torch

    rs = np.random.default_rng(seed)
    L = lmax + 1
    T = np.tril(rs.standard_normal((L, F, F))) + 3.0 * np.eye(F)
    T[min(3, lmax)] = rs.standard_normal((F, F))              # a dense root (eigen branch)
    info = np.zeros(L, dtype=np.int32)
    info[min(3, lmax)] = 1
    return ctx.to_device(T), torch.from_numpy(info).to(ctx.device)


@pytest.mark.parametrize("F,lmax,nu0,nnu,ring_kb", [(8, 40, 0, 8, 1), (24, 70, 0, 24, 64), (72, 150, 0, 72, 700),
                                                    (136, 260, 0, 136, 3000), (256, 300, 64, 64, 5000),
                                                    (256, 300, 192, 64, 1 << 20), (40, 90, 8, 16, 100), (7, 33, 2, 5, 10)])
def test_pcg64_stream_in_l_ranges_equals_the_full_buffer(ctx, F, lmax, nu0, nnu, ring_kb):
    _check_pcg64_ranges(ctx, F, lmax, nu0, nnu, ring_kb)


@pytest.mark.parametrize("F,lmax,nu0,nnu,ring_kb", [(8, 40, 0, 8, 1), (72, 150, 0, 72, 700), (256, 300, 64, 64, 5000)])
def test_pcg64_stream_in_l_ranges_single_pass_form(ctx, monkeypatch, F, lmax, nu0, nnu, ring_kb):
    """The ranged pipeline on the single-pass generator (CORAHIP_ZIG_ONEPASS=1): every range is its own launch that starts
    at the raw position the previous one left on the device (no whole-stream count pass), ranges down to one l."""
    monkeypatch.setenv("CORAHIP_ZIG_ONEPASS", "1")
    _check_pcg64_ranges(ctx, F, lmax, nu0, nnu, ring_kb)


def _check_pcg64_ranges(ctx, F, lmax, nu0, nnu, ring_kb):
    """corahip_draw_alm_numpy (numpy's PCG64 stream emitted one range of multipoles at a time into a two-slot ring, K3
    consuming range by range) against normals_pcg64 + draw_alm on the whole stream: the a_lm bit for bit, the generator
    state after, for ring sizes that put the range edges anywhere inside the generator's 1024-position blocks (1 KB:
    one l per range; 1 GB: one range), full and row-block factors, odd F (generic kernel)."""
    import torch

    from cora_amd import _lib

    Td, infod = _factors(ctx, F, lmax, F * 1000 + lmax)
    rng = np.random.default_rng(4242 + F)
    rng.standard_normal(13)
    st = rng.bit_generator.state["state"]
    n = 2 * F * ((lmax + 1) * (lmax + 2) // 2)
    g, nraw = ctx.normals_pcg64(st["state"], st["inc"], n)
    ref = ctx.draw_alm(Td, infod, g, lmax, F, nu0=nu0, nnu=nnu)
    after_ref = _lib.pcg64_advance(st["state"], st["inc"], nraw)
    for rows in (False, True):
        Tin = Td[:, nu0:nu0 + nnu, :].contiguous() if rows else Td
        alm, after = ctx.draw_alm_numpy(Tin, infod, ("pcg64", st["state"], st["inc"]), lmax, F, nu0=nu0, nnu=nnu, rows=rows,
                                        ring_bytes=ring_kb * 1024)
        assert torch.equal(alm, ref), (rows, (alm - ref).abs().max().item())
        assert after == after_ref
    ref_np = rng.standard_normal(n)                            # (and numpy itself, once)
    assert np.array_equal(g.cpu().numpy().view(np.uint64), ref_np.view(np.uint64))
    assert after_ref == int(rng.bit_generator.state["state"]["state"])


@pytest.mark.parametrize("F,lmax,nu0,nnu,ring_kb,skip", [(8, 40, 0, 8, 1, 0), (24, 70, 0, 24, 64, 1), (72, 150, 0, 72, 700, 3),
                                                         (136, 260, 8, 64, 3000, 1), (256, 300, 192, 64, 1 << 20, 5),
                                                         (7, 33, 2, 5, 10, 1)])
def test_legacy_stream_in_l_ranges_equals_the_full_buffer(ctx, F, lmax, nu0, nnu, ring_kb, skip):
    """The same for numpy's legacy MT19937 + polar-method stream (rng=None): ranges cut anywhere inside the sub-segments
    of the emit pass, with and without a cached value in front (an odd `skip` leaves one: every pair then straddles an
    even element index), the state numpy is left in."""
    import torch

    Td, infod = _factors(ctx, F, lmax, F * 1000 + lmax + 1)
    rs = np.random.RandomState(777 + F)
    rs.standard_normal(skip)
    st = rs.get_state(legacy=False)
    n = 2 * F * ((lmax + 1) * (lmax + 2) // 2)
    g, new_ref = ctx.normals_legacy(st, n)
    ref = ctx.draw_alm(Td, infod, g, lmax, F, nu0=nu0, nnu=nnu)
    d = _ulps(g.cpu().numpy(), rs.standard_normal(n))
    assert d.max() <= _LEGACY_ULP, int(d.max())
    for rows in (False, True):
        Tin = Td[:, nu0:nu0 + nnu, :].contiguous() if rows else Td
        alm, new = ctx.draw_alm_numpy(Tin, infod, ("legacy", st), lmax, F, nu0=nu0, nnu=nnu, rows=rows, ring_bytes=ring_kb * 1024)
        assert torch.equal(alm, ref), (rows, (alm - ref).abs().max().item())
        twin = np.random.RandomState(0)
        twin.set_state(new)
        assert twin.get_state(legacy=False)["has_gauss"] == rs.get_state(legacy=False)["has_gauss"]
        probe = np.random.RandomState(0)
        probe.set_state(rs.get_state(legacy=False))
        assert np.array_equal(twin.random_sample(700), probe.random_sample(700))


def test_ranged_stream_leaves_generators_where_numpy_does(ctx):
    """skysim.draw_numpy_stream: the caller's Generator / numpy's global state after a ranged draw = after the reference's
    loop of complex_std_normal calls."""
    from cora_amd.core import skysim

    F, lmax = 12, 50
    Td, infod = _factors(ctx, F, lmax, 5)
    a, b = np.random.default_rng(99), np.random.default_rng(99)
    skysim.draw_numpy_stream(ctx, Td, infod, a, lmax, F)
    skysim._host_normals(F, lmax, b)
    assert a.bit_generator.state == b.bit_generator.state
    np.random.seed(4321)
    skysim.draw_numpy_stream(ctx, Td, infod, None, lmax, F)
    after = np.random.random_sample(50)
    np.random.seed(4321)
    skysim._host_normals(F, lmax, None)
    assert np.array_equal(after, np.random.random_sample(50))


@pytest.mark.parametrize("F,lmax,chunk,r,N", [(64, 120, 8, 1, 4), (256, 200, 16, 0, 8), (256, 200, 16, 7, 8), (512, 90, 32, 3, 8),
                                              (1024, 40, 64, 2, 8)])
def test_folded_two_chunk_draw_equals_the_columns_of_the_full_draw(ctx, F, lmax, chunk, r, N):
    """K3 on the two chunks of a folded frequency shard (corahip_chanset: chunks r and 2N-1-r, row-block factors in
    local order) = the same channels of the full draw, bit for bit - Philox stream and numpy's stream in l ranges."""
    import torch

    Td, infod = _factors(ctx, F, lmax, 31 * F + lmax)
    chunks = [(r * chunk, chunk), ((2 * N - 1 - r) * chunk, chunk)]
    ch = torch.cat([torch.arange(a, a + n) for a, n in chunks]).to(ctx.device)
    Trows = Td.index_select(1, ch).contiguous()
    full = ctx.alm_dev_to_square(ctx.draw_alm_philox(Td, infod, 99, lmax, F), lmax, F)
    got = ctx.alm_dev_to_square(ctx.draw_alm_philox_chunks(Trows, infod, 99, lmax, F, chunks), lmax, 2 * chunk)
    assert torch.equal(got, full.index_select(0, ch))
    rng = np.random.default_rng(F)
    st = rng.bit_generator.state["state"]
    fulln, after_full = ctx.draw_alm_numpy(Td, infod, ("pcg64", st["state"], st["inc"]), lmax, F)
    gotn, after = ctx.draw_alm_numpy(Trows, infod, ("pcg64", st["state"], st["inc"]), lmax, F, chunks=chunks, ring_bytes=1 << 22)
    assert after == after_full
    assert torch.equal(ctx.alm_dev_to_square(gotn, lmax, 2 * chunk), ctx.alm_dev_to_square(fulln, lmax, F).index_select(0, ch))


# ------------------------------------------------------------------ the generator started ahead of the factors (round 6)
@pytest.mark.parametrize("kind", ["pcg64", "legacy"])
def test_generator_started_ahead_of_the_factors_gives_the_same_draw(ctx, kind):
    """corahip_draw_alm_numpy_prepare / _run: the generator's passes (and the first two ranges of normals) are enqueued
    BEFORE the kernels that make the factors and run beside them - the stream depends on the generator alone
    (cora/util/nputil.py:121-125).  Same a_lm bit for bit and same generator state as the one-call form, for ring sizes
    from one l per range to one range, full and row-block factors; a second session while one is pending is refused;
    a prepared session that is given up leaves the generator untouched."""
    import torch

    from cora_amd import _lib
    from cora_amd.core import skysim

    F, lmax, nu0, nnu = 72, 150, 8, 32
    Td, infod = _factors(ctx, F, lmax, 4711)

    def fresh():
        if kind == "pcg64":
            g = np.random.default_rng(99)
            g.standard_normal(5)
            return g
        rs = np.random.RandomState(98)
        rs.standard_normal(3)                     # (an odd count: a cached value in front)
        return rs

    ref_rng = fresh()
    ref = skysim.draw_numpy_stream(ctx, Td, infod, ref_rng, lmax, F, nu0=nu0, nnu=nnu).clone()
    after_ref = ref_rng.standard_normal(4)
    for rows in (False, True):
        rng = fresh()
        prep = skysim.prepare_numpy_stream(ctx, rng, lmax, F)
        assert prep is not None
        # "the kernels that make the factors": unrelated work on the context's stream while the generator runs
        busy = ctx.to_device(np.ones((2048, 2048)))
        for _ in range(3):
            busy = busy @ busy * 1e-4
        with pytest.raises(_lib.CoraHipError):   # one session per context: a second generator is refused, not interleaved
            ctx.normals_pcg64(1, 3, 10)
        Tin = Td[:, nu0:nu0 + nnu, :].contiguous() if rows else Td
        alm = skysim.draw_numpy_stream(ctx, Tin, infod, rng, lmax, F, nu0=nu0, nnu=nnu, rows=rows, prepared=prep)
        assert torch.equal(alm, ref), (kind, rows)
        assert np.array_equal(rng.standard_normal(4), after_ref), (kind, rows)
    # given up: nothing drawn, generator as it was, the context free for the next session
    rng, twin = fresh(), fresh()
    prep = skysim.prepare_numpy_stream(ctx, rng, lmax, F)
    prep.abort()
    assert np.array_equal(rng.standard_normal(6), twin.standard_normal(6))
    again = skysim.draw_numpy_stream(ctx, Td, infod, fresh(), lmax, F, nu0=nu0, nnu=nnu)
    assert torch.equal(again, ref)
    # a generator that is consumed on the host has nothing to prepare
    assert skysim.prepare_numpy_stream(ctx, np.random.Generator(np.random.MT19937(1)), lmax, F) is None


def test_getsky_and_mkfullsky_start_the_generator_first(ctx, monkeypatch):
    """Sky3d.getsky / skysim.mkfullsky_device order their launches generator first, C_l integration / factorisation second,
    K3 third - and return the maps of the plain order."""
    from cora_amd.core import skysim
    from cora_amd.foreground import galaxy

    order = []
    real_prepare, real_factor, real_clarray = skysim.prepare_numpy_stream, skysim.factor_device, skysim.clarray_device
    monkeypatch.setattr(skysim, "prepare_numpy_stream", lambda *a, **k: (order.append("prepare"), real_prepare(*a, **k))[1])
    monkeypatch.setattr(skysim, "factor_device", lambda *a, **k: (order.append("factor"), real_factor(*a, **k))[1])
    monkeypatch.setattr(skysim, "clarray_device", lambda *a, **k: (order.append("clarray"), real_clarray(*a, **k))[1])
    sky = galaxy.FullSkySynchrotron()
    sky.nside, sky.nu_lower, sky.nu_upper, sky.nu_num = 16, 400.0, 800.0, 8
    sky.lmax = 40
    m1 = sky.getsky(rng=np.random.default_rng(3))
    assert order == ["prepare", "clarray", "factor"], order
    monkeypatch.undo()
    nu = np.asarray(sky._channels(), dtype=np.float64)
    cl = skysim.clarray(sky.angular_powerspectrum, 40, nu.copy(), zromb=sky.oversample)
    m2 = skysim.mkfullsky(cl, 16, rng=np.random.default_rng(3))
    mean = np.asarray(sky.mean_nu(nu), dtype=np.float64) * np.ones(nu.shape)
    assert np.abs(m1 - mean[:, None] - m2).max() <= 1e-9 * np.abs(m2).max()
