"""Counterpart of cora/core/skysim.py: C_l(nu,nu') array + correlated Gaussian sky maps.

Same call surface as the reference (``clarray``, ``mkfullsky``); the work is done by the
HIP kernels behind ``libcorahip.so``:

  clarray   -> K1  fused table-gather + Romberg channel average (21cm table model),
                   outer-product kernel (separable foregrounds), or a device Romberg
                   reduction of host-evaluated samples (any other callable)
  mkfullsky -> K2  batched jittered Cholesky / eigen root per l
               K3  correlated draw a = T g on FP64 MFMA (+ device Philox normals)
               K4  ring-pair Legendre contraction on FP64 MFMA
               K5  per-ring alias fold + FFT, RING-ordered pixels

``mkconstrained`` (cora/core/skysim.py:139-201) is built on the analysis kernels (K5^T, K4^T).
"""
import numpy as np
import scipy.integrate as si

from .. import _lib
from ..util import nputil
from ..util.nputil import DeviceRNG

# l rows evaluated per host call for generic callables (the reference uses chunks of ~5,
# cora/core/skysim.py:51, purely to bound the [5, F zint, F zint] temporaries)
_GENERIC_LCHUNK = 8


def romberg_weights(zromb):
    """Normalised weights (sum = 1) of scipy's Romberg rule on 2**zromb + 1 samples.

    Romberg integration is linear in the samples, so ``si.romb(y, dx)/(2 zhalf)`` of
    cora/core/skysim.py:64-67 is a fixed dot product; the weights are read off scipy itself.
    """
    if zromb == 0:
        return np.ones(1)
    zint = 2**zromb + 1
    return si.romb(np.eye(zint), dx=1.0, axis=0) / 2**zromb


def _plan_of(aps):
    obj = getattr(aps, "__self__", None)
    fn = getattr(obj, "_clarray_plan", None)
    return fn(aps) if fn is not None else None


def clarray_device(aps, lmax, zarray, zromb=3, zwidth=None):
    """As :func:`clarray` but returns the ``[lmax+1, F, F]`` array as a device tensor."""
    zarray = np.asarray(zarray, dtype=np.float64)
    ctx = _lib.get_context()
    zlen = zarray.size
    if zromb == 0:
        zint = 1
        za = zarray.copy()
    else:
        zsort = np.sort(zarray)
        zhalf = np.abs(zsort[1] - zsort[0]) / 2.0 if zwidth is None else zwidth / 2.0
        zint = 2**zromb + 1
        za = (zarray[:, np.newaxis] + np.linspace(-zhalf, zhalf, zint)[np.newaxis, :]).flatten()
        # the reference splits l into lmax // 5 sections and fails for lmax < 5 (skysim.py:51)
        np.array_split(np.arange(lmax + 1), lmax // 5)
    w = ctx.to_device(romberg_weights(zromb))
    larr = np.arange(lmax + 1, dtype=np.float64)

    plan = _plan_of(aps)
    if plan is not None and plan["kind"] == "table21cm":
        p = plan["prepare"](ctx, za)
        lx = np.log10(np.where(larr == 0.0, 1e-10, larr))
        return ctx.clarray_table21cm(p["dd"], p["dv"], p["vv"], p["kperpmin"], p["kperpmax"], p["kparmax"],
                                     ctx.to_device(p["chi"]), ctx.to_device(p["pfd"]), ctx.to_device(p["f"]),
                                     ctx.to_device(p["b"]), zlen, zint, w, ctx.to_device(lx))
    if plan is not None and plan["kind"] == "separable":
        al, bcov = plan["prepare"](larr.copy(), za)
        return ctx.clarray_separable(ctx.to_device(al), ctx.to_device(bcov), zlen, zint, w)

    # generic callable: evaluate on the host in l-chunks, reduce on the device
    import torch

    out = ctx.empty((lmax + 1, zlen, zlen))
    for l0 in range(0, lmax + 1, _GENERIC_LCHUNK):
        lsec = np.arange(l0, min(l0 + _GENERIC_LCHUNK, lmax + 1))
        clt = aps(lsec[:, np.newaxis, np.newaxis].astype(np.float64), za[np.newaxis, :, np.newaxis],
                  za[np.newaxis, np.newaxis, :])
        # (a writable, contiguous copy: broadcast views are read-only and torch refuses to wrap those silently)
        clt = np.array(np.broadcast_to(clt, (len(lsec), za.size, za.size)), dtype=np.float64, order="C", copy=True)
        out[l0 : l0 + len(lsec)] = ctx.romb_reduce(torch.from_numpy(clt).to(ctx.device), len(lsec), zlen, zint, w)
    return out


def clarray(aps, lmax, zarray, zromb=3, zwidth=None):
    """Calculate an array of C_l(z, z') (cora/core/skysim.py:10-69).

    Parameters
    ----------
    aps : function
        The angular power spectrum ``aps(l, z1, z2)``, vectorised.
    lmax : integer
        Maximum l to calculate up to.
    zarray : array_like
        Array of z's (or frequencies) to calculate at.
    zromb : integer
        The Romberg order for integrating over frequency samples.
    zwidth : scalar, optional
        Width of frequency channel to integrate over. If None (default),
        calculate from the separation of the first two bins.

    Returns
    -------
    aps : np.ndarray[lmax+1, len(zarray), len(zarray)]
    """
    return _lib.get_context().to_host(clarray_device(aps, lmax, zarray, zromb=zromb, zwidth=zwidth))


def _host_normals(numz, maxl, rng):
    """The reference's draw order (skysim.py:114-120, nputil.py:104-125): for l ascending,
    F*(l+1) reals then F*(l+1) imaginaries, un-normalised N(0,1)."""
    nalm = (maxl + 1) * (maxl + 2) // 2
    g = np.empty(2 * numz * nalm, dtype=np.float64)
    o = 0
    sn = np.random.standard_normal if rng is None else rng.standard_normal
    for l in range(maxl + 1):
        n = numz * (l + 1)
        g[o : o + n] = sn((numz, l + 1)).ravel()
        g[o + n : o + 2 * n] = sn((numz, l + 1)).ravel()
        o += 2 * n
    return g


def _upload_host_normals(ctx, numz, maxl, rng):
    """The reference's normal stream (see :func:`_host_normals`) as a device array, generated l by l on the host -
    that order IS the seeded-parity contract - into two pinned staging buffers whose uploads (copy stream) overlap
    the generation of the next block: the 8.6 GB of a cfg-3 realisation never exist on the host at once."""
    import torch

    nalm = (maxl + 1) * (maxl + 2) // 2
    total = 2 * numz * nalm
    chunk = 1 << 25                                   # doubles per staging buffer (256 MB)
    if total <= chunk:
        return torch.from_numpy(_host_normals(numz, maxl, rng)).to(ctx.device)
    g = ctx.empty((total,))
    bufs = [torch.empty((chunk,), dtype=torch.float64, pin_memory=True) for _ in range(2)]
    views = [b.numpy() for b in bufs]
    events = [None, None]
    cs = ctx.copy_stream
    # `g` may be the recycled block of the previous realisation's normals, whose draw is still queued on the compute
    # stream: the copies must not start before that stream has reached this point
    cs.wait_stream(torch.cuda.current_stream(ctx.device))
    sn = np.random.standard_normal if rng is None else rng.standard_normal
    cur, fill, dev_off = 0, 0, 0

    def flush():
        nonlocal cur, fill, dev_off
        if fill:
            with torch.cuda.stream(cs):
                g[dev_off:dev_off + fill].copy_(bufs[cur][:fill], non_blocking=True)
                events[cur] = torch.cuda.Event()
                events[cur].record(cs)
            dev_off += fill
        cur, fill = 1 - cur, 0
        if events[cur] is not None:
            events[cur].synchronize()                 # the buffer about to be refilled has left the host

    for l in range(maxl + 1):
        for _part in range(2):                        # real block, then imaginary block (nputil.py:121-125)
            a = sn((numz, l + 1)).ravel()
            o = 0
            while o < a.size:
                m = min(a.size - o, chunk - fill)
                views[cur][fill:fill + m] = a[o:o + m]
                fill += m
                o += m
                if fill == chunk:
                    flush()
    flush()
    done = torch.cuda.Event()
    done.record(cs)
    torch.cuda.current_stream(ctx.device).wait_event(done)
    g.record_stream(cs)
    return g


def _is_pcg64_generator(rng):
    """True for a ``numpy.random.Generator`` on the ``PCG64`` bit generator - what ``default_rng(seed)`` builds and
    what cora's callers pass (cora/signal/lss.py:449-450)."""
    return isinstance(rng, np.random.Generator) and type(rng.bit_generator) is np.random.PCG64


def _legacy_state_of(rng):
    """(get_state, set_state, lock) of a numpy LEGACY generator on MT19937 - ``None`` = the global state behind
    ``np.random.standard_normal`` (cora/util/nputil.py:121-123), or a ``RandomState`` instance - else None."""
    import threading

    if rng is None:
        rs = getattr(np.random.mtrand, "_rand", None)
        if rs is None or np.random.get_state(legacy=False)["bit_generator"] != "MT19937":
            return None
        return np.random.get_state, np.random.set_state, getattr(rs, "_bit_generator", rs).lock if hasattr(
            getattr(rs, "_bit_generator", rs), "lock") else threading.Lock()
    if isinstance(rng, np.random.RandomState) and rng.get_state(legacy=False)["bit_generator"] == "MT19937":
        bg = getattr(rng, "_bit_generator", None)
        return rng.get_state, rng.set_state, bg.lock if bg is not None and hasattr(bg, "lock") else threading.Lock()
    return None


def stream_normals(ctx, numz, maxl, rng):
    """The WHOLE normal stream of one realisation in the reference's draw order (:func:`_host_normals`) as a device
    array of 2 F nalm doubles - for callers that want the numbers themselves (tests, `corahip_draw_alm`); the product
    path does not materialise it any more (:func:`draw_numpy_stream`).

    A ``Generator`` on PCG64 is continued ON THE DEVICE (``corahip_normals_pcg64``: the same PCG64 + ziggurat
    sequence, bit for bit) and left exactly where ``rng.standard_normal`` would have left it: its state is advanced by
    the number of raw draws the normals consumed.  ``rng=None`` - numpy's legacy global MT19937 + polar method, what the
    reference draws from when no generator is passed - and ``RandomState`` instances are continued on the device too
    (``corahip_normals_mt19937_legacy``: the same accepted attempts, generator state and values - glibc's log is restated on the device).  Any
    other generator is consumed on the host and uploaded (:func:`_upload_host_normals`)."""
    n = 2 * numz * ((maxl + 1) * (maxl + 2) // 2)
    legacy = _legacy_state_of(rng)
    if legacy is not None:
        # numpy's legacy stream (MT19937 + polar method): rng=None = the global state, what Sky3d.getsky() draws from
        get_state, set_state, lock = legacy
        with lock:
            g, new = ctx.normals_legacy(get_state(legacy=False), n)
            set_state(new)
        return g
    if not _is_pcg64_generator(rng):
        return _upload_host_normals(ctx, numz, maxl, rng)
    bg = rng.bit_generator
    with bg.lock:                                          # the lock numpy's own draws hold
        st = bg.state
        s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
        g, nraw = ctx.normals_pcg64(s, inc, n)
        st["state"]["state"] = _lib.pcg64_advance(s, inc, nraw)
        bg.state = st                                      # (has_uint32 / uinteger untouched, as standard_normal leaves them)
    return g


class _PreparedStream:
    """A numpy stream whose device generator has been started ahead of the factors (:func:`prepare_numpy_stream`): holds
    the generator's lock until the draw's ``finish()`` - or :meth:`abort` - has run."""

    def __init__(self, rng, legacy, lock, state, handle):
        self.rng, self.legacy, self.lock, self.state, self.handle = rng, legacy, lock, state, handle

    def abort(self):
        """Give the session up (the factors could not be made): the generator is left as it was."""
        try:
            self.handle.abort()
        finally:
            if self.lock is not None:
                self.lock.release()
                self.lock = None


def prepare_numpy_stream(ctx, rng, maxl, numz):
    """Starts the device generator of :func:`draw_numpy_stream` NOW, on the library's generator stream: the passes that
    do not depend on the factors (count + scan of numpy's PCG64 / ziggurat stream, or jump tree + count of the legacy
    MT19937 stream, and the first two ranges of normals) then run BESIDE the kernels the caller enqueues next - the
    C_l integration and the factorisation (the stream is a function of the generator alone: the reference draws it inside
    ``mkfullsky``, cora/util/nputil.py:121-125, but nothing it draws depends on the covariance).  Returns the object
    ``draw_numpy_stream(..., prepared=...)`` takes, or None for a generator that is consumed on the host.  The
    generator's lock is held from here to the end of the draw."""
    legacy = _legacy_state_of(rng)
    pcg = legacy is None and _is_pcg64_generator(rng)
    if legacy is None and not pcg:
        return None
    lock = legacy[2] if legacy is not None else rng.bit_generator.lock
    lock.acquire()
    try:
        if legacy is not None:
            st = None
            spec = ("legacy", legacy[0](legacy=False))
        else:
            st = rng.bit_generator.state
            spec = ("pcg64", int(st["state"]["state"]), int(st["state"]["inc"]))
        handle = ctx.draw_alm_numpy_prepare(spec, maxl, numz)
    except BaseException:
        lock.release()
        raise
    return _PreparedStream(rng, legacy, lock, st, handle)


def draw_numpy_stream(ctx, T, info, rng, maxl, numz, nu0=0, nnu=None, out=None, rows=False, defer=False, chunks=None,
                      prepared=None):
    """K3 with the REFERENCE's normals: ``a_lm = T_l g_l`` for channels ``[nu0, nu0 + nnu)`` where ``g`` is what
    ``complex_std_normal((numz, l + 1), rng)`` returns inside the reference's l loop (cora/core/skysim.py:114-121,
    cora/util/nputil.py:104-125) - drawn the way the reference draws them: range of multipoles by range, never the
    16 F nalm bytes of a whole realisation at once (``corahip_draw_alm_numpy``).

    rng : ``numpy.random.Generator`` on PCG64 (``default_rng(seed)``, cora/signal/lss.py:449-450), ``None`` (numpy's
        legacy global state, what ``Sky3d.getsky()`` draws from, cora/core/maps.py:235-237) or a ``RandomState``:
        continued on the device bit for bit and left where numpy would leave it (for the legacy generators as an
        equivalent (key, pos) pair: the NEXT draws are numpy's, ``get_state()`` itself may differ in representation).
        Any other generator is consumed on the host, l by l, and uploaded.
    T : full factors ``[L, F, F]`` or, with ``rows``, the row block ``[L, nnu, F]`` of a frequency shard
        (``chunks`` = [(first, count), (first, count)]: the two chunks of a FOLDED shard, rows in local order).
    prepared : the object of :func:`prepare_numpy_stream` (the generator already runs; ``rng`` is then ignored).
    defer : return ``(alm, finish)``: the draw is only ENQUEUED; ``finish()`` - to be called after the caller has
        enqueued what follows, e.g. the synthesis - waits for the queue and writes the generator's state back (the
        generator's lock is held until then).  Default: ``alm``, generator already updated.
    Errors: a call the generator's set-up refuses as a STATE error (``CORAHIP_ESTATE``: e.g. a draw still pending on the
    context) is redone on the host stream with the caller's generator untouched; invalid shapes (``CORAHIP_EINVAL``), a
    ring that cannot be allocated (``CORAHIP_ENOMEM``) and HIP errors are raised as they are - the host path would
    materialise the whole 16 F nalm byte stream and hide them.  A status the device generator flags while it runs (a
    margin of > 85 sigma exceeded) is read back at the end: ``finish()`` then raises, the a_lm of that call are not to
    be used and the generator is left untouched."""
    if chunks is not None:
        rows, nu0, nnu = True, chunks[0][0], sum(c[1] for c in chunks)
    nnu = numz if nnu is None else nnu

    def host_path():
        g = _upload_host_normals(ctx, numz, maxl, rng)
        if chunks is not None and len(chunks) > 1:       # (the whole-buffer kernel takes one block: chunk by chunk)
            import torch

            parts, o = [], 0
            for c0, cn in chunks:
                parts.append(ctx.draw_alm_rows(T[:, o:o + cn, :].contiguous(), info, g, maxl, numz, c0, cn))
                o += cn
            alm = torch.cat(parts, dim=1)
            if out is not None:
                out.copy_(alm)
                return out
            return alm
        if rows:
            return ctx.draw_alm_rows(T, info, g, maxl, numz, nu0, nnu, out=out)
        return ctx.draw_alm(T, info, g, maxl, numz, nu0=nu0, nnu=nnu, out=out)

    def done(alm, finish=lambda: None):
        if defer:
            return alm, finish
        finish()
        return alm

    if prepared is not None:
        # the generator was started ahead of the factors (prepare_numpy_stream): its lock is held, K3 runs against T now
        legacy, lock, st = prepared.legacy, prepared.lock, prepared.state
        if legacy is not None:
            get_state, set_state, _ = legacy
        else:
            bg = prepared.rng.bit_generator
        prepared.lock = None                               # (released by finish() below, or here on failure)
        try:
            alm, fin = ctx.draw_alm_numpy(T, info, None, maxl, numz, nu0=nu0, nnu=nnu, out=out, rows=rows, defer=True,
                                          chunks=chunks, prepared=prepared.handle)
        except BaseException:
            lock.release()
            raise
    else:
        legacy = _legacy_state_of(rng)
        pcg = legacy is None and _is_pcg64_generator(rng)
        if legacy is None and not pcg:
            return done(host_path())
        if legacy is not None:
            get_state, set_state, lock = legacy
        else:
            bg = rng.bit_generator
            lock = bg.lock                                     # the lock numpy's own draws hold
        lock.acquire()
        try:
            if legacy is not None:
                spec = ("legacy", get_state(legacy=False))
            else:
                st = bg.state
                spec = ("pcg64", int(st["state"]["state"]), int(st["state"]["inc"]))
            alm, fin = ctx.draw_alm_numpy(T, info, spec, maxl, numz, nu0=nu0, nnu=nnu, out=out, rows=rows, defer=True, chunks=chunks)
        except _lib.CoraHipError as e:
            lock.release()
            if e.status != -3:                                 # only CORAHIP_ESTATE is a reason to take the host stream
                raise
            return done(host_path())
        except BaseException:
            lock.release()
            raise

    def finish():
        try:
            after = fin()
            if legacy is not None:
                set_state(after)
            else:
                st["state"]["state"] = after
                bg.state = st                              # (has_uint32 / uinteger untouched, as standard_normal leaves them)
        finally:
            lock.release()

    return done(alm, finish)


def factor_device(corr):
    """Per-l roots of the jittered covariance blocks (skysim.py:115-119) on the device."""
    ctx = _lib.get_context()
    import torch

    if not isinstance(corr, torch.Tensor):
        corr = ctx.to_device(corr)
    return ctx.factor_batched(corr, jitter_rel=1e-14, eig_thresh=1e-16)


def mkfullsky_device(corr, nside, alms=False, rng=None, factors=None, nu_range=None, prepared=None):
    """Device-resident :func:`mkfullsky`: returns torch tensors and accepts cached factors.

    corr : ndarray or device tensor [lmax+1, F, F] (ignored when ``factors`` is given)
    factors : optional (T, info) from :func:`factor_device`
    nu_range : optional (nu0, nnu): only these channels are synthesised (frequency shard);
        the normals are always the full global stream so shards are consistent.
    prepared : optional object of :func:`prepare_numpy_stream` for ``rng`` (started by the caller ahead of its C_l
        integration, as ``Sky3d.getsky`` does); without it the generator is started here, ahead of the factorisation.
    """
    import torch

    ctx = _lib.get_context()
    prep = prepared
    if factors is None:
        numz = corr.shape[1]
        if corr.shape[2] != numz:
            raise Exception("Correlation matrix is incorrect shape.")
        if prep is None and not isinstance(rng, DeviceRNG):
            # numpy's stream does not depend on the covariance: its generator passes start here, beside the factorisation
            prep = prepare_numpy_stream(ctx, rng, corr.shape[0] - 1, numz)
        try:
            T, info = factor_device(corr)
        except BaseException:
            if prep is not None:
                prep.abort()
            raise
    else:
        T, info = factors
        numz = T.shape[1]
    maxl = T.shape[0] - 1
    nu0, nnu = (0, numz) if nu_range is None else nu_range

    if isinstance(rng, DeviceRNG):
        alm = ctx.draw_alm_philox(T, info, rng.next_seed(), maxl, numz, nu0=nu0, nnu=nnu)
    else:
        # the draw is enqueued, the synthesis behind it; only then is the generator's state waited for
        alm, finish = draw_numpy_stream(ctx, T, info, rng, maxl, numz, nu0=nu0, nnu=nnu, defer=True, prepared=prep)
        try:
            out = ctx.alm_dev_to_square(alm, maxl, nnu) if alms else ctx.alm2map(alm, int(nside), maxl, nnu)
        finally:
            finish()
        return out
    if alms:
        return ctx.alm_dev_to_square(alm, maxl, nnu)
    return ctx.alm2map(alm, int(nside), maxl, nnu)


def mkfullsky(corr, nside, alms=False, rng=None):
    """Construct a set of correlated Healpix maps (cora/core/skysim.py:72-136).

    Parameters
    ----------
    corr : np.ndarray (lmax+1, numz, numz)
        The correlation matrix :math:`C_l(z, z')`.
    nside : integer
        The resolution of the Healpix maps.
    alms : boolean, optional
        If True return the alms ``[numz, 1, lmax+1, lmax+1]`` instead of the sky maps.
    rng : numpy Generator, :class:`cora_amd.DeviceRNG`, optional
        Seeded generator.  A numpy ``Generator`` on PCG64 (``default_rng(seed)``) is continued
        on the GPU - the same values cora draws, in the reference's order, and the generator
        is left in the state cora would leave it in; ``None`` (numpy's legacy global MT19937
        state, what the reference draws from here) and ``RandomState`` instances are continued
        on the GPU as well (an equivalent (key, pos) state is written back: the next draws are
        numpy's); other bit generators are consumed on the host in that order; a ``DeviceRNG``
        is the library's own counter-based stream (not numpy's numbers).  The normals are
        generated range of multipoles by range, as the reference's loop consumes them.

    Returns
    -------
    hpmaps : np.ndarray (numz, npix)
    """
    local = getattr(corr, "local_array", None)
    if local is not None:  # caput MPIArray (skysim.py:97-103)
        gshape = tuple(getattr(corr, "global_shape", local.shape))
        if gshape != tuple(local.shape):
            # l-distributed input: one process per GPU with torch.distributed initialised (the ranks of the MPI
            # job); returns the rank's frequency shard, wrapped like the reference's return value (:132-134)
            import torch.distributed as dist

            if not (dist.is_available() and dist.is_initialized()):
                raise NotImplementedError(
                    "l-distributed MPIArray input needs an initialised torch.distributed process group "
                    "(one process per GPU): see cora_amd.parallel.mkfullsky_sharded")
            from .. import parallel

            out, _ = parallel.mkfullsky_sharded(np.asarray(local), gshape, nside, rng=rng, alms=alms)
            if alms:
                # the reference returns ``alm_array.allgather()`` here (skysim.py:123-125): the FULL [numz, 1, L, L]
                # array on every rank, as a plain ndarray - not a wrapped frequency shard
                return _lib.get_context().to_host(parallel.allgather_channels(out, gshape[1]))
            out = _lib.get_context().to_host(out)
            wrap = getattr(type(corr), "wrap", None)
            return wrap(out, axis=0) if wrap is not None else out
        corr = np.asarray(local)
    corr = np.asarray(corr, dtype=np.float64)
    if corr.shape[2] != corr.shape[1]:
        raise Exception("Correlation matrix is incorrect shape.")
    out = mkfullsky_device(corr, nside, alms=alms, rng=rng)
    return _lib.get_context().to_host(out)


def mkfullsky_stream(corr, nside, rngs, alms=False, factors=None):
    """Generator over realisations of :func:`mkfullsky` delivered to the HOST as numpy arrays, one per entry of
    ``rngs`` (numpy Generators, ``DeviceRNG`` instances or ``None``).  The factors are made once (what repeated seeds
    amortise in the reference's pipelines, cora/signal/lss.py:424-478); the D2H copy of realisation i goes through
    pinned memory on the copy stream while realisation i + 1 is drawn and synthesised, so a consumer that keeps up
    sees the PCIe rate (25.8 GB per cfg-3 realisation), not copy + compute."""
    ctx = _lib.get_context()
    if factors is None:
        corr = np.asarray(corr, dtype=np.float64) if not hasattr(corr, "device") else corr
        if corr.shape[2] != corr.shape[1]:
            raise Exception("Correlation matrix is incorrect shape.")
        factors = factor_device(corr)
    pending = None
    for rng in rngs:
        out = mkfullsky_device(None, nside, alms=alms, rng=rng, factors=factors)
        nxt = ctx.to_host_async(out)                 # queued behind the kernels that make `out`
        del out
        if pending is not None:
            pending[1].synchronize()
            yield pending[0].numpy()
        pending = nxt
    if pending is not None:
        pending[1].synchronize()
        yield pending[0].numpy()


def mkconstrained(corr, constraints, nside):
    """Maps satisfying given constraints on some frequency slices, built from the lowest eigenmodes
    (cora/core/skysim.py:139-205).

    corr [lmax+1, numz, numz]; constraints = [[frequency_index, healpix map], ...]; returns [numz, npix].
    The per-l symmetric eigenproblems and the nmodes x nmodes solves stay on the host (scipy, as the
    reference); the two transforms the reference takes from healpy - ``map2alm(cons, lmax=maxl)`` with
    healpy's defaults (iter=3, no ring weights) and ``alm2map`` of every channel - run on the GPU in
    one batch each."""
    import scipy.linalg as la
    import torch

    from ..util import hputil

    numz = corr.shape[1]
    maxl = corr.shape[0] - 1
    L = maxl + 1
    nmodes = len(constraints)
    f_ind = [c[0] for c in constraints]
    if corr.shape[2] != numz:
        raise Exception("Correlation matrix is incorrect shape.")

    trans = np.zeros((corr.shape[0], nmodes, corr.shape[2]))
    tmat = np.zeros((corr.shape[0], nmodes, nmodes))
    for i in range(L):
        trans[i] = la.eigh(corr[i])[1][:, -nmodes:].T
        tmat[i] = trans[i][:, f_ind]

    ctx = _lib.get_context()
    cons = np.ascontiguousarray(np.stack([np.asarray(c[1], dtype=np.float64) for c in constraints]))
    ns_c = int(round(np.sqrt(cons.shape[1] / 12.0)))
    calm = hputil.map2alm_device(torch.from_numpy(cons).to(ctx.device), ns_c, maxl, use_weights=False, niter=3)
    cmap = ctx.alm_dev_to_square(calm, maxl, nmodes).cpu().numpy()[:, 0]      # [nmodes, l, m]

    cv = np.zeros((numz, L, L), dtype=np.complex128)
    for l in range(1, L):                                                     # l = 0 stays zero (:193-194)
        amp = la.solve(tmat[l].T, cmap[:, l, : l + 1])                        # [nmodes, m]
        cv[:, l, : l + 1] = np.dot(trans[l].T, amp)
    return hputil._synth(cv, nside)
