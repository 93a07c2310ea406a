"""ctypes binding of libcorahip.so (include/corahip.h) + a thin torch-tensor front end.

PyTorch is used for device memory, streams and (elsewhere) torch.distributed only;
all arithmetic of the hot path runs in the hand-written HIP kernels of the library.
There is NO CPU fallback: if the library or a gfx950 GPU is missing, the compute
entry points raise (loudly) instead of computing something else.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CORAHIP_LIB", os.path.join(_HERE, "libcorahip.so"))  # override: diagnostics only

c_int, c_double, c_void_p, c_size_t = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
c_u64, c_char_p = ctypes.c_uint64, ctypes.c_char_p
PTR = c_void_p

# name -> (restype, argtypes): every symbol include/corahip.h declares
SIGNATURES = {
    "corahip_abi_version": (c_int, []),
    "corahip_abi_minor": (c_int, []),
    "corahip_last_error": (c_char_p, []),
    "corahip_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "corahip_ctx_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "corahip_ctx_destroy": (c_int, [c_void_p]),
    "corahip_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "corahip_ctx_sync": (c_int, [c_void_p]),
    "corahip_timer_begin": (c_int, [c_void_p]),
    "corahip_timer_end": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "corahip_profile_enable": (c_int, [c_void_p, c_int]),
    "corahip_profile_get": (c_int, [c_void_p, c_char_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int)]),
    "corahip_profile_reset": (c_int, [c_void_p]),
    "corahip_malloc": (c_int, [c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    "corahip_free": (c_int, [c_void_p, c_void_p]),
    "corahip_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "corahip_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "corahip_clarray_table21cm": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_double, c_double, c_double,
                                          PTR, PTR, PTR, PTR, c_int, c_int, PTR, PTR, c_int, PTR]),
    "corahip_clarray_table21cm_pairs": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_double, c_double, c_double,
                                                PTR, PTR, PTR, PTR, c_int, c_int, PTR, PTR, c_int, c_int, c_int, c_int,
                                                PTR]),
    "corahip_clarray_pairs_finish": (c_int, [c_void_p, PTR, c_int, c_int, c_int, c_int, PTR]),
    "corahip_clarray_tables_pin": (c_int, [c_void_p, PTR, PTR, PTR, c_u64]),
    "corahip_aps_table21cm_points": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_double, c_double, c_double,
                                             ctypes.c_long, PTR, PTR, PTR, PTR, PTR, PTR, PTR]),
    "corahip_clarray_separable": (c_int, [c_void_p, PTR, c_int, PTR, c_int, c_int, PTR, PTR]),
    "corahip_romb_reduce": (c_int, [c_void_p, PTR, c_int, c_int, c_int, PTR, PTR]),
    "corahip_factor_batched": (c_int, [c_void_p, PTR, c_int, c_int, c_double, c_double, PTR, PTR]),
    "corahip_normals_philox": (c_int, [c_void_p, c_u64, c_int, c_int, PTR]),
    "corahip_normals_pcg64": (c_int, [c_void_p, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.c_int64, PTR,
                                      ctypes.POINTER(c_u64)]),
    "corahip_normals_mt19937_legacy": (c_int, [c_void_p, c_void_p, ctypes.c_int64, PTR]),
    "corahip_glibc_exp": (c_int, [c_void_p, PTR, ctypes.c_int64, PTR]),
    "corahip_pcg64_advance": (c_int, [ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), c_u64, ctypes.POINTER(c_u64)]),
    "corahip_draw_alm_philox": (c_int, [c_void_p, PTR, PTR, c_u64, c_int, c_int, c_int, c_int, PTR]),
    "corahip_draw_alm_philox_rows": (c_int, [c_void_p, PTR, PTR, c_u64, c_int, c_int, c_int, c_int, PTR]),
    "corahip_draw_alm": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_int, c_int, PTR]),
    "corahip_draw_alm_rows": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_int, c_int, PTR]),
    "corahip_draw_alm_numpy_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_size_t, ctypes.POINTER(c_void_p)]),
    "corahip_draw_alm_numpy_run": (c_int, [c_void_p, c_void_p, PTR, c_int, PTR, c_void_p, PTR]),
    "corahip_draw_alm_numpy": (c_int, [c_void_p, PTR, c_int, PTR, c_void_p, c_int, c_int, c_int, c_int, PTR, c_size_t]),
    "corahip_draw_alm_numpy_begin": (c_int, [c_void_p, PTR, c_int, PTR, c_void_p, c_int, c_int, c_int, c_int, PTR, c_size_t,
                                             ctypes.POINTER(c_void_p)]),
    "corahip_draw_alm_numpy_end": (c_int, [c_void_p, c_void_p, c_void_p]),
    "corahip_draw_alm_numpy_begin_set": (c_int, [c_void_p, PTR, PTR, c_void_p, c_int, c_int, c_void_p, PTR, c_size_t,
                                                 ctypes.POINTER(c_void_p)]),
    "corahip_draw_alm_philox_rows_set": (c_int, [c_void_p, PTR, PTR, c_u64, c_int, c_int, c_void_p, PTR]),
    "corahip_mkfullsky_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "corahip_mkfullsky": (c_int, [c_void_p, c_void_p, PTR, c_int, c_void_p, c_int, c_int, c_int, PTR, c_void_p, c_size_t]),
    "corahip_shard_plan": (c_int, [c_int, c_int, c_int, c_int, PTR]),
    "corahip_factor_rows_pack": (c_int, [c_void_p, PTR, c_int, c_int, c_int, c_int, PTR]),
    "corahip_factor_rows_unpack": (c_int, [c_void_p, PTR, PTR, c_int, c_int, c_int, c_int, PTR]),
    "corahip_alm_dev_to_square": (c_int, [c_void_p, PTR, c_int, c_int, PTR]),
    "corahip_alm_packed_to_dev": (c_int, [c_void_p, PTR, c_int, c_int, PTR]),
    "corahip_sht_plan_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "corahip_sht_plan_create_ex": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "corahip_sht_plan_cut_exp": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "corahip_sht_plan_k4_mfma_count": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_u64)]),
    "corahip_sht_plan_destroy": (c_int, [c_void_p, c_void_p]),
    "corahip_alm2map_workspace_bytes": (c_int, [c_void_p, c_int, ctypes.POINTER(c_size_t)]),
    "corahip_alm2map": (c_int, [c_void_p, c_void_p, PTR, c_int, PTR, c_void_p, c_size_t]),
    "corahip_map2alm_workspace_bytes": (c_int, [c_void_p, c_int, ctypes.POINTER(c_size_t)]),
    "corahip_map2alm": (c_int, [c_void_p, c_void_p, PTR, c_int, PTR, PTR, c_void_p, c_size_t]),
    "corahip_alm2map_spin2": (c_int, [c_void_p, c_void_p, PTR, c_int, PTR, c_void_p, c_size_t]),
    "corahip_spin2_ring_scale": (c_int, [c_void_p, c_void_p, PTR, c_int, PTR]),
    "corahip_spin2_combine": (c_int, [c_void_p, c_void_p, PTR, c_int, c_int, PTR, c_int]),
    "corahip_xi_table_average": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_double, c_double, PTR, c_int, PTR, PTR,
                                         c_int, c_int, PTR]),
    "corahip_ps_table21cm": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, c_double, PTR, c_int, PTR, c_int, c_double,
                                     PTR, PTR, PTR, PTR]),
    "corahip_dct1_workspace_bytes": (c_int, [ctypes.c_long, c_int, ctypes.POINTER(c_size_t)]),
    "corahip_dct1_rows": (c_int, [c_void_p, PTR, ctypes.c_long, c_int, c_double, c_void_p, c_size_t]),
    "corahip_legendre_project": (c_int, [c_void_p, PTR, PTR, c_int, c_int, PTR, ctypes.c_long, PTR]),
    "corahip_fft_c2c": (c_int, [c_void_p, PTR, c_int, PTR, c_int, c_int]),
    "corahip_irfftn": (c_int, [c_void_p, PTR, c_int, PTR, c_int, PTR]),
    "corahip_rfftn": (c_int, [c_void_p, PTR, c_int, PTR, c_int, PTR]),
    "corahip_randomfield_draw": (c_int, [c_void_p, PTR, ctypes.c_int64, ctypes.c_uint64, PTR]),
    "corahip_fg_mix": (c_int, [c_void_p, PTR, PTR, PTR, c_int, c_int, ctypes.c_int64, PTR]),
    "corahip_randomfield_irfftn": (c_int, [c_void_p, PTR, c_int, PTR, ctypes.c_uint64, PTR, PTR]),
    "corahip_spec_mul_real": (c_int, [c_void_p, PTR, PTR, ctypes.c_int64]),
    "corahip_cube_affine": (c_int, [c_void_p, PTR, PTR, PTR, PTR, PTR, c_int, ctypes.c_int64, PTR]),
    "corahip_raytrace_slices": (c_int, [c_void_p, PTR, c_int, c_int, c_int, PTR, PTR, PTR, PTR, c_double, c_double,
                                        c_int, c_int, c_int, PTR]),
    "corahip_sht_plan_rings": (c_int, [c_void_p, PTR, PTR, PTR, PTR]),
    "corahip_sht_plan_ring_classes": (c_int, [c_void_p, PTR]),
    "corahip_sht_lambda": (c_int, [c_void_p, c_void_p, c_int, c_int, PTR]),
    "corahip_sht_lambda_entry": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, PTR]),
}


class CoraHipError(RuntimeError):
    """An entry point of libcorahip.so returned a non-zero status (``status``: <0 a CORAHIP_E* code of include/corahip.h,
    >0 a hipError_t)."""

    status = 0


_lib = None


def load():
    """Load libcorahip.so and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cora_amd: %s not found. Build it with `make -C cora_amd/csrc` (needs hipcc, gfx950). "
                "There is no CPU fallback for the compute path." % LIB_PATH
            )
        # torch bundles its own libamdhip64.so.7; import it FIRST so that libcorahip.so binds to the
        # same HIP runtime instance (streams and device pointers are shared with torch).  Loading
        # the library first would pull in /opt/rocm's copy and leave two runtimes in the process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.corahip_abi_version() != 1:
            raise ImportError("cora_amd: ABI version mismatch in %s" % LIB_PATH)
        _lib = lib
    return _lib


def pcg64_advance(state, inc, delta):
    """PCG64 state (python int) after ``delta`` steps - numpy's ``bit_generator.advance`` as host arithmetic of the
    library (no GPU needed)."""
    M = 2**64 - 1
    st = (c_u64 * 2)((int(state) >> 64) & M, int(state) & M)
    ic = (c_u64 * 2)((int(inc) >> 64) & M, int(inc) & M)
    out = (c_u64 * 2)()
    _check(load().corahip_pcg64_advance(st, ic, c_u64(int(delta)), out))
    return (int(out[0]) << 64) | int(out[1])


def _check(rc):
    if rc != 0:
        msg = load().corahip_last_error()
        err = CoraHipError("libcorahip status %d: %s" % (rc, msg.decode() if msg else "?"))
        err.status = int(rc)
        raise err


def _torch():
    import torch

    return torch


class _MtState(ctypes.Structure):          # corahip_mt_state
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32), ("gauss", c_double)]


class _Rng(ctypes.Structure):              # corahip_rng
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stream", c_void_p), ("seed", c_u64),
                ("state", c_u64 * 2), ("inc", c_u64 * 2), ("legacy", ctypes.POINTER(_MtState))]


class _ChanSet(ctypes.Structure):          # corahip_chanset
    _fields_ = [("nchunks", ctypes.c_int32), ("chunk_nnu", ctypes.c_int32), ("nu0", ctypes.c_int32 * 2)]


def _chanset(chunks):
    """[(first channel, count)] (one block, or the two equal chunks of a folded shard) -> corahip_chanset."""
    cs = _ChanSet()
    chunks = [(int(a), int(n)) for a, n in chunks]
    if len(chunks) not in (1, 2) or (len(chunks) == 2 and chunks[0][1] != chunks[1][1]):
        raise ValueError("a channel set is one block or two equal chunks")
    cs.nchunks, cs.chunk_nnu = len(chunks), chunks[0][1]
    cs.nu0[0] = chunks[0][0]
    cs.nu0[1] = chunks[1][0] if len(chunks) == 2 else 0
    return cs


def _rng_struct(rng):
    """("pcg64", state, inc) | ("legacy", state dict) -> (corahip_rng, corahip_mt_state or None)."""
    r = _Rng()
    M = 2**64 - 1
    if rng[0] == "pcg64":
        r.kind = 2
        r.state[0], r.state[1] = (int(rng[1]) >> 64) & M, int(rng[1]) & M
        r.inc[0], r.inc[1] = (int(rng[2]) >> 64) & M, int(rng[2]) & M
        return r, None
    if rng[0] != "legacy" or rng[1]["bit_generator"] != "MT19937":
        raise ValueError("numpy stream on the device: ('pcg64', state, inc) or ('legacy', MT19937 state dict)")
    ms = _MtState()
    key = np.ascontiguousarray(rng[1]["state"]["key"], dtype=np.uint32)
    ctypes.memmove(ms.key, key.ctypes.data, 624 * 4)
    ms.pos, ms.has_gauss, ms.gauss = int(rng[1]["state"]["pos"]), int(rng[1]["has_gauss"]), float(rng[1]["gauss"])
    r.kind = 3
    r.legacy = ctypes.pointer(ms)
    return r, ms


class Context:
    """One context per GPU; wraps the C ABI with torch tensors as device arrays."""

    def __init__(self, device=0):
        torch = _torch()
        self.lib = load()
        if not torch.cuda.is_available():
            raise CoraHipError(
                "cora_amd needs an AMD MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                "and there is no CPU fallback"
            )
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = c_void_p()
        _check(self.lib.corahip_ctx_create(device, ctypes.byref(h)))
        self.h = h
        self._plans = {}
        self._workspace = None
        self.use_current_stream()

    # -- plumbing ---------------------------------------------------------------------
    def use_current_stream(self):
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        _check(self.lib.corahip_ctx_set_stream(self.h, c_void_p(s)))

    def sync(self):
        _check(self.lib.corahip_ctx_sync(self.h))

    def timer_begin(self):
        _check(self.lib.corahip_timer_begin(self.h))

    def timer_end(self):
        ms = ctypes.c_float()
        _check(self.lib.corahip_timer_end(self.h, ctypes.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on=True):
        _check(self.lib.corahip_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        _check(self.lib.corahip_profile_reset(self.h))

    def profile_get(self, name):
        ms, n = c_double(), c_int()
        _check(self.lib.corahip_profile_get(self.h, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def empty(self, shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, dtype=dtype or torch.float64, device=self.device)

    def to_device(self, a, dtype=np.float64):
        torch = _torch()
        a = np.ascontiguousarray(a, dtype=dtype)
        return torch.from_numpy(a).to(self.device)

    @staticmethod
    def _p(t):
        assert t.is_contiguous(), "device array must be contiguous"
        return c_void_p(t.data_ptr())

    def _f64(self, t):
        torch = _torch()
        assert t.dtype == torch.float64 and t.device == self.device
        return self._p(t)

    # -- host delivery ------------------------------------------------------------------
    _PINNED_MIN_BYTES = 1 << 24     # below this a pageable copy is as fast as pinning a buffer

    @property
    def copy_stream(self):
        """A second HIP stream for host <-> device copies that overlap the kernels of the library's stream."""
        torch = _torch()
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        return self._copy_stream

    def to_host_async(self, t):
        """Start the D2H copy of device tensor ``t`` into PINNED host memory on the copy stream (after everything
        queued so far on the current stream); returns ``(host_tensor, event)``.  The pinned block comes from torch's
        caching host allocator: the first buffer of a size is page-locked once (seconds for tens of GB), later ones of
        that size are recycled as soon as the caller drops the array."""
        torch = _torch()
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        cs = self.copy_stream
        cs.wait_event(ready)
        with torch.cuda.stream(cs):
            host.copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(cs)
        t.record_stream(cs)
        return host, done

    def to_host(self, t):
        """Device tensor -> numpy array.  Large arrays arrive through pinned memory at PCIe rate (the ndarray is a view
        of the pinned block and keeps it alive); small ones by an ordinary pageable copy."""
        if t.numel() * t.element_size() < self._PINNED_MIN_BYTES:
            return t.cpu().numpy()
        host, done = self.to_host_async(t)
        done.synchronize()
        return host.numpy()

    # -- K1 ---------------------------------------------------------------------------
    def pin_tables(self, dd, dv, vv, generation):
        """The 21cm tables (dd, dv, vv) stay as they are under this generation number: K1 keeps its transposed copy
        of them between calls (corahip_clarray_tables_pin).  The model that owns the tables calls this."""
        _check(self.lib.corahip_clarray_tables_pin(self.h, self._f64(dd), self._f64(dv), self._f64(vv), c_u64(int(generation))))

    def unpin_tables(self, generation=0):
        """Withdraws the pin of ``generation`` (0: whatever is pinned): the owner calls this before its tables are
        freed or replaced - a recycled device address must never be taken for the pinned tables."""
        if self.h:
            _check(self.lib.corahip_clarray_tables_pin(self.h, None, None, None, c_u64(int(generation))))

    def clarray_table21cm(self, dd, dv, vv, kperpmin, kperpmax, kparmax, chi, pfd, f, b, F, zint, w, log10l):
        nl = log10l.numel()
        out = self.empty((nl, F, F))
        nkperp, nkpar = dd.shape
        _check(self.lib.corahip_clarray_table21cm(
            self.h, self._f64(dd), self._f64(dv), self._f64(vv), nkperp, nkpar, kperpmin, kperpmax, kparmax,
            self._f64(chi), self._f64(pfd), self._f64(f), self._f64(b), F, zint, self._f64(w), self._f64(log10l),
            nl, self._f64(out)))
        return out

    def clarray_table21cm_pairs(self, dd, dv, vv, kperpmin, kperpmax, kparmax, chi, pfd, f, b, F, zint, w, log10l,
                                pair_first, pair_step, l_block, nblocks=None):
        """Pair shard of the integration: [nblocks >= ceil(nl / l_block), npl, l_block] slabs (see
        include/corahip.h); nblocks = number of ranks when the slabs feed an all-to-all."""
        torch = _torch()
        nl = log10l.numel()
        npl = (F * (F + 1) // 2 + pair_step - 1) // pair_step
        nblk = (nl + l_block - 1) // l_block
        if nblocks is not None:
            assert nblocks >= nblk
            nblk = nblocks
        out = torch.zeros((nblk, npl, l_block), dtype=torch.float64, device=self.device)
        nkperp, nkpar = dd.shape
        _check(self.lib.corahip_clarray_table21cm_pairs(
            self.h, self._f64(dd), self._f64(dv), self._f64(vv), nkperp, nkpar, kperpmin, kperpmax, kparmax,
            self._f64(chi), self._f64(pfd), self._f64(f), self._f64(b), F, zint, self._f64(w), self._f64(log10l),
            nl, pair_first, pair_step, l_block, self._f64(out)))
        return out

    def clarray_pairs_finish(self, slabs, F, nl):
        """[nranks, npl, l_stride] pair slabs (slab r integrated by rank r) -> C [nl, F, F]."""
        nranks, npl, l_stride = slabs.shape
        assert npl == (F * (F + 1) // 2 + nranks - 1) // nranks and nl <= l_stride
        out = self.empty((nl, F, F))
        _check(self.lib.corahip_clarray_pairs_finish(self.h, self._f64(slabs), F, nranks, l_stride, nl, self._f64(out)))
        return out

    def aps_table21cm_points(self, dd, dv, vv, kperpmin, kperpmax, kparmax, lx, chi1, chi2, cdd, cdv, cvv):
        n = lx.numel()
        out = self.empty((n,))
        nkperp, nkpar = dd.shape
        _check(self.lib.corahip_aps_table21cm_points(
            self.h, self._f64(dd), self._f64(dv), self._f64(vv), nkperp, nkpar, kperpmin, kperpmax, kparmax, n,
            self._f64(lx), self._f64(chi1), self._f64(chi2), self._f64(cdd), self._f64(cdv), self._f64(cvv),
            self._f64(out)))
        return out

    def clarray_separable(self, al, bcov, F, zint, w):
        nl = al.numel()
        out = self.empty((nl, F, F))
        _check(self.lib.corahip_clarray_separable(self.h, self._f64(al), nl, self._f64(bcov), F, zint,
                                                  self._f64(w), self._f64(out)))
        return out

    def romb_reduce(self, clt, nl, F, zint, w):
        out = self.empty((nl, F, F))
        _check(self.lib.corahip_romb_reduce(self.h, self._f64(clt), nl, F, zint, self._f64(w), self._f64(out)))
        return out

    # -- K0 ---------------------------------------------------------------------------
    def ps_table21cm(self, kperp, kpar, spline=None, kstar=0.0, freq_window=0.0, dd=None):
        """(dd, dv, vv) device tables [nkperp, nkpar] BEFORE the DCT: from a spline description
        ``spline = (loglog, x, y, y2)`` (device arrays) or from a host-evaluated ``dd`` (device array)."""
        nkperp, nkpar = kperp.numel(), kpar.numel()
        dv, vv = self.empty((nkperp, nkpar)), self.empty((nkperp, nkpar))
        if dd is None:
            loglog, kx, ky, ky2 = spline
            dd = self.empty((nkperp, nkpar))
            _check(self.lib.corahip_ps_table21cm(self.h, self._f64(kx), self._f64(ky), self._f64(ky2), kx.numel(),
                                                 1 if loglog else 0, float(kstar), self._f64(kperp), nkperp,
                                                 self._f64(kpar), nkpar, float(freq_window), None, self._f64(dd),
                                                 self._f64(dv), self._f64(vv)))
        else:
            _check(self.lib.corahip_ps_table21cm(self.h, None, None, None, 0, 0, 0.0, self._f64(kperp), nkperp,
                                                 self._f64(kpar), nkpar, 0.0, self._f64(dd), None, self._f64(dv),
                                                 self._f64(vv)))
        return dd, dv, vv

    def dct1_rows(self, data, scale=1.0):
        """In place: every row of the device array ``data`` [nrows, n] -> scipy.fftpack.dct(row, type=1) * scale."""
        nrows, n = data.shape
        b = c_size_t()
        _check(self.lib.corahip_dct1_workspace_bytes(nrows, n, ctypes.byref(b)))
        ws = self.workspace(int(b.value))
        _check(self.lib.corahip_dct1_rows(self.h, self._f64(data), nrows, n, float(scale), self._p(ws), int(b.value)))
        return data

    # -- K2 ---------------------------------------------------------------------------
    def factor_batched(self, C, jitter_rel=1e-14, eig_thresh=1e-16):
        torch = _torch()
        nl, F, F2 = C.shape
        assert F == F2
        T = self.empty((nl, F, F))
        info = torch.empty((nl,), dtype=torch.int32, device=self.device)
        _check(self.lib.corahip_factor_batched(self.h, self._f64(C), nl, F, jitter_rel, eig_thresh, self._f64(T),
                                               self._p(info)))
        return T, info

    # -- K3 ---------------------------------------------------------------------------
    def normals_philox(self, seed, lmax, F, out=None):
        n = 2 * F * (lmax + 1) * (lmax + 2) // 2
        g = out if out is not None else self.empty((n,))
        assert g.numel() >= n
        _check(self.lib.corahip_normals_philox(self.h, c_u64(int(seed) & (2**64 - 1)), lmax, F, self._f64(g)))
        return g

    def normals_pcg64(self, state, inc, n, out=None):
        """The next ``n`` values of numpy's ``Generator(PCG64).standard_normal`` from bit-generator ``state`` / ``inc``
        (python ints, 128 bits), generated on the device; returns (g, n_raw): the normals and the number of raw 64-bit
        draws they consumed (see :func:`pcg64_advance`)."""
        g = out if out is not None else self.empty((n,))
        assert g.numel() >= n
        M = 2**64 - 1
        st = (c_u64 * 2)((int(state) >> 64) & M, int(state) & M)
        ic = (c_u64 * 2)((int(inc) >> 64) & M, int(inc) & M)
        nraw = c_u64(0)
        _check(self.lib.corahip_normals_pcg64(self.h, st, ic, int(n), self._f64(g), ctypes.byref(nraw)))
        return g, int(nraw.value)

    def glibc_exp(self, x):
        """exp(x) of a device float64 tensor as the wedge test of the device ziggurat evaluates it (glibc's routine,
        restated): the test hook ``corahip_glibc_exp``."""
        y = self.empty(tuple(x.shape))
        _check(self.lib.corahip_glibc_exp(self.h, self._f64(x), int(x.numel()), self._f64(y)))
        return y

    def normals_legacy(self, state, n, out=None):
        """The next ``n`` values of numpy's LEGACY stream (``np.random.standard_normal`` / ``RandomState``: MT19937 + polar
        method) from ``state`` = ``get_state(legacy=False)``, generated on the device; returns (g, state after) with the
        state as a dict ``set_state`` takes."""
        import numpy as np

        class MtState(ctypes.Structure):
            _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32),
                        ("gauss", c_double)]

        if state["bit_generator"] != "MT19937":
            raise ValueError("legacy stream on the device: MT19937 only")
        ms = MtState()
        key = np.ascontiguousarray(state["state"]["key"], dtype=np.uint32)
        ctypes.memmove(ms.key, key.ctypes.data, 624 * 4)
        ms.pos, ms.has_gauss, ms.gauss = int(state["state"]["pos"]), int(state["has_gauss"]), float(state["gauss"])
        g = out if out is not None else self.empty((n,))
        assert g.numel() >= n
        _check(self.lib.corahip_normals_mt19937_legacy(self.h, ctypes.byref(ms), int(n), self._f64(g)))
        new = {"bit_generator": "MT19937", "state": {"key": np.frombuffer(ms.key, dtype=np.uint32).copy(), "pos": int(ms.pos)},
               "has_gauss": int(ms.has_gauss), "gauss": float(ms.gauss)}
        return g, new

    def draw_alm(self, T, info, g, lmax, F, nu0=0, nnu=None, out=None):
        nnu = F if nnu is None else nnu
        nalm = (lmax + 1) * (lmax + 2) // 2
        G = (nnu + 3) // 4
        alm = out if out is not None else self.empty((nalm, G, 2, 4))
        _check(self.lib.corahip_draw_alm(self.h, self._f64(T), self._p(info) if info is not None else None,
                                         self._f64(g), lmax, F, nu0, nnu, self._f64(alm)))
        return alm

    def draw_alm_rows(self, T_rows, info, g, lmax, F, nu0, nnu, out=None):
        """draw_alm with T_rows [lmax+1, nnu, F] = rows nu0..nu0+nnu-1 of every factor."""
        assert tuple(T_rows.shape) == (lmax + 1, nnu, F), T_rows.shape
        nalm = (lmax + 1) * (lmax + 2) // 2
        G = (nnu + 3) // 4
        alm = out if out is not None else self.empty((nalm, G, 2, 4))
        _check(self.lib.corahip_draw_alm_rows(self.h, self._f64(T_rows), self._p(info) if info is not None else None,
                                              self._f64(g), lmax, F, nu0, nnu, self._f64(alm)))
        return alm

    def draw_alm_numpy_prepare(self, rng, lmax, F, ring_bytes=0):
        """``corahip_draw_alm_numpy_prepare``: the generator's part of :meth:`draw_alm_numpy` - its count / jump passes and
        the first two ranges of normals - enqueued on the library's generator stream NOW, so that it runs beside
        whatever the caller enqueues next (the kernels that make the factors).  Returns the handle ``draw_alm_numpy``
        takes as ``prepared``; ``handle.abort()`` gives the session up (generator untouched)."""
        r, ms = _rng_struct(rng)
        pend = c_void_p()
        _check(self.lib.corahip_draw_alm_numpy_prepare(self.h, ctypes.byref(r), int(lmax), int(F), int(ring_bytes), ctypes.byref(pend)))
        ctx = self

        class Prepared:
            rng_struct, mt_state, pending, shape, live = r, ms, pend, (int(lmax), int(F)), True

            def abort(self):
                if self.live:
                    self.live = False
                    _check(ctx.lib.corahip_draw_alm_numpy_end(ctx.h, self.pending, ctypes.byref(self.rng_struct)))

        return Prepared()

    def draw_alm_numpy(self, T, info, rng, lmax, F, nu0=0, nnu=None, out=None, rows=False, ring_bytes=0, defer=False,
                       chunks=None, prepared=None):
        """``corahip_draw_alm_numpy``: K3 with numpy's own stream generated on the device range by range (no 16 F nalm
        byte buffer).  ``rng``: ("pcg64", state, inc) python ints of a PCG64 bit generator, or ("legacy",
        get_state(legacy=False) dict).  ``rows``: T is the row block [L, nnu, F].  Returns (alm, state after): the PCG64
        state as a python int, or the legacy state dict ``set_state`` takes.  ``defer``: returns (alm, finish) instead -
        everything is enqueued, nothing waited for; ``finish()`` (``corahip_draw_alm_numpy_end``: the one read-back)
        returns the state after and is called once the caller has enqueued what follows (the synthesis).
        ``chunks``: [(first, count), (first, count)] - the two chunks of a folded frequency shard (row-block T in local
        channel order) instead of ``nu0`` / ``nnu``.  ``prepared``: the handle of :meth:`draw_alm_numpy_prepare` (``rng``
        is then ignored: the session carries the generator)."""
        import numpy as np

        if chunks is not None:
            rows, nu0, nnu = True, chunks[0][0], sum(c[1] for c in chunks)
        nnu = F if nnu is None else nnu
        assert tuple(T.shape) == ((lmax + 1, nnu, F) if rows else (lmax + 1, F, F)), T.shape
        nalm = (lmax + 1) * (lmax + 2) // 2
        alm = out if out is not None else self.empty((nalm, (nnu + 3) // 4, 2, 4))
        if prepared is not None:
            # the generator's passes were enqueued earlier (draw_alm_numpy_prepare): K3 against the factors now
            assert prepared.live and prepared.shape == (int(lmax), int(F)), "the prepared session is for another shape"
            r, ms, pend = prepared.rng_struct, prepared.mt_state, prepared.pending
            cs = _chanset(chunks if chunks is not None and len(chunks) > 1 else [(nu0, nnu)])
            prepared.live = False
            try:
                _check(self.lib.corahip_draw_alm_numpy_run(self.h, pend, self._f64(T), 1 if rows else 0,
                                                           self._p(info) if info is not None else None, ctypes.byref(cs), self._f64(alm)))
            except CoraHipError:
                self.lib.corahip_draw_alm_numpy_end(self.h, pend, ctypes.byref(r))       # (frees the session)
                raise
        else:
            r, ms = _rng_struct(rng)
            pend = c_void_p()
            if chunks is not None and len(chunks) > 1:
                cs = _chanset(chunks)
                _check(self.lib.corahip_draw_alm_numpy_begin_set(self.h, self._f64(T), self._p(info) if info is not None else None,
                                                                 ctypes.byref(r), lmax, F, ctypes.byref(cs), self._f64(alm),
                                                                 int(ring_bytes), ctypes.byref(pend)))
            else:
                _check(self.lib.corahip_draw_alm_numpy_begin(self.h, self._f64(T), 1 if rows else 0,
                                                             self._p(info) if info is not None else None, ctypes.byref(r), lmax,
                                                             F, nu0, nnu, self._f64(alm), int(ring_bytes), ctypes.byref(pend)))
        keep = [T, info, alm]           # (alive until the queue has been waited for)

        def finish():
            _check(self.lib.corahip_draw_alm_numpy_end(self.h, pend, ctypes.byref(r)))
            keep.clear()
            if ms is not None:
                return {"bit_generator": "MT19937", "state": {"key": np.frombuffer(ms.key, dtype=np.uint32).copy(), "pos": int(ms.pos)},
                        "has_gauss": int(ms.has_gauss), "gauss": float(ms.gauss)}
            return (int(r.state[0]) << 64) | int(r.state[1])

        if defer:
            return alm, finish
        return alm, finish()

    def draw_alm_philox(self, T, info, seed, lmax, F, nu0=0, nnu=None, out=None):
        nnu = F if nnu is None else nnu
        nalm = (lmax + 1) * (lmax + 2) // 2
        G = (nnu + 3) // 4
        alm = out if out is not None else self.empty((nalm, G, 2, 4))
        _check(self.lib.corahip_draw_alm_philox(self.h, self._f64(T), self._p(info) if info is not None else None,
                                                c_u64(int(seed) & (2**64 - 1)), lmax, F, nu0, nnu, self._f64(alm)))
        return alm

    def draw_alm_philox_rows(self, T_rows, info, seed, lmax, F, nu0, nnu, out=None):
        """draw_alm_philox with T_rows [lmax+1, nnu, F] = rows nu0..nu0+nnu-1 of every factor."""
        assert tuple(T_rows.shape) == (lmax + 1, nnu, F), T_rows.shape
        nalm = (lmax + 1) * (lmax + 2) // 2
        G = (nnu + 3) // 4
        alm = out if out is not None else self.empty((nalm, G, 2, 4))
        _check(self.lib.corahip_draw_alm_philox_rows(self.h, self._f64(T_rows), self._p(info) if info is not None else None,
                                                     c_u64(int(seed) & (2**64 - 1)), lmax, F, nu0, nnu, self._f64(alm)))
        return alm

    def draw_alm_philox_chunks(self, T_rows, info, seed, lmax, F, chunks, out=None):
        """draw_alm_philox_rows for the channels of ``chunks`` = [(first, count), ...]: one block, or the two equal
        chunks of a folded frequency shard; T_rows [lmax+1, sum(count), F] in local channel order."""
        nnu = sum(int(c[1]) for c in chunks)
        assert tuple(T_rows.shape) == (lmax + 1, nnu, F), T_rows.shape
        nalm = (lmax + 1) * (lmax + 2) // 2
        alm = out if out is not None else self.empty((nalm, (nnu + 3) // 4, 2, 4))
        cs = _chanset(chunks)
        _check(self.lib.corahip_draw_alm_philox_rows_set(self.h, self._f64(T_rows), self._p(info) if info is not None else None,
                                                         c_u64(int(seed) & (2**64 - 1)), lmax, F, ctypes.byref(cs),
                                                         self._f64(alm)))
        return alm

    # -- frequency sharding: the data movements around the factor row-block all-to-all ---------
    def factor_rows_pack(self, T_local, l_stride, world):
        """[n_local, F, F] l-sharded factors -> send [world, l_stride, F / world, F] (slab q = rows of rank q's
        channels, l rows past n_local zero)."""
        n_local, F, _ = T_local.shape
        send = self.empty((world, l_stride, F // world, F))
        _check(self.lib.corahip_factor_rows_pack(self.h, self._f64(T_local) if n_local else None, n_local, l_stride, F,
                                                 world, self._f64(send)))
        return send

    def factor_rows_unpack(self, recv, counts):
        """recv [world, l_stride, nnu, F] (slab r from rank r) + the multipoles each rank holds -> T_rows
        [sum(counts), nnu, F], the l blocks in rank order."""
        world, l_stride, nnu, F = recv.shape
        cnt = (ctypes.c_int32 * world)(*[int(c) for c in counts])
        out = self.empty((int(sum(counts)), nnu, F))
        _check(self.lib.corahip_factor_rows_unpack(self.h, self._f64(recv), cnt, world, l_stride, nnu, F, self._f64(out)))
        return out

    def alm_dev_to_square(self, alm, lmax, nnu):
        torch = _torch()
        L = lmax + 1
        sq = torch.empty((nnu, 1, L, L), dtype=torch.complex128, device=self.device)
        _check(self.lib.corahip_alm_dev_to_square(self.h, self._f64(alm), lmax, nnu, self._p(sq)))
        return sq

    def alm_packed_to_dev(self, packed, lmax):
        torch = _torch()
        assert packed.dtype == torch.complex128
        nnu, nalm = packed.shape
        assert nalm == (lmax + 1) * (lmax + 2) // 2
        alm = self.empty((nalm, (nnu + 3) // 4, 2, 4))
        _check(self.lib.corahip_alm_packed_to_dev(self.h, self._p(packed), lmax, nnu, self._f64(alm)))
        return alm

    # -- K4/K5 ------------------------------------------------------------------------
    # truncation exponent of the Legendre sums used by plans made without an explicit one (0 = the library's
    # default, 2^-70); `Context.sht_cut_exp = -900` before the first transform keeps every representable term
    sht_cut_exp = 0

    def sht_plan(self, nside, lmax, cut_exp=None):
        """The (nside, lmax) transform plan; terms of the Legendre sums below 2^cut_exp are dropped (pixel error
        <= 2 sum |a_lm| 2^cut_exp; default -70, see corahip_sht_plan_create_ex)."""
        cut = int(self.sht_cut_exp if cut_exp is None else cut_exp)
        key = (int(nside), int(lmax)) if cut == 0 else (int(nside), int(lmax), cut)
        if key not in self._plans:
            h = c_void_p()
            _check(self.lib.corahip_sht_plan_create_ex(self.h, key[0], key[1], cut, ctypes.byref(h)))
            self._plans[key] = h
        return self._plans[key]

    def k4_mfma_count(self, nside, lmax, nnu, cut_exp=None):
        """FP64 MFMA instructions (v_mfma_f64_16x16x4_f64, 2048 flop each) the Legendre kernel issues for one
        alm2map pass over ``nnu`` channels - from the plan's tables, equal to the SQ_INSTS_VALU_MFMA_F64 counter."""
        n = c_u64()
        _check(self.lib.corahip_sht_plan_k4_mfma_count(self.h, self.sht_plan(nside, lmax, cut_exp), int(nnu), ctypes.byref(n)))
        return int(n.value)

    def alm2map_workspace_bytes(self, plan, nnu):
        b = c_size_t()
        _check(self.lib.corahip_alm2map_workspace_bytes(plan, nnu, ctypes.byref(b)))
        return int(b.value)

    def workspace(self, nbytes):
        torch = _torch()
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = None
            self._workspace = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        return self._workspace

    def alm2map(self, alm, nside, lmax, nnu, out=None, max_workspace_bytes=None, cut_exp=None):
        plan = self.sht_plan(nside, lmax, cut_exp)
        npix = 12 * nside * nside
        maps = out if out is not None else self.empty((nnu, npix))
        need = self.alm2map_workspace_bytes(plan, nnu)
        if max_workspace_bytes is not None:
            need = min(need, int(max_workspace_bytes))
        ws = self.workspace(need)
        _check(self.lib.corahip_alm2map(self.h, plan, self._f64(alm), nnu, self._f64(maps), self._p(ws), need))
        return maps

    def mkfullsky_fused(self, C, nside, rng, nu0=0, nnu=None, alms=False, workspace_bytes=None):
        """``corahip_mkfullsky``: C [L, F, F] (device) -> maps [nnu, npix] or, with ``alms``, a_lm [nnu, 1, L, L] complex,
        in one library call.  ``rng``: ("pcg64", state, inc) python ints of a numpy bit generator - returns the state
        after the draws -, ("stream", g) device normals in stream order, or ("philox", seed).  Returns (out, state)."""
        torch = _torch()
        L, F = int(C.shape[0]), int(C.shape[1])
        lmax = L - 1
        nnu = F if nnu is None else nnu
        plan = self.sht_plan(nside, lmax)

        class MtState(ctypes.Structure):
            _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32),
                        ("gauss", c_double)]

        class Rng(ctypes.Structure):
            _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stream", c_void_p), ("seed", c_u64),
                        ("state", c_u64 * 2), ("inc", c_u64 * 2), ("legacy", ctypes.POINTER(MtState))]

        r = Rng()
        M = 2**64 - 1
        kind = {"stream": 0, "philox": 1, "pcg64": 2, "legacy": 3}[rng[0]]
        ms = None
        if kind == 3:        # ("legacy", np.random.get_state(legacy=False)): returns the state dict after the draws
            import numpy as np

            ms = MtState()
            key = np.ascontiguousarray(rng[1]["state"]["key"], dtype=np.uint32)
            ctypes.memmove(ms.key, key.ctypes.data, 624 * 4)
            ms.pos, ms.has_gauss, ms.gauss = int(rng[1]["state"]["pos"]), int(rng[1]["has_gauss"]), float(rng[1]["gauss"])
            r.legacy = ctypes.pointer(ms)
        r.kind = kind
        if kind == 0:
            r.stream = self._f64(rng[1])
        elif kind == 1:
            r.seed = int(rng[1]) & M
        elif kind == 2:
            r.state[0], r.state[1] = (int(rng[1]) >> 64) & M, int(rng[1]) & M
            r.inc[0], r.inc[1] = (int(rng[2]) >> 64) & M, int(rng[2]) & M
        b = c_size_t()
        _check(self.lib.corahip_mkfullsky_workspace_bytes(plan, F, nu0, nnu, kind, 1 if alms else 0, ctypes.byref(b)))
        need = int(b.value) if workspace_bytes is None else int(workspace_bytes)
        ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=self.device)
        out = (torch.empty((nnu, 1, L, L), dtype=torch.complex128, device=self.device) if alms
               else self.empty((nnu, 12 * nside * nside)))
        _check(self.lib.corahip_mkfullsky(self.h, plan, self._f64(C), F, ctypes.byref(r), nu0, nnu, 1 if alms else 0,
                                          c_void_p(out.data_ptr()), self._p(ws), need))
        if kind == 3:
            import numpy as np

            return out, {"bit_generator": "MT19937", "state": {"key": np.frombuffer(ms.key, dtype=np.uint32).copy(), "pos": int(ms.pos)},
                         "has_gauss": int(ms.has_gauss), "gauss": float(ms.gauss)}
        return out, ((int(r.state[0]) << 64) | int(r.state[1])) if kind == 2 else None

    def map2alm_workspace_bytes(self, plan, nnu):
        b = c_size_t()
        _check(self.lib.corahip_map2alm_workspace_bytes(plan, nnu, ctypes.byref(b)))
        return int(b.value)

    def map2alm(self, maps, nside, lmax, ring_w=None, chunk=None):
        """One weighted quadrature pass maps [nnu, npix] -> alm_dev [nalm, ceil(nnu/4), 2, 4] (K5^T + K4^T).

        ring_w: device [2 nside] north-ring weights or None.  Channels go through in chunks of `chunk`
        (a multiple of 8; default: as many as a 96 GB workspace holds)."""
        torch = _torch()
        plan = self.sht_plan(nside, lmax)
        nnu, npix = maps.shape
        assert npix == 12 * nside * nside
        nalm = (lmax + 1) * (lmax + 2) // 2
        G4 = (nnu + 3) // 4
        if chunk is None:
            per8 = self.map2alm_workspace_bytes(plan, 8)
            chunk = max(8, min((int(96e9 // per8)) * 8, (nnu + 7) // 8 * 8))
        assert chunk % 8 == 0
        out = self.empty((nalm, G4, 2, 4))
        for c0 in range(0, nnu, chunk):
            n = min(chunk, nnu - c0)
            G8 = (n + 7) // 8 * 2
            need = self.map2alm_workspace_bytes(plan, n)
            ws = self.workspace(need)
            part = out if (c0 == 0 and n == nnu and G8 == G4) else self.empty((nalm, G8, 2, 4))
            _check(self.lib.corahip_map2alm(self.h, plan, self._f64(maps[c0:c0 + n]), n,
                                            self._f64(ring_w) if ring_w is not None else None, self._f64(part),
                                            self._p(ws), need))
            if part is not out:
                g0 = c0 // 4
                gn = min(G8, G4 - g0)
                out[:, g0:g0 + gn].copy_(part[:, :gn])
        return out

    def alm2map_spin2(self, alm, nside, lmax, nnu, out=None):
        """alm_dev of nnu = 2 nfreq interleaved (E, B) channels -> maps [nnu, npix] = interleaved (Q, U)."""
        plan = self.sht_plan(nside, lmax)
        maps = out if out is not None else self.empty((nnu, 12 * nside * nside))
        need = self.alm2map_workspace_bytes(plan, nnu)
        ws = self.workspace(need)
        _check(self.lib.corahip_alm2map_spin2(self.h, plan, self._f64(alm), nnu, self._f64(maps), self._p(ws), need))
        return maps

    # -- spin-2 analysis (composition of scalar passes) ---------------------------------------
    def map2alm_spin2(self, maps_qu, nside, lmax, ring_w=None):
        """One quadrature pass (Q_f, U_f interleaved) [2 nf, npix] -> alm_dev [nalm, G, 2, 4] with (E_f, B_f)
        interleaved, G = nnu_pad8(2 nf) / 4 (the layout alm2map_spin2 takes)."""
        plan = self.sht_plan(nside, lmax)
        n2, npix = maps_qu.shape
        assert n2 % 2 == 0 and npix == 12 * nside * nside
        nf = n2 // 2
        nalm = (lmax + 1) * (lmax + 2) // 2
        maps6 = self.empty((6 * nf, npix))
        _check(self.lib.corahip_spin2_ring_scale(self.h, plan, self._f64(maps_qu), nf, self._f64(maps6)))
        a6 = self.map2alm(maps6, nside, lmax, ring_w)
        del maps6
        gout = (2 * nf + 7) // 8 * 2
        out = self.empty((nalm, gout, 2, 4))
        _check(self.lib.corahip_spin2_combine(self.h, plan, self._f64(a6), a6.shape[1], nf, self._f64(out), gout))
        return out

    # -- n3: xi(r) -> C_l --------------------------------------------------------------
    def xi_table_average(self, kx, ky, ky2, kind, x_t, f_t, mu, xa, xw, F, xint):
        nm = mu.numel()
        out = self.empty((nm, F, F))
        _check(self.lib.corahip_xi_table_average(self.h, self._f64(kx), self._f64(ky), self._f64(ky2), kx.numel(), kind,
                                                 float(x_t), float(f_t), self._f64(mu), nm, self._f64(xa), self._f64(xw),
                                                 F, xint, self._f64(out)))
        return out

    def legendre_project(self, mu, wt, lmax, xi):
        nm = mu.numel()
        ncol = xi.numel() // nm
        out = self.empty((lmax + 1, ncol))
        _check(self.lib.corahip_legendre_project(self.h, self._f64(mu), self._f64(wt), nm, lmax, self._f64(xi),
                                                 ncol, self._f64(out)))
        return out

    # -- n4: flat-sky fields -----------------------------------------------------------
    def _c128(self, t):
        torch = _torch()
        assert t.dtype == torch.complex128 and t.device == self.device
        return self._p(t)

    @staticmethod
    def _dims(shape):
        return (ctypes.c_int64 * len(shape))(*[int(x) for x in shape])

    def fft_c2c(self, data, axis, inverse=False):
        """In-place numpy.fft.fft / ifft of a complex128 device array along one axis."""
        axis = axis % data.dim()
        _check(self.lib.corahip_fft_c2c(self.h, self._c128(data), data.dim(), self._dims(data.shape), axis,
                                        1 if inverse else 0))
        return data

    def irfftn(self, spec, naxes=None, last=None):
        """numpy.fft.irfftn over the last ``naxes`` axes (default all); ``spec`` is overwritten."""
        nd = spec.dim()
        naxes = nd if naxes is None else naxes
        rshape = list(spec.shape)
        rshape[-1] = 2 * (spec.shape[-1] - 1) if last is None else int(last)
        if rshape[-1] // 2 + 1 != spec.shape[-1]:
            raise CoraHipError("irfftn: last axis %d does not match %d spectral bins" % (rshape[-1], spec.shape[-1]))
        out = self.empty(tuple(rshape))
        _check(self.lib.corahip_irfftn(self.h, self._c128(spec), nd, self._dims(rshape), naxes, self._f64(out)))
        return out

    def rfftn(self, arr, naxes=None):
        """numpy.fft.rfftn over the last ``naxes`` axes (default all) of a float64 device array."""
        torch = _torch()
        nd = arr.dim()
        naxes = nd if naxes is None else naxes
        cshape = list(arr.shape)
        cshape[-1] = arr.shape[-1] // 2 + 1
        spec = torch.empty(tuple(cshape), dtype=torch.complex128, device=self.device)
        _check(self.lib.corahip_rfftn(self.h, self._f64(arr), nd, self._dims(arr.shape), naxes, self._c128(spec)))
        return spec

    def randomfield_draw(self, kweight, seed):
        """(N(0,1) + i N(0,1)) * kweight from the Philox device stream (counter = flat element index)."""
        torch = _torch()
        spec = torch.empty(tuple(kweight.shape), dtype=torch.complex128, device=self.device)
        _check(self.lib.corahip_randomfield_draw(self.h, self._f64(kweight), kweight.numel(), int(seed),
                                                 self._c128(spec)))
        return spec

    def randomfield_irfftn(self, kweight, seed, last=None, spec=None):
        """``randomfield_draw`` + ``irfftn`` over all axes in one call: the spectrum is generated where the first pass
        loads it (same values, no separate draw pass).  kweight real [..., n/2 + 1]; returns the real field."""
        torch = _torch()
        nd = kweight.dim()
        rshape = list(kweight.shape)
        rshape[-1] = 2 * (kweight.shape[-1] - 1) if last is None else int(last)
        if rshape[-1] // 2 + 1 != kweight.shape[-1]:
            raise CoraHipError("randomfield_irfftn: last axis %d does not match %d spectral bins" % (rshape[-1], kweight.shape[-1]))
        if spec is None:
            spec = torch.empty(tuple(kweight.shape), dtype=torch.complex128, device=self.device)
        out = self.empty(tuple(rshape))
        _check(self.lib.corahip_randomfield_irfftn(self.h, self._f64(kweight), nd, self._dims(rshape), int(seed) & (2**64 - 1),
                                                   self._c128(spec), self._f64(out)))
        return out

    def fg_mix(self, freq_weight, normals, aff):
        """out[f] = aff * sum_c freq_weight[f, c] normals[c]  (complex [F, *aff.shape])."""
        torch = _torch()
        F, ncorr = freq_weight.shape
        assert normals.shape[0] == ncorr and tuple(normals.shape[1:]) == tuple(aff.shape)
        out = torch.empty((F,) + tuple(aff.shape), dtype=torch.complex128, device=self.device)
        _check(self.lib.corahip_fg_mix(self.h, self._f64(freq_weight), self._f64(normals), self._c128(aff), F, ncorr,
                                       aff.numel(), self._c128(out)))
        return out

    def spec_mul_real(self, spec, weight):
        assert tuple(spec.shape) == tuple(weight.shape)
        _check(self.lib.corahip_spec_mul_real(self.h, self._c128(spec), self._f64(weight), spec.numel()))
        return spec

    def cube_affine(self, df, vf, a, b, c):
        """out[z] = a[z] df[z] + b[z] vf[z] + c[z]; ``vf`` may be None."""
        n0 = df.shape[0]
        out = self.empty(tuple(df.shape))
        _check(self.lib.corahip_cube_affine(self.h, self._f64(df), None if vf is None else self._f64(vf), self._f64(a),
                                            None if vf is None else self._f64(b), self._f64(c), n0, df.numel() // n0,
                                            self._f64(out)))
        return out

    def raytrace_slices(self, cube, zc, scale, tx, ty, wx, wy):
        n0, n1, n2 = cube.shape
        numz, numx, numy = zc.numel(), tx.numel(), ty.numel()
        out = self.empty((numz, numx, numy))
        _check(self.lib.corahip_raytrace_slices(self.h, self._f64(cube), n0, n1, n2, self._f64(zc), self._f64(scale),
                                                self._f64(tx), self._f64(ty), float(wx), float(wy), numz, numx, numy,
                                                self._f64(out)))
        return out

    def sht_rings(self, nside, lmax):
        plan = self.sht_plan(nside, lmax)
        nring = 4 * nside - 1
        start = np.zeros(nring, dtype=np.int64)
        nphi = np.zeros(nring, dtype=np.int32)
        z = np.zeros(nring)
        phi0 = np.zeros(nring)
        _check(self.lib.corahip_sht_plan_rings(plan, start.ctypes.data_as(c_void_p), nphi.ctypes.data_as(c_void_p),
                                               z.ctypes.data_as(c_void_p), phi0.ctypes.data_as(c_void_p)))
        return dict(start=start, nphi=nphi, z=z, phi0=phi0)

    def sht_ring_classes(self, nside, lmax):
        """Ring-FFT class of every ring as the synthesis kernels take it: 0 = direct transform, else the Bluestein
        length (a power of two or 3 * 2^k)."""
        cls = np.zeros(4 * nside - 1, dtype=np.int32)
        _check(self.lib.corahip_sht_plan_ring_classes(self.sht_plan(nside, lmax), cls.ctypes.data_as(c_void_p)))
        return cls

    def sht_lambda(self, nside, lmax, m, ring_pair):
        plan = self.sht_plan(nside, lmax)
        out = self.empty((lmax - m + 1,))
        _check(self.lib.corahip_sht_lambda(self.h, plan, m, ring_pair, self._f64(out)))
        return out

    def sht_lambda_entry(self, nside, lmax, m, ring_pair, kq):
        """lambda_lm as lane group ``kq`` of the synthesis kernel forms them (entry at a window start from the plan's
        four entry states per (m, ring))."""
        plan = self.sht_plan(nside, lmax)
        out = self.empty((lmax - m + 1,))
        _check(self.lib.corahip_sht_lambda_entry(self.h, plan, m, ring_pair, kq, self._f64(out)))
        return out


_contexts = {}


def get_context(device=None):
    """Cached Context for `device` (default: torch's current CUDA device)."""
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _contexts:
        _contexts[device] = Context(device)
    ctx = _contexts[device]
    ctx.use_current_stream()
    return ctx
