"""Counterpart of the hot-path part of cora/signal/corr.py: the flat-sky FFT table model
of C_l(z, z') (``RedshiftCorrelation.angular_powerspectrum_fft``, corr.py:891-982).

Setup (host, one-off per instance, exactly the reference's recipe corr.py:909-942):
P(k) on a [500 log k_perp] x [32768 linear k_par] grid, times {1, mu^2, mu^4}, DCT-I along
k_par -> three [500, 32768] tables, uploaded once to HBM (393 MB) and cached.
Evaluation (device): K1 in csrc/clarray.hip, either fused with the Romberg channel
average (through ``skysim.clarray``) or at arbitrary broadcast points (calling
``angular_powerspectrum`` directly).

The flat-sky redshift-space cube (``_realisation_dv`` / ``realisation``, corr.py:562-770) runs on
the GPU through csrc/flatsky.hip: Gaussian field draw, the mu^2 velocity field by rfftn / irfftn,
per-slice growth factors and the ray-traced trilinear resampling.

The multipole correlation functions and ``angular_powerspectrum_full`` of the reference's corr.py
are outside this package's scope.
"""
import math

import numpy as np

from .. import _lib
from ..core import gaussianfield
from ..util import constants, fftutil
from ..util import cubicspline as cs
from ..util.cosmology import Cosmology


class RedshiftCorrelation(object):
    r"""Redshift-space correlations in the linear regime: the C_l(z, z') table model.

    Parameters
    ----------
    ps_vv : function
        Velocity power spectrum P(k) [k in h/Mpc] at ``redshift``.
    ps_dd, ps_dv : function, optional
        Accepted for interface compatibility; the table model uses ``ps_vv`` with ``bias``.
    redshift : scalar, optional
        Redshift at which the power spectra are calculated.
    bias : scalar, optional
        Bias between the observable and the velocities.
    """

    ps_vv = None
    ps_dd = None
    ps_dv = None
    ps_2d = False
    ps_redshift = 0.0
    bias = 1.0
    _vv_only = False

    kperpmin = 1e-4
    kperpmax = 40.0
    nkperp = 500
    kparmax = 20.0
    nkpar = 32768
    _freq_window = 0.0
    _aps_cache = False

    def __init__(self, ps_vv=None, ps_dd=None, ps_dv=None, redshift=0.0, bias=1.0):
        self.ps_vv = ps_vv
        self.ps_dd = ps_dd
        self.ps_dv = ps_dv
        self.ps_redshift = redshift
        self.bias = bias
        self._vv_only = False if ps_dd and ps_dv else True
        self.cosmology = Cosmology()
        self._dev_tables = {}

    # ---- model hooks (corr.py:448-530) --------------------------------------------------
    def bias_z(self, z):
        return self.bias * np.ones_like(z)

    def growth_factor(self, z):
        return 1.0 / (1.0 + z)

    def growth_rate(self, z):
        return 1.0 * np.ones_like(z)

    def prefactor(self, z):
        return 1.0 * np.ones_like(z)

    def mean(self, z):
        return np.ones_like(z) * 0.0

    # ---- flat-sky redshift-space cube (corr.py:547-770) ----------------------------------
    _sigma_v = 0.0

    def sigma_v(self, z):
        """Pairwise velocity dispersion, ``_sigma_v`` km/s -> h^-1 Mpc (corr.py:549-556)."""
        return np.ones_like(z) * (self._sigma_v / 100.0)

    def velocity_damping(self, kpar):
        """Lorentzian damping of the line-of-sight modes (corr.py:558-560)."""
        return (1.0 + (kpar * self.sigma_v(self.ps_redshift)) ** 2.0) ** -1.0

    def _cube_generator(self, d, n):
        # field generator + mu^2 array of one cube geometry, resident on the device and reused between calls
        key = (tuple(float(x) for x in d), tuple(int(x) for x in n))
        cache = self.__dict__.setdefault("_cube_cache", {})
        if key not in cache:
            rfv = gaussianfield.RandomField(npix=np.array(n), wsize=np.array(d))
            rfv.powerspectrum = lambda karray: (self.ps_vv((karray**2).sum(axis=3) ** 0.5)
                                                * self.velocity_damping(karray[..., 0]))
            rfv.generate_kweight()
            kvec = fftutil.rfftfreqn(rfv._n, (rfv._w / rfv._n) / (2 * math.pi))
            with np.errstate(invalid="ignore", divide="ignore"):
                mu2 = kvec[..., 0] ** 2 / (kvec**2).sum(axis=3)
            mu2.flat[0] = 0.0
            cache.clear()
            cache[key] = (rfv, _lib.get_context().to_device(mu2))
        return cache[key]

    def _realisation_dv_device(self, d, n, seed=None):
        """Density and line-of-sight velocity cubes as device tensors (corr.py:562-603).

        ``seed=None`` uses numpy's global state exactly as the reference; an integer draws on the GPU.
        """
        if not self._vv_only:
            raise Exception("Doesn't work for independent fields, I need to think a bit more first.")
        ctx = _lib.get_context()
        rfv, mu2 = self._cube_generator(d, n)
        df = rfv.getfield_device(seed=seed)
        spec = ctx.spec_mul_real(ctx.rfftn(df), mu2)
        vf = ctx.irfftn(spec)
        return df, vf

    def _realisation_dv(self, d, n):
        df, vf = self._realisation_dv_device(d, n)
        return df.cpu().numpy(), vf.cpu().numpy()

    def realisation(self, z1, z2, thetax, thetay, numz, numx, numy, zspace=True, refinement=1,
                    report_physical=False, density_only=False, no_mean=False, no_evolution=False, pad=5,
                    seed=None, device=False):
        """Simulate a redshift-space volume ``[numz, numx, numy]`` in the flat-sky approximation.

        Same arguments and geometry as corr.py:605-770: a padded comoving box between ``z1`` and ``z2`` is
        filled with a Gaussian density field and its mu^2 velocity field, scaled slice by slice with
        D(z), b(z), f(z), the prefactor and the mean, and resampled along the lines of sight onto regular
        angle / redshift (``zspace``) or angle / scale-factor bins by trilinear interpolation.  ``seed`` and
        ``device`` are extensions: device-side normals, and device tensors instead of numpy arrays.
        """
        cosmo = self.cosmology
        d1, d2 = cosmo.proper_distance(z1), cosmo.proper_distance(z2)
        c1, c2 = cosmo.comoving_distance(z1), cosmo.comoving_distance(z2)
        c_center = (c1 + c2) / 2.0
        d = np.array([c2 - c1, thetax * d2 * constants.degree, thetay * d2 * constants.degree])
        n = np.array([numz, int(d2 / d1 * numx), int(d2 / d1 * numy)])
        if (n[-1] + pad) % 2 != 0:
            pad += 1
        d = d * (n + pad).astype(float) / n.astype(float)
        c1 = c_center - (c_center - c1) * (n[0] + pad) / float(n[0])
        c2 = c_center + (c2 - c_center) * (n[0] + pad) / float(n[0])
        n = refinement * (n + pad)

        ctx = _lib.get_context()
        df, vf = self._realisation_dv_device(d, n, seed=seed)
        n = tuple(df.shape)

        # redshift of every slice of the box, then the slice factors
        comoving_inv = inverse_approx(cosmo.comoving_distance, z1, z2)
        za = comoving_inv(np.linspace(c1, c2, n[0], endpoint=True))
        mz = self.mean(za)
        Dz = self.growth_factor(za) / self.growth_factor(self.ps_redshift)
        dfac = Dz * self.prefactor(za) * self.bias_z(za)
        vfac = Dz * self.prefactor(za) * self.growth_rate(za)
        if no_evolution:
            dfac = np.full(n[0], np.mean(dfac))
            vfac = np.full(n[0], np.mean(vfac))
        mean = np.zeros(n[0]) if no_mean else mz * np.ones(n[0])
        rsf = ctx.cube_affine(df, None if density_only else vf, ctx.to_device(dfac), ctx.to_device(vfac),
                              ctx.to_device(mean))

        # lines of sight: regular in redshift or in scale factor
        if zspace:
            za = np.linspace(z1, z2, numz, endpoint=False)
        else:
            za = 1.0 / np.linspace(1.0 / (1 + z2), 1.0 / (1 + z1), numz, endpoint=False)[::-1] - 1.0
        da = cosmo.proper_distance(za)
        xa = cosmo.comoving_distance(za)
        tx = np.linspace(-thetax / 2.0, thetax / 2.0, numx) * constants.degree
        ty = np.linspace(-thetay / 2.0, thetay / 2.0, numy) * constants.degree
        zc = (xa - c1) / (c2 - c1) * (n[0] - 1.0)
        acube = ctx.raytrace_slices(rsf, ctx.to_device(zc), ctx.to_device(da), ctx.to_device(tx), ctx.to_device(ty),
                                    d[1], d[2])
        if not device:
            acube, rsf = acube.cpu().numpy(), (rsf.cpu().numpy() if report_physical else rsf)
        if report_physical:
            return acube, rsf, (c1, c2, d[1], d[2])
        return acube

    # ---- table build / cache (corr.py:870-887, 909-942) ---------------------------------
    # K0: the tables are BUILT ON THE DEVICE (csrc/tables21.hip): P(k) on the 500 x 32768 grid - by the device spline
    # evaluator when ps_vv is the library's own spline form (Corr21cm: exp(-k^2 / 2 k*^2) LogInterpolater), by one
    # vectorised host call of the user's callable otherwise (it IS a Python callback) - then mu^2 / mu^4 and the
    # DCT-I along k_par on the GPU.  They stay in HBM; the host copies (`_aps_dd` ...) are made only when asked for.
    def _ps_spline_plan(self):
        """(interpolater, kstar) when ``ps_vv`` is still the spline-form callable a subclass registered."""
        plan = getattr(self, "_ps_plan", None)
        if plan is None or self.ps_2d or self.ps_vv is not plan["callable"]:
            return None
        return plan["spline"], float(plan["kstar"]())

    def _build_tables(self, ctx=None):
        ctx = ctx if ctx is not None else _lib.get_context()
        kperp = np.logspace(np.log10(self.kperpmin), np.log10(self.kperpmax), self.nkperp)
        kpar = np.linspace(0, self.kparmax, self.nkpar)
        kt_d, kp_d = ctx.to_device(kperp), ctx.to_device(kpar)
        sp = self._ps_spline_plan()
        kind = sp[0]._device_spline()[0] if sp is not None else None
        if sp is not None and kind in (0, 1):
            _, x, y, y2, _, _ = sp[0]._device_spline()
            dd, dv, vv = ctx.ps_table21cm(kt_d, kp_d, spline=(kind == 1, ctx.to_device(x), ctx.to_device(y), ctx.to_device(y2)),
                                          kstar=sp[1], freq_window=self._freq_window)
        else:
            kpar2, kperp2 = kpar[np.newaxis, :], kperp[:, np.newaxis]
            k = (kpar2**2 + kperp2**2) ** 0.5
            window = np.sinc(kpar2 * self._freq_window / (2 * np.pi)) ** 2
            dd_h = (self.ps_vv(k, kpar2 / k) if self.ps_2d else self.ps_vv(k)) * window
            dd, dv, vv = ctx.ps_table21cm(kt_d, kp_d, dd=ctx.to_device(np.ascontiguousarray(dd_h, dtype=np.float64)))
        norm = self.kparmax / (2 * self.nkpar)
        for t in (dd, dv, vv):
            ctx.dct1_rows(t, norm)
        # (the other devices' copies stay: a second context must not evict the first one's tables)
        if getattr(self, "_dev_tables", None) is None:
            self._dev_tables = {}
        pins = self.__dict__.get("_pins")
        if pins and ctx.device.index in pins:              # the set being replaced is pinned: not any more
            self._unpin({ctx.device.index: pins.pop(ctx.device.index)})
        self._dev_tables[ctx.device.index] = (dd, dv, vv)
        self._host_tables = None
        self._aps_cache = True

    def _host_copy(self, i):
        if not self._aps_cache:
            self._build_tables()
        if getattr(self, "_host_tables", None) is None:
            dev = next(iter(self._dev_tables.values()))
            self._host_tables = tuple(t.cpu().numpy() for t in dev)
        return self._host_tables[i]

    def _set_host_table(self, i, value):
        """Assigning a table (what the reference's load_fft_cache does, corr.py:879-887, and user subclasses may):
        it becomes the host copy; the device copies are dropped and re-uploaded on next use."""
        cur = list(self._host_tables) if getattr(self, "_host_tables", None) is not None else (
            [self._host_copy(k) if k != i else None for k in range(3)] if self._aps_cache else [None, None, None])
        cur[i] = np.ascontiguousarray(value, dtype=np.float64)
        self._host_tables = tuple(cur)
        self._drop_dev_tables()
        if all(t is not None for t in cur):
            self.nkperp, self.nkpar = cur[0].shape
            self._aps_cache = True

    _aps_dd = property(lambda self: self._host_copy(0), lambda self, v: self._set_host_table(0, v))
    _aps_dv = property(lambda self: self._host_copy(1), lambda self, v: self._set_host_table(1, v))
    _aps_vv = property(lambda self: self._host_copy(2), lambda self, v: self._set_host_table(2, v))

    def save_fft_cache(self, fname):
        """Save the three lookup tables (corr.py:870-877)."""
        np.savez(fname, dd=self._aps_dd, dv=self._aps_dv, vv=self._aps_vv)

    def load_fft_cache(self, fname):
        """Load lookup tables saved by :meth:`save_fft_cache` (corr.py:879-887)."""
        a = np.load(fname)
        self._host_tables = (a["dd"], a["dv"], a["vv"])
        self.nkperp, self.nkpar = self._host_tables[0].shape
        self._aps_cache = True
        self._drop_dev_tables()

    _table_generation = [0]      # process-wide counter: every (re)built / uploaded table set gets its own number

    @staticmethod
    def _unpin(pins):
        """Withdraws the K1 pins listed in ``pins`` ({device index: (ctx, generation)}): called when the device tables they
        vouch for are dropped - by the setters, ``load_fft_cache``, a rebuild - and by the finaliser of the model (the
        caching allocator hands a freed block to the next same-size request: a stale pin would vouch for foreign data)."""
        for ctx, gen in list(pins.values()):
            try:
                ctx.unpin_tables(gen)
            except Exception:       # (interpreter shutdown: the context may be gone already)
                pass
        pins.clear()

    def _drop_dev_tables(self):
        pins = self.__dict__.get("_pins")
        if pins:
            self._unpin(pins)
        self.__dict__.pop("_dev_table_gen", None)
        self._dev_tables = {}

    def _tables_on(self, ctx):
        dd_dv_vv = self._tables_on_unpinned(ctx)
        # K1 may keep its transposed copy of these tables between calls: they are this instance's cache and change only
        # through _build_tables / load_fft_cache / the _aps_* setters, each of which makes a new entry with a new number
        gens = self.__dict__.setdefault("_dev_table_gen", {})
        key = ctx.device.index
        if gens.get(key, (None, None))[0] is not dd_dv_vv:
            RedshiftCorrelation._table_generation[0] += 1
            gens[key] = (dd_dv_vv, RedshiftCorrelation._table_generation[0])
        ctx.pin_tables(*dd_dv_vv, gens[key][1])
        pins = self.__dict__.get("_pins")
        if pins is None:
            import weakref

            pins = self._pins = {}
            weakref.finalize(self, RedshiftCorrelation._unpin, pins)     # (holds the dict, not the model)
        pins[key] = (ctx, gens[key][1])
        return dd_dv_vv

    def _tables_on_unpinned(self, ctx):
        key = ctx.device.index
        if key not in self._dev_tables:
            ht = getattr(self, "_host_tables", None)
            if self._aps_cache and ht is not None and all(t is not None for t in ht):
                self._dev_tables[key] = tuple(ctx.to_device(t) for t in ht)   # loaded cache
            elif self._aps_cache and self._dev_tables:
                src = next(iter(self._dev_tables.values()))                   # another GPU holds them: device-to-device
                self._dev_tables[key] = tuple(t.to(ctx.device) for t in src)
            else:
                self._build_tables(ctx)
        return self._dev_tables[key]

    # ---- per-redshift quantities (corr.py:944-951) --------------------------------------
    def _z_quantities(self, za):
        za = np.asarray(za, dtype=np.float64)
        chi = self.cosmology.comoving_distance(za)
        pfd = self.prefactor(za) * self.growth_factor(za) / self.growth_factor(self.ps_redshift)
        return chi, pfd, self.growth_rate(za) * np.ones_like(za), self.bias_z(za) * np.ones_like(za)

    def _table_plan(self, za_to_z):
        def prepare(ctx, za):
            dd, dv, vv = self._tables_on(ctx)
            chi, pfd, f, b = self._z_quantities(za_to_z(za))
            return dict(dd=dd, dv=dv, vv=vv, kperpmin=self.kperpmin, kperpmax=self.kperpmax,
                        kparmax=self.kparmax, chi=chi, pfd=pfd, f=f, b=b)

        return dict(kind="table21cm", prepare=prepare)

    def _clarray_plan(self, aps):
        """Protocol used by ``skysim.clarray`` to recognise the table model: only the un-overridden flat-sky
        method qualifies (``angular_powerspectrum`` is an alias of it); any other bound method - a subclass
        override, ``angular_powerspectrum_full``, ... - goes down the generic host-callable path."""
        if getattr(aps, "__func__", None) is not RedshiftCorrelation.angular_powerspectrum_fft:
            return None
        return self._table_plan(lambda z: z)

    # ---- the aps callable ------------------------------------------------------------------
    def angular_powerspectrum_fft(self, la, za1, za2):
        """C_l(z1, z2) in the flat-sky limit, at broadcast points (corr.py:891-982)."""
        import torch

        ctx = _lib.get_context()
        dd, dv, vv = self._tables_on(ctx)
        la, za1, za2 = (np.asarray(v, dtype=np.float64) for v in (la, za1, za2))
        shape = np.broadcast(la, za1, za2).shape
        # per-redshift work on the unique redshifts only (chi is an ODE solve)
        zu, inv = np.unique(np.concatenate([za1.ravel(), za2.ravel()]), return_inverse=True)
        chi, pfd, f, b = self._z_quantities(zu)
        i1 = inv[: za1.size].reshape(za1.shape)
        i2 = inv[za1.size :].reshape(za2.shape)
        la = np.where(la == 0.0, 1e-10, la)

        def full(a):
            return np.array(np.broadcast_to(a, shape), dtype=np.float64, order="C").ravel()

        P = pfd[i1] * pfd[i2]
        args = [np.log10(la), chi[i1], chi[i2], b[i1] * b[i2] * P, (f[i1] * b[i2] + f[i2] * b[i1]) * P,
                f[i1] * f[i2] * P]
        dev = [torch.from_numpy(full(a)).to(ctx.device) for a in args]
        out = ctx.aps_table21cm_points(dd, dv, vv, self.kperpmin, self.kperpmax, self.kparmax, *dev)
        res = out.cpu().numpy().reshape(shape)
        return res if res.ndim else float(res)

    angular_powerspectrum = angular_powerspectrum_fft


def inverse_approx(f, x1, x2):
    """Spline of the inverse of a monotonic ``f`` sampled at 1000 points of [x1, x2] (corr.py:1053-1076)."""
    xa = np.linspace(x1, x2, 1000)
    return cs.Interpolater(f(xa), xa)
