#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box).  Nothing from the reference is copied into the repo: this script
imports the reference's Python from where it lies, compiles its three Cython
natives into a throw-away temp dir, runs the functions on the hot path with
fixed inputs and stores ONLY inputs + outputs as small .npz files.

Third-party modules the reference imports but this image lacks are replaced by
in-memory stand-ins that are part of THIS script (they are not reference code):

* ``caput.astro.constants`` - physical constants (values below; only ``degree``,
  ``c``, ``nu21`` are used on the path; ``mega_parsec`` cancels in chi(z)).
* ``caput.mpiarray`` - single-process stand-in (zeros/enumerate/allgather/
  redistribute/wrap) so ``skysim.mkfullsky(alms=True)`` runs.
* ``healpy`` - stub; never called (the SHT itself is NOT available here, see
  DESIGN.md "parity unpinned at the healpy boundary").

Usage:  python tests/golden/make_golden.py   (takes a few minutes: the 21cm
lookup table build is ~1-2 min of CPU)
"""
import importlib.machinery
import importlib.util
import math
import os
import subprocess
import sys
import sysconfig
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _install_shims():
    caput = types.ModuleType("caput")
    astro = types.ModuleType("caput.astro")
    const = types.ModuleType("caput.astro.constants")
    const.degree = 2 * math.pi / 360
    const.c = 299792458.0
    const.nu21 = 1420.40575177
    const.k_B = 1.3806503e-23
    const.mega_parsec = 3.08568025e22
    const.year = 365.25 * 86400.0
    const.mega_year = 1e6 * const.year
    const.G_n = 6.6742e-11
    const.a_rad = 4 * 5.670400e-8 / const.c
    astro.constants = const
    caput.astro = astro

    mpa = types.ModuleType("caput.mpiarray")

    class MPIArray(np.ndarray):
        @classmethod
        def wrap(cls, a, axis=0):
            o = a.view(cls)
            return o

        @property
        def local_array(self):
            return self.view(np.ndarray)

        @property
        def global_shape(self):
            return self.shape

        def enumerate(self, axis):
            return [(i, i) for i in range(self.shape[axis])]

        def allgather(self):
            return self.view(np.ndarray)

        def redistribute(self, axis):
            return self

    def zeros(shape, dtype=np.float64, axis=0):
        return MPIArray.wrap(np.zeros(shape, dtype=dtype), axis=axis)

    mpa.MPIArray = MPIArray
    mpa.zeros = zeros
    caput.mpiarray = mpa

    hp = types.ModuleType("healpy")
    hp.nside2npix = lambda n: 12 * n * n

    def _absent(*a, **k):
        raise ImportError("healpy is absent in this container")

    hp.alm2map = hp.map2alm = _absent

    sys.modules.update(
        {
            "caput": caput,
            "caput.astro": astro,
            "caput.astro.constants": const,
            "caput.mpiarray": mpa,
            "healpy": hp,
        }
    )


def _build_cython(tmp):
    """cythonize + gcc the reference's natives into `tmp`, register in sys.modules."""
    inc = sysconfig.get_paths()["include"]
    npinc = np.get_include()
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    for name in ("cubicspline", "bilinearmap"):
        c = os.path.join(tmp, name + ".c")
        so = os.path.join(tmp, name + suffix)
        subprocess.check_call(
            ["cython", "-3", os.path.join(REF, "cora/util", name + ".pyx"), "-o", c]
        )
        subprocess.check_call(
            ["gcc", "-O3", "-fno-math-errno", "-fno-trapping-math", "-fopenmp", "-shared",
             "-fPIC", "-I" + inc, "-I" + npinc, c, "-o", so]
        )
        full = "cora.util." + name
        loader = importlib.machinery.ExtensionFileLoader(full, so)
        spec = importlib.util.spec_from_file_location(full, so, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        sys.modules[full] = mod
        import cora.util  # noqa

        setattr(sys.modules["cora.util"], name, mod)


def main():
    _install_shims()
    sys.path.insert(0, REF)
    tmp = tempfile.mkdtemp(prefix="cora_golden_")
    _build_cython(tmp)

    from cora.core import maps, skysim
    from cora.foreground import galaxy, gaussianfg, pointsource  # noqa
    from cora.signal import corr21cm
    from cora.util import cosmology, cubicspline, hputil, nputil

    g = {}

    # ---- (i) KATs of tests/test_corr.py (values + Planck-2013 note) -------------
    kat = dict(
        sig_sum=1.5963772205823096e-09, sig_v1=8.986790805379046e-13,
        sig_v2=1.1939298801340165e-18, fg_sum=75.47681191093129,
        fg_v1=9.690708728692975e-06, fg_v2=0.00017630767166797886,
    )

    # ---- foreground models ------------------------------------------------------
    fg = galaxy.FullSkySynchrotron()
    fa = np.linspace(400.0, 800.0, 64)
    aps1 = fg.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    aps2 = fg.angular_powerspectrum(
        np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :]
    )
    g["fg_kat"] = np.array([aps1.sum(), aps2[400, 40, 40], aps2[200, 10, 40]])
    print("fg KAT", g["fg_kat"], [kat["fg_sum"], kat["fg_v1"], kat["fg_v2"]])

    ps = pointsource.CombinedPointSources._UnresolvedBackground()

    f8 = 400.0 + (np.arange(8) + 0.5) * 50.0
    g["f8"] = f8
    for name, model in (("syn", fg), ("ups", ps)):
        for zr in (0, 3):
            g["cla_%s_F8_l64_zromb%d" % (name, zr)] = skysim.clarray(
                model.angular_powerspectrum, 64, f8.copy(), zromb=zr
            )

    # ---- cosmology / spline -----------------------------------------------------
    cos = cosmology.Cosmology()
    zs = np.array([0.05, 0.3, 0.7755, 1.0, 1.367, 2.0, 2.551])
    g["cosmo_z"] = zs
    g["cosmo_chi"] = cos.comoving_distance(zs)
    g["cosmo_H"] = cos.H(zs)
    c13 = cosmology.Cosmology(omega_b=0.0483, omega_c=0.2589, omega_l=0.6914, H0=67.77)
    g["cosmo13_chi"] = c13.comoving_distance(zs)

    rs = np.random.RandomState(11)
    xk = np.sort(rs.uniform(0.0, 10.0, 12))
    yk = rs.standard_normal(12)
    sp = cubicspline.Interpolater(xk, yk)
    xe = np.concatenate([np.linspace(-2.0, 12.0, 57), xk])
    g["spl_xk"], g["spl_yk"], g["spl_xe"] = xk, yk, xe
    g["spl_ye"] = sp(xe)
    g["spl_y2"] = sp.data()[1]
    lsp = cubicspline.LogInterpolater(np.dstack((xk + 0.5, np.exp(yk)))[0])
    g["lspl_ye"] = lsp(np.abs(xe) + 0.25)
    # SinhInterpolater (cubicspline.pyx:290-345): data may be zero / negative; thresholds x_t, f_t
    ssp = cubicspline.SinhInterpolater(np.dstack((xk, yk))[0], 0.7, 0.05)
    g["sspl_ye"] = ssp(xe)

    # ---- 21cm model -------------------------------------------------------------
    cr = corr21cm.Corr21cm()
    aps1 = cr.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    aps2 = cr.angular_powerspectrum(
        np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :]
    )
    g["sig_kat_default"] = np.array([aps1.sum(), aps2[400, 40, 40], aps2[200, 10, 40]])
    g["sig_aps_800_800"] = aps1[[0, 1, 2, 10, 100, 500, 999]]
    g["sig_aps2_slices"] = aps2[[0, 1, 7, 200, 400, 999]][:, ::9, ::7]
    print("sig KAT (default cosmology)", g["sig_kat_default"])

    # table slices (iii)
    rows = np.array([0, 100, 250, 499])
    cols = np.concatenate([np.arange(64), [1000, 5000, 20000, 32767]])
    g["tab_rows"], g["tab_cols"] = rows, cols
    for nm in ("dd", "dv", "vv"):
        g["tab_" + nm] = getattr(cr, "_aps_" + nm)[np.ix_(rows, cols)]
    kk = np.array([1e-6, 5e-5, 1e-4, 1.2345e-3, 0.1, 1.0, 7.5, 40.0, 44.0, 100.0])
    g["ps_k"] = kk
    g["ps_vv"] = cr.ps_vv(kk)
    zt = np.array([0.8, 1.0, 1.5, 2.0, 2.5])
    g["m21_z"] = zt
    g["m21_Tb"] = cr.T_b(zt)
    g["m21_D"] = cr.growth_factor(zt)
    g["m21_f"] = cr.growth_rate(zt)

    for zr in (0, 1, 3):
        g["cla_21cm_F8_l64_zromb%d" % zr] = skysim.clarray(
            cr.angular_powerspectrum, 64, f8.copy(), zromb=zr
        )
    # narrow channels (the regime of BASELINE cfg 3: 1.5625 MHz channels)
    f6 = 600.0 + (np.arange(6) + 0.5) * 1.5625
    g["f6"] = f6
    g["cla_21cm_F6n_l96_zromb3"] = skysim.clarray(cr.angular_powerspectrum, 96, f6.copy(), zromb=3)
    g["cla_21cm_F6n_l96_zromb2_zw"] = skysim.clarray(
        cr.angular_powerspectrum, 96, f6.copy(), zromb=2, zwidth=1.0
    )

    # Planck-2013 instance reproduces the reference's own KATs
    cr13 = corr21cm.Corr21cm()
    cr13.cosmology = c13
    a1 = cr13.angular_powerspectrum(np.arange(1000), 800.0, 800.0)
    a2 = cr13.angular_powerspectrum(
        np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :]
    )
    g["sig_kat_planck13"] = np.array([a1.sum(), a2[400, 40, 40], a2[200, 10, 40]])
    print("sig KAT (Planck13)", g["sig_kat_planck13"], [kat["sig_sum"], kat["sig_v1"], kat["sig_v2"]])
    g["kat_test_corr"] = np.array([kat[k] for k in ("sig_sum", "sig_v1", "sig_v2", "fg_sum", "fg_v1", "fg_v2")])

    # ---- (iv) matrix_root_manynull, both branches -------------------------------
    A = g["cla_21cm_F8_l64_zromb3"][10]
    g["root_well_in"] = A
    g["root_well_out"] = nputil.matrix_root_manynull(A, truncate=False)
    B = g["cla_syn_F8_l64_zromb0"][10]
    g["root_sing_in"] = B
    try:
        import scipy.linalg as la

        la.cholesky(B, lower=True)
        g["root_sing_branch"] = np.array(0)
    except Exception:
        g["root_sing_branch"] = np.array(1)
    g["root_sing_out"] = nputil.matrix_root_manynull(B, truncate=False)
    r, npos = nputil.matrix_root_manynull(B)
    g["root_sing_trunc"], g["root_sing_npos"] = r, np.array(npos)
    Z = np.zeros((5, 5))
    g["root_zero_out"] = nputil.matrix_root_manynull(Z, truncate=False)
    # rank-deficient PSD with a clear eigh branch
    rs = np.random.RandomState(5)
    V = rs.standard_normal((6, 3))
    R = V @ V.T
    g["root_rank3_in"] = R
    g["root_rank3_out"] = nputil.matrix_root_manynull(R, truncate=False)

    # ---- (v) complex_std_normal order --------------------------------------------
    g["csn_3x5_seed7"] = nputil.complex_std_normal((3, 5), rng=np.random.default_rng(7))

    # ---- (vi) mkfullsky alms ------------------------------------------------------
    f4 = 400.0 + (np.arange(4) + 0.5) * 100.0
    g["f4"] = f4
    cla4 = skysim.clarray(cr.angular_powerspectrum, 16, f4.copy(), zromb=1)
    g["cla_21cm_F4_l16_zromb1"] = cla4
    g["alm_21cm_F4_l16_seed3"] = skysim.mkfullsky(cla4, 8, alms=True, rng=np.random.default_rng(3))
    g["alm_21cm_F8_l64_seed4"] = skysim.mkfullsky(
        g["cla_21cm_F8_l64_zromb3"], 32, alms=True, rng=np.random.default_rng(4)
    )
    g["alm_syn_F8_l64_seed5"] = skysim.mkfullsky(
        g["cla_syn_F8_l64_zromb0"], 32, alms=True, rng=np.random.default_rng(5)
    )
    np.random.seed(1234)
    g["alm_21cm_F4_l16_legacy1234"] = skysim.mkfullsky(cla4, 8, alms=True)

    # ---- (vii) pack_alm ----------------------------------------------------------
    lab = np.zeros((6, 6), dtype=np.complex128)
    for l in range(6):
        for m in range(l + 1):
            lab[l, m] = 100 * l + m + 1j * (l - m)
    g["pack_in"] = lab
    g["pack_out"] = hputil.pack_alm(lab)
    g["unpack_out"] = hputil.unpack_alm(g["pack_out"], 5)
    g["nside_for_lmax"] = np.array([[l, hputil.nside_for_lmax(l)] for l in (1, 2, 5, 95, 96, 383, 384, 2048)])

    # ---- (viii) Map3d frequencies --------------------------------------------------
    m3 = maps.Map3d()
    m3.nu_lower, m3.nu_upper, m3.nu_num = 400.0, 800.0, 32
    g["map3d_freq_400_800_32"] = m3.frequencies
    m3d = maps.Map3d()
    g["map3d_freq_default"] = m3d.frequencies

    for k, v in g.items():
        g[k] = np.asarray(v)
    path = os.path.join(OUT, "reference_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path) / 1024, "KiB,", len(g), "arrays")


if __name__ == "__main__":
    main()
