#!/usr/bin/env python3
"""Golden vectors for the EoR variant of the 21cm model and the leftover flat-sky classes: outputs of the reference's
own ``EoR21cm`` (cora/signal/corr21cm.py:333-385: T_b of Santos et al. 2009, bias 3 - the class behind
``cora-makesky 21cm --eor``, cora/scripts/makesky.py:316-334), of ``skysim.clarray`` driven by it on a 150-200 MHz band
(cora/core/skysim.py:10-69), and of ``Cmb.powerspectrum`` / ``TestF.powerspectrum`` (cora/core/gaussianfield.py:159-191),
obtained by importing the reference in this container with the stand-ins of make_golden.py.

``Cmb()`` without arguments reads ``cora/core/ps_cmb2.dat``, which the reference's tree does not contain: the class is
run on a synthetic (l, l(l+1)C_l/2pi) table written by this script (the same table is stored in the vectors).

Commits data only: tests/golden/eor_vectors.npz.      python tests/golden/make_golden_eor.py   (~2 min: 21cm tables)
"""
import os
import sys
import tempfile

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, OUT)
import make_golden  # noqa: E402  (shims only)


def main():
    make_golden._install_shims()
    sys.path.insert(0, make_golden.REF)
    tmp = tempfile.mkdtemp(prefix="cora_golden_eor_")
    make_golden._build_cython(tmp)

    from cora.core import gaussianfield, skysim
    from cora.signal import corr21cm

    g = {}
    eor = corr21cm.EoR21cm()
    zs = np.array([5.5, 6.1, 7.0, 8.47, 10.0, 13.2])
    g["z"] = zs
    g["T_b"] = np.asarray(eor.T_b(zs), dtype=np.float64)
    g["bias_z"] = eor.bias_z(zs)
    g["omega_HI"] = np.float64(eor.omega_HI(zs))
    g["x_h"] = np.float64(eor.x_h(zs))
    g["prefactor"] = np.asarray(eor.prefactor(zs), dtype=np.float64)
    # the aps itself: the three numbers the reference's own test takes of Corr21cm (tests/test_corr.py:15-31), here of
    # EoR21cm in its band
    fa = np.linspace(150.0, 200.0, 16)
    g["fa"] = fa
    aps1 = eor.angular_powerspectrum(np.arange(1000), 180.0, 180.0)
    aps2 = eor.angular_powerspectrum(np.arange(1000)[:, None, None], fa[None, :, None], fa[None, None, :])
    g["aps_180_180"] = aps1
    g["aps2_samples"] = np.array([aps1.sum(), aps2[400, 10, 10], aps2[200, 3, 10], aps2[0, 5, 6], aps2[999, 15, 0]])
    g["aps2_l200"] = aps2[200]
    # clarray as Sky3d.getsky() drives it (frequencies in MHz, oversample 3), and the plain vectorised call
    f8 = 150.0 + (np.arange(8) + 0.5) * 6.25
    g["f8"] = f8
    for zr in (0, 1, 3):
        g["cla_eor_F8_l64_zromb%d" % zr] = skysim.clarray(eor.angular_powerspectrum, 64, f8.copy(), zromb=zr)
    # a wider, lower band (100-200 MHz: the largest |chi - chi'| the class meets in practice), zromb 3, explicit zwidth
    f6 = np.array([100.5, 113.0, 131.25, 150.0, 177.7, 199.5])
    g["f6"] = f6
    g["cla_eor_F6_l40_zromb2_zw1"] = skysim.clarray(eor.angular_powerspectrum, 40, f6.copy(), zromb=2, zwidth=1.0)

    # ---- Cmb / TestF (gaussianfield.py:159-191) -----------------------------------------------------------------
    l = np.arange(2.0, 2002.0)
    dl = 5000.0 * np.exp(-(((l - 220.0) / 300.0) ** 2)) + 1000.0 * (l / 1000.0) ** -1.5 + 50.0       # l(l+1)C_l/2pi, arbitrary smooth
    tab = np.stack([l, dl], axis=1)
    psfile = os.path.join(tmp, "ps_cmb_synth.dat")
    np.savetxt(psfile, tab)
    g["cmb_table"] = np.loadtxt(psfile)
    karr = np.stack(np.meshgrid(np.linspace(3.0, 900.0, 7), np.linspace(-400.0, 1200.0, 5), indexing="ij"), axis=-1)
    g["cmb_karray"] = karr
    g["cmb_ps_cambnorm"] = gaussianfield.Cmb(psfile=psfile, cambnorm=True).powerspectrum(karr)
    g["cmb_ps_plain"] = gaussianfield.Cmb(psfile=psfile, cambnorm=False).powerspectrum(karr)
    k3 = np.stack(np.meshgrid(np.linspace(-0.1, 0.1, 4), np.linspace(-500.0, 500.0, 5), np.linspace(0.0, 700.0, 3),
                              indexing="ij"), axis=-1)
    g["testf_karray"] = k3
    tf = gaussianfield.TestF.__new__(gaussianfield.TestF)
    g["testf_ps"] = tf.powerspectrum(k3)

    path = os.path.join(OUT, "eor_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, len(g), "arrays")
    for k in ("T_b", "aps2_samples"):
        print(k, g[k])


if __name__ == "__main__":
    main()
