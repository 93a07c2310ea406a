#!/usr/bin/env python3
"""Where the seeded step's time goes: wall time and every stage timer of cold cfg-3 steps with the Philox stream, numpy's
PCG64 stream and numpy's legacy stream (tools/seeded_probe.py [nrep])."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cora_amd import _lib
from cora_amd.parallel import SkyShard
from cora_amd.signal import corr21cm
ctx = _lib.get_context()
F, nside, lmax = 256, 1024, 2048
freq = 400.0 + (np.arange(F) + 0.5) * (400.0 / F)
sh = SkyShard(corr21cm.Corr21cm(), freq, nside, lmax, zromb=3, ctx=ctx)
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
names = ("clarray", "factor", "zig_seek", "zig_count", "zig_scan", "zig_emit", "zig_chain", "mt_jump", "mt_count", "mt_emit", "draw", "legendre", "ringfft")
rng = np.random.default_rng(1)
np.random.seed(2)
st0 = np.random.default_rng(9).bit_generator.state["state"]
gbuf = ctx.empty((2 * F * ((lmax + 1) * (lmax + 2) // 2),))


def philox_after_dummy_stream(i):
    """the Philox step with a whole-stream generation (6.5 ms of integer VALU, 8.6 GB written) in front of the draw: does K4
    slow down behind ANY longer prologue, or only behind the stream-mode K3?"""
    fac = sh.factors()
    ctx.normals_pcg64(st0["state"], st0["inc"], gbuf.numel(), out=gbuf)
    sh.realise(200 + i, fac)


def pcg64_then_idle(i):
    """the seeded step with the queue drained and 20 ms of idle between the draw and the synthesis"""
    fac = sh.factors()
    _, fin = sh.draw_numpy(rng, fac, defer=True)
    fin()
    time.sleep(0.02)
    ctx.alm2map(sh.alm_buf, sh.nside, sh.lmax, sh.nnu, out=sh.maps_buf)


def early(g):
    """the generator started AHEAD of K1 / K2 (corahip_draw_alm_numpy_prepare): its passes run beside them"""
    def f(i):
        prep = sh.prepare_numpy(g)
        sh.realise_numpy(g, sh.factors(), prepared=prep)
    return f


modes = {"philox": lambda i: sh.realise(100 + i, sh.factors()), "pcg64": lambda i: sh.realise_numpy(rng, sh.factors()),
         "pcg64, generator ahead of K1": early(rng),
         "legacy": lambda i: sh.realise_numpy(None, sh.factors()), "legacy, generator ahead of K1": early(None)}
if os.environ.get("PROBE_MORE"):
    modes.update({"philox+dummy stream": philox_after_dummy_stream, "pcg64, idle before K4": pcg64_then_idle})
if os.environ.get("PROBE_RINGS"):       # the seeded step for several ring sizes (MB): does K4's first-run penalty follow the ring?
    def with_ring(mb):
        def f(i):
            os.environ["CORAHIP_RING_MB"] = str(mb)
            sh.realise_numpy(rng, sh.factors())
        return f
    def with_ring_ahead(mb, g):
        def f(i):
            os.environ["CORAHIP_RING_MB"] = str(mb)
            early(g)(i)
        return f
    modes = {"philox": modes["philox"]}
    for mb in os.environ["PROBE_RINGS"].split(","):
        modes["pcg64 ring %s MB" % mb] = with_ring(int(mb))
        modes["pcg64 ring %s MB, generator ahead" % mb] = with_ring_ahead(int(mb), rng)
        modes["legacy ring %s MB, generator ahead" % mb] = with_ring_ahead(int(mb), None)
for tag, fn in modes.items():
    fn(0); fn(1)
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.time()
    for i in range(nrep):
        fn(i)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / nrep * 1e3
    ctx.profile_enable(False)
    st = {n: round(ctx.profile_get(n)[0] / nrep, 3) for n in names if ctx.profile_get(n)[1]}
    serial = sum(v for k, v in st.items() if k in ("clarray", "factor", "draw", "legendre", "ringfft"))
    print("%-22s %.2f ms per step; stages %s; clarray+factor+draw+legendre+ringfft = %.2f" % (tag, ms, st, serial))
