#!/usr/bin/env python3
"""Headline benchmark: sky-maps/s of cora's Gaussian-sky realisation path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2], the config the metric is quoted on): Corr21cm, 256
channels 400-800 MHz, nside = 1024, lmax = 2048, oversample (Romberg order) 3.

One STEP = one full `Sky3d.getsky()`-equivalent realisation with every input already in
HBM: K1 C_l(nu,nu') integration -> K2 per-l factor -> K3 correlated
draw with device Philox normals generated in registers -> K4 Legendre MFMA contraction -> K5 ring FFT -> 256 RING maps in HBM.
Nothing is cached between steps except what the reference itself caches per model
instance (the three DCT lookup tables) and the geometry plan.

N > 1 (frequency sharding, north_star): K1 is sharded over channel pairs, one small RCCL
all-to-all turns pair shards into l shards for K2, a second all-to-all hands every rank the
factor rows of its own channels; every rank generates the same global normal stream and
synthesises F/N channels.  Total work is fixed -> "scaling": "strong".

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (components [(model, zromb)], F, nu_lower, nu_upper, nside, lmax)
    "cfg3": ([("21cm", 3)], 256, 400.0, 800.0, 1024, 2048),
    "cfg2": ([("synchrotron", 3)], 32, 400.0, 800.0, 256, 512),
    # BASELINE configs[3]: three independent Gaussian components summed on one 512-channel grid
    "cfg4": ([("21cm", 3), ("synchrotron", 0), ("pointsource", 0)], 512, 400.0, 800.0, 1024, 2048),
    # BASELINE configs[4]: 412 GB of maps - an 8-GPU configuration; with fewer ranks the work of ONE of eight
    # ranks is timed (--emulate-shard 8 is implied) and the line says so
    "cfg5": ([("21cm", 3)], 1024, 400.0, 800.0, 2048, 4096),
    "tiny": ([("21cm", 1)], 16, 600.0, 625.0, 64, 128),
    "tiny3": ([("21cm", 1), ("synchrotron", 0), ("pointsource", 0)], 16, 600.0, 625.0, 64, 128),
    "tiny32": ([("21cm", 1)], 32, 600.0, 650.0, 64, 128),          # (32 channels: the folded assignment at 2 and 4 ranks)
    "tiny32s": ([("21cm", 1), ("synchrotron", 0), ("pointsource", 0)], 32, 600.0, 650.0, 64, 128),
}
MIN_RANKS = {"cfg5": 8}        # workloads that do not fit fewer GPUs: emulate one rank's share instead

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor spec; profiles/mfma_f64_probe_r01.txt measures 77.4
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: L2 (per XCD, aggregate) ~34.5 TB/s


def build_model(name):
    if name == "21cm":
        from cora_amd.signal import corr21cm

        return corr21cm.Corr21cm()
    if name == "pointsource":
        from cora_amd.foreground import pointsource

        return pointsource.CombinedPointSources._UnresolvedBackground()
    from cora_amd.foreground import galaxy

    return galaxy.FullSkySynchrotron()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain shell: start the N ranks as a CHILD torch.distributed.run job - before
    this process has imported torch or touched HIP (a process that initialised the GPU must never exec another
    program) - relay rank 0's JSON line and return the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if line:
        print(line[-1], flush=True)
    return p.returncode if p.returncode or line else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--warm", action="store_true", help="time the warm path only (factors cached)")
    ap.add_argument("--no-host-delivered", action="store_true", help="skip the PCIe-inclusive (numpy-returning) leg")
    ap.add_argument("--no-seeded-modes", action="store_true", help="skip the numpy-seeded / legacy-RNG legs")
    # test hooks (not used by the driver): run N ranks on ONE GPU over gloo and print per-channel checksums
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--checksum", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (process group, exchanges) even with 1 rank")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="time the share of EVERY rank of an N-rank job in turn on this GPU (no comm); the line is the critical rank's")
    ap.add_argument("--emulate-rank", type=int, default=None, help="with --emulate-shard: time this rank's share only")
    ap.add_argument("--fold", action="store_true",
                    help="folded channel assignment: rank r owns chunks r and 2N-1-r of 2N (balances the triangular draw)")
    ap.add_argument("--sum-mode", default="joint", choices=["joint", "separate"],
                    help="multi-component workloads (cfg4): one factorisation + draw of the summed covariance, or one per component")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line (the JSON): native libraries print banners there (RCCL's version block
    # at communicator creation), so fd 1 is pointed at stderr for the run and the result goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    from cora_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:       # (one rank forced through the group: any free port, runs may follow each other closely)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # a rendezvous or a collective that does not complete ends the job - non-zero exit, the stage named - instead of
        # hanging it: torch's watchdog through `timeout`, and the stage guard of cora_amd.parallel (CORA_DIST_TIMEOUT_S)
        from cora_amd import parallel as _par

        _par.start_watchdog(rank, world)
        with _par.exchange_stage("init_process_group (%s rendezvous at %s:%s)" % (args.dist_backend, os.environ["MASTER_ADDR"],
                                                                                  os.environ["MASTER_PORT"])):
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=_par.dist_timeout())
            else:
                dist.init_process_group(args.dist_backend, timeout=_par.dist_timeout())

    comps, F, nu_lo, nu_hi, nside, lmax = WORKLOADS[args.workload]
    model_name = "+".join(c[0] for c in comps)
    zromb = comps[0][1]
    L = lmax + 1
    nalm = L * (L + 1) // 2
    npix = 12 * nside * nside
    from cora_amd.parallel import SkyShard, SkySum

    ranks_seen = dist.get_world_size() if dist is not None else 1
    # which GPU every rank really sits on: (rank, device index, name, uuid / PCI id) gathered to rank 0 - the line then shows
    # N DISTINCT devices, or says that ranks share one (--same-device test hook)
    def device_id_string():
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = getattr(pr, "uuid", None)
        pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        return "rank %d: cuda:%d %s uuid=%s pci=%s" % (rank, torch.cuda.current_device(), pr.name, ident, pci)

    rank_devices = [device_id_string()]
    if dist is not None:
        gathered = [None] * ranks_seen
        with _par.exchange_stage("all_gather_object of the ranks' device ids"):
            dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered
    distinct_devices = len({d.split(" ", 3)[3] for d in rank_devices})
    if world < MIN_RANKS.get(args.workload, 1) and args.emulate_shard <= 1:
        if world != 1:
            raise SystemExit("%s needs %d ranks (or 1 rank emulating one of them)" % (args.workload, MIN_RANKS[args.workload]))
        args.emulate_shard = MIN_RANKS[args.workload]
    ctx = _lib.get_context(local_rank)
    freq = nu_lo + (np.arange(F) + 0.5) * ((nu_hi - nu_lo) / F)

    seed_box = [1000]

    def barrier():
        if dist is not None:
            with _par.exchange_stage("barrier + device synchronisation (everything enqueued before it)", sync=False):
                dist.barrier()
                torch.cuda.synchronize()
        else:
            torch.cuda.synchronize()

    def measure(emulate_rank):
        """Build the rank's pipeline object (untimed: everything the timed region reads is put in HBM - tables, plan,
        buffers) and time `steps` steps after `warmup`; returns the shard, its step function, wall time, stage times."""
        t_setup = time.time()
        kw = dict(rank=rank, world=world, ctx=ctx, distributed=multi, emulate_world=args.emulate_shard, fold=args.fold,
                  emulate_rank=emulate_rank)
        if len(comps) == 1:
            shard = SkyShard(build_model(comps[0][0]), freq, nside, lmax, zromb=zromb, **kw)
        else:
            shard = SkySum([(build_model(m), z) for m, z in comps], freq, nside, lmax, mode=args.sum_mode, **kw)
        torch.cuda.synchronize()
        t_setup = time.time() - t_setup
        cold_factors = shard.factors
        cached = {}

        def step():
            if args.warm:
                if "f" not in cached:
                    cached["f"] = cold_factors()
                fac = cached["f"]
            else:
                fac = cold_factors()     # K1 (+ all-to-all) -> K2 (+ all-to-all)
            seed_box[0] += 1
            # K3 (device Philox normals generated inside the draw kernel: no 8.6 GB normal buffer) -> K4 -> K5
            shard.realise(seed_box[0], fac)

        for _ in range(args.warmup):
            step()
        barrier()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.time()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.time() - t0
        ctx.profile_enable(False)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        stages = {}
        for name in ("clarray", "factor", "normals", "draw", "legendre", "ringfft"):
            ms, n = ctx.profile_get(name)
            if n:
                stages[name] = {"ms_per_launch": ms / n, "launches": n, "ms_per_step": ms / args.steps}
        return dict(shard=shard, dt=dt, stages=stages, t_setup=t_setup, rank=emulate_rank)

    # --emulate-shard N: the share of every rank of an N-rank job, one after the other on this GPU (no exchanges); the
    # line below is the CRITICAL rank's (the slowest), `emulated_ranks` lists them all
    emulated_ranks = None
    if args.emulate_shard > 1 and args.emulate_rank is None:
        runs = []
        for r in range(args.emulate_shard):
            m = measure(r)
            runs.append({"rank": r, "ms_per_step": m["dt"] / args.steps * 1e3, "channels": [list(c) for c in m["shard"].chunks],
                         "stages_ms": {k: round(v["ms_per_step"], 3) for k, v in m["stages"].items()}})
            keep = m
            if r + 1 < args.emulate_shard:
                del m, keep
                torch.cuda.empty_cache()
        crit = max(runs, key=lambda x: x["ms_per_step"])
        emulated_ranks = {"per_rank": runs, "per_rank_ms": [round(x["ms_per_step"], 3) for x in runs],
                          "critical_rank": crit["rank"], "critical_path_ms": crit["ms_per_step"],
                          "job_maps_per_s_no_exchange": F / (crit["ms_per_step"] * 1e-3),
                          "fold": bool(args.fold),
                          "note": "one GPU doing the share of each of %d ranks in turn, exchanges excluded: unmeasured on "
                                  "multi-GPU hardware" % args.emulate_shard}
        emulated_ranks["exchange_model"] = exchange_model(comps, F, L, args.emulate_shard, args.sum_mode, crit["ms_per_step"])
        if crit["rank"] != args.emulate_shard - 1:
            # (both names dropped: two resident rank pipelines would be an OOM risk at cfg 5)
            del m, keep
            torch.cuda.empty_cache()
            keep = measure(crit["rank"])
            # one number per rank in the line: the critical rank's entry IS the run the headline is taken from
            redo = {"rank": crit["rank"], "ms_per_step": keep["dt"] / args.steps * 1e3, "channels": [list(c) for c in keep["shard"].chunks],
                    "stages_ms": {k: round(v["ms_per_step"], 3) for k, v in keep["stages"].items()}}
            runs[crit["rank"]] = redo
            emulated_ranks["per_rank_ms"] = [round(x["ms_per_step"], 3) for x in runs]
            emulated_ranks["critical_path_ms"] = redo["ms_per_step"]
            emulated_ranks["job_maps_per_s_no_exchange"] = F / (redo["ms_per_step"] * 1e-3)
            emulated_ranks["exchange_model"] = exchange_model(comps, F, L, args.emulate_shard, args.sum_mode, redo["ms_per_step"])
        m = keep
    else:
        m = measure(args.emulate_rank)
    shard, dt, stages, t_setup = m["shard"], m["dt"], m["stages"], m["t_setup"]
    nnu, nu0 = shard.nnu, shard.nu0
    maps_buf = shard.maps_buf
    cold_factors = shard.factors

    # warm path (factors cached: draw + synthesis only - what repeated seeds amortise, cora/signal/lss.py:424-478),
    # measured after the timed region and reported next to the headline (cold) number
    warm_ms = None
    if not args.warm and args.emulate_shard <= 1:
        fac = cold_factors()
        nw = max(1, min(3, args.steps))
        barrier()
        tw = time.time()
        for _ in range(nw):
            seed_box[0] += 1
            shard.realise(seed_box[0], fac)
        barrier()
        warm_ms = (time.time() - tw) / nw * 1e3
        del fac

    # the REFERENCE's seeded call (rng = numpy Generator, cora/core/skysim.py:72,120; cora/signal/lss.py:449-450): the same
    # cold step with numpy's PCG64 + ziggurat stream generated on the device (bit-identical to numpy, the generator left
    # where numpy would leave it) instead of the library's Philox stream; and the legacy mode (rng=None: numpy's global
    # MT19937 + polar method, what Sky3d.getsky() draws from), generated on the device as well
    # Since round 5 the stream is generated range of multipoles by range (no 16 F nalm byte buffer), so the legs run on
    # every line: multi-component workloads (separate: the components draw one after the other from the one generator),
    # emulated shards and real ranks (every rank consumes the whole stream for its own rows of the factors).
    seeded_numpy = legacy_rng = None
    if not (args.no_seeded_modes or args.checksum or args.warm):
        seeded_numpy, legacy_rng = seeded_modes(ctx, shard, cold_factors, F, lmax, barrier, args.steps, len(comps)
                                                if args.sum_mode == "separate" else 1, nnu)

    # host-delivered rate: the reference's own signature returns numpy arrays (cora/core/skysim.py:130-136), i.e. every
    # realisation crosses PCIe.  skysim.mkfullsky_stream double-buffers that copy (pinned memory, copy stream) behind the
    # next realisation; measured after the timed region, reported next to the HBM-resident headline, never as `value`.
    host_delivered = None
    if rank == 0 and world == 1 and args.emulate_shard <= 1 and len(comps) == 1 and not args.no_host_delivered:
        try:
            host_delivered = host_delivered_rate(ctx, shard, nside, F, npix)
        except Exception as e:      # (e.g. a box that cannot page-lock 80 GB): the leg is extra, the headline must not die with it
            host_delivered = {"skipped": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()

    # the literal drop-in call: what a cora user sees from ONE `skysim.mkfullsky(C_numpy, nside, rng=default_rng(s))`
    # (cora/core/skysim.py:72-136; the second half of Sky3d.getsky(), cora/core/maps.py:227-237) - H2D of the 1.07 GB
    # C_l, factor + seeded draw + synthesis, D2H of the 25.8 GB of maps into a fresh numpy array.  Outside the timed region.
    drop_in = None
    if rank == 0 and world == 1 and args.emulate_shard <= 1 and len(comps) == 1 and not (args.no_host_delivered or args.checksum or args.warm):
        try:
            drop_in = drop_in_call(build_model(comps[0][0]), freq, nside, lmax, zromb, F, npix)
        except Exception as e:      # (extra leg: the headline must not die with it)
            drop_in = {"skipped": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()

    # sanity: the maps of the last step are finite and have the expected variance scale
    chk = float(maps_buf[0, ::4097].std().item())
    assert os.environ.get("CORAHIP_LIB") or (np.isfinite(chk) and chk > 0)  # (diagnostic builds skip the check)

    if args.checksum:
        # per-channel (mean, rms) of the last realisation, gathered in channel order: identical for any N
        st = torch.stack([maps_buf.mean(dim=1), (maps_buf**2).mean(dim=1).sqrt()], dim=1).cpu()
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, (shard.channels, st))
            full = torch.empty((F, 2), dtype=st.dtype)
            for chn, pst in parts:          # (a folded shard holds two chunks: every row goes to its channel)
                full[torch.as_tensor(chn)] = pst
            st = full
        if rank == 0:
            import hashlib

            print("CHECKSUM", hashlib.sha1(st.numpy().round(decimals=14).tobytes()).hexdigest(), float(st[:, 1].sum()), file=sys.stderr)
        # the reference's seeded call and its rng=None call: EVERY PIXEL of every channel hashed (sha1 of the raw
        # bytes per channel, gathered in channel order) - a sharded run must reproduce the single-rank maps bit for bit
        import hashlib

        for tag, mk in (("CHECKSUM_SEEDED", lambda: np.random.default_rng(77)), ("CHECKSUM_LEGACY", lambda: np.random.seed(78))):
            g = mk()
            # (the generator started ahead of the factors, as the timed seeded legs run it - where the shard offers it)
            prep = shard.prepare_numpy(g) if hasattr(shard, "prepare_numpy") else None
            try:
                fac = cold_factors()
            except BaseException:
                if prep is not None:
                    prep.abort()
                raise
            m = (shard.realise_numpy(g, fac, prepared=prep) if prep is not None else shard.realise_numpy(g, fac)).cpu().numpy()
            hs = [hashlib.sha1(m[i].tobytes()).hexdigest() for i in range(m.shape[0])]
            if dist is not None:
                parts = [None] * world
                dist.all_gather_object(parts, (shard.channels, hs))
                byc = {int(c): h for chn, ph in parts for c, h in zip(chn, ph)}
                hs = [byc[c] for c in range(F)]
            if rank == 0:
                print(tag, hashlib.sha1("".join(hs).encode()).hexdigest(), file=sys.stderr)

    result = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        emu = args.emulate_shard if args.emulate_shard > 1 else 0
        # emulated shard: only the nnu maps of that one rank were made
        value = (nnu if emu else F) * args.steps / dt
        # dominant kernel: K4 Legendre contraction, FP64 MFMA bound.  `achieved` prices the flops the kernel EXECUTES:
        # the plan counts, from its first-contributing-l tables, the v_mfma_f64_16x16x4_f64 instructions (2048 flop
        # each) this launch issues (corahip_sht_plan_k4_mfma_count - the number SQ_INSTS_VALU_MFMA_F64 reads, checked
        # against the committed PMC profile by tests/test_gpu_fullsize.py).  The algorithmic 8 nside nalm F of SURVEY
        # 8(d) counts Legendre terms below the plan's 2^-70 cut that nobody has to compute (pixel error bound
        # 2 sum |a_lm| 2^-70): it is reported as `algorithmic_tflops`, never as a fraction of peak.
        leg = stages.get("legendre", {"ms_per_launch": float("nan")})
        flops_alg = 8.0 * nside * nalm * nnu           # per launch, SURVEY 8(d) / DESIGN.md
        mfma_insts = ctx.k4_mfma_count(nside, lmax, nnu)
        flops_exec = 2048.0 * mfma_insts
        ach = flops_exec / (leg["ms_per_launch"] * 1e-3) / 1e12
        alg_bytes = 8.0 * npix * nnu + 32.0 * nalm * nnu + 8.0 * L * F * F   # warm path, SURVEY 8(d)
        # HBM bytes of the dominant kernel from the PMC passes of the same command (collected separately
        # with rocprofv3 --pmc and committed under profiles/; bench.py cannot read counters itself)
        traffic = traffic_source = pmc_mfma = None
        for pmc_name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
            pmc_file = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc_file) and world == 1 and not emu and args.workload == "cfg3":
                pmc = json.load(open(pmc_file)).get("kernels", {})
                k4 = next((v for k, v in pmc.items() if k.startswith("legendre_kernel")), None)
                if k4:
                    traffic = k4["hbm_bytes_per_launch"]
                    traffic_source = ("profiles/%s (rocprofv3 --pmc passes of this command, 2 x FETCH_SIZE + WRITE_SIZE; "
                                      "not measured in this run)" % pmc_name)
                    pmc_mfma = k4.get("SQ_INSTS_VALU_MFMA_F64_per_launch")
                    break
        result = {
            "metric": "sky-maps/sec (nside=%d, lmax=%d, %d freq)" % (nside, lmax, F),
            "value": value,
            "unit": "maps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "rank_devices": rank_devices,
            "distinct_devices": distinct_devices,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %s, %d channels %g-%g MHz, nside=%d, lmax=%d, zromb=%s, %s path, device Philox normals%s%s"
                            % (args.workload, model_name, F, nu_lo, nu_hi, nside, lmax, "/".join(str(c[1]) for c in comps),
                               "warm (cached factors)" if args.warm else "cold (C_l integration + factor + draw + synthesis)",
                               "" if len(comps) == 1 else (", components summed as ONE Gaussian field (covariances added, one factorisation + one draw)"
                                                           if args.sum_mode == "joint" else ", one draw per component, a_lm added"),
                               (" - ONE GPU doing the share of rank %s of %d ranks (%d channels%s, no exchanges), the slowest of "
                                "the %d shares timed in turn: value counts those maps only"
                                % (m["rank"] if m["rank"] is not None else emu - 1, emu, nnu, ", folded" if args.fold else "", emu))
                               if emu else ""),
                "sum_mode": (args.sum_mode + (": ONE draw of the summed covariance - same distribution as, not the same numbers per seed as, "
                                              "the reference's one getsky() per component" if args.sum_mode == "joint"
                                              else ": one factorisation + draw per component, as the reference's per-component getsky() calls"))
                            if len(comps) > 1 else None,
                "parallelism": "freq-shard x%d (pair-sharded C_l -> all-to-all -> l-sharded factor -> all-to-all of factor row blocks)" % world if world > 1 else "single GPU",
                "channel_assignment": ("folded: rank r owns chunks r and 2N-1-r of 2N" if args.fold else "contiguous blocks (the reference's split)"),
                "emulated_ranks": emulated_ranks,
                "realisations_per_s": args.steps / dt,
                "warm_path": None if warm_ms is None else {"ms_per_step": warm_ms, "maps_per_s": F / (warm_ms * 1e-3)},
                "setup_s": t_setup,
                "host_delivered": host_delivered,
                "drop_in_call": drop_in,
                "seeded_numpy_mode": seeded_numpy,
                "legacy_rng_mode": legacy_rng,
            },
            "stages_ms": {k: round(v["ms_per_step"], 3) for k, v in stages.items()},
            "roofline": {
                "bound": "mfma",
                "kernel": "legendre_kernel (K4)",
                "achieved": ach,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "note": "achieved = EXECUTED flops / measured time: 2048 x the FP64 MFMA instructions this launch issues, counted "
                        "in this run from the plan's first-contributing-l tables (= SQ_INSTS_VALU_MFMA_F64); Legendre terms "
                        "below 2^-70 are skipped (pixel error <= 2 sum |a_lm| 2^-70), so the algorithmic count of SURVEY 8(d) is larger",
                "executed_mfma_instructions_per_launch": mfma_insts,
                "executed_flops_per_launch": flops_exec,
                "pmc_mfma_instructions_per_launch": pmc_mfma,
                "algorithmic_flops_per_launch": flops_alg,
                "algorithmic_tflops": flops_alg / (leg["ms_per_launch"] * 1e-3) / 1e12,
                "algorithmic_bytes": 16.0 * nalm * nnu + 16.0 * (4 * nside - 1) * L * nnu,
            },
            "roofline_k5": {
                "bound": "hbm",
                "kernel": "ringfft_direct_ct / ringfft_blu_ct (K5)",
                "achieved": (16.0 * (4 * nside - 1) * L * nnu + 8.0 * npix * nnu) / (stages.get("ringfft", {"ms_per_launch": float("nan")})["ms_per_launch"] * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (16.0 * (4 * nside - 1) * L * nnu + 8.0 * npix * nnu) / (stages.get("ringfft", {"ms_per_launch": float("nan")})["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            # every stage against the roof that bounds it (algorithmic work of SURVEY 8(d) / DESIGN section 3 per launch)
            "stage_rooflines": stage_rooflines(stages, comps, F, nside, lmax, nnu, nu0, max(world, emu, 1), args.sum_mode,
                                               flops_exec, shard.channels if shard.folded else None),
            "hbm_roofline_whole_step": {
                "algorithmic_GB": alg_bytes / 1e9,
                "achieved_GBs": alg_bytes / 1e9 / (ms_step * 1e-3),
                "peak_GBs": HBM_PEAK_GBS,
                "frac": alg_bytes / 1e9 / (ms_step * 1e-3) / HBM_PEAK_GBS,
            },
        }
        if not args.no_cpu_baseline and world == 1 and len(comps) == 1 and not emu:
            result["cpu_baseline"] = cpu_baseline(model_name, F, freq, nside, lmax, zromb)
            result["config"]["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    if dist is not None:
        dist.destroy_process_group()
    return result


def exchange_model(comps, F, L, N, sum_mode, critical_ms):
    """A MODELLED cost of the two exchange steps of an N-rank step (cora_amd/parallel.py; the counterpart of the
    reference's allgather / redistribute, cora/core/skysim.py:97-134), to stand next to the exchange-free emulated
    critical path - nothing here is measured.  Bytes one rank SENDS:
      all-to-all #1  its pair shard of every table component's C_l, cut by destination multipole shard:
                     8 (F (F + 1) / 2 / N) L (N - 1) / N per component that is integrated per pair
      all-to-all #2  the factor row blocks of the other ranks' channels: 8 (L / N) F^2 (N - 1) / N per factored component
                     (joint mode: the one summed covariance)
    xGMI is point to point, one link per peer (MI355X: 7 links of ~153 GB/s peak each way): the N - 1 messages of a rank go
    out in parallel, each over its own link, so the time of an exchange is (bytes sent / (N - 1)) / link rate + a launch
    latency.  Two link rates bracket it: 50 GB/s (small messages, protocol overhead) and 150 GB/s (peak); latency 50 us per
    collective (RCCL launch + synchronisation), all stated in the record."""
    kinds = ["table" if m == "21cm" else "separable" for m, _ in comps]
    joint = len(comps) > 1 and sum_mode == "joint"
    ntab = sum(1 for k in kinds if k == "table")
    nfac = 1 if joint else ntab           # components factored per l (a separable one is factored once, on every rank)
    a2a1 = ntab * 8.0 * (F * (F + 1) / 2.0 / N) * L * (N - 1) / N
    a2a2 = nfac * 8.0 * (L / float(N)) * F * F * (N - 1) / N
    lat_ms = 0.05
    out = {"modelled": True, "bytes_sent_per_rank": {"all_to_all_pair_slabs": a2a1, "all_to_all_factor_rows": a2a2},
           "links_per_rank": N - 1, "link_GBs_bracket": [50.0, 150.0], "latency_ms_per_collective": lat_ms,
           "note": "modelled, not measured: (bytes sent / (N - 1) links) / link rate + latency per exchange; RCCL over xGMI has never "
                   "run with more than one rank on this pool"}
    ms = []
    for rate in (50.0, 150.0):
        t = sum((b / max(N - 1, 1)) / (rate * 1e9) * 1e3 + (lat_ms if b else 0.0) for b in (a2a1, a2a2))
        ms.append(t)
    out["exchange_ms_bracket"] = [ms[1], ms[0]]
    out["critical_path_ms_with_exchange_bracket"] = [critical_ms + ms[1], critical_ms + ms[0]]
    return out


def stage_work(comps, F, nside, lmax, nnu, nu0, nranks, sum_mode, legendre_executed_flops, channels=None):
    """Algorithmic work of ONE RANK per step, per stage: {stage: (bound, flops, bytes, note)} (DESIGN
    section 3).  `comps` = [(model name, zromb)], `nranks` = ranks the work is cut over (the world size, or the N of
    --emulate-shard N), channels [nu0, nu0 + nnu) are this rank's.
      clarray   pair-sharded: F (F + 1) / 2 / nranks channel pairs x zint^2 sub-sample pairs; its roof is the L2 -> CU traffic
                of the profile build (6 table values per k_perp row and sub-sample pair), priced against the L2 peak; the
                NOMINAL 60 flop per evaluation of the table model stay in the record as TFLOPs but are no fraction of
                anything; separable components cost an outer product: bytes = the C_l rows written, HBM bound
      factor    l-sharded: L / nranks blocks of F^3 / 3 for every component that is factored per l (the table model; the
                summed covariance in joint mode); a separable component is ONE F^3 / 3 (its root serves every l)
      draw      triangular factors: channel nu takes nu + 1 columns, re and im: 4 nalm sum_{nu in the shard} (nu + 1) flop
                per draw (joint mode: one draw; separate: one per component)
      legendre  the MFMAs the kernel executes (the plan's count); ringfft: the cells read + the pixels written"""
    L = lmax + 1
    nalm = L * (L + 1) // 2
    npix = 12 * nside * nside
    kinds = ["table" if m == "21cm" else "separable" for m, _ in comps]
    k1_flops = k1_bytes = k1_l2 = 0.0
    # table rows of the k_perp axis a profile spans: x = log10(l) 499 / log10(4e5) puts 89.07 rows in a decade of l; the
    # first 25 multipoles are more than two rows apart and get their two rows each, the dense range runs from l = 25 on
    rows = 2.0 * min(25, L) + 89.07 * np.log10(max(lmax, 25) / 25.0)
    for (m, z), kind in zip(comps, kinds):
        zint = 2**z + 1 if z else 1
        if kind == "table":
            k1_flops += 60.0 * L * (F * (F + 1) / 2.0 / nranks) * zint * zint
            # L2 -> CU bytes of the profile build: every sub-sample pair reads its two k_par columns of three tables on
            # every row (6 values of 8 bytes; cfg 3: 2.66e6 sub-sample pairs x 220 rows x 48 B = 28 GB, the figure the
            # ablations and the TCP counters of round 5 gave)
            k1_l2 += (F * (F + 1) / 2.0 / nranks) * zint * zint * rows * 48.0
        k1_bytes += 8.0 * L * F * F / nranks
    joint = len(comps) > 1 and sum_mode == "joint"
    per_l = 1 if joint else sum(1 for k in kinds if k == "table")
    once = 0 if joint else sum(1 for k in kinds if k == "separable")
    k2_flops = per_l * (L / nranks) * F**3 / 3.0 + once * F**3 / 3.0
    k2_bytes = 16.0 * (per_l * (L / nranks) + once) * F * F
    ndraw = 1 if joint else len(comps)
    # (channels: the global indices of the rank's channels when they are not the block [nu0, nu0 + nnu) - a folded shard)
    k3_flops = ndraw * 4.0 * nalm * (nnu * (nu0 + 0.5 * (nnu + 1)) if channels is None else float(sum(int(c) + 1 for c in channels)))
    k3_bytes = ndraw * (8.0 * L * F * nnu + 16.0 * nalm * nnu)
    return {
        "clarray": ("l2" if k1_l2 else "hbm", k1_flops, k1_l2 if k1_l2 else k1_bytes,
                    "bound by the table reads of the profile build (L2 -> CU bytes by the model of bench.stage_work, 28 GB at cfg 3); "
                    "TFLOPs is a NOMINAL 60 flop per table evaluation, never a fraction; pairs of this rank only"),
        "factor": ("valu", k2_flops, k2_bytes, "multipoles of this rank only"),
        "draw": ("mfma", k3_flops, k3_bytes, "rows of this rank's channels; %d draw(s) per step" % ndraw),
        "legendre": ("mfma", legendre_executed_flops, 16.0 * nalm * nnu + 16.0 * (4 * nside - 1) * L * nnu, "executed MFMAs"),
        "ringfft": ("hbm", 2.5 * npix * np.log2(4 * nside) * nnu, 16.0 * (4 * nside - 1) * L * nnu + 8.0 * npix * nnu, ""),
    }


def stage_rooflines(stages, comps, F, nside, lmax, nnu, nu0, nranks, sum_mode, legendre_executed_flops, channels=None):
    """Every stage of the step against the roof that bounds it: the rank's algorithmic work per STEP (stage_work) over the
    stage's measured time per step.  All fractions are <= 1 by construction of the work model for any rank count,
    --emulate-shard and multi-component workloads (tests/test_host.py checks the committed lines)."""
    out = {}
    for k, (bound, flops, nbytes, note) in stage_work(comps, F, nside, lmax, nnu, nu0, nranks, sum_mode,
                                                      legendre_executed_flops, channels).items():
        if k not in stages:
            continue
        sec = stages[k]["ms_per_step"] * 1e-3
        e = {"ms": stages[k]["ms_per_step"], "bound": bound, "TFLOPs": flops / sec / 1e12, "GBs": nbytes / sec / 1e9}
        if bound == "hbm":
            e["frac"] = e["GBs"] / HBM_PEAK_GBS
        elif bound == "l2":                   # K1: table bytes through the L2s against their aggregate peak
            e["peak_GBs"] = L2_PEAK_GBS
            e["frac"] = e["GBs"] / L2_PEAK_GBS
            e["nominal_TFLOPs"] = e.pop("TFLOPs")   # (60 flop per evaluation by convention: may exceed the FP64 peak, prices nothing)
        else:
            e["frac"] = e["TFLOPs"] / FP64_MFMA_PEAK_TFLOPS
        if note:
            e["work"] = note
        out[k] = e
    return out


def seeded_modes(ctx, shard, cold_factors, F, lmax, barrier, steps, ndraw=1, nmaps=None):
    """(seeded_numpy_mode, legacy_rng_mode) of the bench line.

    seeded_numpy_mode: HBM-resident ms per COLD step when the caller passes ``rng = numpy.random.default_rng(seed)`` as
    cora's callers do - the normals are numpy's own PCG64 + ziggurat sequence, generated on the device.
    legacy_rng_mode: the same with ``rng=None`` (numpy's global MT19937 + polar method, cora/util/nputil.py:121-123;
    ``Sky3d.getsky()``'s default) - since round 4 also continued on the device; numpy's own rate on the host (1e7
    normals timed) is reported beside it."""
    import torch

    nrep = max(1, min(3, steps))
    rng = np.random.default_rng(2024)

    def cold_step(g):
        # one cold step as the library's own getsky() runs it: the generator's passes - which depend on the generator
        # alone - are started first and run beside the C_l integration and the factorisation (shard.prepare_numpy)
        prep = shard.prepare_numpy(g) if hasattr(shard, "prepare_numpy") else None
        if prep is None:
            return shard.realise_numpy(g, cold_factors())
        try:
            fac = cold_factors()
        except BaseException:
            prep.abort()
            raise
        return shard.realise_numpy(g, fac, prepared=prep)

    cold_step(rng)                                            # (workspace growth, not timed)
    barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.time()
    for _ in range(nrep):
        cold_step(rng)
    barrier()
    ms = (time.time() - t0) / nrep * 1e3
    ctx.profile_enable(False)
    st = {}
    for name in ("zig_seek", "zig_count", "zig_scan", "zig_emit", "draw"):
        t, n = ctx.profile_get(name)
        if n:
            st[name] = round(t / nrep, 3)
    nnorm = ndraw * 2 * F * ((lmax + 1) * (lmax + 2) // 2)
    nmaps = F if nmaps is None else nmaps
    seeded = {"ms_per_step": ms, "maps_per_s": nmaps / (ms * 1e-3), "stages_ms": st, "normals_per_step": nnorm,
              "rng": "numpy.random.default_rng(seed): PCG64 + ziggurat standard_normal continued on the device, bit-identical to "
                     "numpy, emitted one range of multipoles at a time into a two-slot ring that K3 consumes "
                     "(corahip_draw_alm_numpy: no 16 F nalm byte stream buffer; 'draw' spans the emit + K3 pipeline); the "
                     "Generator's state is advanced as numpy would; since round 6 the generator's own passes (count + scan, the "
                     "first two ranges) are started ahead of the C_l integration and run beside it (corahip_draw_alm_numpy_prepare)"}
    # rng=None: numpy's legacy global state, continued on the device (corahip_normals_mt19937_legacy)
    np.random.seed(12345)
    cold_step(None)
    barrier()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.time()
    for _ in range(nrep):
        cold_step(None)
    barrier()
    lms = (time.time() - t0) / nrep * 1e3
    ctx.profile_enable(False)
    lst = {}
    for name in ("mt_jump", "mt_count", "mt_emit", "draw"):
        t, n = ctx.profile_get(name)
        if n:
            lst[name] = round(t / nrep, 3)
    nsample = 10_000_000
    t0 = time.time()
    np.random.standard_normal(nsample)
    rate = nsample / (time.time() - t0)
    legacy = {"ms_per_step": lms, "maps_per_s": nmaps / (lms * 1e-3), "stages_ms": lst,
              "rng": "rng=None: numpy's legacy global MT19937 + polar method (what Sky3d.getsky() draws from), continued on the "
                     "device: MT19937 cut into segments by GF(2) jump-ahead polynomials; same accepted attempts and generator "
                     "state as numpy and - glibc's log restated operation by operation - the same values bit for bit; emitted "
                     "one range of multipoles at a time into the ring K3 consumes (corahip_draw_alm_numpy)",
              "host_numpy_would_take_s": nnorm / rate, "host_normals_per_s": rate}
    torch.cuda.synchronize()
    return seeded, legacy


def host_delivered_rate(ctx, shard, nside, F, npix, nrep=4):
    """maps/s through skysim.mkfullsky_stream: realisations delivered to the host as numpy arrays (pinned memory,
    D2H on the copy stream overlapped with the next realisation), factors cached.  Skipped when the host cannot hold
    three realisations in RAM."""
    import torch

    from cora_amd.core import skysim
    from cora_amd.util.nputil import DeviceRNG

    need = 3.2 * F * npix * 8
    try:
        avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except Exception:
        avail = 0
    if avail < need + 32e9:
        return {"skipped": "host has %.0f GB available, the leg needs %.0f GB of pinned memory" % (avail / 1e9, need / 1e9)}
    T, info, rows = shard.factors()
    if rows and T.shape[1] != T.shape[2]:
        return {"skipped": "row-sliced factors"}
    t0 = time.time()
    nwarm = 3                              # three pinned blocks rotate (in flight, delivered, being filled): page-locked once
    gen = skysim.mkfullsky_stream(None, nside, [DeviceRNG(7000 + i) for i in range(nrep + nwarm)], factors=(T, info))
    chk = 0.0
    for _ in range(nwarm):
        first = next(gen)
        chk += float(first[0, 0])
        del first
    t_first = time.time() - t0
    t0 = time.time()
    n = 0
    for m in gen:
        chk += float(m[-1, -1])
        n += 1
        del m
    dt = time.time() - t0
    torch.cuda.synchronize()
    return {"maps_per_s": F * n / dt, "GB_per_s": F * n * npix * 8 / dt / 1e9, "realisations": n,
            "warmup_s_incl_page_locking": t_first, "finite": bool(np.isfinite(chk)),
            "path": "skysim.mkfullsky_stream: device Philox draw + synthesis, pinned double-buffered D2H on a copy stream"}


def drop_in_call(model, freq, nside, lmax, zromb, F, npix):
    """Wall seconds of the reference's own call forms, numpy in / numpy out, as a cora user would type them
    (cora/core/maps.py:227-237: ``cla = skysim.clarray(aps, lmax, freqs, zromb)``; ``skysim.mkfullsky(cla, nside, rng)``):
    ``clarray`` (K1 + 1.07 GB to the host), then the FIRST and a LATER ``mkfullsky(C_numpy, nside, rng=default_rng(s))``
    (1.07 GB up, K2 + numpy's PCG64 stream + K3 + K4 + K5 on the device, 25.8 GB down into a new ndarray; the first call
    also grows the library's workspaces).  PCIe- and page-fault-bound, never `value`."""
    import torch

    from cora_amd.core import skysim

    need = 1.15 * F * npix * 8
    try:
        avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except Exception:
        avail = 0
    if avail < need + 16e9:
        return {"skipped": "host has %.0f GB available, one returned realisation needs %.0f GB" % (avail / 1e9, need / 1e9)}
    torch.cuda.synchronize()
    t0 = time.time()
    cla = skysim.clarray(model.angular_powerspectrum, lmax, freq.copy(), zromb=zromb)
    t_cla = time.time() - t0
    times = []
    chk = 0.0
    for s in (11, 12, 13):
        rng = np.random.default_rng(s)
        t0 = time.time()
        sky = skysim.mkfullsky(cla, nside, rng=rng)
        times.append(time.time() - t0)
        chk += float(sky[0, 0]) + float(sky[-1, -1])
        assert sky.shape == (F, npix) and isinstance(sky, np.ndarray)
        del sky
    return {"clarray_s": t_cla, "mkfullsky_first_s": times[0], "mkfullsky_later_s": min(times[1:]), "maps_per_s_later": F / min(times[1:]),
            "GB_returned": F * npix * 8 / 1e9, "finite": bool(np.isfinite(chk)),
            "call": "skysim.clarray(Corr21cm().angular_powerspectrum, lmax, freqs, zromb) -> ndarray; skysim.mkfullsky(ndarray[L,F,F], "
                    "nside, rng=numpy.random.default_rng(s)) -> ndarray[F,npix]: H2D + factor + seeded draw + synthesis + D2H "
                    "into a page-locked block the returned ndarray views"}


def cpu_baseline(model_name, F, freq, nside, lmax, zromb):
    """The oracle (a port of the reference's algorithm) timed on this box's host cores on a bounded sample of the same
    workload, each leg scaled linearly.  Legs and their threading:
      C_l integration  numpy + C/OpenMP table lookups (bilinearmap.pyx's loop), l in sections of 5 as the reference
                       (cora/core/skysim.py:51-67): numpy parts 1 thread, lookups `cores` threads
      factor + draw    scipy Cholesky of the jittered block + numpy normals + ONE real GEMM [F,F] x [F, 2(l+1)] per l
                       (the reference multiplies a complex array: twice the flops): BLAS threads = `cores`
      synthesis        C/OpenMP Legendre recurrence + C/OpenMP ring stage (radix-2 / Bluestein FFT per ring) per channel,
                       channels one after the other as hputil.py:525-529: `cores` threads
    """
    import scipy.integrate as si
    import scipy.linalg as la

    from oracle import models
    from oracle import sht

    ncore = len(os.sched_getaffinity(0))
    L = lmax + 1
    rng = np.random.default_rng(0)
    if model_name == "21cm":
        om = models.Corr21cm()
        t0 = time.time()
        om.tables()
        t_tables = time.time() - t0
    else:
        om = models.FullSkySynchrotron()
        t_tables = 0.0
    zint = 2**zromb + 1
    zhalf = abs(freq[1] - freq[0]) / 2.0
    za = (freq[:, None] + np.linspace(-zhalf, zhalf, zint)[None, :]).ravel()
    # K1 sample: three sections of 5 multipoles spread over the range, all channels
    nsec, per = 3, 5
    secs = [np.arange(l0, l0 + per, dtype=np.float64) for l0 in np.linspace(1, lmax - per, nsec).astype(int)]
    zspace = 2.0 * zhalf / 2**zromb
    t0 = time.time()
    cl_s, l_s = [], []
    for lsec in secs:
        clt = om.angular_powerspectrum(lsec[:, None, None], za[None, :, None], za[None, None, :])
        clt = clt.reshape(-1, F, zint, F, zint)
        clt = si.romb(si.romb(clt, dx=zspace, axis=4), dx=zspace, axis=2) / (2 * zhalf) ** 2
        cl_s.extend(list(clt))
        l_s.extend(int(v) for v in lsec)
    t_k1 = (time.time() - t0) * L / (nsec * per)
    # K2 + K3 sample: factor + draw for the sampled l
    t0 = time.time()
    nd = 0
    for C, l in zip(cl_s, l_s):
        cm = C + np.identity(F) * C.diagonal().max() * 1e-14
        T = la.cholesky(cm, lower=True)
        g = rng.standard_normal((F, 2 * (l + 1)))            # (re | im) blocks: one real GEMM
        np.dot(T, g) * 0.7071067811865476
        nd += l + 1
    t_k23 = (time.time() - t0) * (L * (L + 1) / 2) / nd
    # K4 + K5 sample: nmap channels through the C/OpenMP synthesis
    nmap = 8 if nside >= 1024 else 4
    nalm = L * (L + 1) // 2
    a = rng.standard_normal(nalm) + 1j * rng.standard_normal(nalm)
    sht.alm2map(a, nside, lmax, rings_c=True)                 # (thread pool / page warm-up, not timed)
    t0 = time.time()
    for _ in range(nmap):
        sht.alm2map(a, nside, lmax, rings_c=True)
    t_sht = (time.time() - t0) * F / nmap
    t_real = t_k1 + t_k23 + t_sht
    return {
        "value": F / t_real,
        "unit": "maps/s",
        "cores": ncore,
        "kind": "port",
        "sample": "C_l integration on %d of %d l in sections of 5 (numpy 1 thread + C/OpenMP table lookups %d threads), "
                  "Cholesky + normals + one real GEMM per l on those l (BLAS %d threads, scaled by nalm), C/OpenMP Legendre + "
                  "C/OpenMP ring FFTs of %d of %d channels (%d threads); each leg scaled linearly; per-realisation seconds: "
                  "clarray %.1f, factor+draw %.1f, synthesis %.1f (one-off 21cm table build %.1f s not counted)"
                  % (nsec * per, L, ncore, ncore, nmap, F, ncore, t_k1, t_k23, t_sht, t_tables),
    }


if __name__ == "__main__":
    main()
