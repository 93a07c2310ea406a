"""Multi-rank paths on ONE GPU (ranks share cuda:0 and talk over gloo): bench.py's self-launching --gpus N, the
three-component workload of BASELINE configs[3] through parallel.SkySum, and skysim.mkfullsky on an l-distributed
MPIArray-like input (cora/core/skysim.py:97-134).  Run with -m gpu."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _bench(args):
    p = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=_plain_env(), capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout          # stdout carries exactly one line: the JSON
    m = re.search(r"CHECKSUM (\w+)", p.stderr)
    _bench.seeded = tuple(re.findall(r"CHECKSUM_(?:SEEDED|LEGACY) (\w+)", p.stderr))
    return json.loads(lines[0]), (m.group(1) if m else None)


@pytest.mark.parametrize("workload", ["tiny", "tiny3"])
def test_bench_launches_its_own_ranks(workload):
    """`python bench.py --gpus 2` from a plain shell (no torch.distributed.run around it): the ranks are started as a
    child job, one JSON line comes back, the process group really had 2 ranks, and the per-channel checksums of the
    2-rank realisation equal the single-rank ones (tiny3: three summed components, parallel.SkySum)."""
    common = ["--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--checksum"]
    one, c1 = _bench(common)
    s1 = _bench.seeded
    two, c2 = _bench(["--gpus", "2", "--dist-backend", "gloo", "--same-device"] + common)
    s2 = _bench.seeded
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2
    assert c1 is not None and c1 == c2, (c1, c2)
    assert two["value"] > 0 and two["metric"] == one["metric"]
    # the reference's seeded call (rng = default_rng(seed)) and its rng=None call through the sharded path - every rank
    # generates numpy's stream range by range for its own factor rows: every pixel of every channel bit for bit
    assert len(s1) == 2 and s1 == s2, (s1, s2)


@pytest.mark.parametrize("workload,world", [("tiny32", 2), ("tiny32", 4), ("tiny32s", 2)])
def test_folded_channel_assignment_reproduces_the_single_rank_realisation(workload, world):
    """bench.py --fold: rank r owns chunks r and 2N-1-r of 2N (the draw's triangular work balanced over the ranks).
    The union of the folded shards - Philox stream, the reference's seeded stream and its rng=None stream - equals the
    single-rank realisation: per-channel statistics and every pixel (sha1) of every channel."""
    common = ["--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--checksum"]
    one, c1 = _bench(common)
    s1 = _bench.seeded
    two, c2 = _bench(["--gpus", str(world), "--dist-backend", "gloo", "--same-device", "--fold"] + common)
    s2 = _bench.seeded
    assert two["ranks_seen"] == world and "folded" in two["config"]["channel_assignment"]
    assert c1 is not None and c1 == c2, (c1, c2)
    assert len(s1) == 2 and s1 == s2, (s1, s2)


@pytest.mark.parametrize("workload,extra", [("cfg2", []), ("tiny32s", ["--fold"]), ("tiny", [])])
def test_rccl_code_path_with_one_rank(workload, extra):
    """Everything of the N-rank path that can run on one GPU, on the REAL backend: `bench.py --gpus 1 --force-dist` makes
    a one-rank nccl (= RCCL) process group (init_process_group with device_id and the timeout of cora_amd.parallel), runs
    all-to-all #1 / #2 through `all_to_all_single`, the `all_gather_into_tensor` of the info vectors and the barrier +
    max-over-ranks timing, for the separable model (cfg2: l-sharded factor + all-gather), the three-component sum with the
    folded channel assignment (tiny32s --fold) and the pair-sharded 21cm model (tiny).  Same per-channel checksums and
    same pixels (seeded / legacy streams) as the plain single-process run; the line names the rank's device."""
    common = ["--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--checksum"]
    one, c1 = _bench(common + extra)
    s1 = _bench.seeded
    nc, c2 = _bench(["--gpus", "1", "--force-dist", "--dist-backend", "nccl"] + common + extra)
    s2 = _bench.seeded
    assert nc["ranks_seen"] == 1 and nc["distinct_devices"] == 1 and len(nc["rank_devices"]) == 1
    assert nc["rank_devices"][0].startswith("rank 0: cuda:0 ") and "uuid=" in nc["rank_devices"][0]
    assert c1 is not None and c1 == c2, (c1, c2)
    assert len(s1) == 2 and s1 == s2, (s1, s2)


def test_mkfullsky_l_distributed_mpiarray():
    """skysim.mkfullsky(MPIArray-like l-distributed corr): 2 and 3 ranks (uneven l blocks, F = 8 and 7) return the
    frequency shards of the single-process realisation (DeviceRNG and identically seeded numpy Generators)."""
    port = 29871
    for n in (2, 3):
        port += 1
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                            "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join("tests", "_mpiarray_worker.py")], cwd=ROOT, env=_plain_env(),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "MPIARRAY OK" in p.stderr, p.stderr[-3000:]


def _ngpu():
    import torch

    return torch.cuda.device_count()      # (does not initialise the GPU in this process)


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs: RCCL with world > 1")
@pytest.mark.parametrize("workload", ["tiny", "tiny3"])
def test_bench_two_devices_over_rccl(workload):
    """bench.py --gpus 2 as the driver's scaling tier runs it: one rank per DEVICE over RCCL (backend nccl) -
    parallel._all_to_all's all_to_all_single branch with world > 1, the pair-slab and factor-row exchanges over xGMI.
    The per-channel checksums must equal the single-GPU ones.  Skips on the 1-GPU boxes of this pool and runs the
    first time a multi-GPU node executes the suite, i.e. before the SCALE bench does."""
    common = ["--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--checksum"]
    one, c1 = _bench(common)
    two, c2 = _bench(["--gpus", "2"] + common)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2
    assert c1 is not None and c1 == c2, (c1, c2)


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs: RCCL with world > 1")
def test_mkfullsky_l_distributed_mpiarray_over_rccl():
    """skysim.mkfullsky on an l-distributed MPIArray-like input with one rank per device over RCCL (2 ranks)."""
    env = _plain_env()
    env["CORA_TEST_BACKEND"] = "nccl"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29891",
                        os.path.join("tests", "_mpiarray_worker.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0 and "MPIARRAY OK" in p.stderr, p.stderr[-3000:]
