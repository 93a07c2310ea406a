"""Worker of tests/test_gpu_multirank.py::test_mkfullsky_l_distributed_mpiarray (started by torch.distributed.run, all
ranks on GPU 0 over gloo): skysim.mkfullsky on an l-distributed MPIArray-like input (cora/core/skysim.py:97-103,
132-134) must give every rank its frequency shard of the single-process realisation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeMPIArray:
    """The three members of caput.mpiarray.MPIArray the reference touches on this path."""

    def __init__(self, local, global_shape):
        self.local_array = local
        self.global_shape = tuple(global_shape)

    @staticmethod
    def wrap(arr, axis=0):
        out = FakeMPIArray(arr, arr.shape)
        out.wrapped_axis = axis
        return out


def main():
    import torch
    import torch.distributed as dist

    # default: all ranks share GPU 0 and talk over gloo (the 1-GPU boxes of the pool); CORA_TEST_BACKEND=nccl: one
    # device per rank over RCCL (parallel._all_to_all's all_to_all_single branch with world > 1)
    backend = os.environ.get("CORA_TEST_BACKEND", "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
    else:
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    from cora_amd.core import skysim
    from cora_amd.util.nputil import DeviceRNG

    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
    Cfull = g["cla_21cm_F8_l64_zromb3"]
    nside = 16
    for F in (8, 7):                      # 8: row blocks by all-to-all (world 2); 7: all-gather of the factors
        C = np.ascontiguousarray(Cfull[:, :F, :F])
        L = C.shape[0]
        base, extra = divmod(L, world)    # caput's axis-0 split: the first L % world ranks hold one more
        n = base + (1 if rank < extra else 0)
        lo = rank * base + min(rank, extra)
        for mode in ("device", "numpy", "alms"):
            mk = (lambda: DeviceRNG(5)) if mode == "device" else (lambda: np.random.default_rng(5))
            out = skysim.mkfullsky(FakeMPIArray(C[lo:lo + n], C.shape), nside, alms=(mode == "alms"), rng=mk())
            ref = skysim.mkfullsky(C, nside, alms=(mode == "alms"), rng=mk())      # single-process result, on every rank
            if mode == "alms":
                # the reference returns alm_array.allgather() (cora/core/skysim.py:123-125): the FULL array on EVERY
                # rank, a plain ndarray - each rank's return value is compared directly
                assert isinstance(out, np.ndarray) and out.shape == ref.shape == (F, 1, L, L), (type(out), out.shape)
                err = np.abs(out - ref).max() / np.abs(ref).max()
                assert err <= 1e-13, (F, mode, rank, err)
                continue
            assert isinstance(out, FakeMPIArray) and out.wrapped_axis == 0
            fb, fe = divmod(F, world)     # the frequency shard of this rank (axis-0 split of F)
            f0, fn = rank * fb + min(rank, fe), fb + (1 if rank < fe else 0)
            assert out.local_array.shape == ref[f0:f0 + fn].shape, (out.local_array.shape, ref.shape)
            err = np.abs(out.local_array - ref[f0:f0 + fn]).max() / np.abs(ref).max()
            assert err <= 1e-13, (F, mode, rank, err)
        # rng=None: one realisation with a broadcast seed - finite, right shape, same on reruns of the shape check
        out = skysim.mkfullsky(FakeMPIArray(C[lo:lo + n], C.shape), nside)
        assert np.all(np.isfinite(out.local_array)) and out.local_array.shape[1] == 12 * nside * nside
    dist.barrier()
    if rank == 0:
        print("MPIARRAY OK", file=sys.stderr)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
