"""Counterpart of cora/util/nputil.py on the hot path: matrix root + normal draws.

``matrix_root_manynull`` runs the library's batched factor kernel (K2) on a batch of one;
``complex_std_normal`` consumes a numpy generator exactly as the reference does (that IS
the definition of the seeded stream); ``DeviceRNG`` is this package's
counter-based alternative that never leaves the GPU.
"""
import numpy as np

from .. import _lib


class DeviceRNG:
    """Counter-based (Philox4x32-10 + Box-Muller) N(0,1) stream generated on the GPU.

    Pass an instance as ``rng`` to ``skysim.mkfullsky`` to keep the draw on the device.
    Element k of the stream (in the order of cora/core/skysim.py:114-121, see
    include/corahip.h "stream order") depends only on (seed, k): reproducible for any
    number of GPUs.  Each ``mkfullsky`` call advances ``seed`` by one.
    """

    def __init__(self, seed=0):
        self.seed = int(seed)

    def next_seed(self):
        s = self.seed
        self.seed += 1
        return s


def matrix_root_manynull(mat, threshold=1e-16, truncate=True):
    """Square root of a symmetric PSD matrix (cora/util/nputil.py:51-101).

    Cholesky first; if a pivot is not positive, eigen-decomposition with eigenvalues
    below ``max * threshold`` zeroed.  With ``truncate`` the root keeps only the columns
    of the non-zero eigenvalues and ``(root, num_pos)`` is returned.
    """
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    if mat.ndim != 2 or mat.shape[0] != mat.shape[1]:
        raise ValueError("expected square matrix")
    ctx = _lib.get_context()
    n = mat.shape[0]
    T, info = ctx.factor_batched(ctx.to_device(mat[np.newaxis]), jitter_rel=0.0, eig_thresh=threshold)
    root = T[0].cpu().numpy()
    branch = int(info[0].item())
    if branch == 0:
        num_pos = n
    else:
        nz = np.flatnonzero(np.abs(root).sum(axis=0) > 0)
        num_pos = len(nz)
        if truncate:
            # columns are in ascending-eigenvalue order (scipy.linalg.eigh order): keep the last num_pos
            root = root[:, n - num_pos:] if num_pos > 0 else root[:, n:]
            root = root[np.newaxis]  # the reference returns [1, N, num_pos] here (nputil.py:92-96 quirk)
    if truncate:
        return root, num_pos
    return root


def complex_std_normal(shape, rng=None):
    """Complex standard normals (cora/util/nputil.py:104-125): the real block is drawn
    first, then the imaginary block; ``rng=None`` uses numpy's legacy global state."""
    if rng is None:
        re = np.random.standard_normal(shape)
        im = np.random.standard_normal(shape)
    else:
        re = rng.standard_normal(shape)
        im = rng.standard_normal(shape)
    return (re + 1.0j * im) / 2**0.5


def save_ndarray_list(fname, la):
    """cora/util/nputil.py:12-27."""
    np.savez(fname, **{repr(i): v for i, v in enumerate(la)})


def load_ndarray_list(fname):
    """cora/util/nputil.py:30-48."""
    d = np.load(fname)
    return [v for _, v in sorted(d.items(), key=lambda kv: int(kv[0]))]
