"""n-D real-FFT frequency grids and the rfftn / irfftn pair, on the GPU.

API counterpart of cora/util/fftutil.py.  The transforms run in ``csrc/flatsky.hip``
(LDS line FFTs, Bluestein for lengths that are not powers of two); there is no host FFT here.
"""
import warnings

import numpy as np

from .. import _lib


def rfftfreqn(n, d=None):
    """Frequency vector of every sample of an n-D real FFT, shape ``n[:-1] + (n[-1]//2+1, len(n))``.

    Follows fftutil.py:14-61 literally: the leading axes hold ``fftshift(arange(-x/2, x/2))``, which for an
    odd ``x`` is the half-integer grid the reference produces (not ``numpy.fft.fftfreq``).  ``d`` is the
    sample spacing per axis; unlike the reference the caller's array is not scaled in place.
    """
    n = np.array(n)
    if d is None:
        scale = n.astype(np.float64)
    else:
        if len(d) != len(n):
            raise Exception("Sample spacing array is the wrong length.")
        scale = np.asarray(d, dtype=np.float64) * n
    per_axis = [np.fft.fftshift(np.arange(-x / 2, x / 2, 1.0)) for x in n[:-1]]
    per_axis.append(np.arange(0, n[-1] // 2 + 1, 1.0))
    return np.stack(np.meshgrid(*per_axis, indexing="ij"), axis=-1) / scale


def rfftn_device(arr):
    """``rfftn`` of a float64 torch tensor on the GPU -> complex128 tensor."""
    return _lib.get_context().rfftn(arr.contiguous())


def irfftn_device(spec):
    """``irfftn`` of a complex128 torch tensor on the GPU (``spec`` is consumed) -> float64 tensor."""
    return _lib.get_context().irfftn(spec.contiguous())


def rfftn(arr):
    """numpy-in / numpy-out ``rfftn`` (fftutil.py:64-77)."""
    arr = np.asarray(arr, dtype=np.float64)
    if arr.shape[-1] % 2 != 0:
        warnings.warn("Last axis length not multiple of 2. fftutil.irfftn will not reproduce this exactly.")
    ctx = _lib.get_context()
    return ctx.rfftn(ctx.to_device(arr)).cpu().numpy()


def irfftn(arr):
    """numpy-in / numpy-out ``irfftn`` (fftutil.py:80-87); the output's last axis has ``2 (m - 1)`` samples."""
    ctx = _lib.get_context()
    spec = ctx.to_device(np.asarray(arr), dtype=np.complex128)
    return ctx.irfftn(spec).cpu().numpy()
