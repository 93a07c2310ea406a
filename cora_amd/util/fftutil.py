"""Counterpart of cora/util/fftutil.py: n-D real-FFT frequency grids and thin FFT shims."""
import numpy as np


def rfftfreqn(n, d=None):
    """Frequency vectors of an n-D real FFT, shape ``n[:-1] + (n[-1]//2+1, len(n))`` (fftutil.py:14-61)."""
    n = np.array(n)
    d = np.ones_like(n, dtype=np.float64) if d is None else np.array(d)
    if n.shape != d.shape:
        raise Exception("Sample spacing array is the wrong length.")
    axes = [np.fft.fftfreq(int(ni), di) for ni, di in zip(n[:-1], d[:-1])]
    axes.append(np.abs(np.fft.rfftfreq(int(n[-1]), d[-1])))
    grids = np.meshgrid(*axes, indexing="ij")
    return np.stack(grids, axis=-1)


def rfftn(arr):
    return np.fft.rfftn(arr)


def irfftn(arr):
    return np.fft.irfftn(arr)
