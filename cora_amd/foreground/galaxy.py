"""Counterpart of the Gaussian part of cora/foreground/galaxy.py: the full-sky synchrotron
parameter sets (galaxy.py:20-40).  ``ConstrainedGalaxy`` (Haslam-constrained, needs
``skydata.npz`` and healpy smoothing/rotation) is outside this package's scope."""
from . import gaussianfg


class FullSkySynchrotron(gaussianfg.Synchrotron):
    """Synchrotron amplitudes of La Porta et al. 2008 for |b| > 5 deg."""

    A = 6.6e-3
    beta = 2.8
    nu_0 = 408.0
    l_0 = 100.0


class FullSkyPolarisedSynchrotron(gaussianfg.Synchrotron):
    """Polarised synchrotron: same spectral shape, polarisation fraction 0.5 and a short
    frequency correlation length from Faraday rotation."""

    A = 1.65e-3
    beta = 2.8
    nu_0 = 408.0
    l_0 = 100.0
    zeta = 0.04
