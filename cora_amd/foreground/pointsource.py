"""Counterpart of the Gaussian part of cora/foreground/pointsource.py: the unresolved
point-source background used by ``CombinedPointSources`` (pointsource.py:541-546).
The Poisson / catalogue populations are not Gaussian and are outside this package's scope."""
from . import gaussianfg


class UnresolvedBackground(gaussianfg.PointSources):
    """``CombinedPointSources._UnresolvedBackground``: Gaussian approximation for S < 0.1 Jy."""

    A = 3.55e-5
    nu_0 = 408.0
    l_0 = 100.0

    oversample = 0


class CombinedPointSources(object):
    """Namespace kept for drop-in access to ``CombinedPointSources._UnresolvedBackground``."""

    _UnresolvedBackground = UnresolvedBackground
