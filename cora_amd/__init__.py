"""cora_amd - MI355X (gfx950) implementation of cora's Gaussian-sky realisation path.

Mirrors the Python call surface of radiocosmology/cora for that path
(``core.skysim``, ``core.maps``, ``core.gaussianfield``, ``util.nputil``, ``util.hputil``,
``signal.corr21cm``, ``foreground.gaussianfg`` ...) on top of hand-written HIP kernels
reached through the C ABI of ``libcorahip.so`` (``include/corahip.h``).

Importing the package does not need a GPU; every compute entry point does, and raises
``CoraHipError`` / ``ImportError`` when the library or the GPU is missing (no CPU fallback).
"""
from ._lib import CoraHipError, get_context  # noqa: F401
from .util.nputil import DeviceRNG  # noqa: F401

__version__ = "0.1.0"
