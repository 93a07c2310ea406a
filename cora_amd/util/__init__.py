"""Counterpart of cora.util: nputil, hputil, cosmology, cubicspline, bilinearmap, fftutil."""
