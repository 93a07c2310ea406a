"""Counterpart of cora.core: skysim (hot path), maps, gaussianfield."""
