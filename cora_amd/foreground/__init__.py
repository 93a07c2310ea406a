"""Counterpart of cora.foreground: gaussianfg, galaxy (parameter classes), pointsource (unresolved background)."""
