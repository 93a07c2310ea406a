"""Counterpart of cora.signal: corr (flat-sky FFT C_l table model), corr21cm."""
