"""Physical constants the path needs (the reference takes them from caput.astro.constants,
a third-party module absent from /root/reference: cora/signal/corr21cm.py:3,
cora/util/cosmology.py:16, cora/core/maps.py:3).  SI units unless noted."""
import math

degree = 2 * math.pi / 360
c = 299792458.0
nu21 = 1420.40575177  # MHz
k_B = 1.3806503e-23
mega_parsec = 3.08568025e22
year = 365.25 * 86400.0
mega_year = 1e6 * year
