"""CPU oracle: a plain numpy / C restatement of cora's Gaussian-sky hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``cora_amd`` (the product) may import,
call, link or execute anything from this package.  Allowed users: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Each function cites the reference file:line (relative to /root/reference) it
follows.  Pinning status:

* ``clarray``, the 21cm and foreground ``aps`` models, ``Cosmology``, the cubic
  spline, ``matrix_root_manynull``, ``complex_std_normal``, ``mkfullsky(alms=True)``,
  ``pack_alm``: PINNED against outputs of the reference itself, captured by
  ``tests/golden/make_golden.py`` into ``tests/golden/reference_vectors.npz``
  and against the reference's own known-answer tests (tests/test_corr.py).
* ``alm2map`` (the reference delegates to ``healpy.alm2map``, a third-party
  dependency that is absent from /root/reference and from this image,
  ``healpy>=1.17``, pyproject.toml:33): **PARITY UNPINNED** against healpy.  It
  is pinned only at definition level: brute-force ``sum a_lm Y_lm`` with
  ``scipy.special.sph_harm_y`` at HEALPix RING pixel centres, analytic
  single-mode maps, and mpmath spot values of the normalised Legendre functions.
"""
