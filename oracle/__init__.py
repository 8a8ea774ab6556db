"""CPU oracle for the HypAD TadGAN train/score hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker (or as the timed
CPU baseline).  Nothing under ``hypad_amd/`` imports it; the product path fails
loudly when the HIP extension is missing.

What it is: a plain PyTorch-CPU / NumPy restatement of the reference algorithm
(``/root/reference``), each function citing the reference file:line it follows.

Pinning status:
* networks, hyperbolic ops, the three training iterations (losses + every
  parameter gradient), Adam steps, row-wise / pair-wise Poincare distance,
  point error / un-roll median / rolling mean / z-score / score combination /
  KDE critic smoothing: PINNED against outputs of the reference itself, run in
  the build container by ``tests/golden/gen_fixtures.py`` and committed as
  ``tests/golden/*.npz`` (the reference has no tests or golden vectors of its own).
* ``geoopt.optim.RiemannianAdam`` (geoopt==0.5.0, not vendored in the
  reference): PARITY UNPINNED for the 100-element ball-valued bias; the
  Euclidean branch is pinned against ``torch.optim.Adam(weight_decay=...)``.
* ``pyts.metrics.dtw`` (pyts==0.12.0, not vendored): PARITY UNPINNED; restated
  from the published classic-DTW recurrence and pinned only against brute-force
  and hand-computed cases.
* ``_area_error``: the reference calls ``scipy.integrate.trapz`` which no longer
  exists in the SciPy of this image; restated with ``numpy.trapezoid`` and pinned
  against pandas ``rolling().apply`` run here.
"""
