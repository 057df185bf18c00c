"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A from-scratch CPU restatement (NumPy float64 / torch-CPU) of the reference's hot-path
algorithms (mhw32/neural-navier-stokes), each function citing the reference file:line it
follows.  It exists to CHECK the HIP path, never to replace it:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import anything from this package;
  * nothing under ``neural-navier-stokes_amd/`` imports it (a test enforces that);
  * parity pinning: the restatement is pinned against golden vectors captured from the
    reference itself in the build container by ``oracle/capture.py`` (committed under
    ``tests/golden/``).  The periodic-box residual (FD 5/9-point + Fourier spectral) has NO
    reference counterpart: for those functions parity is
    "unpinned by the reference" and is pinned by analytic known-answers (Taylor-Green vortex)
    instead -- see ``oracle/periodic.py``.
"""
