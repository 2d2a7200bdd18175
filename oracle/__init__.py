"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path
(``bilateral_driving_amd``) never imports this package and fails loudly when the
HIP library is missing.

Two halves:

* ``oracle.bilagrid_oracle``  -- bilateral-grid slice / multi-scale affine / TV.
  PINNED: checked against golden vectors produced by importing the reference's
  own Python (``oracle/gen_golden_bilateral.py`` -> ``tests/golden/*.npz``).
* ``oracle.gs_oracle`` -- SH, projection, tile binning, sort, alpha compositing.
  PARITY UNPINNED: the arithmetic lives in gsplat v1.3.0 (pip git dependency,
  /root/reference/README.md:81), which is absent from the reference tree and
  from this image.  The restatement follows the published 3DGS / gsplat
  algorithm and the reference's call-site contract; it is pinned only by this
  repo's own known-answer / finite-difference / invariance tests.
"""
