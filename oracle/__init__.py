"""CPU oracle for the recman CTR forward+backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``recman_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline - never as the thing shipped.

What it is: a restatement, in numpy (float64 / float32) and CPU PyTorch
(float32 + autograd), of the arithmetic of the reference's TensorFlow layers
(``recman/tf/core/layers.py``, ``utils.py``, ``xDeepFM.py``, ``DeepFM.py``,
``DCN.py``) - the reference's own PyTorch path ``recman/th`` is an empty stub
(``recman/th/layers.py`` is 0 bytes, ``recman/th/DeepFM.py:12-13`` is
``class DeepFM(...): pass``).  Every function cites the reference file:line it
follows.

PARITY UNPINNED: the reference has no tests, no golden vectors and no stored
outputs (``tests/utils.py`` is 0 bytes), and its implementation needs
TensorFlow, which is not installed here and cannot be fetched, so the oracle
cannot be checked against a run of the reference.  What pins it instead:

* the CIN walk-through of ``recman/notes/xDeepFM.ipynb`` cell 6 (inputs in the
  notebook, outputs hand-derived: the notebook stores none) - tests/test_oracle.py
  (test_cin_notebook_kat, test_cin_z_layout_is_i_major);
* two independent restatements (numpy float64 and torch float32) that must
  agree, closed-form identities (FM pairwise-dot, cross network on integer
  data) and finite-difference gradient checks;
* ``CrossNet`` does not exist in the reference at all (``recman/tf/core/DCN.py:7``
  has the import commented out, ``DCN.py:134-137`` uses it): its arithmetic is
  restated from eq. (3) of arXiv 1708.05123, the paper ``README.md:6`` cites.
"""
