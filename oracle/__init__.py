"""CPU oracle for the HPCS hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU / numpy restatement of the reference
algorithms on the north-star path (SURVEY.md section 8a, rows a1-a15 plus the
RAdam "next" row).  It exists so that the HIP kernels in ``hpcs_amd`` can be
checked against something that runs without a GPU.

Rules (enforced by tests/test_abi_and_rules.py::test_oracle_is_test_infrastructure_only):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * ``hpcs_amd`` never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * a1-a7, a9, a11 + Poincare helpers + RAdam: pinned against outputs of the
    reference itself, captured in this container by
    ``tests/golden/make_golden.py`` and committed under ``tests/golden/``.
  * round 2 (``tests/golden/make_golden_r2.py``): ``hierarchical_loss`` (f2),
    ``get_triplets`` (a12's miner-less fallback) and the extractor on the
    reference's own three kNN graphs are pinned the same way.
  * a8, a10, a12, a13, a14 (behaviour that lives in the absent third-party
    ``pytorch-metric-learning==1.6.3``): restated from its published algorithm
    and the reference's call sites -- PARITY UNPINNED for those rows; their
    oracle statements are cross-checked by analytic identities in
    ``tests/test_oracle_identities.py``.
"""
from . import extractor, hyperbolic, metric  # noqa: F401
