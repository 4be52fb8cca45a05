"""CPU oracle for the DeSCo hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package, and only as the checker / reported baseline.  The product package
(``desco_amd``) never imports it and fails loudly when its HIP library is missing.

Pinning status:
  * integer path (``oracle.partition``): PINNED.  Checked against golden vectors produced by
    importing the reference's own pure-Python functions in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.json|npz``).
  * float path (``oracle.model``): PARITY UNPINNED by the reference.  The reference ships no tests
    or golden vectors and its float path cannot be imported here (torch_geometric /
    pytorch_lightning / torch_scatter are absent and not installable).  ``oracle.model`` is a
    pure-torch CPU restatement of the reference semantics (file:line cited per function) that is
    cross-checked only against independent dense-algebra formulas on tiny graphs
    (``tests/test_oracle_model.py``).
"""
