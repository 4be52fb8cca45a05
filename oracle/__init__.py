"""CPU oracle for the DeSCo hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package, and only as the checker / reported baseline.  The product package
(``desco_amd``) never imports it and fails loudly when its HIP library is missing.

Pinning status:
  * integer path (``oracle.partition``): PINNED.  Checked against golden vectors produced by
    importing the reference's own pure-Python functions in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.json|npz``).
  * float path (``oracle.model``): PARTLY PINNED.  The reference ships no tests or golden vectors
    and its float path cannot be imported as a whole (torch_geometric / pytorch_lightning /
    torch_scatter are absent and not installable).  Pinned by vectors the reference's OWN code
    produced (its pure-torch methods called unbound with a stand-in ``self``,
    ``tests/golden/float_pieces.npz``, bit-exact): the count head (``embed_to_count``), both
    criteria, ``GossipConv.message`` / ``update`` / ``_gate_value``.  PARITY UNPINNED for what only
    PyG can execute: ``MessagePassing.propagate`` (scatter-add), ``to_hetero`` (per-type copies,
    pairwise sum), ``global_add_pool``, ``remove_self_loops`` / ``to_undirected`` -- restated from
    PyG 2.2.0 semantics and cross-checked against independent dense-algebra
    formulas on tiny graphs (``tests/test_oracle_model.py``).
"""
