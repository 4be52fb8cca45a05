"""CPU oracle for the DeSCo hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package, and only as the checker / reported baseline.  The product package
(``desco_amd``) never imports it and fails loudly when its HIP library is missing.

Pinning status:
  * integer path (``oracle.partition``): PINNED.  Checked against golden vectors produced by
    importing the reference's own pure-Python functions in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.json|npz``).
  * float path (``oracle.model``): PINNED for every line of the reference's own code that is
    executable without PyG / Lightning; PARITY UNPINNED only for the PyG internals listed below.
    Pinned bit-exactly (rtol = 0) by vectors the reference's OWN code produced -- its methods called
    unbound with a stand-in ``self`` (``tests/golden/make_golden.py`` -> ``float_pieces.npz``,
    ``float_flow.npz``; ``tests/test_oracle_float_golden.py``):
      - ``BaseGNNCore.forward`` SAGE loop, gnn_model.py:230-277 (pre_mp, convs[i], updates[i](cat),
        relu, dropout, running cat)                              -> ``gnn_core_hetero``
      - ``BaseGNN.forward`` gossip path, gnn_model.py:58-109 (no anchor, no pooling, post_mp)
                                                                 -> ``post_mp``
      - ``embed_to_count`` + query loop, ``graph_to_count`` / ``train_forward`` / ``test_forward`` /
        ``criterion``, lightning_model.py:176-289                -> ``head_logits``, ``count_from_logits``,
                                                                    ``train_loss_from_logits``, ``eval_loss_from_logits``
      - ``GossipCountingModel.graph_to_count`` / ``train_forward`` / ``criterion``,
        lightning_model.py:585-635                               -> ``gossip_query_loop``, ``gossip_loss_from_pred``
      - ``GossipConv.message`` / ``update`` / ``_gate_value``, gnn_model.py:335-359
                                                                 -> body of ``gossip_single_query``, ``gossip_gate``
      - ``apply_neighborhood_count`` / ``aggregate_neighborhood_count`` / ``apply_truth_from_dataset``,
        workload.py:107-112, 296-324                             -> ``apply_neighborhood_count``,
                                                                    ``aggregate_by_index``, ``apply_truth``
    Still [EXT] (held by torch_geometric 2.2.0 / torch_scatter 2.0.9, not importable here, restated
    from their documented semantics -- SURVEY.md Appendix C -- and cross-checked against independent
    dense-algebra formulas on tiny graphs, ``tests/test_oracle_model.py``):
      - ``MessagePassing.propagate`` as used at gnn_model.py:326-333, 392-394 (x_j = x[edge_index[0]],
        scatter-add at edge_index[1])
      - ``pyg.nn.to_hetero`` (lightning_model.py:371-421): per-type module copies, bipartite (x_s, x_d)
        calls, pairwise-queue sum over edge types with a common destination
      - ``remove_self_loops`` / ``to_undirected`` (gnn_model.py:246-247, 315, 389-390)
      - ``HeteroData.to_homogeneous`` + ``global_add_pool`` (gnn_model.py:88-89, 107)
      - ``torch_scatter.segment_csr`` (workload.py:136-148)
"""
