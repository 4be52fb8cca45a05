"""Exact canonical-count ground truth via the native enumerator (replaces the reference's VF2
process pool, workload.py:551-726)."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import _lib
from .graphs import GraphSet


def canonical_counts(graphs: GraphSet, queries: Sequence, num_threads: int = 0) -> torch.Tensor:
    """[num_nodes, num_queries] float tensor of canonical counts (the reference stores doubles).
    ``queries``: networkx graphs or (n, edges) pairs, connected, 2..6 nodes."""
    flat = []
    for q in queries:
        if hasattr(q, "nodes"):
            nodes = list(q.nodes)
            idx = {v: i for i, v in enumerate(nodes)}
            flat.append((len(nodes), [(idx[a], idx[b]) for a, b in q.edges()]))
        else:
            flat.append((int(q[0]), [tuple(e) for e in q[1]]))
    q_nodes = np.array([n for n, _ in flat], dtype=np.int32)
    q_edge_ptr = np.concatenate([[0], np.cumsum([len(e) for _, e in flat])]).astype(np.int32)
    q_edges = np.array([x for _, es in flat for e in es for x in e], dtype=np.int32)
    out = np.zeros((graphs.num_nodes, len(flat)), dtype=np.int64)
    L = _lib.lib()
    _lib.check(L.desco_canonical_counts(graphs.graph_ptr.ctypes.data, graphs.num_graphs,
                                        graphs.rowptr.ctypes.data, graphs.col.ctypes.data,
                                        q_nodes.ctypes.data, q_edge_ptr.ctypes.data,
                                        q_edges.ctypes.data if len(q_edges) else None, len(flat),
                                        num_threads, out.ctypes.data), "desco_canonical_counts")
    return torch.from_numpy(out).double()
