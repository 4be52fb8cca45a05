"""Exact canonical-count ground truth via the native enumerator (replaces the reference's VF2
process pool, workload.py:551-726)."""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np
import torch

from . import _lib
from .graphs import GraphSet


def _flatten_queries(queries: Sequence):
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
    return flat, q_nodes, q_edge_ptr, q_edges


# the device path keeps one adjacency bitset row per node: bounded so that a huge single graph
# (n^2 / 8 bytes) goes to the host enumerator instead
_DEVICE_BITSET_LIMIT_WORDS = 1 << 28        # 2 GiB


def _device_eligible(graphs: GraphSet, q_nodes: np.ndarray) -> bool:
    if not torch.cuda.is_available() or len(q_nodes) == 0 or len(q_nodes) > 32:
        return False
    if q_nodes.min() < 2 or q_nodes.max() > 5:
        return False
    n = np.diff(graphs.graph_ptr).astype(np.int64)
    return int((n * ((n + 63) // 64)).sum()) <= _DEVICE_BITSET_LIMIT_WORDS


def canonical_counts_device(graphs: GraphSet, queries: Sequence, device="cuda") -> torch.Tensor:
    """The counts of ``canonical_counts`` computed on the MI355X (csrc/groundtruth_dev.hip):
    queries of 2..5 nodes, at most 32, pairwise non-isomorphic.  Returns a [num_nodes, num_queries]
    int64 tensor on ``device``."""
    _, q_nodes, q_edge_ptr, q_edges = _flatten_queries(queries)
    L = _lib.lib()
    table = np.empty(1098, dtype=np.int16)
    kmax = ctypes.c_int(0)
    _lib.check(L.desco_canonical_class_table(q_nodes.ctypes.data, q_edge_ptr.ctypes.data,
                                             q_edges.ctypes.data if len(q_edges) else None,
                                             len(q_nodes), table.ctypes.data, ctypes.byref(kmax)),
               "desco_canonical_class_table")
    dev = torch.device(device)
    n = np.diff(graphs.graph_ptr).astype(np.int64)
    words = n * ((n + 63) // 64)
    bit_off = np.concatenate([[0], np.cumsum(words)]).astype(np.int64)
    N, G, Q = graphs.num_nodes, graphs.num_graphs, len(q_nodes)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)      # noqa: E731
    graph_ptr, rowptr = t(graphs.graph_ptr, np.int64), t(graphs.rowptr, np.int64)
    col, node_graph = t(graphs.col, np.int32), t(graphs.node_graph_ids(), np.int32)
    bit_off_d, cls = t(bit_off[:-1] if G else bit_off, np.int64), t(table, np.int16)
    bits = torch.empty(max(int(bit_off[-1]), 1), dtype=torch.int64, device=dev)
    out = torch.empty((N, Q), dtype=torch.int64, device=dev)
    if N == 0 or Q == 0:
        return out.zero_()
    with torch.cuda.device(dev):
        _lib.check(L.desco_canonical_counts_dev(
            graph_ptr.data_ptr(), G, N, rowptr.data_ptr(), int(graphs.col.shape[0]),
            col.data_ptr() if col.numel() else None, node_graph.data_ptr(), bit_off_d.data_ptr(),
            bits.data_ptr(), int(bit_off[-1]), cls.data_ptr(), int(kmax.value), Q, out.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream), "desco_canonical_counts_dev")
    return out


def canonical_counts(graphs: GraphSet, queries: Sequence, num_threads: int = 0,
                     backend: str = "auto") -> torch.Tensor:
    """[num_nodes, num_queries] float tensor of canonical counts (the reference stores doubles).
    ``queries``: networkx graphs or (n, edges) pairs, connected, 2..6 nodes.
    ``backend``: "host" (OpenMP enumerator), "device" (HIP kernel), or "auto": the device when a GPU
    is present and the queries fit its path (2..5 nodes, <= 32, distinct classes), else the host."""
    flat, q_nodes, q_edge_ptr, q_edges = _flatten_queries(queries)
    if backend == "device" or (backend == "auto" and _device_eligible(graphs, q_nodes)):
        try:
            return canonical_counts_device(graphs, queries).cpu().double()
        except RuntimeError as e:
            if backend == "device" or "isomorphic" not in str(e):
                raise                           # (duplicate query classes: host path below)
    out = np.zeros((graphs.num_nodes, len(flat)), dtype=np.int64)
    L = _lib.lib()
    _lib.check(L.desco_canonical_counts(graphs.graph_ptr.ctypes.data, graphs.num_graphs,
                                        graphs.rowptr.ctypes.data, graphs.col.ctypes.data,
                                        q_nodes.ctypes.data, q_edge_ptr.ctypes.data,
                                        q_edges.ctypes.data if len(q_edges) else None, len(flat),
                                        num_threads, out.ctypes.data), "desco_canonical_counts")
    return torch.from_numpy(out).double()


def canonical_counts_labelled(graphs: GraphSet, queries: Sequence, node_feat_key: str = "feat") -> torch.Tensor:
    """Canonical counts of LABELLED queries (--use_node_feature): the reference's own procedure,
    networkx VF2 with ``node_match`` on the feature (workload.py:327-348) divided by the labelled
    symmetry factor (data.py:61-68).  The native enumerators count unlabelled patterns; labelled
    ground truth is an offline, one-time step and stays on the host like the reference's."""
    import networkx as nx
    if graphs.node_feat is None:
        raise ValueError("labelled ground truth needs GraphSet.node_feat")
    match = lambda a, b: a[node_feat_key] == b[node_feat_key]     # noqa: E731
    targets = []
    for g, (n, edges) in enumerate(graphs.edge_lists()):
        t = nx.Graph()
        base = int(graphs.graph_ptr[g])
        for v in range(n):
            t.add_node(v, **{node_feat_key: [float(x) for x in graphs.node_feat[base + v]]})
        t.add_edges_from(edges)
        targets.append(t)
    out = torch.zeros((graphs.num_nodes, len(queries)), dtype=torch.double)
    for qi, q in enumerate(queries):
        qq = nx.Graph()
        for v in q.nodes:
            qq.add_node(v, **{node_feat_key: [float(x) for x in np.asarray(q.nodes[v][node_feat_key]).reshape(-1)]})
        qq.add_edges_from(q.edges())
        sym = sum(1 for _ in nx.algorithms.isomorphism.GraphMatcher(qq, qq, node_match=match)
                  .subgraph_isomorphisms_iter())
        for g, t in enumerate(targets):
            base = int(graphs.graph_ptr[g])
            gm = nx.algorithms.isomorphism.GraphMatcher(t, qq, node_match=match)
            for vmap in gm.subgraph_isomorphisms_iter():
                out[base + max(vmap.keys()), qi] += 1
        out[:, qi] /= sym
    return out
