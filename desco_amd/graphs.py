"""Flat CSR containers for target graphs (replaces PyG ``Data``/``InMemoryDataset`` storage on the
hot path; the reference keeps graphs as PyG objects and converts to networkx, workload.py:222-233).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import numpy as np


class GraphSet:
    """G undirected simple graphs in one CSR over global node ids.

    graph g owns nodes ``graph_ptr[g] .. graph_ptr[g+1]-1`` (ascending id = the reference's
    networkx node order).  Adjacency is symmetric, loop free, deduplicated, rows sorted ascending
    (== remove_self_loops + to_undirected + coalesce of gnn_model.py:246-247).
    """

    def __init__(self, graph_ptr: np.ndarray, rowptr: np.ndarray, col: np.ndarray,
                 node_feat: "np.ndarray | None" = None):
        self.graph_ptr = np.ascontiguousarray(graph_ptr, dtype=np.int64)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        assert self.rowptr.shape[0] == self.graph_ptr[-1] + 1
        # optional node features [num_nodes, F] fp32 (the PyG ``x`` the reference feeds as "feat" when
        # --use_node_feature is on, workload.py:222-233); None == ZeroNodeFeat
        self.node_feat = None if node_feat is None else np.ascontiguousarray(node_feat, dtype=np.float32)
        if self.node_feat is not None:
            assert self.node_feat.ndim == 2 and self.node_feat.shape[0] == self.graph_ptr[-1]

    # ---- constructors ---------------------------------------------------------------------
    @classmethod
    def from_edge_lists(cls, graphs: Sequence[Tuple[int, Iterable[Tuple[int, int]]]],
                        node_feat=None) -> "GraphSet":
        """graphs: sequence of (num_nodes, undirected edge pairs with graph-local ids);
        ``node_feat``: optional [total nodes, F] array or a list of per-graph [n, F] arrays."""
        sizes = np.array([n for n, _ in graphs], dtype=np.int64)
        graph_ptr = np.concatenate([[0], np.cumsum(sizes)])
        srcs, dsts = [], []
        for g, (n, edges) in enumerate(graphs):
            e = np.asarray(list(edges), dtype=np.int64).reshape(-1, 2)
            if e.size:
                if e.min() < 0 or e.max() >= n:
                    raise ValueError(f"graph {g}: edge endpoint out of range")
                srcs.append(e[:, 0] + graph_ptr[g])
                dsts.append(e[:, 1] + graph_ptr[g])
        src = np.concatenate(srcs) if srcs else np.zeros(0, dtype=np.int64)
        dst = np.concatenate(dsts) if dsts else np.zeros(0, dtype=np.int64)
        gs = cls._from_global_pairs(graph_ptr, src, dst)
        if node_feat is not None:
            nf = np.concatenate([np.asarray(f, dtype=np.float32).reshape(len(f), -1) for f in node_feat]) \
                if isinstance(node_feat, (list, tuple)) else np.asarray(node_feat, dtype=np.float32)
            gs = cls(gs.graph_ptr, gs.rowptr, gs.col, nf)
        return gs

    @classmethod
    def _from_global_pairs(cls, graph_ptr, src, dst) -> "GraphSet":
        n = int(graph_ptr[-1])
        keep = src != dst
        src, dst = src[keep], dst[keep]
        r = np.concatenate([src, dst])
        c = np.concatenate([dst, src])
        key = np.unique(r * n + c) if n else np.zeros(0, dtype=np.int64)
        r, c = key // max(n, 1), key % max(n, 1)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(rowptr, r + 1, 1)
        rowptr = np.cumsum(rowptr)
        return cls(graph_ptr, rowptr, c.astype(np.int32))

    @classmethod
    def from_networkx(cls, graphs) -> "GraphSet":
        out = []
        for g in graphs:
            nodes = list(g.nodes)
            idx = {v: i for i, v in enumerate(nodes)}
            out.append((len(nodes), [(idx[a], idx[b]) for a, b in g.edges()]))
        feats = None
        if len(out) and all("feat" in g.nodes[v] for g in graphs for v in g.nodes):
            feats = [np.asarray([np.asarray(g.nodes[v]["feat"], dtype=np.float32).reshape(-1) for v in g.nodes])
                     for g in graphs]
        return cls.from_edge_lists(out, node_feat=feats)

    # ---- views ------------------------------------------------------------------------------
    @property
    def num_graphs(self) -> int:
        return len(self.graph_ptr) - 1

    @property
    def num_nodes(self) -> int:
        return int(self.graph_ptr[-1])

    @property
    def num_directed_edges(self) -> int:
        return int(self.rowptr[-1])

    def __len__(self):
        return self.num_graphs

    def edge_lists(self) -> List[Tuple[int, List[Tuple[int, int]]]]:
        """Back to [(n, undirected local edges a<b)] (tests, CPU baseline input)."""
        out = []
        for g in range(self.num_graphs):
            b0, b1 = int(self.graph_ptr[g]), int(self.graph_ptr[g + 1])
            edges = []
            for v in range(b0, b1):
                for w in self.col[self.rowptr[v]:self.rowptr[v + 1]]:
                    if v < w:
                        edges.append((v - b0, int(w) - b0))
            out.append((b1 - b0, edges))
        return out

    def subset(self, g0: int, g1: int) -> "GraphSet":
        """Graphs [g0, g1) as a new GraphSet (re-based ids)."""
        n0, n1 = int(self.graph_ptr[g0]), int(self.graph_ptr[g1])
        e0, e1 = int(self.rowptr[n0]), int(self.rowptr[n1])
        return GraphSet(self.graph_ptr[g0:g1 + 1] - n0, self.rowptr[n0:n1 + 1] - e0,
                        self.col[e0:e1] - n0,
                        None if self.node_feat is None else self.node_feat[n0:n1])

    def replicate(self, times: int) -> "GraphSet":
        """The same graphs ``times`` times over (dataset replication for saturation benchmarks)."""
        if times == 1:
            return self
        n, e = self.num_nodes, self.num_directed_edges
        gp = np.concatenate([self.graph_ptr[:-1] + k * n for k in range(times)] + [[times * n]])
        rp = np.concatenate([self.rowptr[:-1] + k * e for k in range(times)] + [[times * e]])
        col = np.concatenate([self.col.astype(np.int64) + k * n for k in range(times)])
        return GraphSet(gp, rp, col.astype(np.int32),
                        None if self.node_feat is None else np.tile(self.node_feat, (times, 1)))

    def node_graph_ids(self) -> np.ndarray:
        return np.repeat(np.arange(self.num_graphs, dtype=np.int64), np.diff(self.graph_ptr))
