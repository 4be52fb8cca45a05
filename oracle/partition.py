"""Oracle, integer path: canonical partition -> hetero neighborhood -> triangle split -> collate.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain Python/numpy restatement of the reference's
algorithm, small inputs only.  Every function cites the reference lines it follows
(paths relative to /root/reference).

Conventions used throughout this repo (and fixed here):
  * a target graph is ``(n, edges)``: nodes ``0..n-1``, ``edges`` an iterable of undirected pairs;
  * inside a neighborhood, nodes are ordered by ASCENDING original id, so the canonical node
    (the maximum id, data.py:385) is always last.  The reference's order is whatever networkx's
    ``FilterAtlas.__iter__`` yields, which for neighborhoods smaller than half the graph is
    CPython ``set`` iteration order (not ascending in general; measured in this container).
    Node SETS, edge SETS, ``nx_neighs_index`` and ``nx_neighs_indicator`` are order independent
    and are checked bit-exactly against the reference; local ids are defined by this repo.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

NODE_TYPES = ("count", "canonical")
# metadata order of lightning_model.py:376-383 (to_hetero_old, tconv_target=True)
EDGE_TYPES = (
    ("count", "union_triangle", "count"),
    ("count", "union_tride", "count"),
    ("count", "union_triangle", "canonical"),
    ("count", "union_tride", "canonical"),
    ("canonical", "union_triangle", "count"),
    ("canonical", "union_tride", "count"),
)
QUERY_EDGE_TYPES = (
    ("union_node", "union_triangle", "union_node"),
    ("union_node", "union_tride", "union_node"),
)


def adjacency(n: int, edges: Iterable[Tuple[int, int]]) -> List[List[int]]:
    """Undirected simple adjacency lists (self loops and duplicates dropped), sorted ascending."""
    adj = [set() for _ in range(n)]
    for a, b in edges:
        a, b = int(a), int(b)
        if a == b:
            continue
        adj[a].add(b)
        adj[b].add(a)
    return [sorted(s) for s in adj]


def k_neigh(adj: Sequence[Sequence[int]], start: int, k: int) -> set:
    """BFS ball of radius k in the FULL graph.  Follows data.py:329-338."""
    neighs = {start}
    fronts = {start}
    for _ in range(k):
        add_node = set()
        for n in fronts:
            add_node.update(adj[n])
        fronts = add_node - neighs
        neighs = neighs | fronts
    return neighs


def get_neigh_hetero(adj, node: int, radius: int):
    """Canonical neighborhood of ``node``.  Follows data.py:375-396.

    ball(radius) in the full graph -> keep ids <= node (filter applied AFTER the BFS, :385) ->
    induced subgraph -> connected component containing ``node`` (:387-390).
    Returns (nodes ascending [canonical last], undirected induced edges (a<b) sorted).
    """
    keep = {v for v in k_neigh(adj, node, radius) if v <= node}
    comp = {node}
    stack = [node]
    while stack:
        u = stack.pop()
        for w in adj[u]:
            if w in keep and w not in comp:
                comp.add(w)
                stack.append(w)
    nodes = sorted(comp)
    edges = sorted((a, b) for a in nodes for b in adj[a] if b in comp and a < b)
    return nodes, edges


def neighborhood_dataset(graphs, depth: int):
    """Driver loop of NeighborhoodDataset.process, workload.py:243-260.

    Returns (nx_neighs_index int[#neigh,2], nx_neighs_indicator bool[#nodes], neighs) where
    ``neighs[k] = (nodes, edges)`` in ORIGINAL ids.  A neighborhood with 0 edges is skipped and
    its indicator is False (:252-256).
    """
    index, indicator, neighs = [], [], []
    for gid, (n, edges) in enumerate(graphs):
        adj = adjacency(n, edges)
        for node in range(n):
            nodes, nedges = get_neigh_hetero(adj, node, depth)
            if len(nedges) == 0:
                indicator.append(False)
            else:
                indicator.append(True)
                index.append((gid, node))
                neighs.append((nodes, nedges))
    return (np.array(index, dtype=np.int64).reshape(-1, 2),
            np.array(indicator, dtype=bool), neighs)


def networkx_to_hetero(nodes: Sequence[int], edges, canonical=None) -> Dict:
    """nx neighborhood -> per-type local ids and per-type directed edge lists.

    Follows transforms.py:319-412 (+ the missing-type fill of workload.py:275-282): node ids per
    type in node order (:342-348); every undirected edge yields both directions (``to_directed``,
    :331); edge type = (type(src), "union", type(dst)) (:351-367).  ``canonical=None`` gives the
    single-type query graph ("union_node", :343-344).  node_feature is zeros [n_t,1] (:380-384).
    """
    if canonical is None:
        ntype = {v: "union_node" for v in nodes}
        types = ("union_node",)
    else:
        ntype = {v: ("canonical" if v == canonical else "count") for v in nodes}
        types = NODE_TYPES
    local: Dict[str, Dict[int, int]] = {t: {} for t in types}
    for v in nodes:
        local[ntype[v]][v] = len(local[ntype[v]])
    adjl = {v: [] for v in nodes}
    for a, b in edges:
        adjl[a].append(b)
        adjl[b].append(a)
    edge_lists: Dict[Tuple[str, str, str], List[Tuple[int, int]]] = {}
    if canonical is None:
        edge_lists[("union_node", "union", "union_node")] = []
    else:
        for (s, d) in (("count", "count"), ("count", "canonical"), ("canonical", "count")):
            edge_lists[(s, "union", d)] = []
    for n0 in nodes:
        for n1 in sorted(adjl[n0]):
            et = (ntype[n0], "union", ntype[n1])
            edge_lists[et].append((local[ntype[n0]][n0], local[ntype[n1]][n1]))
    return {
        "num_nodes": {t: len(local[t]) for t in types},
        "orig_ids": {t: [v for v in nodes if ntype[v] == t] for t in types},
        "edge_index": {et: np.array(el, dtype=np.int64).reshape(-1, 2).T
                       for et, el in edge_lists.items()},
    }


def to_tconv_hetero(h: Dict) -> Dict:
    """Triangle split of every edge type.  Follows transforms.py:180-255 / :258-289.

    Homogenise (node types in store order, :262-267), A = 0/1 adjacency, T = A*(A@A) + A
    (:201-211); an edge is a *triangle* edge iff T > 1 (:221), i.e. its endpoints share at least
    one neighbour inside the neighborhood.  (s,"union",d) -> (s,"union_triangle",d) +
    (s,"union_tride",d), intra-type edge order preserved (:241-250).
    """
    types = list(h["num_nodes"].keys())
    off, c = {}, 0
    for t in types:
        off[t] = c
        c += h["num_nodes"][t]
    A = np.zeros((c, c), dtype=np.int64)
    for (s, _, d), ei in h["edge_index"].items():
        A[ei[0] + off[s], ei[1] + off[d]] = 1
    T = A * (A @ A) + A
    out = {"num_nodes": dict(h["num_nodes"]), "orig_ids": h["orig_ids"], "edge_index": {}}
    for (s, r, d), ei in h["edge_index"].items():
        tri = T[ei[0] + off[s], ei[1] + off[d]] > 1 if ei.shape[1] else np.zeros(0, dtype=bool)
        out["edge_index"][(s, r + "_triangle", d)] = ei[:, tri]
        out["edge_index"][(s, r + "_tride", d)] = ei[:, ~tri]
    return out


def collate(items: Sequence[Dict]) -> Dict:
    """PyG hetero collate [EXT: torch_geometric 2.2.0 Batch.from_data_list], SURVEY App. C.

    Per node type concatenation; per edge type ``edge_index`` offset by the cumulative node
    counts of its own source / destination types; ``batch`` vector per node type.
    """
    types = list(items[0]["num_nodes"].keys())
    etypes = list(items[0]["edge_index"].keys())
    cum = {t: 0 for t in types}
    batch = {t: [] for t in types}
    eis = {et: [] for et in etypes}
    for g, it in enumerate(items):
        for et in etypes:
            ei = it["edge_index"][et]
            s, _, d = et
            eis[et].append(ei + np.array([[cum[s]], [cum[d]]], dtype=np.int64))
        for t in types:
            batch[t].extend([g] * it["num_nodes"][t])
            cum[t] += it["num_nodes"][t]
    return {
        "num_graphs": len(items),
        "num_nodes": cum,
        "batch": {t: np.array(batch[t], dtype=np.int64) for t in types},
        "edge_index": {et: (np.concatenate(eis[et], axis=1) if eis[et] else
                            np.zeros((2, 0), dtype=np.int64)) for et in etypes},
    }


def neighborhood_batch(neighs, tconv: bool = True) -> Dict:
    """(nodes, edges) neighborhoods -> collated hetero batch (A2 + A3 + A4 of SURVEY 8a)."""
    items = []
    for nodes, edges in neighs:
        h = networkx_to_hetero(nodes, edges, canonical=nodes[-1])
        items.append(to_tconv_hetero(h) if tconv else h)
    return collate(items)


def query_batch(queries, tconv: bool = True) -> Dict:
    """The query graphs as one single-type batch (lightning_model.py:37-87, 291-309)."""
    items = []
    for n, edges in queries:
        h = networkx_to_hetero(list(range(n)), sorted(tuple(sorted(e)) for e in edges))
        items.append(to_tconv_hetero(h) if tconv else h)
    return collate(items)


def gossip_edge_index(n: int, edge_index: np.ndarray):
    """GossipConv preprocessing, gnn_model.py:246-248.

    remove_self_loops, to_undirected (= concat both directions, coalesce: sort by (row, col),
    dedupe [EXT]) and ``edge_weight = row < col``.
    """
    row, col = edge_index
    keep = row != col
    row, col = row[keep], col[keep]
    r = np.concatenate([row, col])
    c = np.concatenate([col, row])
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    return np.stack([r, c]), r < c


def canonical_counts_bruteforce(n: int, edges, queries) -> np.ndarray:
    """Induced-subgraph canonical counts, symmetry-normalised (workload.py:327-348 divided by
    data.py:61-67): count[v, q] = #{node subsets S with max(S)=v, G[S] isomorphic to query q}.
    Exponential brute force over connected subsets; only for tiny graphs / tests.
    """
    import itertools
    adj = adjacency(n, edges)
    aset = [set(a) for a in adj]

    def canon_form(k, es):
        best = None
        for perm in itertools.permutations(range(k)):
            f = tuple(sorted(tuple(sorted((perm[a], perm[b]))) for a, b in es))
            if best is None or f < best:
                best = f
        return best

    qforms = [(qn, canon_form(qn, qe)) for qn, qe in queries]
    out = np.zeros((n, len(queries)), dtype=np.int64)
    sizes = sorted({qn for qn, _ in queries})
    for k in sizes:
        for S in itertools.combinations(range(n), k):
            idx = {v: i for i, v in enumerate(S)}
            es = [(idx[a], idx[b]) for a in S for b in aset[a] if b in idx and a < b]
            if len(es) < k - 1:
                continue
            f = canon_form(k, es)
            for qi, (qn, qf) in enumerate(qforms):
                if qn == k and qf == f:
                    out[S[-1], qi] += 1
    return out
