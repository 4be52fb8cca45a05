"""Query ids, atlas graphs and neighborhood extraction entry points with the reference's names
(subgraph_counting/data.py:37-58, 329-396; workload.py:1128-1671 graph_atlas_plus)."""
from __future__ import annotations

from collections import defaultdict
from typing import List

import networkx as nx
import numpy as np

from .graphs import GraphSet
from .partition import build_partition

# gen_query_ids([3,4,5]) of the reference (tests/golden/queries.json pins this list)
STANDARD_QUERY_IDS = [6, 7, 13, 14, 15, 16, 17, 18, 29, 30, 31, 34, 35, 36, 37, 38, 40, 41, 42, 43,
                      44, 45, 46, 47, 48, 49, 50, 51, 52]


def gen_query_ids(query_size: List[int]) -> List[int]:
    """Connected graph-atlas ids whose size is in ``query_size`` (data.py:37-58)."""
    query_ids = defaultdict(list)
    for i in range(6, 209):
        g = nx.graph_atlas(i)
        if nx.is_connected(g):
            query_ids[len(g)].append(i)
        if len(g) > max(query_size):
            break
    out = []
    for size, ids in query_ids.items():
        if size in query_size:
            out.extend(ids)
    return out


def graph_atlas_plus(atlas_id: int) -> nx.Graph:
    """Atlas graph by id (workload.py:1128-1671).  The reference additionally hard-codes 60 large
    (8-14 node) patterns under ids 8000-14001; those literals are not reproduced -- pass such
    patterns explicitly through ``queries=[nx.Graph, ...]``."""
    if atlas_id < 1253:
        return nx.graph_atlas(atlas_id)
    raise NotImplementedError(
        f"atlas id {atlas_id}: the reference's hand-coded 8-14 node patterns are not bundled; "
        "pass them via queries=[...]")


def k_neigh(G: nx.Graph, start_node, k):
    """BFS ball of radius k (data.py:329-338)."""
    neighs, fronts = {start_node}, {start_node}
    for _ in range(k):
        add = set()
        for n in fronts:
            add.update(G.neighbors(n))
        fronts = add - neighs
        neighs |= fronts
    return list(neighs)


def get_neigh_hetero(graph: nx.Graph, node, radius: int) -> nx.Graph:
    """Canonical neighborhood of one node as a networkx graph (data.py:375-396), computed by the
    native builder on the single graph.  Nodes are in ascending id order, canonical last."""
    nodes = list(graph.nodes)
    idx = {v: i for i, v in enumerate(nodes)}
    gs = GraphSet.from_edge_lists([(len(nodes), [(idx[a], idx[b]) for a, b in graph.edges()])])
    part = build_partition(gs, radius)
    v = idx[node]
    out = nx.Graph()
    hit = np.nonzero(part.neigh_index[:, 1] == v)[0]
    if len(hit) == 0:
        out.add_node(node, type="canonical")
        return out
    b = int(hit[0])
    c0, c1 = int(part.count_ptr[b]), int(part.count_ptr[b + 1])
    members = [nodes[int(i)] for i in part.count_orig[c0:c1]] + [node]
    for m in members:
        out.add_node(m, type="count")
    out.nodes[node]["type"] = "canonical"
    mset = set(members)
    out.add_edges_from((a, b2) for a, b2 in graph.edges() if a in mset and b2 in mset)
    return out
