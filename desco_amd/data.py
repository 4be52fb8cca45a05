"""Query ids, atlas graphs and neighborhood extraction entry points with the reference's names
(subgraph_counting/data.py:37-58, 329-396; workload.py:1128-1671 graph_atlas_plus)."""
from __future__ import annotations

from typing import List

import networkx as nx
import numpy as np

from .graphs import GraphSet
from .partition import build_partition

# gen_query_ids([3,4,5]) of the reference (tests/golden/queries.json pins this list)
STANDARD_QUERY_IDS = [6, 7, 13, 14, 15, 16, 17, 18, 29, 30, 31, 34, 35, 36, 37, 38, 40, 41, 42, 43,
                      44, 45, 46, 47, 48, 49, 50, 51, 52]


# first and one-past-last atlas id of the graphs with 3..6 nodes (the atlas is ordered by node count)
_ATLAS_SPAN = {3: (4, 8), 4: (8, 19), 5: (19, 53), 6: (53, 209)}


def gen_query_ids(query_size: List[int]) -> List[int]:
    """The connected graph-atlas patterns of the requested sizes, by ascending size then id -- the contract of
    data.py:37-58 (which scans ids 6..208, so sizes 3..6 exist and nothing else does; ``gen_query_ids([3, 4, 5])`` is
    STANDARD_QUERY_IDS, pinned by tests/golden/queries.json)."""
    ids: List[int] = []
    for size in sorted(set(int(s) for s in query_size)):
        lo, hi = _ATLAS_SPAN.get(size, (0, 0))
        if size <= 5:
            ids += [i for i in STANDARD_QUERY_IDS if lo <= i < hi]
        else:
            ids += [i for i in range(lo, hi) if nx.is_connected(nx.graph_atlas(i))]
    return ids


def graph_atlas_plus(atlas_id: int) -> nx.Graph:
    """Atlas graph by id (workload.py:1128-1671).  The reference additionally hard-codes 60 large
    (8-14 node) patterns under ids 8000-14001; those literals are not reproduced -- pass such
    patterns explicitly through ``queries=[nx.Graph, ...]``."""
    if atlas_id < 1253:
        return nx.graph_atlas(atlas_id)
    raise NotImplementedError(
        f"atlas id {atlas_id}: the reference's hand-coded 8-14 node patterns are not bundled; "
        "pass them via queries=[...]")


def add_node_feat_to_networkx(graph: nx.Graph, node_feats, node_feat_key: str = "feat"):
    """All ``len(node_feats) ** n`` labelled copies of ``graph`` (utils.py:258-272): copy i carries
    the i-th element of ``itertools.product(node_feats, repeat=n)``, node k gets its k-th entry."""
    import itertools
    n = len(graph.nodes)
    out = []
    for feats in itertools.product(node_feats, repeat=n):
        g = graph.copy()
        for k, f in enumerate(feats):
            g.nodes[k][node_feat_key] = f
        out.append(g)
    return out


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


# ------------------------------------------------------------------------------------------------
# dataset loading (subgraph_counting/data.py:91-232)
# ------------------------------------------------------------------------------------------------
_SYNTHETIC_BY_NAME = {"MUTAG": "mutag", "COX2": "cox2", "Syn_1827": "syn_1827",
                      "MSRC_21+IMDB-BINARY": "msrc_imdb"}


def _read_tu_raw(raw_dir: str, name: str) -> GraphSet:
    """TU dataset text format: <name>_A.txt ("i, j", 1-based global ids, both directions) and
    <name>_graph_indicator.txt (graph id of node i, 1-based)."""
    import os
    a = np.loadtxt(os.path.join(raw_dir, name + "_A.txt"), delimiter=",", dtype=np.int64).reshape(-1, 2) - 1
    gi = np.loadtxt(os.path.join(raw_dir, name + "_graph_indicator.txt"), dtype=np.int64).reshape(-1) - 1
    if not (np.diff(gi) >= 0).all():
        raise ValueError("graph_indicator must be sorted")
    graph_ptr = np.concatenate([[0], np.cumsum(np.bincount(gi))])
    gs = GraphSet._from_global_pairs(graph_ptr, a[:, 0], a[:, 1])
    lab = os.path.join(raw_dir, name + "_node_labels.txt")
    if os.path.exists(lab):      # PyG's TUDataset: x = one-hot node labels
        l = np.loadtxt(lab, dtype=np.int64).reshape(-1)
        l = l - l.min()
        gs = GraphSet(gs.graph_ptr, gs.rowptr, gs.col, np.eye(int(l.max()) + 1, dtype=np.float32)[l])
    return gs


def relabel(graphs: GraphSet, mode: str) -> GraphSet:
    """Per-graph node re-indexing (transforms.py:415-442 Relabel): "decreasing_degree",
    "increasing_degree" (stable w.r.t. the original order) or "random" (seed 0)."""
    rng = np.random.default_rng(0)
    out, feats = [], []
    for gi, (n, edges) in enumerate(graphs.edge_lists()):
        deg = np.zeros(n, dtype=np.int64)
        for a, b in edges:
            deg[a] += 1
            deg[b] += 1
        if mode == "random":
            order = rng.permutation(n)
        else:
            order = np.argsort(-deg if mode == "decreasing_degree" else deg, kind="stable")
        new = np.empty(n, dtype=np.int64)
        new[order] = np.arange(n)
        out.append((n, [(int(new[a]), int(new[b])) for a, b in edges]))
        if graphs.node_feat is not None:
            f = graphs.node_feat[graphs.graph_ptr[gi]:graphs.graph_ptr[gi + 1]]
            feats.append(f[order])                       # new id k holds old node order[k]
    return GraphSet.from_edge_lists(out, node_feat=feats if graphs.node_feat is not None else None)


def read_syn_edgelist(edgelist_path: str, indicator_path: str) -> GraphSet:
    """The reference's synthetic-dataset text format (``DeSCoSyntheticDataset``, data.py:644-750):
    ``<name>_edgelist.txt`` = ``# nodes edges`` + one ``u v`` line per undirected edge with global
    node ids, graph after graph; ``<name>_graph_indicator.txt`` = ``# graphs`` + the number of
    edges of every graph.  As in ``process()`` (``from_networkx`` of ``add_edges_from``), a graph's
    local node ids are the order of FIRST APPEARANCE of its nodes in its edge lines."""
    edge_counts = [int(l) for l in open(indicator_path, "rt") if l.strip() and not l.startswith("#")]
    graphs, g, local, edges = [], 0, {}, []
    with open(edgelist_path, "rt") as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            while g < len(edge_counts) and edge_counts[g] == 0:      # (the generator never emits these)
                graphs.append((0, []))
                g += 1
            a, b = line.split()[:2]
            u = local.setdefault(int(a), len(local))
            v = local.setdefault(int(b), len(local))
            edges.append((u, v))
            if len(edges) == edge_counts[g]:
                graphs.append((len(local), edges))
                g, local, edges = g + 1, {}, []
    if edges or g != len(edge_counts):
        raise ValueError(f"{edgelist_path}: edge lines do not add up to the graph indicator")
    return GraphSet.from_edge_lists(graphs)


def write_syn_edgelist(graphs: GraphSet, edgelist_path: str, indicator_path: str) -> None:
    """Writer of the same format (``download()``, data.py:666-704): global ids, ``u < v`` per line."""
    lists = graphs.edge_lists()
    with open(edgelist_path, "w") as f:
        f.write("# {:d} {:d}\n".format(graphs.num_nodes, sum(len(e) for _, e in lists)))
        base = 0
        for n, edges in lists:
            for u, v in edges:
                f.write("{} {}\n".format(base + u, base + v))
            base += n
    with open(indicator_path, "w") as f:
        f.write("# {:d}\n".format(len(lists)))
        for _, edges in lists:
            f.write("{:d}\n".format(len(edges)))


def load_data(dataset_name: str, root_folder="data", n_neighborhoods=-1, transform=None,
              train_split=0.25, val_split=0.25, test_split=0.5) -> GraphSet:
    """Target graphs by name, with the reference's name mini-DSL (data.py:104-137, 206-227):
    ``_train`` / ``_val`` / ``_test`` select the 25/25/50 % split of ``random.seed(0);
    random.shuffle`` and ``_decreaseByDegree`` / ``_increaseByDegree`` / ``_random`` relabel nodes.

    Source, in order: TU text files under ``<root_folder>/<name>/raw`` (no download is attempted:
    there is no network), else a seeded shape-matched synthetic stand-in (desco_amd.synthetic)."""
    import os
    import random
    import warnings
    split = None
    for tag in ("train", "val", "test"):
        if tag in dataset_name:
            split = tag
            dataset_name = dataset_name.replace("_" + tag, "")
            break
    mode = None
    for suffix, m in (("_decreaseByDegree", "decreasing_degree"), ("_increaseByDegree", "increasing_degree"),
                      ("_random", "random")):
        if suffix in dataset_name:
            mode = m
            dataset_name = dataset_name.replace(suffix, "")
    raw = os.path.join(root_folder, dataset_name, "raw")
    syn_name = None
    if dataset_name.split("_")[0] == "Syn" and dataset_name.split("_")[1:2]:
        # "Syn_<graph_num>" -> the reference's raw file names (data.py:188-197, 636-638)
        syn_name = "Synthetic_size_min_10_max_500_graph_num_{:d}".format(int(dataset_name.split("_")[1]))
    if os.path.exists(os.path.join(raw, dataset_name + "_A.txt")):
        graphs = _read_tu_raw(raw, dataset_name)
    elif syn_name and os.path.exists(os.path.join(raw, syn_name + "_edgelist.txt")):
        graphs = read_syn_edgelist(os.path.join(raw, syn_name + "_edgelist.txt"),
                                   os.path.join(raw, syn_name + "_graph_indicator.txt"))
    elif dataset_name in _SYNTHETIC_BY_NAME:
        from . import synthetic
        warnings.warn(f"{dataset_name}: no raw files under {raw}; using the seeded shape-matched "
                      "synthetic stand-in (desco_amd.synthetic)")
        graphs = synthetic.WORKLOADS[_SYNTHETIC_BY_NAME[dataset_name]]()
    else:
        raise FileNotFoundError(f"{dataset_name}: place TU-format text files under {raw} "
                                "(datasets are not downloaded: no network)")
    if mode is not None:
        graphs = relabel(graphs, mode)
    if split is None:
        return graphs
    G = graphs.num_graphs
    idx = list(range(G))
    random.seed(0)
    random.shuffle(idx)                                  # same permutation as shuffling the graphs
    train_len, val_len = int(G * train_split), int(G * val_split)
    sel = {"train": idx[:train_len], "val": idx[train_len:train_len + val_len],
           "test": idx[train_len + val_len:]}[split]
    lists = graphs.edge_lists()
    feats = None
    if graphs.node_feat is not None:
        feats = [graphs.node_feat[graphs.graph_ptr[i]:graphs.graph_ptr[i + 1]] for i in sel]
    return GraphSet.from_edge_lists([lists[i] for i in sel], node_feat=feats)
