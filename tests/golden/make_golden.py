#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own functions.

Run ONLY in the build container (needs /root/reference, read-only):
    cd tests/golden && python -B make_golden.py
The outputs (*.json) are data: seeded inputs and the reference's outputs on them.  No reference
source is copied.  The GPU box never runs this script (it has no /root/reference).

Reference functions executed (unmodified, via _ref_import's inert third-party stubs):
  data.py:37-58    gen_query_ids          data.py:61-88   SymmetricFactor / GenVMap
  data.py:329-338  k_neigh                data.py:375-396 get_neigh_hetero
  workload.py:327-348 MatchSubgraphWorker workload.py:1128-1671 graph_atlas_plus
  analysis.py:22-83 norm_mse / mse / mae  config.py:185-400 parse_* defaults
"""
import argparse
import io
import json
import contextlib
import os

import networkx as nx
import numpy as np

import _ref_import

HERE = os.path.dirname(os.path.abspath(__file__))
DEPTH = 4


def seeded_graphs():
    """~24 small graphs covering tree / ring / dense / star / disconnected-after-filter cases."""
    gs = []
    # SURVEY Appendix B toy case
    gs.append(("toy_appendix_b", 8, [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (3, 4), (4, 5), (5, 6), (6, 7), (7, 2)]))
    gs.append(("path12", 12, [(i, i + 1) for i in range(11)]))
    gs.append(("ring9", 9, [(i, (i + 1) % 9) for i in range(9)]))
    gs.append(("star_hub_low", 10, [(0, i) for i in range(1, 10)]))
    gs.append(("star_hub_high", 10, [(9, i) for i in range(0, 9)]))
    gs.append(("k6", 6, [(i, j) for i in range(6) for j in range(i + 1, 6)]))
    gs.append(("two_components", 11, [(0, 1), (1, 2), (2, 0), (3, 4), (4, 5), (5, 6), (6, 3), (7, 8), (9, 10)]))
    gs.append(("isolated_nodes", 7, [(1, 2), (2, 5)]))
    # path whose low-id part is only reachable through a higher id (disconnected after the id filter)
    gs.append(("filter_disconnects", 8, [(0, 7), (7, 1), (1, 6), (6, 2), (2, 5), (5, 3), (3, 4)]))
    # long ring: retained nodes can be > DEPTH hops from the canonical node inside the neighborhood
    gs.append(("ring20_perm", 20, [((7 * i) % 20, (7 * (i + 1)) % 20) for i in range(20)]))
    rng = np.random.default_rng(20240817)
    for k, (n, m) in enumerate([(15, 20), (25, 30), (30, 60), (40, 44), (41, 45), (60, 70), (60, 150), (35, 34)]):
        g = nx.gnm_random_graph(n, m, seed=int(rng.integers(1 << 30)))
        gs.append((f"gnm_{n}_{m}", n, sorted(g.edges())))
    for k, (n, mm) in enumerate([(30, 2), (50, 3)]):
        g = nx.barabasi_albert_graph(n, mm, seed=int(rng.integers(1 << 30)))
        perm = rng.permutation(n)
        gs.append((f"ba_{n}_{mm}_relabel", n, sorted((int(min(perm[a], perm[b])), int(max(perm[a], perm[b]))) for a, b in g.edges())))
    g = nx.random_labeled_tree(45, seed=7)
    gs.append(("tree45", 45, sorted(g.edges())))
    g = nx.grid_2d_graph(5, 6)
    g = nx.convert_node_labels_to_integers(g)
    gs.append(("grid5x6", 30, sorted(g.edges())))
    # larger graphs: exercise the reference's hash-ordered node iteration (order-independent checks)
    g = nx.gnm_random_graph(200, 260, seed=11)
    gs.append(("gnm_200_260", 200, sorted(g.edges())))
    g = nx.connected_watts_strogatz_graph(120, 4, 0.2, seed=5)
    gs.append(("ws_120", 120, sorted(g.edges())))
    return gs


def to_nx(n, edges):
    G = nx.Graph()
    G.add_nodes_from(range(n))          # pyg.utils.to_networkx adds nodes 0..n-1 in order
    G.add_edges_from(edges)
    return G


def main():
    ref_data, ref_workload, ref_analysis, ref_config = _ref_import.load()

    # ---- (ii) queries -------------------------------------------------------------------
    query_ids = ref_data.gen_query_ids([3, 4, 5])
    queries = []
    for qid in query_ids:
        q = ref_workload.graph_atlas_plus(qid)
        queries.append({
            "atlas_id": int(qid), "n": q.number_of_nodes(),
            "edges": sorted((int(min(a, b)), int(max(a, b))) for a, b in q.edges()),
            "diameter": int(nx.diameter(q)),
            "symmetry_factor": int(ref_data.SymmetricFactor(q)),
        })
    with open(os.path.join(HERE, "queries.json"), "w") as f:
        json.dump({"query_ids": [int(i) for i in query_ids], "queries": queries}, f)

    # ---- (i) canonical partition ----------------------------------------------------------
    out = {"depth": DEPTH, "graphs": []}
    for name, n, edges in seeded_graphs():
        G = to_nx(n, edges)
        index, indicator, neighs = [], [], []
        ascending, canon_last = True, True
        for node in G.nodes:
            ng = ref_data.get_neigh_hetero(G, node, DEPTH)
            order = [int(v) for v in ng.nodes]
            if len(ng.edges) == 0:                      # workload.py:252-256
                indicator.append(False)
                continue
            indicator.append(True)
            index.append(int(node))
            ascending &= order == sorted(order)
            canon_last &= order[-1] == node
            types = {int(v): ng.nodes[v]["type"] for v in ng.nodes}
            assert types[node] == "canonical" and sum(t == "canonical" for t in types.values()) == 1
            neighs.append({
                "canonical": int(node),
                "nodes": sorted(order),
                "edges": sorted((int(min(a, b)), int(max(a, b))) for a, b in ng.edges()),
                "ref_node_order": order,
            })
        out["graphs"].append({
            "name": name, "n": n, "edges": [[int(a), int(b)] for a, b in edges],
            "indicator": indicator, "index_nodes": index, "neighs": neighs,
            "ref_order_is_ascending": bool(ascending), "ref_canonical_is_last": bool(canon_last),
        })
    with open(os.path.join(HERE, "partition_golden.json"), "w") as f:
        json.dump(out, f)

    # ---- (iii) canonical ground-truth counts ---------------------------------------------------
    nx_queries = [ref_workload.graph_atlas_plus(q) for q in query_ids]
    sym = [q["symmetry_factor"] for q in queries]
    truth = []
    for name, n, edges in seeded_graphs():
        if n > 41:
            continue
        G = to_nx(n, edges)
        cnt = np.zeros((n, len(query_ids)), dtype=np.int64)
        for qi, q in enumerate(nx_queries):
            _, _, items = ref_workload.MatchSubgraphWorker((0, G, qi, q, None))
            for node, c in items:
                assert c % sym[qi] == 0
                cnt[node, qi] = c // sym[qi]
        truth.append({"name": name, "count": cnt.tolist()})
    with open(os.path.join(HERE, "canonical_counts.json"), "w") as f:
        json.dump(truth, f)

    # ---- (iv) metrics --------------------------------------------------------------------------
    rng = np.random.default_rng(3)
    pred = rng.gamma(2.0, 5.0, size=(37, 29)).astype(np.float32)
    tru = np.round(rng.gamma(2.0, 5.0, size=(37, 29))).astype(np.float32)
    groups = [[0, 1], list(range(2, 8)), list(range(8, 29))]
    with contextlib.redirect_stdout(io.StringIO()):
        m = {
            "norm_mse": ref_analysis.norm_mse(pred, tru, groups),
            "mse": ref_analysis.mse(pred, tru, groups),
            "mae": [float(v) for v in ref_analysis.mae(pred, tru, groups)],
            "norm_mse_all": ref_analysis.norm_mse(pred, tru),
        }
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump({"pred": pred.tolist(), "truth": tru.tolist(), "groups": groups, **m}, f)

    # ---- (v) config defaults -------------------------------------------------------------------
    parser = argparse.ArgumentParser()
    ref_config.parse_neighborhood(parser)
    ref_config.parse_gossip(parser)
    ref_config.parse_optimizer(parser)
    ns = parser.parse_args([])

    def jsonable(v):
        return v if isinstance(v, (int, float, str, bool, type(None), list)) else repr(v)
    with open(os.path.join(HERE, "config_defaults.json"), "w") as f:
        json.dump({k: jsonable(v) for k, v in sorted(vars(ns).items())}, f, indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
