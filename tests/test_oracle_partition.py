"""Oracle integer path vs. golden vectors produced by the reference's own functions."""
import numpy as np
import pytest

from oracle import partition as P


def test_partition_matches_reference_golden(partition_golden):
    depth = partition_golden["depth"]
    for g in partition_golden["graphs"]:
        graphs = [(g["n"], [tuple(e) for e in g["edges"]])]
        index, indicator, neighs = P.neighborhood_dataset(graphs, depth)
        assert indicator.tolist() == g["indicator"], g["name"]
        assert index[:, 1].tolist() == g["index_nodes"], g["name"]
        assert (index[:, 0] == 0).all()
        assert len(neighs) == len(g["neighs"])
        for (nodes, edges), ref in zip(neighs, g["neighs"]):
            assert nodes == ref["nodes"], g["name"]
            assert nodes[-1] == ref["canonical"]
            assert [list(e) for e in edges] == ref["edges"], g["name"]
            # the reference's node order is a permutation of ours (hash order, see oracle docstring)
            assert sorted(ref["ref_node_order"]) == nodes


def test_appendix_b_toy(partition_golden):
    g = partition_golden["graphs"][0]
    assert g["name"] == "toy_appendix_b"
    assert g["indicator"] == [False] + [True] * 7
    _, _, neighs = P.neighborhood_dataset([(g["n"], [tuple(e) for e in g["edges"]])], 4)
    # v=3: all five edges are triangle edges; v=4 adds the tride edge (3,4)  
    h3 = P.to_tconv_hetero(P.networkx_to_hetero(*neighs[2], canonical=3))
    assert sum(ei.shape[1] for (s, r, d), ei in h3["edge_index"].items() if r == "union_tride") == 0
    assert sum(ei.shape[1] for (s, r, d), ei in h3["edge_index"].items() if r == "union_triangle") == 10
    h4 = P.to_tconv_hetero(P.networkx_to_hetero(*neighs[3], canonical=4))
    tride = {et: ei for et, ei in h4["edge_index"].items() if et[1] == "union_tride"}
    assert sum(ei.shape[1] for ei in tride.values()) == 2
    assert tride[("count", "union_tride", "canonical")].T.tolist() == [[3, 0]]
    assert tride[("canonical", "union_tride", "count")].T.tolist() == [[0, 3]]


def test_triangle_mask_bruteforce(partition_golden):
    """T>1  <=>  endpoints share a neighbour inside the neighborhood (transforms.py:201-221)."""
    for g in partition_golden["graphs"][:20]:
        _, _, neighs = P.neighborhood_dataset([(g["n"], [tuple(e) for e in g["edges"]])], 4)
        for nodes, edges in neighs:
            adj = {v: set() for v in nodes}
            for a, b in edges:
                adj[a].add(b)
                adj[b].add(a)
            h = P.to_tconv_hetero(P.networkx_to_hetero(nodes, edges, canonical=nodes[-1]))
            got = {}
            for (s, r, d), ei in h["edge_index"].items():
                for a, b in ei.T.tolist():
                    got[(h["orig_ids"][s][a], h["orig_ids"][d][b])] = r == "union_triangle"
            assert len(got) == 2 * len(edges)
            for (a, b), tri in got.items():
                assert tri == (len(adj[a] & adj[b]) > 0)


def test_collate_offsets():
    n1 = ([0, 1, 2], [(0, 1), (0, 2), (1, 2)])
    n2 = ([3, 5], [(3, 5)])
    b = P.neighborhood_batch([n1, n2])
    assert b["num_nodes"] == {"count": 3, "canonical": 2}
    assert b["batch"]["count"].tolist() == [0, 0, 1]
    assert b["batch"]["canonical"].tolist() == [0, 1]
    assert b["edge_index"][("count", "union_triangle", "count")].T.tolist() == [[0, 1], [1, 0]]
    assert b["edge_index"][("count", "union_tride", "canonical")].T.tolist() == [[2, 1]]
    assert b["edge_index"][("canonical", "union_tride", "count")].T.tolist() == [[1, 2]]
    assert b["edge_index"][("count", "union_triangle", "canonical")].T.tolist() == [[0, 0], [1, 0]]


def test_queries_and_bruteforce_counts(queries_golden, counts_golden, partition_golden):
    qs = [(q["n"], [tuple(e) for e in q["edges"]]) for q in queries_golden["queries"]]
    assert queries_golden["query_ids"] == [6, 7] + list(range(13, 19)) + [29, 30, 31] + \
        list(range(34, 39)) + list(range(40, 53))
    assert sum(q["n"] for q in queries_golden["queries"]) == 135
    by_name = {g["name"]: g for g in partition_golden["graphs"]}
    for item in counts_golden:
        g = by_name[item["name"]]
        if g["n"] > 15:
            continue
        got = P.canonical_counts_bruteforce(g["n"], [tuple(e) for e in g["edges"]], qs)
        assert got.tolist() == item["count"], item["name"]


def test_gossip_edge_index():
    ei = np.array([[0, 1, 2, 2, 3], [1, 0, 2, 3, 1]])
    e, w = P.gossip_edge_index(4, ei)
    assert e.T.tolist() == [[0, 1], [1, 0], [1, 3], [2, 3], [3, 1], [3, 2]]
    assert w.tolist() == [True, False, True, True, False, False]
