"""Native canonical-partition builder (host C++ behind the C ABI) vs. the golden vectors of the
reference and vs. the oracle's typed edge sets; batch slicing; the remove_self_loops quirk."""
import numpy as np
import pytest

from desco_amd.graphs import GraphSet
from desco_amd.partition import build_partition
from oracle import partition as OP


def _graphs(golden):
    return [(g["n"], [tuple(e) for e in g["edges"]]) for g in golden["graphs"]]


def _typed_sets(ed):
    return {et: sorted(map(tuple, np.asarray(ei).T.tolist())) for et, ei in ed.items()}


def test_builder_matches_reference_golden(partition_golden):
    graphs = _graphs(partition_golden)
    part = build_partition(GraphSet.from_edge_lists(graphs), partition_golden["depth"])
    ind, idx = [], []
    for gid, g in enumerate(partition_golden["graphs"]):
        ind += g["indicator"]
        idx += [(gid, v) for v in g["index_nodes"]]
    assert part.indicator.tolist() == ind                       # bit-exact nx_neighs_indicator
    assert part.neigh_index.tolist() == [list(t) for t in idx]   # bit-exact nx_neighs_index
    # node sets of every neighborhood, in original ids
    b = 0
    gs_ptr = np.concatenate([[0], np.cumsum([g["n"] for g in partition_golden["graphs"]])])
    for gid, g in enumerate(partition_golden["graphs"]):
        for ref in g["neighs"]:
            c0, c1 = part.count_ptr[b], part.count_ptr[b + 1]
            nodes = (part.count_orig[c0:c1] - gs_ptr[gid]).tolist() + [ref["canonical"]]
            assert nodes == ref["nodes"]
            b += 1
    assert b == part.num_neigh


def test_builder_edges_match_oracle_and_slices(partition_golden):
    graphs = _graphs(partition_golden)
    part = build_partition(GraphSet.from_edge_lists(graphs), 4, num_threads=3)
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    assert _typed_sets(part.edge_index_dict()) == _typed_sets(OP.neighborhood_batch(neighs)["edge_index"])
    assert part.num_edges == 2 * sum(len(e) for _, e in neighs)
    for b0, b1 in [(0, 1), (7, 130), (600, 10 ** 6)]:
        sl = part.slice(b0, b1)
        ob = OP.neighborhood_batch(neighs[b0:b1])
        assert _typed_sets(sl.edge_index_dict()) == _typed_sets(ob["edge_index"])
        assert sl.num_count == ob["num_nodes"]["count"] and sl.num_neigh == ob["num_nodes"]["canonical"]


@pytest.mark.parametrize("qb", [1, 3, 16, 512])
def test_selfloop_quirk_matches_pyg_semantics(partition_golden, qb):
    """quirk_batch drops exactly the edges remove_self_loops would drop on the bipartite types of
    each reference batch (gnn_model.py:389-390)."""
    graphs = _graphs(partition_golden)[:12]
    part = build_partition(GraphSet.from_edge_lists(graphs), 4, quirk_batch=qb)
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    got = _typed_sets(part.edge_index_dict())
    want = {et: [] for et in OP.EDGE_TYPES}
    cbase = 0
    for b0 in range(0, len(neighs), qb):
        ob = OP.neighborhood_batch(neighs[b0:b0 + qb])
        for et, ei in ob["edge_index"].items():
            s, _, d = et
            keep = ei[:, ei[0] != ei[1]] if s != d else ei
            off = np.array([[cbase if s == "count" else b0], [cbase if d == "count" else b0]])
            want[et] += list(map(tuple, (keep + off).T.tolist()))
        cbase += ob["num_nodes"]["count"]
    assert got == {et: sorted(v) for et, v in want.items()}
    full = build_partition(GraphSet.from_edge_lists(graphs), 4)
    assert part.num_edges < full.num_edges


@pytest.mark.parametrize("seed,depth", [(21, 4), (22, 2)])
def test_random_graph_families_match_oracle(seed, depth):
    """Stars, wheels, paths, cycles with chords, cliques with tails, grids, barbells, random trees and G(n,p) at three
    densities with shuffled node ids: index, indicator and the six typed edge sets of the host builder equal the oracle's."""
    from helpers import random_family_graphs
    graphs = random_family_graphs(seed, 44)
    part = build_partition(GraphSet.from_edge_lists(graphs), depth, num_threads=2)
    idx, ind, neighs = OP.neighborhood_dataset(graphs, depth)
    assert (part.neigh_index == idx).all() and (part.indicator == ind).all()
    assert _typed_sets(part.edge_index_dict()) == _typed_sets(OP.neighborhood_batch(neighs)["edge_index"])
    assert part.num_edges == 2 * sum(len(e) for _, e in neighs)


def test_empty_and_edgeless_inputs():
    part = build_partition(GraphSet.from_edge_lists([]), 4)
    assert part.num_neigh == 0 and part.num_edges == 0
    part = build_partition(GraphSet.from_edge_lists([(5, []), (1, []), (0, [])]), 4)
    assert part.num_neigh == 0 and part.indicator.tolist() == [False] * 6
    part = build_partition(GraphSet.from_edge_lists([(3, [(0, 1), (1, 1), (0, 1), (1, 0)])]), 4)
    assert part.neigh_index.tolist() == [[0, 1]] and part.num_edges == 2


def test_depth_is_respected():
    path = [(12, [(i, i + 1) for i in range(11)])]
    for depth in (1, 2, 4):
        part = build_partition(GraphSet.from_edge_lists(path), depth)
        sizes = np.diff(part.count_ptr) + 1
        assert sizes.max() == depth + 1
        _, _, neighs = OP.neighborhood_dataset(path, depth)
        assert [len(n) for n, _ in neighs] == sizes.tolist()


def test_graphset_roundtrip_and_replicate():
    graphs = [(4, [(0, 1), (1, 2), (2, 3)]), (3, [(0, 2)])]
    gs = GraphSet.from_edge_lists(graphs)
    assert gs.edge_lists() == graphs
    r = gs.replicate(3)
    assert r.num_graphs == 6 and r.edge_lists() == graphs * 3
    assert r.subset(2, 4).edge_lists() == graphs
