"""Device-side canonical-partition builder (csrc/partition_dev.hip) vs the host builder and the
golden vectors: integer work, bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import golden_graphs

pytestmark = pytest.mark.gpu

from desco_amd import synthetic
from desco_amd.batch import NeighborhoodBatch
from desco_amd.graphs import GraphSet
from desco_amd.partition import build_partition, build_partition_device

FIELDS = ("neigh_index", "indicator", "count_ptr", "count_orig", "vrowptr", "vcol")


def _same(a, b):
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert x.shape == y.shape, (f, x.shape, y.shape)
        assert np.array_equal(x, y), f


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_golden_graphs_bit_exact(depth):
    gs = GraphSet.from_edge_lists(golden_graphs())
    _same(build_partition_device(gs, depth), build_partition(gs, depth))


@pytest.mark.parametrize("workload,count", [("mutag", 188), ("cox2", 200), ("msrc_imdb", 120), ("syn_1827", 60)])
def test_synthetic_shapes_bit_exact(workload, count):
    gs = synthetic.WORKLOADS[workload]()
    gs = gs.subset(0, min(count, gs.num_graphs))
    dev, host = build_partition_device(gs, 4), build_partition(gs, 4)
    _same(dev, host)
    # the device arrays are reused by the batch (no re-upload) and equal the host copies
    b = NeighborhoodBatch(dev, "cuda")
    assert b.vcol.data_ptr() == dev.device_arrays["vcol"].data_ptr()
    assert torch.equal(b.vrowptr.cpu(), torch.from_numpy(host.vrowptr))


@pytest.mark.parametrize("seed,depth", [(31, 4), (32, 3)])
def test_random_graph_families_bit_exact(seed, depth):
    from helpers import random_family_graphs
    gs = GraphSet.from_edge_lists(random_family_graphs(seed, 66))
    _same(build_partition_device(gs, depth), build_partition(gs, depth))


def test_edge_cases():
    # isolated nodes, a single edge, a triangle, a star with a hub of degree 70 (> one wave of lanes)
    graphs = [(3, []), (2, [(0, 1)]), (3, [(0, 1), (1, 2), (0, 2)]),
              (72, [(0, i) for i in range(1, 72)]), (1, [])]
    gs = GraphSet.from_edge_lists(graphs)
    _same(build_partition_device(gs, 4), build_partition(gs, 4))
    empty = GraphSet.from_edge_lists([(2, [])])
    d = build_partition_device(empty, 4)
    assert d.num_neigh == 0 and d.num_edges == 0 and not d.indicator.any()


def test_few_waves_and_many_waves_agree():
    gs = GraphSet.from_edge_lists(golden_graphs(max_n=60))
    a = build_partition_device(gs, 4, num_waves=4)
    b = build_partition_device(gs, 4, num_waves=4096)
    _same(a, b)


def test_large_graphs_multiword_bitmaps_and_host_fallback():
    """A 3000-node graph (94 bitmap words per wave, > 64 KB of dynamic LDS per block) is built on the
    device; a 6000-node graph exceeds the per-wave LDS workspace: the device builder refuses it and
    the pipeline's partition falls back to the host builder."""
    rng = np.random.default_rng(5)
    def tree_plus(n, extra):
        edges = [(i, int(rng.integers(0, i))) for i in range(1, n)]
        edges += [(int(a), int(b)) for a, b in rng.integers(0, n, size=(extra, 2)) if a != b]
        return (n, edges)
    mid = GraphSet.from_edge_lists([tree_plus(3000, 1500), tree_plus(40, 10)])
    _same(build_partition_device(mid, 3), build_partition(mid, 3))
    big = GraphSet.from_edge_lists([tree_plus(6000, 100)])
    with pytest.raises(RuntimeError, match="does not fit the LDS workspace"):
        build_partition_device(big, 2)
    from helpers import make_models, standard_queries
    from desco_amd.pipeline import InferencePipeline
    nm, gm = make_models(seed=0)
    nm, gm = nm.to("cuda"), gm.to("cuda")
    nm.set_queries(standard_queries()[0])
    pipe = InferencePipeline(nm, gm, big, depth=2, device="cuda")
    assert pipe.partition_backend == "host"
    assert torch.isfinite(pipe.run()["graph_gossip_count"]).all()


def test_device_builder_against_reference_golden(partition_golden):
    """The HIP builder DIRECTLY against tests/golden/partition_golden.json (written from the reference's own
    get_neigh_hetero by tests/golden/make_golden.py) -- not through the host builder: nx_neighs_indicator,
    nx_neighs_index, the node set of every neighborhood, its induced edge set (the union of the six typed edge sets,
    mapped back to graph ids, each undirected edge present in both directions), and the triangle / non-triangle type
    of every edge against the definition (ToTconvHetero: the endpoints share a neighbour inside the neighborhood)."""
    graphs = [(g["n"], [tuple(e) for e in g["edges"]]) for g in partition_golden["graphs"]]
    part = build_partition_device(GraphSet.from_edge_lists(graphs), partition_golden["depth"])
    ind, idx = [], []
    for gid, g in enumerate(partition_golden["graphs"]):
        ind += g["indicator"]
        idx += [[gid, v] for v in g["index_nodes"]]
    assert part.indicator.astype(bool).tolist() == ind
    assert part.neigh_index.tolist() == idx
    gs_ptr = np.concatenate([[0], np.cumsum([g["n"] for g in partition_golden["graphs"]])])
    Nc = part.num_count
    # local row (count rows first, then one canonical row per neighborhood) -> (neighborhood, graph-local node id)
    owner = np.concatenate([np.repeat(np.arange(part.num_neigh), np.diff(part.count_ptr)), np.arange(part.num_neigh)])
    gid_of = part.neigh_index[:, 0]
    orig = np.concatenate([part.count_orig - gs_ptr[gid_of[owner[:Nc]]], part.neigh_index[:, 1]])
    got_edges = [dict() for _ in range(part.num_neigh)]          # neighborhood -> {(a, b): is_triangle}
    for (s, rel, d), ei in part.edge_index_dict().items():
        src = ei[0] + (Nc if s == "canonical" else 0)
        dst = ei[1] + (Nc if d == "canonical" else 0)
        assert (owner[src] == owner[dst]).all(), (s, rel, d)
        for b, a_, b_ in zip(owner[src].tolist(), orig[src].tolist(), orig[dst].tolist()):
            assert (a_, b_) not in got_edges[b]
            got_edges[b][(a_, b_)] = rel == "union_triangle"
    b = 0
    for gid, g in enumerate(partition_golden["graphs"]):
        for ref in g["neighs"]:
            c0, c1 = part.count_ptr[b], part.count_ptr[b + 1]
            assert (part.count_orig[c0:c1] - gs_ptr[gid]).tolist() + [ref["canonical"]] == ref["nodes"]
            adj = {v: set() for v in ref["nodes"]}
            for u, v in ref["edges"]:
                adj[u].add(v)
                adj[v].add(u)
            want = {}
            for u, v in ref["edges"]:
                tri = bool(adj[u] & adj[v])
                want[(u, v)] = want[(v, u)] = tri
            assert got_edges[b] == want, (gid, ref["canonical"])
            b += 1
    assert b == part.num_neigh
