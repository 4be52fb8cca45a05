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
