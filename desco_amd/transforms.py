"""Markers with the reference's transform names (subgraph_counting/transforms.py:18-42, 168-255): the partition
builder emits the triangle / tride split itself and ``node_feature=None`` means zeros, so the Workload only inspects
which markers it was given (workload._check_transforms refuses any other callable)."""
from __future__ import annotations

from .batch import tconv_split  # noqa: F401  (exported: query-graph triangle split)


class _Marker:
    def __call__(self, data):
        return data


class ToTconvHetero(_Marker):
    def __init__(self, node_attr: str = "x"):
        self.node_attr = node_attr


class ZeroNodeFeat(_Marker):
    def __init__(self, node_feat_name: str = "x", node_feat_len: int = None):
        self.node_feat_name = node_feat_name
        self.node_feat_len = 1 if node_feat_len is None else node_feat_len
