"""Transform objects with the reference's names (subgraph_counting/transforms.py).

In the reference these are per-item PyG transforms executed in DataLoader workers on every
``__getitem__`` (workload.py:443-449).  Here the work is done once, in bulk, by the native
partition builder (triangle split) or is implicit in the batch containers (zero features), so the
classes are thin markers that the Workload inspects; any OTHER transform / pre_transform / pre_filter callable is refused
by the datasets (workload._check_transforms) instead of being ignored."""
from __future__ import annotations

from .batch import tconv_split  # noqa: F401  (exported: query-graph triangle split)


class ToTconvHetero:
    """Marker for the SHMP triangle / tride edge split (transforms.py:168-255).  The canonical-
    partition builder always emits the split as relation slots; models built with
    ``use_tconv=False`` simply tie the two slots to one weight."""

    def __init__(self, node_attr: str = "x"):
        self.node_attr = node_attr

    def __call__(self, data):
        return data


class ZeroNodeFeat:
    """Marker for all-zero ``[n, node_feat_len]`` node features (transforms.py:18-42): batches
    carry ``node_feature=None`` and the kernels treat it as zeros."""

    def __init__(self, node_feat_name: str = "x", node_feat_len: int = None):
        self.node_feat_name = node_feat_name
        self.node_feat_len = 1 if node_feat_len is None else node_feat_len

    def __call__(self, data):
        return data
