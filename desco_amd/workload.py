"""Workload / NeighborhoodDataset / GossipDataset with the reference's API
(subgraph_counting/workload.py:48-324, 363-747), over flat CSR storage.

Differences from the reference that are deliberate:
  * the target dataset is a ``GraphSet`` (or anything ``GraphSet.from_networkx`` accepts);
  * neighborhoods are built in bulk by the native builder and cached as ``.npz``; the
    ``neighs_index_depth_{d}.npy`` / ``neighs_indicator_depth_{d}.npy`` pair keeps the
    reference's names, dtypes and contents (workload.py:197-213);
  * datasets hand out device-ready batch containers instead of PyG ``HeteroData``.
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional

import numpy as np
import torch

from .batch import GossipBatch, NeighborhoodBatch
from .graphs import GraphSet
from .partition import NeighborhoodPartition, build_partition


def _as_graphset(dataset) -> GraphSet:
    if isinstance(dataset, GraphSet):
        return dataset
    return GraphSet.from_networkx(list(dataset))


def _check_transforms(who, transform, pre_transform, pre_filter):
    """The reference runs ``transform`` per item in DataLoader workers (workload.py:443-449) and ``pre_transform`` /
    ``pre_filter`` once over the PyG data list (:84-85, 287-288).  The native pipeline has no per-item PyG objects to
    hand to a callable: the two transforms main.py uses are built in (transforms.py markers), anything else would be
    silently ignored -- so it is refused."""
    from .transforms import ToTconvHetero, ZeroNodeFeat
    if transform is not None and not isinstance(transform, (ToTconvHetero, ZeroNodeFeat)):
        raise NotImplementedError(
            f"{who}: transform={type(transform).__name__} is not executed by the native pipeline (supported: None, "
            "ToTconvHetero, ZeroNodeFeat -- both are built into the partition builder / batch containers)")
    if pre_transform is not None or pre_filter is not None:
        raise NotImplementedError(f"{who}: pre_transform / pre_filter callables are not supported (no PyG data list)")


class NeighborhoodDataset:
    """One canonical neighborhood per node with >= 1 edge in it (workload.py:153-324)."""

    def __init__(self, depth_neigh, root, dataset=None, nx_targets=None, transform=None,
                 pre_transform=None, pre_filter=None, hetero_graph=True, node_feat=False,
                 node_feat_key="feat", quirk_batch: int = 0, num_threads: int = 0):
        if not hetero_graph:
            raise NotImplementedError("hetero_graph=False (ablation) is outside the hot path")
        if dataset is None and nx_targets is None:
            raise AttributeError("must create Neighborhood dataset with a dataset")
        _check_transforms("NeighborhoodDataset", transform, pre_transform, pre_filter)
        self.dataset = _as_graphset(dataset if dataset is not None else nx_targets)
        self.depth_neigh, self.root, self.transform = depth_neigh, root, transform
        self.quirk_batch = quirk_batch
        self.node_feat = bool(node_feat)
        if self.node_feat and self.dataset.node_feat is None:
            raise ValueError("node_feat=True needs a dataset with node features (GraphSet.node_feat)")
        self.y: Optional[torch.Tensor] = None
        pdir = os.path.join(root, "processed") if root else None
        names = self.processed_file_names
        paths = [os.path.join(pdir, n) for n in names] if pdir else None
        cached = None
        if paths and all(os.path.exists(p) for p in paths):
            cached = np.load(paths[0])
            if int(cached["quirk_batch"]) != int(quirk_batch):
                cached = None          # a cache built for another PyG-quirk emulation setting: rebuild
        if cached is not None:
            z = cached
            self.partition = NeighborhoodPartition(
                np.load(paths[1]), np.load(paths[2]), z["count_ptr"], z["count_orig"],
                z["vrowptr"], z["vcol"], depth_neigh, int(z["quirk_batch"]))
        else:
            self.partition = build_partition(self.dataset, depth_neigh, quirk_batch, num_threads)
            if paths:
                os.makedirs(pdir, exist_ok=True)
                p = self.partition
                np.savez(paths[0], count_ptr=p.count_ptr, count_orig=p.count_orig,
                         vrowptr=p.vrowptr, vcol=p.vcol, quirk_batch=np.int64(p.quirk_batch))
                np.save(paths[1], p.neigh_index.astype(int))          # workload.py:293
                np.save(paths[2], p.indicator.astype(bool))           # workload.py:294
        self.nx_neighs_index = self.partition.neigh_index
        self.nx_neighs_indicator = self.partition.indicator

    @property
    def processed_file_names(self) -> List[str]:
        d = str(self.depth_neigh)
        return ["neighs_csr_depth_" + d + ".npz", "neighs_index_depth_" + d + ".npy",
                "neighs_indicator_depth_" + d + ".npy"]

    def __len__(self):
        return self.partition.num_neigh

    def batch(self, b0: int, b1: int, device="cpu") -> NeighborhoodBatch:
        y = None if self.y is None else self.y[b0:b1]
        part = self.partition.slice(b0, b1)
        feat = None
        if self.node_feat:
            # --use_node_feature: rows = the count nodes' features (count_orig = their global node
            # ids) followed by the canonical nodes' (NetworkxToHetero "feat", transforms.py:380-384)
            nf = self.dataset.node_feat
            canon = self.dataset.graph_ptr[part.neigh_index[:, 0]] + part.neigh_index[:, 1]
            feat = torch.from_numpy(np.concatenate([nf[part.count_orig], nf[canon]]))
        return NeighborhoodBatch(part, device, node_feature=feat, y=y)

    def batches(self, batch_size: int, device="cpu") -> Iterator[NeighborhoodBatch]:
        for b0 in range(0, len(self), batch_size):
            yield self.batch(b0, min(b0 + batch_size, len(self)), device)

    def apply_truth_from_dataset(self, truth):                                   # :296-301
        self.y = torch.as_tensor(truth)[torch.from_numpy(self.nx_neighs_indicator)]

    def aggregate_neighborhood_count(self, count: torch.Tensor) -> torch.Tensor:  # :303-324
        count = count.detach().clone().cpu()
        gid = torch.from_numpy(self.nx_neighs_index[:, 0].astype(np.int64))
        out = torch.zeros((self.dataset.num_graphs, count.shape[1]), dtype=torch.float)
        out.index_add_(0, gid, count.float())
        return out


class GossipDataset:
    """The original graphs with x := neighborhood predictions (workload.py:48-150)."""

    def __init__(self, dataset, root=None, transform=None, pre_transform=None, pre_filter=None,
                 hetero_graph=True):
        _check_transforms("GossipDataset", transform, pre_transform, pre_filter)
        self.dataset = _as_graphset(dataset)
        self.root = root
        self.x: Optional[torch.Tensor] = None
        self.y: Optional[torch.Tensor] = None

    def __len__(self):
        return self.dataset.num_graphs

    def apply_truth_from_dataset(self, truth):                                   # :92-105
        self.y = torch.as_tensor(truth)

    def apply_neighborhood_count(self, count: torch.Tensor, neighborhood_indicator):   # :107-126
        ind = torch.as_tensor(np.asarray(neighborhood_indicator, dtype=bool))
        self.x = torch.zeros((len(ind), count.shape[1]))
        self.x[ind, :] = count.detach().cpu().float()

    def apply_neighborhood_embeddings(self, embedding: torch.Tensor, neighborhood_indicator):
        ind = torch.as_tensor(np.asarray(neighborhood_indicator, dtype=bool))
        e = torch.zeros((len(ind), embedding.shape[1]))
        e[ind, :] = embedding.detach().cpu().float()
        self.x = e if self.x is None else torch.cat([self.x, e], dim=1)

    def aggregate_neighborhood_count(self, count: torch.Tensor) -> torch.Tensor:  # :136-148
        count = count.detach().cpu()
        gid = torch.from_numpy(self.dataset.node_graph_ids())
        out = torch.zeros((self.dataset.num_graphs, count.shape[1]), dtype=count.dtype)
        out.index_add_(0, gid, count)
        return out

    def batch(self, g0: int, g1: int, device="cpu") -> GossipBatch:
        n0, n1 = int(self.dataset.graph_ptr[g0]), int(self.dataset.graph_ptr[g1])
        x = None if self.x is None else self.x[n0:n1]
        y = None if self.y is None else self.y[n0:n1]
        return GossipBatch(self.dataset.subset(g0, g1), device, x=x, y=y)

    def batches(self, batch_size: int, device="cpu") -> Iterator[GossipBatch]:
        for g0 in range(0, len(self), batch_size):
            yield self.batch(g0, min(g0 + batch_size, len(self)), device)


class Workload:
    """Owns the target dataset, ground truth and the two pipeline datasets (workload.py:363-747)."""

    def __init__(self, dataset, root: str, hetero_graph: bool = True, node_feat_len: int = -1,
                 node_feat_key: str = "feat", **kwargs):
        self.dataset = _as_graphset(dataset)
        self.root = root
        self.hetero_graph = hetero_graph
        self.use_node_feat = node_feat_len != -1                                   # workload.py:383
        if self.use_node_feat:
            nf = self.dataset.node_feat
            if nf is None or nf.shape[1] != node_feat_len:
                raise ValueError(f"node_feat_len={node_feat_len}: the dataset carries "
                                 f"{'no node features' if nf is None else f'{nf.shape[1]}-wide features'}")
            self.node_feat_len = node_feat_len
        else:
            self.node_feat_len = 1
        self.node_feat_key = "feat"
        self.queries, self.query_ids = [], []
        self.canonical_count_truth = torch.tensor([[]])
        self.neighborhood_dataset: Optional[NeighborhoodDataset] = None
        self.gossip_dataset: Optional[GossipDataset] = None

    def generate_pipeline_datasets(self, depth_neigh, neighborhood_transform=None,
                                   gossip_transform=None, pre_transform=None, pre_filter=None,
                                   quirk_batch: int = 0):                         # :422-471
        _check_transforms("Workload.generate_pipeline_datasets", None, pre_transform, pre_filter)
        self.neighborhood_dataset = NeighborhoodDataset(
            depth_neigh=depth_neigh,
            root=os.path.join(self.root, "NeighborhoodDataset") if self.root else None,
            dataset=self.dataset, transform=neighborhood_transform, hetero_graph=self.hetero_graph,
            node_feat=self.use_node_feat, quirk_batch=quirk_batch)
        self.gossip_dataset = GossipDataset(
            dataset=self.dataset,
            root=os.path.join(self.root, "GossipDataset") if self.root else None,
            transform=gossip_transform, hetero_graph=self.hetero_graph)
        if self.canonical_count_truth.shape[1] != 0:
            self.neighborhood_dataset.apply_truth_from_dataset(self.canonical_count_truth)
            self.gossip_dataset.apply_truth_from_dataset(self.canonical_count_truth)

    # ---- ground truth ----------------------------------------------------------------------------
    def _truth_path(self, query_ids, queries=None):
        import networkx as nx
        from .data import graph_atlas_plus
        qs = queries if queries is not None else [graph_atlas_plus(q) for q in query_ids]
        name = "query_num_{:d}_query_len_sum_{:d}.pt".format(len(qs), sum(len(q) for q in qs))  # :485-493
        return os.path.join(self.root, "CanonicalCountTruth", name)

    def exist_groundtruth(self, query_ids, queries=None) -> bool:                 # :512-549
        return os.path.exists(self._truth_path(query_ids, queries))

    def load_groundtruth(self, query_ids, queries=None) -> torch.Tensor:         # :473-510
        self.canonical_count_truth = torch.load(self._truth_path(query_ids, queries))
        self.query_ids = query_ids
        return self.canonical_count_truth

    def compute_groundtruth(self, query_ids=None, queries=None, num_workers=-1,
                            save_to_file=True) -> torch.Tensor:                   # :551-726
        from .groundtruth import canonical_counts, canonical_counts_labelled
        from .data import add_node_feat_to_networkx, graph_atlas_plus
        qs = queries if queries is not None else [graph_atlas_plus(q) for q in query_ids]
        if self.use_node_feat and queries is None:                                # :564-577
            eye = [t for t in np.eye(self.node_feat_len).tolist()]
            qs = [g for q in qs for g in add_node_feat_to_networkx(q, eye, self.node_feat_key)]
        if self.use_node_feat:
            truth = canonical_counts_labelled(self.dataset, qs, self.node_feat_key)
        else:
            truth = canonical_counts(self.dataset, qs, num_threads=max(num_workers, 0))
        self.canonical_count_truth = truth
        self.query_ids = query_ids
        if save_to_file and self.root:
            path = self._truth_path(query_ids, queries)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(truth, path)
        return truth

    # ---- stage glue --------------------------------------------------------------------------------
    def apply_neighborhood_count(self, count):                                    # :728-731
        self.gossip_dataset.apply_neighborhood_count(
            count, self.neighborhood_dataset.nx_neighs_indicator)

    def apply_neighborhood_embeddings(self, embeddings):                          # :733-736
        self.gossip_dataset.apply_neighborhood_embeddings(
            embeddings, self.neighborhood_dataset.nx_neighs_indicator)

    def to_networkx(self):                                                        # :738-747
        import networkx as nx
        out = []
        for n, edges in self.dataset.edge_lists():
            g = nx.Graph()
            g.add_nodes_from(range(n))
            g.add_edges_from(edges)
            out.append(g)
        return out
