"""Two-stage inference over a whole dataset, resident in HBM (the hot path of main.py:296-302,
417-423): neighborhood counting -> apply_neighborhood_count -> gossip -> graph-level aggregation.

The reference round-trips GPU -> CPU -> GPU between the stages and walks fixed 512/256-item
DataLoader batches; here the dataset shard is packed once into a few large device blocks sized by
a row budget ("per-GPU dynamic graph-batch packing"), and every stage stays on the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import gnn_model as GMOD
from . import ops
from .batch import GossipBatch, NeighborhoodBatch
from .graphs import GraphSet
from .partition import NeighborhoodPartition, build_partition, build_partition_device


def _split_by_budget(weights: np.ndarray, budget: int) -> List[int]:
    """Cut points so that every chunk's total weight <= budget (single items may exceed it)."""
    cuts, acc, start = [0], 0, 0
    csum = np.concatenate([[0], np.cumsum(weights, dtype=np.int64)])
    n = len(weights)
    while start < n:
        end = int(np.searchsorted(csum, csum[start] + budget, side="right")) - 1
        end = max(end, start + 1)
        end = min(end, n)
        cuts.append(end)
        start = end
    return cuts


class InferencePipeline:
    def __init__(self, neigh_model, gossip_model, graphs: GraphSet, depth: int = 4,
                 device="cuda", quirk_batch: int = 0, max_neigh_rows: int = 48_000_000,
                 max_gossip_rows: int = 48_000_000, num_threads: int = 0,
                 partition: Optional[NeighborhoodPartition] = None,
                 partition_backend: str = "device", rank: Optional[int] = None,
                 world: Optional[int] = None, graph_replay_rows: int = 400_000,
                 chunks: Optional[int] = None, degree_sort: Optional[bool] = None):
        """``degree_sort`` (default on; ``DESCO_DEGREE_SORT=0`` turns it off for A/B runs): the count rows of every
        neighborhood of a block are re-ordered by their number of count -> count sources
        (``NeighborhoodPartition.degree_sorted``), which saves gather steps in the layer kernel on dense shapes; the
        order inside a neighborhood is not observable in any result except through fp32 summation order.
        ``rank`` / ``world`` (default: the initialised torch.distributed group): this process
        keeps the ``rank``-th of ``world`` contiguous, cost-balanced graph ranges
        (distributed.shard_graphs) with all their neighborhoods -- no data-path collective; the
        results of all ranks are assembled in dataset order by ``gather()``.
        ``graph_replay_rows``: ``step()`` replays the pass from a hipGraph when the shard has fewer
        neighborhood rows than this (a 467-graph COX2 pass is ~70 launches of a few microseconds:
        launch-bound when issued eagerly).
        ``chunks`` (None, or a multiple of ``world``): placement-independent mode.  The WHOLE dataset is
        first cut into ``chunks`` contiguous cost-balanced graph ranges -- a decomposition that does not
        depend on the number of ranks -- and every chunk is processed through launches of its own
        (neighborhood block, gossip blocks), exactly as a stand-alone pipeline on that chunk would; rank r
        takes chunks [r chunks / world, (r + 1) chunks / world), which is the same graph range
        ``shard_graphs`` gives it.  A chunk's launches see the same rows in the same tiles whichever rank
        runs them, so N ranks reproduce the 1-rank result BIT FOR BIT (without it they agree to fp32
        rounding: a neighborhood's pooled partial sums depend on where 16-row tiles cut it).
        Costs one launch set per chunk instead of one per 48 M-row block: off by default."""
        from . import distributed as D
        self.nm, self.gm = neigh_model, gossip_model
        # rank / world taken from the process group: every rank builds its pipeline, and rank 0 decides the cuts for all
        agree = rank is None and world is None
        self.rank = D.rank() if rank is None else int(rank)
        self.world = D.world_size() if world is None else int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError(f"rank {self.rank} outside world of {self.world}")
        self.num_graphs_total = graphs.num_graphs
        self.graph_range = (0, graphs.num_graphs)
        self.chunks = None if chunks is None else int(chunks)
        chunk_cuts = None                       # graph boundaries of this rank's chunks, relative to its shard
        Q0 = len(getattr(neigh_model, "queries_flat", [])) or 29
        if self.chunks is not None:
            if self.chunks < 1 or self.chunks % self.world:
                raise ValueError(f"chunks = {self.chunks} must be a positive multiple of world = {self.world}")
            if partition is not None:
                raise ValueError("chunks: let the pipeline build the partition")
            ranges = D.shard_cuts(graphs, self.chunks, Q0, device, depth, agree)
            per = self.chunks // self.world
            mine = ranges[self.rank * per:(self.rank + 1) * per]
            self.graph_range = (mine[0][0], mine[-1][1])
            chunk_cuts = [mine[0][0] - self.graph_range[0]] + [b - self.graph_range[0] for _, b in mine]
            graphs = graphs.subset(*self.graph_range)
        elif self.world > 1:
            if partition is not None:
                raise ValueError("pass the partition of the local shard, or let the pipeline build it")
            graphs, self.graph_range = D.shard_graphs(graphs, self.rank, self.world, Q0, device, depth, agree)
        self.graphs = graphs
        self._chunk_cuts = chunk_cuts
        from .batch import _norm_device
        self.device = _norm_device(device)
        device = self.device
        if device.type == "cuda":
            torch.cuda.set_device(device)      # the C ABI launches on the current device / stream
        self._graph = None
        self._graph_replay_rows = graph_replay_rows
        self._empty = graphs.num_graphs == 0       # more ranks than graphs: this shard has nothing to do
        if self._empty:
            self.partition_backend = "none"
            self.partition = NeighborhoodPartition(
                np.zeros((0, 2), np.int64), np.zeros(0, bool), np.zeros(1, np.int32), np.zeros(0, np.int32),
                np.zeros(1, np.int32), np.zeros(0, np.int32), depth, 0)
            self.neigh_batches, self.gossip_batches, self.num_queries = [], [], None
            return
        # canonical partition: built on the GPU (csrc/partition_dev.hip) unless the PyG quirk
        # emulation is requested or a graph exceeds the device builder's per-wave LDS workspace
        # (then the host C++ builder, same output)
        self.partition_backend = "given"
        if partition is None:
            partition = None
            if partition_backend == "device" and quirk_batch == 0:
                try:
                    partition = build_partition_device(graphs, depth, device)
                    self.partition_backend = "device"
                except RuntimeError as e:
                    if "does not fit the LDS workspace" not in str(e):
                        raise
            if partition is None:
                partition = build_partition(graphs, depth, quirk_batch, num_threads)
                self.partition_backend = "host"
        self.partition = partition
        part = self.partition
        self.num_queries = None
        # neighborhood blocks by row budget
        rows_per_neigh = np.diff(part.count_ptr).astype(np.int64) + 1
        per_graph = np.bincount(part.neigh_index[:, 0], minlength=graphs.num_graphs)
        ngp = np.concatenate([[0], np.cumsum(per_graph)]).astype(np.int64)     # neighborhoods are ordered by graph
        if self._chunk_cuts is None:
            cuts = _split_by_budget(rows_per_neigh, max_neigh_rows)
        else:       # a block never crosses a chunk boundary, and is cut by the budget from the chunk's start
            cuts = [0]
            for ga, gb in zip(self._chunk_cuts[:-1], self._chunk_cuts[1:]):
                a, b = int(ngp[ga]), int(ngp[gb])
                if b > a:
                    cuts += [a + c for c in _split_by_budget(rows_per_neigh[a:b], max_neigh_rows)[1:]]
        import os as _os
        if degree_sort is None:
            degree_sort = _os.environ.get("DESCO_DEGREE_SORT", "1") != "0"
        self.degree_sort = bool(degree_sort)
        self.neigh_batches = [
            NeighborhoodBatch(part.slice(b0, b1).degree_sorted(num_threads) if self.degree_sort else part.slice(b0, b1),
                              device)
            for b0, b1 in zip(cuts[:-1], cuts[1:]) if b1 > b0]
        # neighborhood b -> node row of the gossip x matrix (apply_neighborhood_count)
        rows = graphs.graph_ptr[part.neigh_index[:, 0]] + part.neigh_index[:, 1]
        self.scatter_index = torch.from_numpy(rows.astype(np.int32)).to(device)
        # neighborhoods are ordered by graph: segment pointer for aggregate_neighborhood_count
        self.neigh_graph_ptr = torch.from_numpy(ngp.astype(np.int32)).to(device)
        self.node_graph_ptr = torch.from_numpy(graphs.graph_ptr.astype(np.int32)).to(device)
        # gossip blocks by (node x query) row budget, cut on graph boundaries
        self._gossip_cuts_budget = max_gossip_rows
        self.gossip_batches = None

    def _ensure_gossip_batches(self, Q: int):
        if self.gossip_batches is not None and self.num_queries == Q:
            return
        sizes = np.diff(self.graphs.graph_ptr) * Q
        if self._chunk_cuts is None:
            cuts = _split_by_budget(sizes, self._gossip_cuts_budget)
        else:
            cuts = [0]
            for ga, gb in zip(self._chunk_cuts[:-1], self._chunk_cuts[1:]):
                if gb > ga:
                    cuts += [ga + c for c in _split_by_budget(sizes[ga:gb], self._gossip_cuts_budget)[1:]]
        self.gossip_batches = []
        for g0, g1 in zip(cuts[:-1], cuts[1:]):
            n0, n1 = int(self.graphs.graph_ptr[g0]), int(self.graphs.graph_ptr[g1])
            self.gossip_batches.append((n0, n1, GossipBatch(self.graphs.subset(g0, g1), self.device)))
            self.gossip_batches[-1][2].tile_perm        # built here, outside any stream capture
            self.gossip_batches[-1][2].work_queue
        self.num_queries = Q

    # ---- hipGraph replay --------------------------------------------------------------------------
    def capture(self, gossip: bool = True):
        """Record one full pass into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and
        replay it in ``run_graph()``.  The C-ABI launches only enqueue on the current stream and
        never allocate or synchronise, so the whole two-stage pass (about 70 launches) is
        capturable; small datasets (a 467-graph COX2 pass is ~2 ms) stop paying per-launch host
        overhead.  Weight folding and query embeddings are warmed up outside the capture."""
        self.run(gossip)                                   # warm caches (packing, query embeddings)
        torch.cuda.synchronize(self.device)
        stream = torch.cuda.Stream(self.device)
        stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(stream):
            self.run(gossip)                               # allocator warm-up on the side stream
        torch.cuda.current_stream(self.device).wait_stream(stream)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_out = self.run(gossip)
        self._graph_gossip = gossip
        return self

    def step(self, gossip: bool = True) -> Dict[str, torch.Tensor]:
        """One pass, launch mode chosen by shard size: small shards replay a hipGraph captured on
        the first call (the returned tensors are then the graph's static outputs, overwritten by
        the next ``step()``), large ones launch eagerly."""
        if self._empty or self.partition.num_rows >= self._graph_replay_rows or self.partition.num_neigh == 0:
            return self.run(gossip)
        if self._graph is None or self._graph_gossip != gossip:
            self.capture(gossip)
        return self.run_graph()

    def gather(self, out: Dict[str, torch.Tensor], dst: int = 0, node_level: bool = False):
        """The one exchange of the inference path: the per-graph counts of every rank on
        ``dst`` in dataset order (``[G,29]``, 54 KB for COX2; with ``node_level`` also the per-node
        and per-neighborhood predictions for the CSV dumps).  Returns None on the other ranks."""
        from . import distributed as D
        keys = [k for k in ("graph_neigh_count", "graph_gossip_count") if k in out]
        if node_level:
            keys += [k for k in ("neigh_count", "node_count", "x") if k in out]
        res = {k: D.gather_rows(out[k], dst) for k in keys}
        return res if self.rank == dst or self.world == 1 else None

    def run_graph(self) -> Dict[str, torch.Tensor]:
        """Replay the captured pass; the returned tensors are the graph's static outputs."""
        self._graph.replay()
        return self._graph_out

    @torch.no_grad()
    def run(self, gossip: bool = True) -> Dict[str, torch.Tensor]:
        nm, gm = self.nm, self.gm
        if self._empty:
            Q = len(nm.queries_flat)
            z = torch.zeros((0, Q), device=self.device)
            out = {"neigh_count": z, "graph_neigh_count": z.clone()}
            if gossip:
                out.update({"x": z.clone(), "node_count": z.clone(), "graph_gossip_count": z.clone()})
            return out
        # Persistent result buffers (allocated on the first pass, re-used by every later one: the returned tensors are
        # overwritten by the next run(), like a captured graph's static outputs).  Every block's launches write their
        # slice directly, so a pass contains no torch.cat / fill / copy kernel -- only this library's launches.
        Q = len(nm.queries_flat)
        G, N = self.graphs.num_graphs, self.graphs.num_nodes
        bufs = self.__dict__.setdefault("_result_bufs", {})
        if bufs.get("Q") != Q:
            nb = sum(b.num_graphs for b in self.neigh_batches)
            bufs.update(Q=Q, neigh=torch.empty((nb, Q), device=self.device),
                        x=torch.zeros((N, Q), device=self.device), node=torch.empty((N, Q), device=self.device))
            r = 0
            for b in self.neigh_batches:
                b.out_buf = bufs["neigh"][r:r + b.num_graphs] if Q <= 32 else None
                r += b.num_graphs
        counts = [nm.graph_to_count(b) for b in self.neigh_batches]            # main.py:296-301
        # (the persistent buffer is the result only if every block's launch really wrote its slice of it)
        if counts and all(getattr(b, "out_buf", None) is not None and c.data_ptr() == b.out_buf.data_ptr()
                          for b, c in zip(self.neigh_batches, counts)):
            neigh_count = bufs["neigh"]
        elif not counts:      # no node has a non-empty canonical neighborhood
            neigh_count = torch.zeros((0, Q), device=self.device)
        else:
            neigh_count = counts[0] if len(counts) == 1 else torch.cat(counts)
        out = {"neigh_count": neigh_count}
        out["graph_neigh_count"] = ops.segment_sum(neigh_count, self.neigh_graph_ptr, G)   # :400-404
        if not gossip:
            return out
        # workload.py:107-112: x = zeros; x[indicator] = count.  The rows outside the indicator are never written, so
        # the zeros of the first pass stay; the indicator rows are overwritten by every pass.
        x = bufs["x"]
        ops.scatter_rows(neigh_count, self.scatter_index, x)
        gm.set_query_emb(nm.get_query_emb())                                  # main.py:334
        self._ensure_gossip_batches(Q)
        node = []
        for n0, n1, gb in self.gossip_batches:
            gb.x = x[n0:n1]
            gb.out_buf = bufs["node"][n0:n1] if Q <= 64 else None
            node.append(gm.graph_to_count(gb))                                # main.py:417-420
        if node and all(gb.out_buf is not None and c.data_ptr() == gb.out_buf.data_ptr()
                        for (_, _, gb), c in zip(self.gossip_batches, node)):
            node_count = bufs["node"]
        else:
            node_count = node[0] if len(node) == 1 else torch.cat(node)
        out["x"] = x
        out["node_count"] = node_count
        out["graph_gossip_count"] = ops.segment_sum(node_count, self.node_graph_ptr, G)    # :421-423
        return out
