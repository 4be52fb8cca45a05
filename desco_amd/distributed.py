"""Multi-GPU support: the path shards by independent target graphs.

One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).
Inference: cost-balanced contiguous graph ranges, no data-path collective, one gather of the
[G,29] graph-level counts.  Training: flat-bucket gradient all-reduce (``allreduce_grads``).
The reference only has Lightning DDP for neighborhood training (main.py:242-255) and runs
inference on one device.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graphs import GraphSet


def graph_costs(graphs: GraphSet, num_queries: int = 29) -> np.ndarray:
    """Cheap per-graph cost proxy c(g) ~ neighborhood work + gossip work:
    sum over nodes of (1 + deg)^2 bounded, plus Q*(n + e)."""
    deg = np.diff(graphs.rowptr).astype(np.float64)
    node_cost = np.minimum((1.0 + deg) ** 2, 4096.0)
    gid = graphs.node_graph_ids()
    c = np.zeros(graphs.num_graphs)
    np.add.at(c, gid, node_cost + num_queries * (1.0 + deg))
    return c


def contiguous_shards(costs: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Cut [0,G) into world_size contiguous ranges of ~equal total cost (keeps the dataset order,
    so concatenating rank outputs in rank order reproduces the single-GPU output order)."""
    G = len(costs)
    csum = np.concatenate([[0.0], np.cumsum(costs)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world_size):
        t = total * r / world_size
        k = int(np.searchsorted(csum, t, side="left"))
        cuts.append(min(max(k, cuts[-1]), G))
    cuts.append(G)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def shard_graphs(graphs: GraphSet, rank: int, world_size: int, num_queries: int = 29):
    lo, hi = contiguous_shards(graph_costs(graphs, num_queries), world_size)[rank]
    return graphs.subset(lo, hi), (lo, hi)


def gather_rows(local: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Concatenate per-rank row blocks (different row counts allowed) on ``dst`` in rank order."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def allreduce_grads(params: Sequence[torch.nn.Parameter], mode: str = "mean") -> None:
    """Single flat-bucket gradient all-reduce (the whole model is 5.24 MB fp32: one collective is
    latency-optimal over xGMI).  Parameters without a gradient (the never-used query-side
    ``anchor_mlp``) contribute zeros -- DDP's find_unused_parameters semantics.
    mode="mean" for the neighborhood loss (a mean), "sum" for the gossip loss (a sum)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    params = [p for p in params if p.requires_grad]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if mode == "mean":
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
