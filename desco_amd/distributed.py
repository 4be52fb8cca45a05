"""Multi-GPU support: the path shards by independent target graphs.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU
tests and in the shared-GPU test mode).  Nothing here is a translation of the reference's NCCL
usage: the reference only has Lightning DDP for neighborhood training (main.py:242-255), refuses
it for the gossip model (main.py:353-356) and runs inference on one device.

  launch / init_from_env   start N ranks BEFORE any GPU call / join the process group
  shard_graphs             inference: cost-balanced contiguous graph ranges, no data-path collective
  allgather_rows / gather_rows   the one exchange of the inference path ([G,29] counts, 54 KB on COX2)
  GradBuckets              training: gradients live in a few flat buckets (p.grad are views), each
                           bucket's all-reduce is issued from an autograd hook as soon as its last
                           gradient is accumulated, i.e. overlapped with the rest of backward
  step_groups              equal-work packing of the reference's batch stream over the ranks with
                           count weights, so that the N-rank gradient equals the gradient of the
                           single-process loss over the union batch (the loss is a mean)

Environment: DESCO_SHARE_GPU=1 (tests on a 1-GPU box): every rank uses cuda:0 and the "gloo"
backend; collectives on device tensors are staged through host memory.
"""
from __future__ import annotations

import datetime
import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graphs import GraphSet


# ------------------------------------------------------------------------------------------------
# process management
# ------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def share_gpu() -> bool:
    return os.environ.get("DESCO_SHARE_GPU") == "1" or os.environ.get("DESCO_BENCH_SHARE_GPU") == "1"


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun-style environment (1 process: 0, 1, 0)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def is_initialized() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def force_collectives() -> bool:
    """DESCO_FORCE_COLLECTIVES=1: run the data-parallel code path -- process group, gradient buckets, all-reduces,
    broadcasts -- also in a world of ONE rank.  A 1-GPU box can then execute the RCCL path end to end
    (``init_process_group("nccl", device_id=...)``, asynchronous work handles, stream ordering against the training
    stream): bench.py's ``ddp_world1_nccl`` leg."""
    return os.environ.get("DESCO_FORCE_COLLECTIVES") == "1"


def collectives_on() -> bool:
    """A process group exists and its collectives are to be executed (more than one rank, or forced)."""
    import torch.distributed as dist
    return is_initialized() and (dist.get_world_size() > 1 or force_collectives())


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if is_initialized() else 1


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if is_initialized() else 0


def local_device(devices: Optional[Sequence[int]] = None, accelerator: str = "gpu") -> torch.device:
    """This rank's device: devices[LOCAL_RANK] when the list covers the world, else cuda:LOCAL_RANK
    (cuda:0 for every rank in the shared-GPU test mode; devices[0] in a single process)."""
    if accelerator == "cpu":
        return torch.device("cpu")
    if share_gpu():
        return torch.device("cuda", 0)
    _, world, lr = env_world()
    devs = [int(d) for d in devices] if isinstance(devices, (list, tuple)) else []
    if not devs and os.environ.get("DESCO_DEVICES"):      # exported by ``launch(devices=...)``
        devs = [int(d) for d in os.environ["DESCO_DEVICES"].split(",") if d != ""]
    if world > 1:
        return torch.device("cuda", devs[lr] if len(devs) >= world else lr)
    return torch.device("cuda", devs[0] if devs else 0)


def init_from_env(device: Optional[torch.device] = None, backend: Optional[str] = None) -> bool:
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (set by ``launch``, by
    ``python -m torch.distributed.run`` or by the driver).  Returns True when world_size > 1.
    Call it before the first collective; the GPU is bound with torch.cuda.set_device first."""
    import torch.distributed as dist
    r, w, _ = env_world()
    if device is not None and device.type == "cuda":
        torch.cuda.set_device(device)
    if w <= 1 and not force_collectives():
        return False
    if dist.is_initialized():
        if dist.get_world_size() != w:
            raise RuntimeError(f"process group has {dist.get_world_size()} ranks, environment says {w}")
        return True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: start the ranks with "
                           "desco_amd.distributed.launch, main.py --gpu 0 1 .., or torch.distributed.run")
    if backend is None:
        cpu = device is None or device.type != "cuda"
        backend = "gloo" if (cpu or share_gpu()) else "nccl"
    # collectives of this package are short; what is long is host work BETWEEN them (rank 0 building
    # ground truth / partitions while the others wait): that phase waits on the store
    # (``wait_for_rank0``), not in a collective, and the group timeout is generous on top
    timeout = datetime.timedelta(seconds=float(os.environ.get("DESCO_PG_TIMEOUT_S", "7200")))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=r, world_size=w, device_id=device, timeout=timeout)
    else:
        dist.init_process_group(backend, rank=r, world_size=w, timeout=timeout)
    if dist.get_world_size() != w:      # pragma: no cover
        raise RuntimeError("process group size mismatch")
    return w > 1 or force_collectives()


def launch(argv: Sequence[str], nprocs: int, env: Optional[dict] = None,
           devices: Optional[Sequence[int]] = None, timeout: Optional[float] = None) -> int:
    """Start ``nprocs`` copies of ``python argv...`` with the torchrun environment (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) and wait for them -- what Lightning's "ddp"
    strategy does for main.py:242-255.  The caller must not have touched the GPU: children are
    fresh processes (never an exec of a process that initialised HIP).  Returns the worst exit
    code; on a failure the remaining ranks are terminated."""
    port = free_port()
    procs = []
    for r in range(nprocs):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nprocs),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                  "HSA_ENABLE_IPC_MODE_LEGACY": e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if devices is not None:
            e["DESCO_DEVICES"] = ",".join(str(int(d)) for d in devices)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e))
    # poll ALL children: a rank that dies while the others sit in a collective must end the job
    # (waiting for rank 0 first would hang until a watchdog fires -- never, under gloo)
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:       # pragma: no cover
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------
# collectives (device tensors are staged through the host under gloo)
# ------------------------------------------------------------------------------------------------
def _staged(t: torch.Tensor) -> bool:
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


class _StagedWork:
    """Handle of an asynchronous all-reduce staged through the host (gloo on device tensors)."""

    def __init__(self, work, host: torch.Tensor, dev: torch.Tensor):
        self.work, self.host, self.dev = work, host, dev

    def wait(self):
        self.work.wait()
        self.dev.copy_(self.host)


def all_reduce_(t: torch.Tensor, op: str = "sum", async_op: bool = False):
    """In-place all-reduce; returns a handle with .wait() when async_op (None if already done)."""
    import torch.distributed as dist
    if not collectives_on():
        return None
    rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
    if _staged(t):
        h = t.detach().cpu()
        if async_op:        # same contract as the device path: the caller may go on; .wait() lands it
            return _StagedWork(dist.all_reduce(h, op=rop, async_op=True), h, t)
        dist.all_reduce(h, op=rop)
        t.copy_(h)
        return None
    if async_op:
        return dist.all_reduce(t, op=rop, async_op=True)
    dist.all_reduce(t, op=rop)
    return None


def barrier():
    import torch.distributed as dist
    if collectives_on():
        dist.barrier()


_RANK0_FIRST_CALLS = 0


def rank0_first(fn: Callable, key: str = "desco_rank0_done", timeout_s: float = 7 * 86400.0):
    """``fn()`` on rank 0 first, then on the other ranks (which then find rank 0's on-disk caches).
    The others wait on the process group's STORE, not in a collective: a store wait has its own
    (long) timeout and no NCCL / gloo watchdog behind it, so a cold-cache ground-truth or partition
    build on rank 0 cannot abort the job (main.py workload construction).  A failure on rank 0 is
    passed on instead of leaving the others waiting."""
    import torch.distributed as dist
    if not is_initialized() or dist.get_world_size() == 1:
        return fn()
    store = dist.distributed_c10d._get_default_store()
    # one key per CALL (every rank counts its calls identically): a second use in the same process group must not
    # find the first call's "ok" / "fail" and run ahead of rank 0
    global _RANK0_FIRST_CALLS
    _RANK0_FIRST_CALLS += 1
    key = f"{key}:{_RANK0_FIRST_CALLS}"
    if dist.get_rank() == 0:
        try:
            out = fn()
        except BaseException:
            store.set(key, "fail")
            raise
        store.set(key, "ok")
        return out
    store.wait([key], datetime.timedelta(seconds=timeout_s))
    if store.get(key) != b"ok":
        raise RuntimeError("rank 0 failed while building the shared caches")
    return fn()


def broadcast_object(obj, src: int = 0):
    import torch.distributed as dist
    if not collectives_on():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def _row_sizes(local: torch.Tensor) -> List[int]:
    import torch.distributed as dist
    world = dist.get_world_size()
    staged = _staged(local)
    n = torch.tensor([local.shape[0]], device="cpu" if staged else local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    return [int(s.item()) for s in sizes]


def allgather_rows(local: torch.Tensor) -> torch.Tensor:
    """Concatenate per-rank row blocks (different row counts allowed) in rank order, on EVERY rank."""
    import torch.distributed as dist
    if not is_initialized() or dist.get_world_size() == 1:
        return local
    sizes = _row_sizes(local)
    staged = _staged(local)
    src = local.detach().cpu() if staged else local.detach()
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(src.shape[1:]), device=src.device, dtype=src.dtype)
    pad[:src.shape[0]] = src
    bufs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(bufs, pad)
    out = torch.cat([b[:s] for b, s in zip(bufs, sizes)])
    return out.to(local.device) if staged else out


def gather_rows(local: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Concatenate per-rank row blocks (different row counts allowed) on ``dst`` in rank order."""
    import torch.distributed as dist
    if not is_initialized() or dist.get_world_size() == 1:
        return local
    world, r = dist.get_world_size(), dist.get_rank()
    sizes = _row_sizes(local)
    staged = _staged(local)
    src = local.detach().cpu() if staged else local.detach()
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(src.shape[1:]), device=src.device, dtype=src.dtype)
    pad[:src.shape[0]] = src
    bufs = [torch.empty_like(pad) for _ in range(world)] if r == dst else None
    dist.gather(pad, bufs, dst=dst)
    if r != dst:
        return None
    out = torch.cat([b[:s] for b, s in zip(bufs, sizes)])
    return out.to(local.device) if staged else out


# ------------------------------------------------------------------------------------------------
# inference: graph sharding
# ------------------------------------------------------------------------------------------------
# weights of (directed neighborhood edge, neighborhood, node at 29 queries) in units of one neighborhood row;
# DESCO_COST_WEIGHTS="e,n,v" overrides them for tuning runs (bench.py secondary.*.strong_scaling_8 measures a choice)
COST_WEIGHTS = tuple(float(x) for x in os.environ.get("DESCO_COST_WEIGHTS", "0.5,24,12").split(","))


def graph_costs(graphs: GraphSet, num_queries: int = 29, device=None, depth: int = 4) -> np.ndarray:
    """Per-graph cost c(g) of one inference pass, for cutting a dataset into ranks' shards, in units of one
    neighborhood row:   c = rows + 0.5 edges + 24 neighborhoods + 12 (Q / 29) nodes      (COST_WEIGHTS).

    The pass is dominated by the neighborhood stage: every node of every canonical neighborhood is a ROW of every SHMP
    layer (a 64-wide output row and its share of the folded GEMM), every directed neighborhood EDGE a gathered source
    row, every NEIGHBORHOOD a row of the anchor GEMM, the post-MLP and the head; the gossip stage adds Q rows per NODE
    and each graph its share of the fixed per-launch work.  Weights from MI355X timings of cost-balanced shards run one
    after another (bench.py secondary.*.strong_scaling_8; sweep in
    profiles/r5_i_cost_weights.log): with the rows alone equalised, a Syn_1827 shard of 1 607 small graphs (178 k
    neighborhoods, 45 M edges) took 10.5 ms against 8.7 ms for 54 large ones (61 k, 36 M); with (0.5, 8, 4.8) the slowest
    of 8 shards was 5.8 % above the mean (predicted 8-GPU efficiency 0.945), with (0.5, 24, 12) 1.9 % (0.981; MSRC-21 +
    IMDB 0.985).  With a CUDA ``device`` all four counts are EXACT: the device partition builder (csrc/partition_dev.hip;
    0.1 s for Syn_1827) delivers them -- a size proxy cannot (the neighborhood of a node of a dense 700-node graph has
    hundreds of rows, a molecule's nine: round 4's degree proxy gave the 8 Syn_1827 shards times between 2.7 and
    16.5 ms, a predicted 8-GPU efficiency of 0.54).  Without a device (CPU tests, planning tools) a bound stands in:
    rows = sum over nodes of min(rank + 1, 4-hop ball bound), one neighborhood per node, edges = 0.7 mean degree per row."""
    deg = np.diff(graphs.rowptr).astype(np.float64)
    gid = graphs.node_graph_ids()
    G = graphs.num_graphs
    n_g = np.diff(graphs.graph_ptr).astype(np.float64)
    rows = neigh = edges = None
    if device is not None and str(device).startswith("cuda"):
        from .partition import build_partition_device
        try:
            part = build_partition_device(graphs, depth, device)
            ng = part.neigh_index[:, 0]
            B = part.num_neigh
            rows = np.bincount(ng, weights=np.diff(part.count_ptr).astype(np.float64) + 1.0, minlength=G)
            neigh = np.bincount(ng, minlength=G).astype(np.float64)
            per_row = np.diff(part.vrowptr.astype(np.int64)).reshape(-1, 4).sum(1).astype(np.float64)
            owner = np.concatenate([np.repeat(np.arange(B), np.diff(part.count_ptr)), np.arange(B)])
            edges = np.bincount(ng[owner], weights=per_row, minlength=G)
        except RuntimeError as e:
            # ONLY "a graph beyond the device builder's per-wave LDS workspace" falls back to the bound; anything else
            # (out of memory, a HIP error, a bad argument) is an error of this rank and must not silently give it other
            # costs -- and so other shard cuts -- than its peers (shard_cuts lets one rank decide for all besides)
            if isinstance(e, torch.cuda.OutOfMemoryError) or "does not fit the LDS workspace" not in str(e):
                raise
            rows = None
    if rows is None:
        rank = np.arange(graphs.num_nodes, dtype=np.float64) - graphs.graph_ptr[:-1][gid] + 1.0
        d = np.maximum(deg, 1.0)
        ball = np.minimum(1.0 + d * (1.0 + (d - 1.0) * (1.0 + (d - 1.0) * (1.0 + (d - 1.0)))), n_g[gid])
        rows = np.zeros(G)
        np.add.at(rows, gid, np.minimum(rank, ball))
        neigh = n_g
        mean_deg = np.zeros(G)
        np.add.at(mean_deg, gid, deg)
        edges = rows * 0.7 * mean_deg / np.maximum(n_g, 1.0)
    w_e, w_n, w_v = COST_WEIGHTS
    return rows + w_e * edges + w_n * neigh + w_v * (num_queries / 29.0) * n_g


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


def shard_cuts(graphs: GraphSet, parts: int, num_queries: int = 29, device=None, depth: int = 4,
               agree: bool = False) -> List[Tuple[int, int]]:
    """The ``parts`` contiguous graph ranges of equal cost.  ``agree=True`` (callers that act as a rank of the process
    group: every rank makes this call): rank 0 computes the cuts and broadcasts them, so that all ranks use the SAME
    cuts whatever happens to one of them (a rank that fell back to the cost bound, or that runs another weight setting,
    would otherwise drop or duplicate the graphs between its cuts and its neighbours'); a failure on rank 0 is raised
    on every rank instead of leaving the others in the collective."""
    if not (agree and is_initialized() and world_size() > 1):
        return contiguous_shards(graph_costs(graphs, num_queries, device, depth), parts)
    res = None
    if rank() == 0:
        try:
            res = ("ok", contiguous_shards(graph_costs(graphs, num_queries, device, depth), parts))
        except Exception as e:          # noqa: BLE001  (handed to every rank below)
            res = ("error", f"{type(e).__name__}: {e}")
    res = broadcast_object(res, 0)
    if res[0] != "ok":
        raise RuntimeError("rank 0 could not compute the shard cuts: " + res[1])
    cuts = [tuple(c) for c in res[1]]
    if len(cuts) != parts or cuts[0][0] != 0 or cuts[-1][1] != graphs.num_graphs:
        raise RuntimeError(f"shard cuts from rank 0 do not cover this rank's dataset ({graphs.num_graphs} graphs): "
                           "the ranks hold different datasets")
    return cuts


def shard_graphs(graphs: GraphSet, rank: int, world_size: int, num_queries: int = 29, device=None, depth: int = 4,
                 agree: bool = False):
    lo, hi = shard_cuts(graphs, world_size, num_queries, device, depth, agree)[rank]
    return graphs.subset(lo, hi), (lo, hi)


# ------------------------------------------------------------------------------------------------
# training: batch packing and gradient buckets
# ------------------------------------------------------------------------------------------------
def step_groups(batch_sizes: Sequence[int], world: int) -> List[List[Optional[int]]]:
    """Pack the reference's batch stream (shuffle=False, main.py:195) into optimisation steps of
    ``world`` consecutive batches: group k = batches [k*world, (k+1)*world), rank r takes the r-th.
    A short last group is padded with None: that rank runs no forward but still joins the step's
    collectives with zero gradients, so every rank executes the same number of steps (Lightning's
    DistributedSampler pads by repeating samples instead, which double-counts them)."""
    n = len(batch_sizes)
    groups = []
    for k in range(0, n, world):
        g: List[Optional[int]] = list(range(k, min(k + world, n)))
        g += [None] * (world - len(g))
        groups.append(g)
    return groups


def mean_loss_weight(batch_sizes: Sequence[int], group: Sequence[Optional[int]], r: int) -> float:
    """Weight of rank r's (mean) loss in a step so that the SUM all-reduce of the weighted
    gradients is the gradient of the mean loss over the union of the group's batches:
    B_r / sum_r' B_r' (equal batches: 1 / world, DDP's mean)."""
    tot = sum(batch_sizes[i] for i in group if i is not None)
    i = group[r]
    return 0.0 if i is None or tot == 0 else batch_sizes[i] / tot


_HOOKED_STEPS = 0      # GradBuckets between zero() and finish(): their hooks issue all-reduces from inside backward


def hooks_active() -> bool:
    """A hook-driven data-parallel step is in progress (GradBuckets.zero() ... finish()): autograd hooks issue bucket
    all-reduces on whatever stream a gradient is accumulated on, so a forward that forks part of the model onto a second
    stream must not do so now (NeighborhoodCountingModel.train_forward).  The replayed form (trainer.DDPReplay) packs
    and reduces after the backward has joined its streams: there the fork is safe."""
    return _HOOKED_STEPS > 0


class GradBuckets:
    """Gradients of a model in ``num_buckets`` flat fp32 buffers; ``p.grad`` are views.

    The whole neighborhood model is 5.24 MB (gossip 0.58 MB): over xGMI a ring all-reduce of that
    size is latency-bound, so a few large buckets beat per-tensor collectives.  Buckets are filled
    in REVERSE registration order (the order backward produces gradients: head first, layer 0
    last); a post-accumulate hook counts a bucket's gradients and issues its asynchronous
    all-reduce the moment it is complete -- the rest of the backward overlaps it.  With the fused training
    trunk (autograd.ShmpTrunk, default) the gradients of ALL SHMP-layer weights appear at once, at the end of the
    trunk's backward, so only the head / post-MLP buckets overlap compute; the layer buckets (most of the 5.24 MB)
    are issued behind the trunk.  At 5 MB per step against ~5 ms of kernels that exposes < 0.1 ms per step over
    xGMI; per-op autograd (gnn_model.FUSED_TRAIN_TRUNK = False) restores the overlap at
    1.5x the step time.
    ``finish()`` issues the buckets whose parameters received no gradient (the never-used
    query-side ``anchor_mlp``, SURVEY A10: DDP's find_unused_parameters semantics, they reduce
    zeros) and waits.  After the first step those parameters are moved into ONE last bucket, so that
    from step 1 on every other bucket completes -- and is issued -- from the hooks.  Collective order is identical on every rank because the autograd graph is.
    """

    def __init__(self, params: Sequence[torch.nn.Parameter], num_buckets: int = 4):
        self.params = [p for p in params if p.requires_grad]
        self.num_buckets = num_buckets
        self._hooks = []
        self._steps = 0
        self._layout(list(reversed(self.params)), [])
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.active = False

    def _layout(self, used: List[torch.nn.Parameter], unused: List[torch.nn.Parameter]):
        """(Re)build the flat buffers: ``used`` (in gradient-production order) cut into
        ``num_buckets`` buckets, then ONE last bucket of the parameters known to receive no gradient
        -- so that no never-completing bucket sits in front of the others (buckets are issued in
        index order).  Existing gradient values are carried over."""
        old = {id(p): p.grad for p in self.params if p.grad is not None}
        total = sum(p.numel() for p in used)
        target = max(1, -(-total // max(1, self.num_buckets)))
        self.buckets: List[torch.Tensor] = []
        self._members: List[List[torch.nn.Parameter]] = []
        cur: List[torch.nn.Parameter] = []
        cur_n = 0
        for p in used:
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= target:
                self._members.append(cur)
                cur, cur_n = [], 0
        if cur:
            self._members.append(cur)
        if unused:
            self._members.append(list(unused))
        self._bucket_of = {}
        for b, members in enumerate(self._members):
            # 16-byte aligned slices keep every view usable by vectorised kernels
            offs, off = [], 0
            for p in members:
                offs.append(off)
                off += (p.numel() + 3) // 4 * 4
            flat = torch.zeros(off, device=members[0].device, dtype=members[0].dtype)
            self.buckets.append(flat)
            for p, o in zip(members, offs):
                g = flat[o:o + p.numel()].view_as(p)
                if id(p) in old:
                    g.copy_(old[id(p)])
                p.grad = g
                self._bucket_of[id(p)] = b
        self._pending = [0] * len(self.buckets)
        self._next = 0
        self._handles = []
        self._seen = set()

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # -- per step ---------------------------------------------------------------------------------
    def zero(self):
        """Replaces optimizer.zero_grad(): the views stay attached to the buckets."""
        for flat in self.buckets:
            flat.zero_()
        self._pending = [len(m) for m in self._members]
        self._next = 0                      # buckets are issued strictly in index order on every
        self._handles = []                  # rank (collectives are matched by order)
        if not self.active:
            global _HOOKED_STEPS
            _HOOKED_STEPS += 1
        self.active = True

    # -- the replayable form of a step (Trainer's DDPReplay): gradients land in fresh tensors (p.grad = None before the
    #    backward: autograd stores, it does not accumulate -- no add launch per parameter), ONE copy launch per 24
    #    tensors packs them into the buckets, the buckets are all-reduced between the two captured graphs, and the
    #    optimizer reads the bucket views -------------------------------------------------------------------------
    def views(self):
        """{id(p): the view of p's gradient inside its bucket} (the layout of this moment)."""
        out = {}
        for flat, members in zip(self.buckets, self._members):
            off = 0
            for p in members:
                out[id(p)] = flat[off:off + p.numel()].view_as(p)
                off += (p.numel() + 3) // 4 * 4
        return out

    def fill_zero(self):
        """Zero every bucket (parameters without a gradient contribute zeros); on the GPU one launch of this library's
        per bucket, capturable."""
        for flat in self.buckets:
            if flat.is_cuda:
                from . import ops
                ops.fill(flat, 0.0)
            else:
                flat.zero_()

    def pack_from_grads(self):
        """bucket view of p = p.grad for every parameter that has one (copy2d launches on the GPU: 24 tensors each)."""
        vs = self.views()
        pairs = [(p.grad, vs[id(p)]) for p in self.params if p.grad is not None and p.grad.data_ptr() != vs[id(p)].data_ptr()]
        if not pairs:
            return
        if pairs[0][1].is_cuda:
            from . import ops
            as2d = lambda t: t.reshape(1, -1) if t.dim() != 2 else t          # noqa: E731
            ops.copy2d_multi([(as2d(g.contiguous()), as2d(v)) for g, v in pairs])
        else:
            for g, v in pairs:
                v.copy_(g)

    def attach_views(self):
        """p.grad = its bucket view (what the optimizer is to read)."""
        vs = self.views()
        for p in self.params:
            p.grad = vs[id(p)]

    def allreduce_all(self):
        """All buckets, asynchronously in index order, then wait (the un-overlapped form: a captured backward cannot
        issue collectives from its hooks)."""
        hs = [all_reduce_(flat, "sum", async_op=True) for flat in self.buckets]
        for h in hs:
            if h is not None:
                h.wait()

    def _issue_ready(self, force: bool = False):
        while self._next < len(self.buckets) and (force or self._pending[self._next] == 0):
            h = all_reduce_(self.buckets[self._next], "sum", async_op=True)
            if h is not None:
                self._handles.append(h)
            self._next += 1

    def _on_grad(self, p):
        if not self.active:
            return
        if self._steps == 0:
            self._seen.add(id(p))
        self._pending[self._bucket_of[id(p)]] -= 1
        self._issue_ready()

    def _demote_unused(self):
        """After the FIRST step: parameters no rank produced a gradient for (agreed by a MAX
        all-reduce of the per-parameter flags -- a rank without a batch in this step saw none) move
        to the last bucket."""
        flags = torch.tensor([1 if id(p) in self._seen else 0 for p in self.params],
                             device=self.buckets[0].device, dtype=torch.int32)
        all_reduce_(flags, "max")
        flags = flags.tolist()
        unused = [p for p, f in zip(self.params, flags) if not f]
        last = len(self._members) - 1
        if unused and len(unused) < len(self.params) and any(self._bucket_of[id(p)] != last for p in unused):
            gone = {id(p) for p in unused}
            self._layout([p for p in reversed(self.params) if id(p) not in gone], unused)

    def finish(self, scale: Optional[float] = None):
        """Issue what is left, wait for every bucket, optionally scale (mean = 1 / world)."""
        self._issue_ready(force=True)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.active:
            global _HOOKED_STEPS
            _HOOKED_STEPS -= 1
        self.active = False
        if self._steps == 0:
            self._demote_unused()
        self._steps += 1
        if scale is not None and scale != 1.0:
            for flat in self.buckets:
                flat.mul_(scale)


def broadcast_params(module: torch.nn.Module, src: int = 0) -> None:
    """Every rank starts from rank ``src``'s parameters and buffers (what DDP does at construction)."""
    import torch.distributed as dist
    if not collectives_on():
        return
    ts = [t for t in list(module.parameters()) + list(module.buffers())]
    if not ts:
        return
    flat = torch.cat([t.detach().reshape(-1).float() for t in ts])
    if _staged(flat):
        h = flat.cpu()
        dist.broadcast(h, src=src)
        flat = h.to(flat.device)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
            off += n


def allreduce_grads(params: Sequence[torch.nn.Parameter], mode: str = "mean") -> None:
    """One-shot (blocking) gradient all-reduce for callers without ``GradBuckets``: a single flat
    buffer.  mode="mean" for the neighborhood loss (a mean), "sum" for the gossip loss (a sum).
    Parameters without a gradient contribute zeros."""
    if world_size() == 1:
        return
    params = [p for p in params if p.requires_grad]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in params])
    all_reduce_(flat, "sum")
    if mode == "mean":
        flat /= world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
