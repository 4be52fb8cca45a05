"""Device-resident batch containers (replace PyG ``HeteroData``/``Data`` batches on the hot path).

The reference moves a collated ``HeteroData`` to the GPU per batch (Lightning) and the model reads
``node_feature_dict`` / ``edge_index_dict`` (gnn_model.py:60-63).  Here a batch is a handful of
flat int32/fp32 tensors in HBM in the layout the kernels consume; the PyG-style dict views are
kept as (lazy, host-side) properties for interop and tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graphs import GraphSet
from .partition import NeighborhoodPartition


def _norm_device(device) -> torch.device:
    """torch.device("cuda") and torch.device("cuda:0") compare unequal; always carry the index."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def _i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


def _transpose_index(vrowptr: np.ndarray, vcol: np.ndarray, n_src: int):
    """For every source row, the virtual rows that read it (the index of the backward gather)."""
    cnt = np.diff(vrowptr.astype(np.int64))
    vrow_of_edge = np.repeat(np.arange(len(cnt), dtype=np.int64), cnt)
    src = vcol.astype(np.int64)
    order = np.argsort(src, kind="stable")
    t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n_src))])
    return t_rowptr, vrow_of_edge[order]


class _TrainIndexMixin:
    """Lazily built device indices the backward pass needs (transposed CSR, segment ids)."""

    def train_index(self):
        if getattr(self, "_train_index", None) is None:
            from . import ops
            # both built on the device (desco_vcsr_transpose_sym / desco_segment_ids): the blocks
            # are symmetric, so the transposed index is row-local (include/desco_hip.h)
            num_count = getattr(self, "num_count", self.num_rows)
            t_rowptr, t_col = ops.vcsr_transpose_sym(self.vrowptr, self.vcol, self.num_rows,
                                                     self.slots, num_count)
            seg_ptr = self._seg_ptr_device()
            n_seg_rows = num_count if self.slots == 4 else self.num_rows
            seg_id = ops.segment_ids(seg_ptr, n_seg_rows)
            dev = self.vrowptr.device
            S = self.slots
            self._train_index = {
                "t_rowptr": t_rowptr, "t_col": t_col, "seg_id": seg_id,
                "ident_ptr": torch.arange(n_seg_rows + 1, device=dev, dtype=torch.int32),
                # the same transposed index for gradient rows laid out [row][S + 1 blocks of 64] (S relation
                # slots + the self block, one GEMM output of the fused training trunk): virtual row
                # k S + s -> k (S + 1) + s
                "t_col_s1": (t_col + torch.div(t_col, S, rounding_mode="floor")).to(torch.int32),
            }
        return self._train_index


class NeighborhoodBatch(_TrainIndexMixin):
    """B canonical neighborhoods: N_c count rows followed by B canonical rows, 4-slot CSR."""

    slots = 4

    def __init__(self, part: NeighborhoodPartition, device, node_feature: Optional[torch.Tensor] = None,
                 y: Optional[torch.Tensor] = None, input_dim: int = 1):
        self.part = part
        self.device = _norm_device(device)
        device = self.device
        self.num_graphs = part.num_neigh
        self.num_count = part.num_count
        self.num_rows = part.num_rows
        da = getattr(part, "device_arrays", None)
        if da is not None and _norm_device(da["device"]) == device:    # built on this device already
            self.count_ptr, self.vrowptr, self.vcol = da["count_ptr"], da["vrowptr"], da["vcol"]
        else:
            self.count_ptr = _i32(part.count_ptr, device)
            self.vrowptr = _i32(part.vrowptr, device)
            self.vcol = _i32(part.vcol, device)
        self.input_dim = input_dim if node_feature is None else node_feature.shape[1]
        # None == all-zero features (ZeroNodeFeat, workload.py:431-440): pre_mp output is its bias
        self.node_feature = None if node_feature is None else node_feature.to(device).float()
        self.y = None if y is None else y.to(device)

    def to(self, device):
        if _norm_device(device) == self.device:
            return self
        return NeighborhoodBatch(self.part, device, self.node_feature, self.y, self.input_dim)

    def _seg_ptr_host(self):
        return self.part.count_ptr.astype(np.int64)

    def _seg_ptr_device(self):
        return self.count_ptr

    def max_count_rows(self) -> int:
        """the largest number of count rows of a neighborhood of this batch (host data, cached)"""
        m = self.__dict__.get("_max_count_rows")
        if m is None:
            d = np.diff(self.part.count_ptr)
            m = self.__dict__["_max_count_rows"] = int(d.max()) if len(d) else 0
        return m

    def pool_index(self, tile_rows: Optional[int] = None):
        """(pool_bits, pool_slot, num_slots) of the fused pooling (desco_shmp_layer_pool_bf16x6_f32):
        per wave tile (``tile_rows`` = 16 or 32 count rows; default: what the library's layer kernel
        uses), the bitmap of rows that END a neighborhood and the first partial slot of the tile (a
        tile uses one slot per neighborhood that has a row in it)."""
        if tile_rows is None:
            from . import ops
            tile_rows = ops.pool_tile_rows()
        if tile_rows not in (16, 32):
            raise ValueError("tile_rows must be 16 or 32")
        cache = self.__dict__.setdefault("_pool_index", {})
        if tile_rows not in cache:
            TR, sh = tile_rows, 4 if tile_rows == 16 else 5
            cp = self.part.count_ptr.astype(np.int64)
            nc = int(cp[-1])
            if (np.diff(cp) <= 0).any():
                raise ValueError("fused pooling needs at least one count row per neighborhood")
            nt = (nc + TR - 1) // TR
            ends = cp[1:] - 1
            bits = np.zeros(nt, dtype=np.uint32)
            np.bitwise_or.at(bits, ends >> sh, (np.uint32(1) << (ends & (TR - 1)).astype(np.uint32)))
            pop = np.zeros(nt, dtype=np.int64)
            np.add.at(pop, ends >> sh, 1)
            last_row = np.minimum(TR * np.arange(nt, dtype=np.int64) + TR - 1, nc - 1)
            carry = ((bits >> (last_row & (TR - 1)).astype(np.uint32)) & 1) == 0    # a segment runs on into the next tile
            nseg = pop + carry
            slot = np.concatenate([[0], np.cumsum(nseg)])
            if slot[-1] >= 2 ** 31:
                raise ValueError("too many pooling slots for int32")
            dev = self.device
            cache[tile_rows] = (torch.from_numpy(bits.view(np.int32)).to(dev),
                                torch.from_numpy(slot[:-1].astype(np.int32)).to(dev), int(slot[-1]))
        return cache[tile_rows]

    def degree_table_index(self, max_rows: int = 1 << 16):
        """The count rows' S slot degrees as an index into their DISTINCT tuples (built once per batch, on the device):
        ``(uptr, row_id, vcol_t)`` or None.  The closed-form first layer (ZeroNodeFeat: every node of a type has the same
        input row) makes X_1[i] a function of row i's degree tuple alone, so X_1 = T[row_id] for the table T of the U
        distinct tuples -- ``uptr`` [U S + 1] int32 is their row-pointer array (what desco_degree_affine_f32 turns into
        T), ``row_id`` [num_count] int32 each count row's table row, and ``vcol_t`` = vcol with the sources of the
        relation slots 0 and 1 (count rows) replaced by their table rows, so that the second layer's launches gather
        from T (desco_shmp_layer_pool_table_f16x3_f32, which recomputes the launch's own rows from their degrees).  None when the batch has more than ``max_rows`` distinct tuples
        (the table must stay cache-resident to pay)."""
        if "_degree_table" in self.__dict__:
            return self.__dict__["_degree_table"]
        res = None
        S, nc, n = self.slots, self.num_count, self.num_rows
        if S == 4 and nc > 0:
            vr = self.vrowptr.to(torch.int64)
            deg = (vr[1:] - vr[:-1]).view(n, S)
            dc = deg[:nc]
            m = int(dc.max().item()) + 1
            if m ** S < 2 ** 62:
                key = ((dc[:, 0] * m + dc[:, 1]) * m + dc[:, 2]) * m + dc[:, 3]
                uniq, inv = torch.unique(key, return_inverse=True)
                if uniq.numel() <= max_rows:
                    ut = torch.stack([uniq // (m ** 3), (uniq // (m ** 2)) % m, (uniq // m) % m, uniq % m], dim=1)
                    uptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=ut.device), ut.reshape(-1).cumsum(0)])
                    # slot of every CSR entry: entry e belongs to (row, slot) pair rs(e); sources of slots 0, 1 are count rows
                    rs = torch.repeat_interleave(torch.arange(n * S, device=vr.device), deg.reshape(-1))
                    col = self.vcol.to(torch.int64)
                    low = (rs % S) < 2
                    if bool((col[low] < nc).all()):
                        row_id = inv.to(torch.int32)
                        vcol_t = torch.where(low, inv[col.clamp(max=nc - 1)], col).to(torch.int32)
                        res = (uptr.to(torch.int32).contiguous(), row_id.contiguous(), vcol_t.contiguous())
        self.__dict__["_degree_table"] = res
        return res

    # PyG-style views -----------------------------------------------------------------------
    @property
    def node_feature_dict(self) -> Dict[str, torch.Tensor]:
        f = self.node_feature
        if f is None:
            f = torch.zeros((self.num_rows, self.input_dim), device=self.device)
        return {"count": f[:self.num_count], "canonical": f[self.num_count:]}

    @property
    def edge_index_dict(self):
        return {k: torch.from_numpy(v) for k, v in self.part.edge_index_dict().items()}

    @property
    def batch_dict(self):
        b = np.repeat(np.arange(self.num_graphs), np.diff(self.part.count_ptr))
        return {"count": torch.from_numpy(b), "canonical": torch.arange(self.num_graphs)}


def tconv_split(n: int, edges) -> Tuple[np.ndarray, np.ndarray]:
    """Directed edges (src, dst) of an undirected graph and their tride flag.

    tride = NOT(endpoints share a neighbour) -- ToTconvHetero, transforms.py:201-221
    (T = A*(A@A) + A, triangle iff T > 1)."""
    A = np.zeros((n, n), dtype=np.int64)
    for a, b in edges:
        if a != b:
            A[a, b] = A[b, a] = 1
    T = A * (A @ A) + A
    src, dst = np.nonzero(A)
    return np.stack([src, dst]), T[src, dst] <= 1


class QueryBatch(_TrainIndexMixin):
    """The query graphs as one single-type ("union_node") block with 2 relation slots
    (union_triangle, union_tride) -- lightning_model.py:37-87, 291-309."""

    slots = 2

    def __init__(self, queries: Sequence[Tuple[int, Sequence[Tuple[int, int]]]], device,
                 input_dim: int = 1, node_feature: Optional[torch.Tensor] = None):
        """``node_feature`` [sum of query sizes, input_dim]: the labelled queries of --use_node_feature
        (one-hot rows, main.py:51-62); None = zeros (the unlabelled standard queries)."""
        self.queries = [(int(n), [tuple(e) for e in es]) for n, es in queries]
        self.device = _norm_device(device)
        device = self.device
        self.input_dim = input_dim
        sizes = np.array([n for n, _ in self.queries], dtype=np.int64)
        gp = np.concatenate([[0], np.cumsum(sizes)])
        N = int(gp[-1])
        cnt = np.zeros(2 * N, dtype=np.int64)
        ents = []
        for g, (n, es) in enumerate(self.queries):
            ei, tride = tconv_split(n, es)
            for (s, d), t in zip(ei.T.tolist(), tride.tolist()):
                ents.append(((d + gp[g]) * 2 + int(t), s + gp[g]))
        ents.sort()
        for v, _ in ents:
            cnt[v] += 1
        self.num_graphs = len(self.queries)
        self.num_rows = N
        self.graph_ptr_host = gp
        self.vrowptr = _i32(np.concatenate([[0], np.cumsum(cnt)]), device)
        self.vcol = _i32(np.array([c for _, c in ents], dtype=np.int64), device)
        self.graph_ptr = _i32(gp, device)
        self.node_feature = None
        if node_feature is not None:
            nf = torch.as_tensor(node_feature, dtype=torch.float32)
            if nf.shape != (N, input_dim):
                raise ValueError(f"query node_feature must be [{N}, {input_dim}], got {tuple(nf.shape)}")
            self.node_feature = nf.to(device).contiguous()

    def _seg_ptr_host(self):
        return self.graph_ptr_host

    def _seg_ptr_device(self):
        return self.graph_ptr


class GossipBatch:
    """Whole target graphs for the gossip stage: symmetric CSR (ascending cols) + x [N,Q].

    Equivalent of the PyG ``Data`` batch of GossipDataset (workload.py:48-150) after the
    per-call canonicalisation of gnn_model.py:246-248, done once."""

    def __init__(self, graphs: GraphSet, device, x: Optional[torch.Tensor] = None,
                 y: Optional[torch.Tensor] = None):
        self.graphs = graphs
        self.device = _norm_device(device)
        device = self.device
        self.num_graphs = graphs.num_graphs
        self.num_nodes = graphs.num_nodes
        if graphs.num_directed_edges >= 2 ** 31:
            raise ValueError("gossip batch too large for int32 edge offsets; split it")
        self.rowptr = _i32(graphs.rowptr, device)
        self.col = _i32(graphs.col, device)
        self.graph_ptr = _i32(graphs.graph_ptr, device)
        self.x = None if x is None else x.to(device).float().contiguous()
        self.y = None if y is None else y.to(device)
        self._tile_perm = None
        self._work_queue = None

    @property
    def work_queue(self):
        """Two zeroed int64 words: the work-item tickets of this batch's fused gossip launches (the kernel leaves
        them zero).  One per batch object: launches of one batch are stream-ordered; batches that run concurrently
        (other streams, other graph replays) each have their own."""
        if self._work_queue is None:
            self._work_queue = torch.zeros(2, dtype=torch.int64, device=self.device)
        return self._work_queue

    @property
    def tile_perm(self):
        """Degree-balanced row order of the fused gossip kernel's neighbour-sum phase per 128-node tile
        (ops.gossip_tile_order): a function of the CSR alone, computed on first use."""
        if self._tile_perm is None and self.device.type == "cuda":
            from . import ops
            self._tile_perm = ops.gossip_tile_order(self.rowptr, self.num_nodes)
        return self._tile_perm

    def to(self, device):
        if _norm_device(device) == self.device:
            return self
        return GossipBatch(self.graphs, device, self.x, self.y)

    @property
    def edge_index(self):
        rp = self.graphs.rowptr
        dst = np.repeat(np.arange(self.num_nodes, dtype=np.int64), np.diff(rp))
        return torch.from_numpy(np.stack([self.graphs.col.astype(np.int64), dst]))

    @property
    def batch(self):
        return torch.from_numpy(self.graphs.node_graph_ids())
