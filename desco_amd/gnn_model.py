"""GNN building blocks with the reference's names, parameters and state-dict keys
(subgraph_counting/gnn_model.py:18-419), executing on MI355X through libdesco_hip.so.

The modules only HOLD parameters in the reference's layout (so the authors' checkpoints load);
``forward`` does not run them as torch layers.  Instead the weights are folded ("packed") into the
operands of the HIP kernels:

  SHMP layer (SAGEConv x edge types + to_hetero sum + updates Linear, gnn_model.py:262-264, 395):
      x'_d = relu( sum_s agg_s (U_n W_s)^T + x_d U_x^T + (U_n sum_s b_s + c) )
      -> one gather kernel + one MFMA GEMM with K = (S+1)*64 per destination type.
  Gossip (GossipConv, gnn_model.py:280-350): see ``gossip_forward`` and DESIGN.md section 4.2.

There is no CPU / eager fallback: inputs must live on the GPU.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch
import torch.nn as nn

from . import ops
from .batch import GossipBatch, NeighborhoodBatch, QueryBatch

H = 64
# one fused gather+MFMA launch per destination type and layer (False: gather kernel + GEMM)
FUSED_SHMP_LAYER = True
# scalars pre-pass + one on-chip kernel for the whole gossip network (False: 7 launches via HBM)
FUSED_GOSSIP = True
# the fused gossip pass in the three-product fp16 form (csrc/gossip_f16.hip); False: the six-product bf16 kernel
# (csrc/gossip_fused.hip), kept as its cross-check
GOSSIP_F16X3 = os.environ.get("DESCO_GOSSIP_F16X3", "1") != "0"
# inference GEMMs (anchor, post MLP, head, canonical table) on the bf16 matrix pipe with fp32-level
# accuracy (bf16x6 split, csrc/gemm_split.hip); False: v_mfma_f32_32x32x2_f32 (gemm_f32.hip)
GEMM_BF16X6 = True
# the large inference GEMM (anchor MLP) in the THREE-product fp16 hi/lo form (csrc/gemm_f16x3.hip): half the MFMAs and
# two operand planes instead of three, power-of-two scales per weight matrix and per activation row; measured error
# below the f32 MFMA's (tests/test_kernels_gpu.py::test_gemm_f16x3_is_fp32_accurate).  Needs GEMM_BF16X6.
GEMM_F16X3 = os.environ.get("DESCO_GEMM_F16X3", "1") != "0"
# True: the fused SHMP layer's MFMA blocks also run as the bf16x6 split (csrc/shmp_layer.hip, K <= 192)
SHMP_BF16X6 = True
# ... in the three-product fp16 form with per-row power-of-two scales (csrc/shmp_layer16.hip, F16 instantiations); needs
# SHMP_BF16X6 and 16-row wave tiles
SHMP_F16X3 = os.environ.get("DESCO_SHMP_F16X3", "1") != "0"
# training: the query model's trunk (<= 144 rows) as one single-workgroup launch per direction (csrc/shmp_small.hip);
# False: the general per-layer launches of autograd.ShmpTrunk (its cross-check in the tests)
SMALL_TRUNK_KERNEL = True
# True: global_add_pool of the count rows fused into the layer kernel's epilogue (partials per
# (32-row tile, neighborhood) + a small reduce) instead of one segment_sum pass over X_l per layer
FUSED_POOLING = True
# ... and the layers' partials reduced by ONE launch at the end of the layer loop (False: one launch per layer)
POOL_REDUCE_MULTI = os.environ.get("DESCO_POOL_REDUCE_MULTI", "1") != "0"
# ... and the closed-form first layer's count launch leaves its partial sums too (desco_degree_affine_pool_f32) instead
# of a segment-sum pass over the rows it has just written
POOL_FIRST_LAYER = os.environ.get("DESCO_POOL_FIRST_LAYER", "1") != "0"
# the canonical rows of every layer are stored ONCE, in their column block of the anchor operand [B, 64 (L + 1)]: the
# canonical launches read their own rows from there (desco_shmp_layer_f16x3_f32: xself) and the table products too,
# instead of from a second copy behind the count rows of X_l (False: both copies, rounds 2-5)
CANON_ROWS_ONCE = os.environ.get("DESCO_CANON_ROWS_ONCE", "1") != "0"
# Training: the SHMP layer loop + anchor + pooling as ONE autograd node whose forward and backward are C-ABI
# launches on its own buffers (autograd.ShmpTrunk); False: one autograd Function per op (round 2; kept for
# --neigh_dropout > 0 and as the cross-check of the fused node's gradients)
FUSED_TRAIN_TRUNK = True
# degree-balanced row order inside the fused gossip kernel's 128-node tiles (desco_gossip_tile_order); results are
# bit-identical with and without it
GOSSIP_TILE_ORDER = os.environ.get("DESCO_GOSSIP_TILE_ORDER", "1") != "0"
_RELEASED = object()        # placeholder of a layer's rows that shmp_forward has released

TARGET_NODE_TYPES = ["count", "canonical"]
# metadata of to_hetero_old(tconv_target=True), lightning_model.py:376-383
TARGET_EDGE_TYPES_TCONV = [
    ("count", "union_triangle", "count"),
    ("count", "union_tride", "count"),
    ("count", "union_triangle", "canonical"),
    ("count", "union_tride", "canonical"),
    ("canonical", "union_triangle", "count"),
    ("canonical", "union_tride", "count"),
]
# lightning_model.py:392-397
TARGET_EDGE_TYPES_UNION = [
    ("count", "union", "canonical"),
    ("canonical", "union", "count"),
    ("count", "union", "count"),
]
QUERY_NODE_TYPES = ["union_node"]
QUERY_EDGE_TYPES_TCONV = [
    ("union_node", "union_triangle", "union_node"),
    ("union_node", "union_tride", "union_node"),
]
QUERY_EDGE_TYPES_UNION = [("union_node", "union", "union_node")]


def _require_hidden(hidden_dim):
    if hidden_dim != H:
        raise NotImplementedError(
            f"desco_amd kernels are specialised for hidden_dim == {H} (reference default, "
            f"config.py:250); got {hidden_dim}")


class SAGEConv(nn.Module):
    """Sum-aggregate then Linear (gnn_model.py:362-419).  Holds ``lin``."""

    def __init__(self, in_channels, out_channels, aggr="add", **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels)

    def reset_parameters(self):
        self.lin.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None, size=None, res_n_id=None):
        """Stand-alone call with PyG semantics (gnn_model.py:372-400) on device tensors."""
        if isinstance(x, torch.Tensor):
            x = (x, x)
        x_src, x_dst = x
        n_dst = x_dst.shape[0] if size is None else size[1]
        if edge_index is None:
            edge_index = torch.zeros((2, 0), dtype=torch.long, device=x_src.device)
        if edge_index.numel() != 0:
            edge_index = edge_index[:, edge_index[0] != edge_index[1]]     # :389-390
        dst, order = torch.sort(edge_index[1])
        col = edge_index[0][order].to(torch.int32)
        rowptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=x_src.device)
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_dst), 0)
        agg = ops.csr_gather_sum(x_src.float().contiguous(), rowptr.to(torch.int32), col, n_dst, 1)
        return ops.gemm(agg, self.lin.weight.t().contiguous(), self.lin.bias)

    def __repr__(self):
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)


class GossipConv(nn.Module):
    """Direction-gated message passing conditioned on the query embedding (gnn_model.py:280-359).
    Holds ``lin_com``, ``lin_update``, ``lin_gate``; executed by ``gossip_forward``."""

    def __init__(self, in_channels, out_channels, emb_channels, aggr="add", **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_com = nn.Linear(in_channels, out_channels)
        self.lin_update = nn.Linear(out_channels + in_channels, out_channels)
        self.lin_gate = nn.Sequential(
            nn.Linear(emb_channels, out_channels), nn.Sigmoid(),
            nn.Linear(out_channels, 1), nn.Sigmoid(), nn.LeakyReLU())

    def _gate_value(self, query_emb: torch.Tensor):
        """lin_gate(query_emb) (gnn_model.py:294-301, 357-359) -> [Q, 1].  Written out: the 64 -> 1 Linear as a product
        and a row sum instead of a matrix-vector product (rocBLAS gemv reduces with float atomics: see ``_mv``)."""
        l0, l2 = self.lin_gate[0], self.lin_gate[2]
        h = torch.sigmoid(torch.nn.functional.linear(query_emb, l0.weight, l0.bias))
        g = torch.sigmoid((h * l2.weight[0]).sum(-1, keepdim=True) + l2.bias)
        return torch.nn.functional.leaky_relu(g, self.lin_gate[4].negative_slope)

    def forward(self, x, edge_index, edge_weight=None, size=None, res_n_id=None, query_emb=None):
        """Stand-alone layer call with the reference's semantics (gnn_model.py:303-350) on device
        tensors, for ONE query: ``out = lin_update([sum_j w_ji * lin_com(x_j) | x])`` with
        ``w_ji = gate`` where ``edge_weight`` is True and ``1 - gate`` elsewhere.  (The model path
        does not call this: ``BaseGNN.forward`` batches all queries and both layers into the fused
        kernels.)  ``edge_weight`` must be the direction flag ``src < dst`` the reference computes
        (gnn_model.py:246-248); it is derived when omitted."""
        if edge_index.numel():
            edge_index = edge_index[:, edge_index[0] != edge_index[1]]          # remove_self_loops
        src, dst = edge_index[0], edge_index[1]
        if edge_weight is None:
            both = torch.cat([edge_index, edge_index.flip(0)], 1)               # to_undirected
            both = torch.unique(both, dim=1)
            src, dst = both[0], both[1]
        elif not torch.equal(edge_weight.bool(), src < dst):
            raise NotImplementedError("GossipConv.forward: edge_weight must be the flag src < dst")
        n = x.shape[0]
        gate = 0.5 if query_emb is None else float(self.lin_gate(query_emb.reshape(1, -1)).detach().reshape(()))
        order = torch.argsort(dst, stable=True)
        col = src[order].to(torch.int32)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
        msg = ops.gemm(x.float().contiguous(), self.lin_com.weight.t().contiguous(), self.lin_com.bias)
        g = torch.tensor([gate], device=x.device, dtype=torch.float32)
        agg = ops.gossip_gather(msg, rowptr.to(torch.int32), col, n, 1, g)      # [n, 64]
        return ops.gemm(agg, self.lin_update.weight.t().contiguous(), self.lin_update.bias,
                        a2=x.float().contiguous())

    def __repr__(self):
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)


class BaseGNNCore(nn.Module):
    """pre_mp + L x (conv, update) (gnn_model.py:115-277).  ``to_hetero`` re-creates PyG's
    per-node-type / per-edge-type module copies with its state-dict naming
    ("__".join(edge_type))."""

    def __init__(self, input_dim, hidden_dim, output_dim, args, **kwargs):
        super().__init__()
        _require_hidden(hidden_dim)
        self.dropout = args.dropout
        self.layer_num = args.layer_num
        self.conv_type = args.conv_type
        self.use_hetero = args.use_hetero
        self.kwargs = kwargs
        self.input_dim = input_dim
        pre_dim_out = hidden_dim
        self.pre_mp = nn.Sequential(nn.Linear(input_dim, pre_dim_out))          # :131
        self.input_pattern_emb = "input_pattern_emb" in kwargs                  # :144-153
        if self.input_pattern_emb:
            pre_dim_out += kwargs["emb_channels"]
        self.convs = nn.ModuleList()
        self.updates = nn.ModuleList()
        for l in range(args.layer_num):
            hidden_input_dim = hidden_dim
            if l == 0 and self.input_pattern_emb:
                hidden_input_dim = hidden_dim + kwargs["emb_channels"]
            if args.conv_type == "GOSSIP":
                self.convs.append(GossipConv(hidden_input_dim, hidden_dim,
                                             emb_channels=kwargs["emb_channels"]))        # :178-183
            elif args.conv_type == "SAGE":
                self.convs.append(SAGEConv(hidden_input_dim, hidden_dim, aggr="add"))     # :187
                self.updates.append(nn.Linear(2 * hidden_dim, hidden_dim))                # :190
            else:
                raise NotImplementedError(
                    f"conv_type {args.conv_type!r}: only SAGE (SHMP) and GOSSIP are on the hot path")
        self.post_input_dim = hidden_dim * args.layer_num + pre_dim_out          # :207
        self.node_types: Optional[List[str]] = None
        self.edge_types: Optional[List[Tuple[str, str, str]]] = None

    def to_hetero(self, node_types: Sequence[str], edge_types: Sequence[Tuple[str, str, str]]):
        """pyg.nn.to_hetero(aggr="sum") equivalent for this module [EXT, SURVEY App. C]."""
        if self.conv_type != "SAGE":
            raise NotImplementedError("to_hetero is only defined for the SAGE (SHMP) core")
        if self.node_types is not None:
            raise RuntimeError("model is already heterogeneous")
        in_dim = self.pre_mp[0].in_features
        self.pre_mp = nn.Sequential(nn.ModuleDict({t: nn.Linear(in_dim, H) for t in node_types}))
        self.convs = nn.ModuleList([
            nn.ModuleDict({"__".join(et): SAGEConv(H, H) for et in edge_types})
            for _ in range(self.layer_num)])
        self.updates = nn.ModuleList([
            nn.ModuleDict({t: nn.Linear(2 * H, H) for t in node_types})
            for _ in range(self.layer_num)])
        self.node_types, self.edge_types = list(node_types), [tuple(e) for e in edge_types]
        return self

    def slot_keys(self, dst_type: str) -> List[str]:
        """Module keys of the relation slots (triangle, tride) x (source types) feeding dst_type."""
        keys = []
        srcs = [dst_type] if len(self.node_types) == 1 else ["count", "canonical"]
        for src in srcs:
            if src == "canonical" and dst_type == "canonical":
                continue
            for rel in ("union_triangle", "union_tride"):
                et = (src, rel, dst_type)
                if et in self.edge_types:
                    keys.append("__".join(et))
                elif (src, "union", dst_type) in self.edge_types:   # use_tconv=False: one weight
                    keys.append("__".join((src, "union", dst_type)))
                else:
                    raise KeyError(f"edge type {et} not in model metadata")
        return keys

    def forward(self, x, edge_index, query_emb=None):
        """``BaseGNNCore.forward(x, edge_index, query_emb=None)`` with the reference's arguments and return value
        (gnn_model.py:230-277): pre_mp, then per layer conv -> (SAGE: updates[i](cat(x_neigh, x))) -> relu -> dropout
        -> running cat, op by op on this library's kernels (the stand-alone ``SAGEConv.forward`` /
        ``GossipConv.forward`` + ``ops.gemm``).  Inference only (no autograd); ``BaseGNN.forward`` -- what the models
        call -- runs the same layers as fused kernels on the canonical-partition batches.

        SAGE after ``to_hetero``: ``x`` = {node type: [n_t, input_dim]}, ``edge_index`` = {(src, rel, dst): [2, E]} with
        indices local to their node types; returns {node type: [n_t, 64 (L + 1)]}; the relations into one destination
        type are summed pairwise in metadata order (pyg.nn.to_hetero(aggr="sum")).  GOSSIP: ``x`` [N, input_dim],
        ``edge_index`` [2, E], ``query_emb`` [1, 64] of ONE query; returns [N, 64 (L + 2)]."""
        from collections import deque
        with torch.no_grad():
            if self.conv_type == "GOSSIP":
                dev = x.device
                lin = self.pre_mp[0]
                h = ops.linear_smallk(x.float().contiguous(), ops.transposed(lin.weight), lin.bias)       # :231
                if self.input_pattern_emb:                                                                # :233-240
                    if query_emb is None:
                        raise AssertionError("query_emb is required (input_pattern_emb)")
                    h = torch.cat((query_emb.reshape(1, -1).float().to(dev).expand(h.shape[0], -1), h), dim=-1)
                ei = edge_index[:, edge_index[0] != edge_index[1]] if edge_index.numel() else edge_index  # :246
                ei = torch.unique(torch.cat([ei, ei.flip(0)], 1), dim=1) if ei.numel() else ei            # :247
                ew = ei[0] < ei[1]                                                                        # :248
                emb = h
                for l, conv in enumerate(self.convs):
                    h = conv(h.contiguous(), ei, edge_weight=ew, query_emb=query_emb)                     # :257-260
                    h = self._relu_dropout(h, l)                                                          # :273-274
                    emb = torch.cat((emb, h), 1)                                                          # :275
                return emb
            if self.node_types is None:
                raise NotImplementedError("homogeneous SAGE (ablation) is out of the hot path; call to_hetero first")
            xs = {t: ops.linear_smallk(x[t].float().contiguous(), ops.transposed(self.pre_mp[0][t].weight),
                                       self.pre_mp[0][t].bias) for t in self.node_types}
            emb = dict(xs)
            for l in range(self.layer_num):
                outs = {t: deque() for t in self.node_types}
                for (s, r, d) in self.edge_types:
                    ei = edge_index.get((s, r, d))
                    outs[d].append(self.convs[l]["__".join((s, r, d))]((xs[s], xs[d]), ei))               # :262
                new = {}
                for t in self.node_types:
                    q = outs[t]
                    while len(q) >= 2:                      # to_hetero's pairwise sum over a common destination
                        q.append(q.popleft() + q.popleft())
                    up = self.updates[l][t]
                    hcat = ops.gemm(q[0], ops.transposed(up.weight), up.bias, a2=xs[t].contiguous())      # :264
                    new[t] = self._relu_dropout(hcat, l, t)
                xs = new
                emb = {t: torch.cat((emb[t], xs[t]), dim=1) for t in self.node_types}
            return emb

    def _relu_dropout(self, h, layer, node_type=None):
        """relu (gnn_model.py:273) and, in training mode with dropout > 0, F.dropout (:274) as this library's
        counter-based factor (ops.dropout_mask) -- one key per call, the site id separates layers / node types."""
        h = torch.relu(h)
        p = float(self.dropout or 0.0)
        if self.training and p > 0.0 and h.numel():
            site = 16 * layer + (0 if node_type is None else 1 + self.node_types.index(node_type))
            h = h * ops.dropout_mask(ops.DropSite(ops.rng_next(h.device), site % 256, p), h.shape[0], h.shape[1])
        return h


class BaseGNN(nn.Module):
    """core + anchor MLP + pool + post MLP (gnn_model.py:18-112)."""

    def __init__(self, input_dim, hidden_dim, output_dim, args, **kwargs):
        super().__init__()
        self.dropout = args.dropout
        self.layer_num = args.layer_num
        self.conv_type = args.conv_type
        self.use_hetero = args.use_hetero
        self.args, self.kwargs = args, kwargs
        self.output_dim = output_dim
        self.gnn_core = BaseGNNCore(input_dim, hidden_dim, output_dim, args, **kwargs)
        p = self.gnn_core.post_input_dim
        self.anchor_mlp = nn.Sequential(nn.Linear(p, p), nn.LeakyReLU(0.1))              # :40-42
        self.post_mp = nn.Sequential(                                                    # :44-53
            nn.Linear(p, hidden_dim), nn.Dropout(args.dropout), nn.LeakyReLU(0.1),
            nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, 256), nn.ReLU(),
            nn.Linear(256, output_dim))
        self._pack_cache = None

    # -- weight folding ---------------------------------------------------------------------------
    def _param_version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self):
        """Kernel operands folded from the parameters; cached until a parameter changes."""
        ver = self._param_version()
        if self._pack_cache is None or self._pack_cache[0] != ver:
            with torch.no_grad():
                pk = pack_gossip(self) if self.conv_type == "GOSSIP" else pack_shmp(self)
            self._pack_cache = (ver, pk)
        return self._pack_cache[1]

    def forward(self, data, query_emb=None, drop_key=None):
        if self.conv_type == "GOSSIP":
            if not isinstance(data, GossipBatch):
                raise TypeError("gossip BaseGNN.forward expects a desco_amd.batch.GossipBatch")
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                return gossip_forward_train(self, data, query_emb)
            return gossip_forward(self, data, query_emb)
        if not isinstance(data, (NeighborhoodBatch, QueryBatch)):
            raise TypeError("BaseGNN.forward expects a NeighborhoodBatch or QueryBatch")
        if self.gnn_core.node_types is None:
            raise NotImplementedError(
                "homogeneous SAGE (ablation, hetero_graph=False) is out of the hot path; call "
                "to_hetero_old()/to_hetero() first (main.py:221-224)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return shmp_forward_train(self, data, drop_key)
        return shmp_forward(self, data)


# -------------------------------------------------------------------------------------------------
# SHMP (neighborhood / query) path
# -------------------------------------------------------------------------------------------------
def _mv(A: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """A [..., m, k] times v [..., k] -> [..., m] as a broadcast product and a row sum.  torch routes a matrix-vector
    product to rocBLAS gemv, whose split reduction uses float atomics: the folded weights -- and with them a seeded
    training run -- were not reproducible from run to run.  This form is a fixed-order reduction."""
    return (A * v.unsqueeze(-2)).sum(-1)


def _vm(v: torch.Tensor, A: torch.Tensor) -> torch.Tensor:
    """v [k] times A [k, n] -> [n], same reason as ``_mv``."""
    return (v.unsqueeze(-1) * A).sum(0)


def _lin_t(lin: nn.Linear):
    return lin.weight.t().contiguous(), lin.bias.contiguous()


def pack_shmp(gnn: BaseGNN, bf16_planes: bool = True) -> dict:
    """bf16_planes: also emit the pre-split weight planes of the bf16x6 GEMMs (inference only)."""
    core = gnn.gnn_core
    pk = {"pre": {}, "layers": []}
    for t in core.node_types:
        pk["pre"][t] = _lin_t(core.pre_mp[0][t])
    # Folding (U_n W_s)^T for every (layer, destination type, slot) and U_n sum_s b_s + c for every
    # (layer, type): two batched matmuls instead of ~100 tiny ones (the training step re-folds the
    # weights every step; per-product launches made it host-bound).  Differentiable.
    items, un_items, bias_items, un_rows = [], [], [], []
    for l in range(core.layer_num):
        for t in core.node_types:
            Un = core.updates[l][t].weight[:, :H]
            bsum, seen = 0, set()
            for key in core.slot_keys(t):
                conv = core.convs[l][key]
                items.append(conv.lin.weight)
                un_items.append(Un)
                if key not in seen:      # one bias per edge TYPE (use_tconv=False ties two slots)
                    bsum = bsum + conv.lin.bias
                    seen.add(key)
            bias_items.append(bsum)
            un_rows.append(Un)
    folded = torch.bmm(torch.stack(un_items), torch.stack(items)).transpose(1, 2)    # (U_n W_s)^T
    fbias = _mv(torch.stack(un_rows), torch.stack(bias_items))
    it = ib = 0
    for l in range(core.layer_num):
        per_type = {}
        for t in core.node_types:
            U, c = core.updates[l][t].weight, core.updates[l][t].bias
            Ux = U[:, H:]
            nslots = len(core.slot_keys(t))
            blocks = [folded[it + k] for k in range(nslots)]
            it += nslots
            blocks.append(Ux.t())
            fb = fbias[ib] + c
            ib += 1
            entry = {"wt": torch.cat(blocks, 0).contiguous(), "b": fb.contiguous()}
            if len(blocks) == 5:
                # count destinations: the two canonical->count relations have at most one source
                # per row; apply them from a pre-transformed table (K 320 -> 192)
                entry["wt_mfma"] = torch.cat([blocks[0], blocks[1], blocks[4]], 0).contiguous()
                entry["wt_tab"] = torch.cat([blocks[2], blocks[3]], 1).contiguous()      # [64,128]
                if bf16_planes and GEMM_BF16X6:
                    entry["wt_tab_l64"] = ops.linear64_planes(entry["wt_tab"].t())     # [2,3,64,64]
            if bf16_planes and SHMP_BF16X6:
                # n-major operand planes of the MFMA blocks of the fused layer (K <= 192): fp16 (hi, lo) + one scale
                # per matrix, or bf16 (hi, mid, lo)
                f16 = SHMP_F16X3 and ops.pool_tile_rows() == 16
                for name in ("wt_mfma", "wt"):
                    if name in entry and entry[name].shape[0] <= 192:
                        entry[name + "_x6"] = (ops.split_f16_planes if f16 else ops.split_bf16_planes)(entry[name].t())
            per_type[t] = entry
        pk["layers"].append(per_type)
    pk["anchor"] = _lin_t(gnn.anchor_mlp[0])
    pk["post"] = [_lin_t(gnn.post_mp[i]) for i in (0, 3, 5, 7)]
    if not (bf16_planes and GEMM_BF16X6):
        return pk
    # n-major ([out, in]) pre-split operands of the bf16x6 GEMM
    _split = ops.split_f16_planes if GEMM_F16X3 else ops.split_bf16_planes
    pk["anchor_nk"] = (_split(gnn.anchor_mlp[0].weight), gnn.anchor_mlp[0].bias.contiguous())
    # (64-input layers run on the streaming row-wise kernel: planes per 64-column output block)
    pk["post_nk"] = [((ops.linear64_planes(gnn.post_mp[i].weight) if gnn.post_mp[i].in_features == 64
                       else ops.split_bf16_planes(gnn.post_mp[i].weight)),
                      gnn.post_mp[i].bias.contiguous()) for i in (0, 3, 5, 7)]
    # post_mp.3 -> .5 -> .7 in one launch (ops.post_mp_tail): the three matrices as fp16 (hi, lo) planes
    dims = [tuple(gnn.post_mp[i].weight.shape) for i in (3, 5, 7)]
    if POST_TAIL_FUSED and GEMM_F16X3 and dims == [(64, 64), (256, 64), (64, 256)]:
        pk["post_tail"] = [v for i in (3, 5, 7)
                           for v in (ops.split_f16_planes(gnn.post_mp[i].weight), gnn.post_mp[i].bias.contiguous())]
    return pk


# post_mp.3 -> .5 -> .7 in one launch (desco_post_mp_tail_f16x3_f32): the [B, 64] and [B, 256] intermediates never reach HBM
POST_TAIL_FUSED = os.environ.get("DESCO_POST_TAIL_FUSED", "1") != "0"


class _PostMp0(object):
    """post_mp.0's output computed by _shmp_pooled itself (the pooled operand never materialised)"""
    __slots__ = ("h0",)

    def __init__(self, h0):
        self.h0 = h0


def _post_mp(pk, pooled):
    if GEMM_BF16X6 and "post_nk" in pk:
        def lin(a, w, b, act=ops.ACT_NONE, slope=0.0):
            f = ops.linear64 if w.dim() == 4 else ops.gemm_split
            return f(a, w, b, act=act, slope=slope)
        (w0, b0), (w3, b3), (w5, b5), (w7, b7) = pk["post_nk"]
        h = pooled.h0 if isinstance(pooled, _PostMp0) else lin(pooled, w0, b0, ops.ACT_LEAKY, 0.1)
        if POST_TAIL_FUSED and "post_tail" in pk:
            return ops.post_mp_tail(h, *pk["post_tail"])
        h = lin(h, w3, b3, ops.ACT_RELU)
        h = lin(h, w5, b5, ops.ACT_RELU)
        return lin(h, w7, b7)
    (w0, b0), (w3, b3), (w5, b5), (w7, b7) = pk["post"]
    h = ops.gemm(pooled, w0, b0, act=ops.ACT_LEAKY, slope=0.1)
    h = ops.gemm(h, w3, b3, act=ops.ACT_RELU)
    h = ops.gemm(h, w5, b5, act=ops.ACT_RELU)
    return ops.gemm(h, w7, b7)


def _gemm_planes(a, w, b, **kw):
    """A GEMM on pre-split weight planes: f16x3 (ops.F16Planes) or bf16x6 (int16 [3, n, k])."""
    return (ops.gemm_f16x3 if isinstance(w, ops.F16Planes) else ops.gemm_split)(a, w, b, **kw)


def _first_layer_coef(pk, t, su, S, x0, src_of_slot, dev):
    """[S+1, 64] coefficients of the closed-form first layer for destination type ``t`` (folded once per
    weight version): rows s < su = x0_src(s) W_s, unused slots zero, last row = x0_t W_self + bias."""
    ck = ("layer0_coef", t, S)
    if ck not in pk:
        e = pk["layers"][0][t]
        wt = e["wt"]                                    # [(su+1)*64, 64]
        rows = [_vm(x0[src_of_slot(t, s)], wt[s * H:(s + 1) * H]) for s in range(su)]
        rows += [torch.zeros(H, device=dev)] * (S - su)  # unused slots of this type
        rows.append(_vm(x0[t], wt[su * H:(su + 1) * H]) + e["b"])
        pk[ck] = torch.stack(rows).contiguous()
    return pk[ck]


def _anchor_const_input(pk, gnn, canon, row_bound=None):
    """anchor_mlp (gnn_model.py:69-73) on emb["canonical"] when the input layer is constant (all-zero node
    features: x^0 of every canonical row is pre_mp's bias): the first 64-column block of the operand is the
    same row for every neighborhood, so its product is folded into the bias and the GEMM runs with
    K = 512 instead of 576 (one ninth of the largest dense product of the pass)."""
    if "anchor_nk_const" not in pk:
        w, b = gnn.anchor_mlp[0].weight, gnn.anchor_mlp[0].bias
        x0 = pk["pre"]["canonical"][1]
        _split = ops.split_f16_planes if GEMM_F16X3 else ops.split_bf16_planes
        pk["anchor_nk_const"] = (_split(w[:, H:].contiguous()), (b + _mv(w[:, :H], x0)).contiguous())
    kw = {"row_scale": row_bound} if (row_bound is not None and isinstance(pk["anchor_nk_const"][0], ops.F16Planes)) else {}
    return _gemm_planes(canon[:, H:], *pk["anchor_nk_const"], act=ops.ACT_LEAKY, slope=0.1, **kw)


# X_1's count rows are never written: the closed-form first layer's output is a function of a row's four slot degrees, so
# the second layer gathers the few thousand DISTINCT rows from a table and recomputes its own rows from their degrees
# (NeighborhoodBatch.degree_table_index, desco_shmp_layer_pool_table_f16x3_f32); bit-identical to the launch on the
# materialised tensor
FIRST_LAYER_TABLE = os.environ.get("DESCO_FIRST_LAYER_TABLE", "1") != "0"

# the pooled embeddings [B, 64 (L + 1)] are never written: post_mp.0 forms its operand's chunks from the anchor rows and
# the fused pooling's partial sums in its load phase (desco_pool_post_bf16x6_f32; neighborhoods of at most 33 count rows)
POOL_POST_FUSED = os.environ.get("DESCO_POOL_POST_FUSED", "1") != "0"


def shmp_forward(gnn: BaseGNN, batch) -> torch.Tensor:
    """BaseGNN.forward, hetero path (gnn_model.py:58-109) -> graph embeddings [B, 64]."""
    return _post_mp(gnn.packed(), _shmp_pooled(gnn, batch, fuse_post0=True))                 # :108


def _shmp_pooled(gnn: BaseGNN, batch, fuse_post0: bool = False):
    """The pooled embeddings [B, 64 (L+1)] of BaseGNN.forward before post_mp (gnn_model.py:58-107)."""
    pk = gnn.packed()
    core = gnn.gnn_core
    dev = batch.vrowptr.device
    N, S = batch.num_rows, batch.slots
    if isinstance(batch, NeighborhoodBatch):
        Nc = batch.num_count
        groups = [("count", 0, Nc, 4), ("canonical", Nc, N, 2)]
    else:
        Nc = N
        groups = [("union_node", 0, N, 2)]
    feat = batch.node_feature
    const_input = feat is None and FUSED_SHMP_LAYER and core.layer_num >= 1
    tab1 = table1 = None
    if const_input:
        # ZeroNodeFeat (workload.py:431-440): pre_mp(x) is its bias, identical for every node of a
        # type, so X_0 is never materialised and layer 0 is a degree-affine map (desco_hip.h).
        x0 = {t: pk["pre"][t][1] for t, *_ in groups}
        src_of_slot = (lambda t, s: ("count" if s < 2 else "canonical")) if len(groups) == 2 else \
            (lambda t, s: t)
        xn = torch.empty((N, H), device=dev)
        # (canonical rows only in the anchor operand: the f16x3 fused path with its direct column-block writes)
        canon_once = (CANON_ROWS_ONCE and isinstance(batch, NeighborhoodBatch) and GEMM_BF16X6 and GEMM_F16X3
                      and SHMP_BF16X6 and SHMP_F16X3 and N > Nc > 0)
        # the count rows' launch leaves their pooled partial sums like the fused layers' launches do (round 6: saves the
        # one read of X_1 that its segment sum cost)
        pool1 = None
        if (POOL_FIRST_LAYER and FUSED_POOLING and SHMP_BF16X6 and GEMM_BF16X6 and isinstance(batch, NeighborhoodBatch)
                and Nc > 0 and ops.pool_tile_rows() == 16):
            pbits1, pslot1, nslots1 = batch.pool_index()
            pool1 = (pbits1, pslot1, torch.empty((nslots1, H), device=dev))
        # ... and with the table form of the second layer's launches they are not stored either
        if (FIRST_LAYER_TABLE and pool1 is not None and canon_once and core.layer_num >= 2 and SHMP_F16X3
                and "wt_tab" in pk["layers"][1]["count"]
                and isinstance(pk["layers"][1]["count"].get("wt_mfma_x6"), ops.F16Planes)):
            tab1 = batch.degree_table_index()
        if tab1 is not None:
            coef_c = _first_layer_coef(pk, "count", 4, S, x0, src_of_slot, dev)
            table1 = torch.empty((tab1[0].numel() // S, H), device=dev)
            ops.degree_affine(tab1[0], 0, table1.shape[0], S, coef_c, ops.ACT_RELU, 0.0, table1)
            xn = None
        for t, r0, r1, su in groups:
            if r1 <= r0:
                continue
            coef = _first_layer_coef(pk, t, su, S, x0, src_of_slot, dev)
            if pool1 is not None and t == "count" and r0 == 0:
                ops.degree_affine_pool(batch.vrowptr, r1, S, coef, ops.ACT_RELU, 0.0, xn, pool1)
            elif t == "canonical" and canon_once:
                pass                                   # (written below, straight into the anchor operand's block 1)
            else:
                ops.degree_affine(batch.vrowptr, r0, r1 - r0, S, coef, ops.ACT_RELU, 0.0, xn)
        X = [None, xn]
        first = 1
    else:
        if feat is None:
            feat = torch.zeros((N, core.input_dim), device=dev)
        x = torch.empty((N, H), device=dev)
        for t, r0, r1, _ in groups:
            wt, b = pk["pre"][t]
            ops.linear_smallk(feat[r0:r1], wt, b, out=x[r0:r1])               # :231
        X = [x]
        first = 0
        canon_once = False
    B = batch.num_graphs
    P = H * (core.layer_num + 1)
    # emb["canonical"] [B, P] (operand of the anchor MLP): the fused canonical launches write their
    # column block directly (out2), so no concatenation pass is needed
    direct_canon = FUSED_SHMP_LAYER and isinstance(batch, NeighborhoodBatch)
    canon = torch.empty((B, P), device=dev) if direct_canon else None
    # per-row bound of the anchor operand, left by the launches that write its column blocks (saves the f16x3 GEMM's
    # pre-pass over the operand): only when every block comes from such a launch (constant input, fp16 layer form)
    canon_max = None
    if direct_canon and const_input and GEMM_BF16X6 and GEMM_F16X3 and SHMP_BF16X6 and SHMP_F16X3 and first == 1:
        canon_max = torch.empty((B,), device=dev)
    if direct_canon and const_input and first == 1:
        # the closed-form first layer once more for the canonical rows, straight into its column block of the anchor
        # operand (a 1/9-size launch of our own instead of a strided torch copy per pass); it WRITES the row bound the
        # canonical launches below accumulate into, so it runs before them
        t, r0, r1, su = groups[1]
        ops.degree_affine(batch.vrowptr, r0, r1 - r0, S, _first_layer_coef(pk, t, su, S, x0, src_of_slot, dev),
                          ops.ACT_RELU, 0.0, canon[:, H:2 * H], out_row0=0, row_absmax=canon_max)
    # fused pooling: the count launches leave partial neighborhood sums, reduced after the anchor MLP
    fpool = (FUSED_POOLING and FUSED_SHMP_LAYER and SHMP_BF16X6 and GEMM_BF16X6
             and isinstance(batch, NeighborhoodBatch) and Nc > 0)
    pool_parts = {}
    if fpool:
        pbits, pslot, nslots = batch.pool_index()
        if const_input and first == 1 and pool1 is not None:
            pool_parts[1] = pool1[2]
    for l in range(first, core.layer_num):
        last = l == core.layer_num - 1
        # the last layer's count rows feed nothing but the pooling: with fused pooling they are
        # never stored (the canonical rows still are, they sit at the end of the same tensor)
        xn = torch.empty((N, H), device=dev)
        # layer input, column ids and (count rows) self index of this layer's launches: X_l itself, or -- for the second
        # layer when X_1's count rows exist as a table of distinct rows only -- that table
        x_src, vcol_l, coef_l = X[-1], batch.vcol, None
        if const_input and first == 1 and l == 1 and tab1 is not None:
            x_src, vcol_l, coef_l = table1, tab1[2], coef_c
        if FUSED_SHMP_LAYER:
            for t, r0, r1, su in groups:                                   # :262-264, :273, :389-395
                if r1 <= r0:
                    continue
                e = pk["layers"][l][t]
                if "wt_tab" in e:
                    crows = canon[:, l * H:(l + 1) * H] if canon_once else X[-1][Nc:]     # canonical rows of X_l
                    ytab = (ops.linear64(crows, e["wt_tab_l64"]) if GEMM_BF16X6 else
                            ops.gemm(crows, e["wt_tab"]))                 # canonical rows x [W2|W3]
                    pool = None
                    if fpool and "wt_mfma_x6" in e:
                        pool_parts[l + 1] = torch.empty((nslots, H), device=dev)
                        pool = (pbits, pslot, pool_parts[l + 1])
                    ops.shmp_layer(x_src, batch.vrowptr, vcol_l, r0, r1 - r0, S, 2,
                                   e.get("wt_mfma_x6", e["wt_mfma"]) if SHMP_BF16X6 else e["wt_mfma"],
                                   e["b"], None if (pool is not None and last) else xn, ytab=ytab,
                                   ytab_row0=Nc, pool=pool, self_coef=coef_l)
                else:
                    once = canon_once and t == "canonical" and isinstance(e.get("wt_x6"), ops.F16Planes)
                    ops.shmp_layer(x_src, batch.vrowptr, vcol_l, r0, r1 - r0, S, su,
                                   e.get("wt_x6", e["wt"]) if SHMP_BF16X6 else e["wt"], e["b"], None if once else xn,
                                   out2=(canon[:, (l + 1) * H:(l + 2) * H]
                                         if direct_canon and t == "canonical" else None),
                                   row_absmax=canon_max if (direct_canon and t == "canonical" and
                                                            isinstance(e.get("wt_x6"), ops.F16Planes)) else None,
                                   xself=canon[:, l * H:(l + 1) * H] if once else None)
        else:
            agg = ops.csr_gather_sum(X[-1], batch.vrowptr, batch.vcol, N, S)   # [N, S*64]
            for t, r0, r1, su in groups:
                if r1 > r0:
                    e = pk["layers"][l][t]
                    ops.gemm(agg[r0:r1, :su * H], e["wt"], e["b"], a2=X[-1][r0:r1],
                             act=ops.ACT_RELU, out=xn[r0:r1])
        X.append(xn)
        if fpool and direct_canon and l >= 1 and l in pool_parts:
            # layer l's rows have been consumed (their pooled sums sit in pool_parts[l], their canonical
            # rows in `canon`): release them -- a block then holds three [N, 64] tensors instead of nine,
            # which is what lets InferencePipeline run blocks of tens of millions of rows
            X[l] = _RELEASED
    pooled = torch.empty((B, P), device=dev)
    if isinstance(batch, NeighborhoodBatch):
        c0 = x0["canonical"].expand(B, H) if const_input else X[0][Nc:]
        folded_x0 = GEMM_BF16X6 and const_input and direct_canon and first == 1    # (_anchor_const_input: K = 512)
        if direct_canon:
            if not folded_x0:
                canon[:, :H] = c0
            for l in range(1, first + 1):        # layers produced outside the fused launches
                if not (const_input and l == 1):     # (that one was written above, before the layer loop)
                    canon[:, l * H:(l + 1) * H] = X[l][Nc:]
        else:
            canon = torch.cat([c0] + [xl[Nc:] for xl in X[1:]], dim=1)        # emb["canonical"] [B,P]
        aw, ab = pk["anchor"]
        if folded_x0:
            anch = _anchor_const_input(pk, gnn, canon, row_bound=canon_max)
        elif GEMM_BF16X6:
            anch = _gemm_planes(canon, *pk["anchor_nk"], act=ops.ACT_LEAKY, slope=0.1)
        else:
            anch = ops.gemm(canon, aw, ab, act=ops.ACT_LEAKY, slope=0.1)   # :69-73
        seg_ptr = batch.count_ptr
    else:
        anch = None                                  # query graphs: no canonical node, no anchor
        seg_ptr = batch.graph_ptr
    if (fuse_post0 and POOL_POST_FUSED and anch is not None and const_input and first == 1 and GEMM_BF16X6
            and "post_nk" in pk and sorted(pool_parts) == list(range(1, core.layer_num + 1)) and core.layer_num <= 8
            and ops.pool_tile_rows() == 16 and batch.max_count_rows() <= 33):
        w0, b0 = pk["post_nk"][0]
        return _PostMp0(ops.pool_post(anch, [pool_parts[l] for l in range(1, core.layer_num + 1)], pbits, pslot, seg_ptr,
                                      x0[groups[0][0]], w0, b0, ops.ACT_LEAKY, 0.1))
    if pool_parts and POOL_REDUCE_MULTI:
        # the layers' partial sums, reduced together: one launch for (up to eight of) them instead of one per layer
        ls = sorted(pool_parts)
        ops.pool_reduce_multi([pool_parts[l] for l in ls], pbits, pslot, seg_ptr, B,
                              [None if anch is None else anch[:, l * H:(l + 1) * H] for l in ls],
                              [pooled[:, l * H:(l + 1) * H] for l in ls])
    for l, xl in enumerate(X):                                             # :88-89, :107
        extra = None if anch is None else anch[:, l * H:(l + 1) * H]
        out_l = pooled[:, l * H:(l + 1) * H]
        if l in pool_parts:
            if not POOL_REDUCE_MULTI:
                ops.pool_reduce(pool_parts[l], pbits, pslot, seg_ptr, B, extra=extra, out=out_l)
        elif xl is None:     # constant X_0: the segment sum is (rows in segment) * x0
            t0 = groups[0][0]
            ck = ("pool0_coef", t0)
            if ck not in pk:
                pk[ck] = torch.stack([x0[t0], torch.zeros(H, device=dev)]).contiguous()
            ops.degree_affine(seg_ptr, 0, B, 1, pk[ck], ops.ACT_NONE, 0.0, out_l, extra=extra)
        else:
            ops.segment_sum(xl[:Nc], seg_ptr, B, extra=extra, out=out_l)
    return pooled


def pack_shmp_stacked(gnn: BaseGNN) -> dict:
    """The folding of pack_shmp -- (U_n W_s)^T per slot, U_x^T, U_n sum_s b_s + c -- for the fused
    training trunk, differentiable and STACKED over the layers: per node type (Wt [L, (S_t+1) 64, 64],
    bias [L, 64]) from a dozen batched torch ops.  (pack_shmp's per-(layer, type) entries cost ~100 small
    differentiable select / cat / add ops per step, and their backward as many zero-fills, copies and full-size
    accumulations: half of the replayed step's GPU time in round 2.)"""
    core = gnn.gnn_core
    L = core.layer_num
    out = {}
    for t in core.node_types:
        keys = core.slot_keys(t)
        uniq = list(dict.fromkeys(keys))                 # one bias per edge TYPE (use_tconv=False ties two slots)
        U = torch.stack([core.updates[l][t].weight for l in range(L)])                   # [L, 64, 128]
        c = torch.stack([core.updates[l][t].bias for l in range(L)])                     # [L, 64]
        Un, Ux = U[:, :, :H], U[:, :, H:]
        W = torch.stack([core.convs[l][k].lin.weight for l in range(L) for k in keys]).view(L, len(keys), H, H)
        bs = torch.stack([core.convs[l][k].lin.bias for l in range(L) for k in uniq]).view(L, len(uniq), H).sum(1)
        folded = torch.matmul(Un.unsqueeze(1), W).transpose(-1, -2)                       # (U_n W_s)^T
        Wt = torch.cat([folded, Ux.transpose(-1, -2).unsqueeze(1)], dim=1).reshape(L, (len(keys) + 1) * H, H)
        fb = _mv(Un, bs) + c
        out[t] = (Wt, fb)
    return out


def fold_shmp_native(gnn: BaseGNN, t: str):
    """(Wt [L, (S_t+1) 64, 64], fb [L, 64]) of row type ``t`` -- pack_shmp_stacked's folding -- by desco_fold_shmp_fwd
    from the raw parameters, differentiable through autograd.FoldShmp (gradients in one flat buffer, no torch op)."""
    from . import autograd as AG
    specs = gnn.__dict__.setdefault("_fold_specs", {})
    sp = specs.get(t)
    if sp is None or not sp.valid():
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("fold_shmp_native: the parameter address table has to be (re)built, which uploads it -- "
                               "not inside a hipGraph capture; run one eager step after replacing parameters")
        sp = specs[t] = AG.FoldSpec(gnn.gnn_core, t)
    return AG.FoldShmp.apply(sp, *sp.params)


POST_DROP_SITE = 200      # dropout site of post_mp.1 (the layers use 2 l + row type)


def shmp_forward_train(gnn: BaseGNN, batch, drop_key: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable twin of ``shmp_forward`` (same math, un-fused kernels, autograd Functions from
    desco_amd.autograd; every forward and backward op is a C-ABI kernel launch).  ``drop_key``: the (seed, step) key of
    this pass's dropout (ops.rng_next), drawn here when None -- callers that run two models on two streams draw both
    keys first, on one stream (NeighborhoodCountingModel.train_forward)."""
    from . import autograd as AG
    core = gnn.gnn_core
    dev = batch.vrowptr.device
    N, S = batch.num_rows, batch.slots
    import torch.nn.functional as F
    # --neigh_dropout > 0 (default 0.0, config.py:251): F.dropout after every layer's relu (gnn_model.py:274) and the
    # nn.Dropout of post_mp.1 (:46), in training mode: counter-based factors inside the fused nodes' epilogues
    # (autograd.ShmpTrunk / ShmpTrunkSmall / Mlp), as in the gossip model
    p_layer = float(core.dropout or 0.0) if gnn.training else 0.0
    p_post = float(gnn.post_mp[1].p or 0.0) if gnn.training else 0.0
    drop = p_layer > 0.0 or p_post > 0.0
    if isinstance(batch, NeighborhoodBatch):
        Nc = batch.num_count
        groups = [("count", 0, Nc, 4), ("canonical", Nc, N, 2)]
        seg_ptr = batch.count_ptr
    else:
        Nc = N
        groups = [("union_node", 0, N, 2)]
        seg_ptr = batch.graph_ptr
    feat = batch.node_feature
    if feat is None:
        feat = batch.__dict__.get("_zero_feat")          # ZeroNodeFeat: a constant of the batch, made once
        if feat is None or feat.shape[1] != core.input_dim:
            feat = batch.__dict__["_zero_feat"] = torch.zeros((N, core.input_dim), device=dev)
    small = (SMALL_TRUNK_KERNEL and not isinstance(batch, NeighborhoodBatch) and S == 2 and core.layer_num >= 1
             and 0 < N <= ops.shmp_trunk_small_max_rows())
    if small and drop and not AG.ShmpTrunkSmall.per_graph(batch):
        small = False                                   # (only the per-graph kernels carry the dropout factors)
    if FUSED_TRAIN_TRUNK and all(len(core.slot_keys(t)) == su for t, _, _, su in groups):
        if drop and drop_key is None:
            drop_key = ops.rng_next(dev)
        ldrop = (drop_key, p_layer) if p_layer > 0.0 else None
        # The whole layer loop + anchor + pooling as one autograd node (autograd.ShmpTrunk) on weights folded in
        # stacked form.  Everything between the parameters and that node is this library's kernels too (round 5): the
        # folding reads the parameters through an address table (autograd.FoldShmp), the K-major copies of pre_mp /
        # anchor_mlp / post_mp are one copy2d launch (autograd.TransposedMany), pre_mp writes one buffer (PreLinear).
        has_anchor = isinstance(batch, NeighborhoodBatch)
        lins = [core.pre_mp[0][t] for t, *_ in groups] + ([gnn.anchor_mlp[0]] if has_anchor else []) + \
               [gnn.post_mp[i] for i in (0, 3, 5, 7)]
        wts = AG.TransposedMany.apply(*[m.weight for m in lins])
        ng = len(groups)
        pre = []
        for g in range(ng):
            pre += [wts[g], lins[g].bias]
        x = AG.PreLinear.apply(feat, groups, *pre)
        flat = [wts[ng], gnn.anchor_mlp[0].bias] if has_anchor else []
        for t, *_ in groups:
            flat += list(fold_shmp_native(gnn, t))
        if small and len(groups) == 1:
            # the query graphs (135 rows): the whole trunk in one launch per direction (one workgroup per graph)
            pooled = AG.ShmpTrunkSmall.apply(x, batch, ldrop, *flat)
        else:
            pooled = AG.ShmpTrunk.apply(x, batch, groups, has_anchor, ldrop, *flat)
        pw = wts[ng + (1 if has_anchor else 0):]
        post = {"post": [(pw[j], gnn.post_mp[i].bias) for j, i in enumerate((0, 3, 5, 7))]}
        pdrop = ops.DropSite(drop_key, POST_DROP_SITE, p_post) if p_post > 0.0 else None
        return _post_mp_train(AG, post, gnn, pooled, pdrop)
    pk = pack_shmp(gnn, bf16_planes=False)   # differentiable folding: grads reach the raw parameters
    ti = batch.train_index()
    x = torch.cat([AG.SmallKLinear.apply(feat[r0:r1], *pk["pre"][t]) for t, r0, r1, _ in groups], 0)
    X = [x]
    for l in range(core.layer_num):
        agg = AG.GatherSum.apply(X[-1], batch.vrowptr, batch.vcol, ti["t_rowptr"], ti["t_col"], N, S)
        parts = []
        for t, r0, r1, su in groups:
            e = pk["layers"][l][t]
            parts.append(AG.Linear.apply(agg[r0:r1, :su * H], X[-1][r0:r1], e["wt"], e["b"],
                                         ops.ACT_RELU, 0.0))
        xl = torch.cat(parts, 0)
        if drop:                                                           # gnn_model.py:274
            xl = F.dropout(xl, p=core.dropout, training=True)
        X.append(xl)
    if isinstance(batch, NeighborhoodBatch):
        canon = torch.cat([xl[Nc:] for xl in X], dim=1)
        aw, ab = pk["anchor"]
        anch = AG.Linear.apply(canon, None, aw, ab, ops.ACT_LEAKY, 0.1)
        pooled = torch.cat([AG.SegmentSum.apply(xl[:Nc], seg_ptr, ti["seg_id"], ti["ident_ptr"],
                                                anch[:, l * H:(l + 1) * H].contiguous())
                            for l, xl in enumerate(X)], dim=1)
    else:
        pooled = torch.cat([AG.SegmentSum.apply(xl, seg_ptr, ti["seg_id"], ti["ident_ptr"], None)
                            for xl in X], dim=1)
    return _post_mp_train(AG, pk, gnn, pooled, drop)


def _post_mp_train(AG, pk, gnn, pooled, drop):
    """post_mp of a training pass.  ``drop``: None / False (no dropout), an ops.DropSite (post_mp.1 as the counter-based
    factor inside the fused node), or True (the per-op cross-check path: torch's F.dropout)."""
    import torch.nn.functional as F
    (w0, b0), (w3, b3), (w5, b5), (w7, b7) = pk["post"]
    if not drop or isinstance(drop, ops.DropSite):
        # the four Linears and their backward as one autograd node (activation derivatives in the GEMM epilogues, one
        # launch pair for all weight gradients)
        return AG.Mlp.apply(pooled, ((ops.ACT_LEAKY, 0.1), (ops.ACT_RELU, 0.0), (ops.ACT_RELU, 0.0), (ops.ACT_NONE, 0.0)),
                            tuple(gnn.post_mp[i].weight for i in (0, 3, 5, 7)), drop if drop else None,
                            w0, b0, w3, b3, w5, b5, w7, b7)
    if drop:                                                               # post_mp.1 (gnn_model.py:46)
        h = AG.Linear.apply(pooled, None, w0, b0, ops.ACT_NONE, 0.0)
        h = F.leaky_relu(F.dropout(h, p=gnn.post_mp[1].p, training=True), 0.1)
    else:
        h = AG.Linear.apply(pooled, None, w0, b0, ops.ACT_LEAKY, 0.1)
    h = AG.Linear.apply(h, None, w3, b3, ops.ACT_RELU, 0.0)
    h = AG.Linear.apply(h, None, w5, b5, ops.ACT_RELU, 0.0)
    return AG.Linear.apply(h, None, w7, b7, ops.ACT_NONE, 0.0)


# -------------------------------------------------------------------------------------------------
# gossip path
# -------------------------------------------------------------------------------------------------
# test switch: run the dropout form of the training kernels at p = 0 too (must not change a bit: factor 1 everywhere)
DROPOUT_AT_ZERO = False


def pack_gossip(gnn: BaseGNN, bf16_planes: bool = True) -> dict:
    core = gnn.gnn_core
    if core.layer_num != 2 or not core.input_pattern_emb or core.input_dim != 1:
        raise NotImplementedError(
            "gossip kernels implement the reference configuration: 2 GossipConv layers, "
            "input_dim 1, query embedding as input (config.py:312-322, main.py:316-325)")
    pre = core.pre_mp[0]
    pk = {"w_pre": pre.weight[:, 0].contiguous(), "b_pre": pre.bias.contiguous()}
    c0, c1 = core.convs[0], core.convs[1]
    pk["C0"], pk["c0"] = c0.lin_com.weight, c0.lin_com.bias
    pk["D0"], pk["d0"] = c0.lin_update.weight, c0.lin_update.bias
    D1 = c1.lin_update.weight
    D1a, D1b = D1[:, :H], D1[:, H:]
    pk["wt1"] = torch.cat([(D1a @ c1.lin_com.weight).t(), D1b.t()], 0).contiguous()   # [128,64]
    pk["ws1"] = torch.stack([_mv(D1a, c1.lin_com.bias), torch.zeros_like(c1.lin_com.bias)]).contiguous()
    pk["d1"] = c1.lin_update.bias.contiguous()
    P0, p0 = gnn.post_mp[0].weight, gnn.post_mp[0].bias
    pk["P0"], pk["p0"] = P0, p0
    pk["wtp"] = torch.cat([P0[:, 2 * H:3 * H].t(), P0[:, 3 * H:4 * H].t()], 0).contiguous()
    pk["wsp"] = torch.stack([torch.zeros(H, device=P0.device), _mv(P0[:, H:2 * H], pk["w_pre"])]).contiguous()
    pk["post"] = [_lin_t(gnn.post_mp[i]) for i in (3, 5)]
    pk["w7"] = gnn.post_mp[7].weight[0].contiguous()
    pk["b7"] = float(gnn.post_mp[7].bias[0])
    pk["fused_w1"] = pk["wt1"].t().contiguous()                 # [64,128]
    pk["fused_wp"] = pk["wtp"].t().contiguous()                 # [64,128]
    pk["fused_w3"] = gnn.post_mp[3].weight.contiguous()         # [64,64]  (already [out, in])
    pk["fused_w5"] = gnn.post_mp[5].weight.contiguous()         # [256,64]
    if bf16_planes:     # bf16 planes (hi, mid, lo) of the n-major matrices: the fused kernel's operands
        for k in ("fused_w1", "fused_wp", "fused_w3", "fused_w5"):
            pk[k + "s"] = ops.split_bf16_planes(pk[k])
        # ... and the fp16 (hi, lo) weight stream of the three-product kernel
        pk["wstream"], pk["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(pk[k]) for k in
                                                            ("fused_w1", "fused_wp", "fused_w3", "fused_w5")])
    pk["qcache"] = None
    return pk


def _gossip_query_terms(gnn: BaseGNN, pk: dict, query_emb: torch.Tensor) -> dict:
    """Everything that depends on (weights, query embeddings) only: gates and folded vectors."""
    key = (query_emb.data_ptr(), query_emb._version, tuple(query_emb.shape))
    if pk["qcache"] is not None and pk["qcache"][0] == key:
        return pk["qcache"][1]
    core = gnn.gnn_core
    E = query_emb.float()
    C0, D0 = pk["C0"], pk["D0"]
    w_pre, b_pre = pk["w_pre"], pk["b_pre"]
    q = {}
    q["g0"] = core.convs[0]._gate_value(E).reshape(-1).contiguous()          # gnn_model.py:340
    q["g1"] = core.convs[1]._gate_value(E).reshape(-1).contiguous()
    a_q = E @ C0[:, :H].t() + (_mv(C0[:, H:], b_pre) + pk["c0"])             # lin_com(h0) const part
    v = _mv(C0[:, H:], w_pre)
    D0a, D0b, D0c = D0[:, :H], D0[:, H:2 * H], D0[:, 2 * H:]
    q["p"] = (a_q @ D0a.t()).contiguous()
    q["r"] = _mv(D0a, v).contiguous()
    q["t"] = _mv(D0c, w_pre).contiguous()
    q["z"] = (E @ D0b.t() + (_mv(D0c, b_pre) + pk["d0"])).contiguous()
    P0 = pk["P0"]
    q["zp"] = (E @ P0[:, :H].t() + (_mv(P0[:, H:2 * H], b_pre) + pk["p0"])).contiguous()     # [Q,64]
    pk["qcache"] = (key, q)
    return q


def gossip_forward(gnn: BaseGNN, batch: GossipBatch, query_emb: torch.Tensor) -> torch.Tensor:
    """All-queries gossip correction + residual: returns pred [N, Q] = x + post_mp(emb)
    (BaseGNN.forward gossip path gnn_model.py:58-103 looped over queries as in
    lightning_model.py:613-628, here batched over the query axis)."""
    pk = gnn.packed()
    with torch.no_grad():
        q = _gossip_query_terms(gnn, pk, query_emb)
    x = batch.x
    N, Q = x.shape
    if Q != query_emb.shape[0]:
        raise ValueError("batch.x has a different number of query columns than query_emb rows")
    if FUSED_GOSSIP:
        (w3, b3), (w5, b5) = pk["post"]
        outs = []
        # the scalars pre-pass maps one lane to one query: more than 64 queries (the labelled queries
        # of --use_node_feature) go in column groups
        for q0 in range(0, Q, 64):
            q1 = min(q0 + 64, Q)
            sl = (lambda t: t) if (q0 == 0 and q1 == Q) else (lambda t: t[q0:q1].contiguous())
            xs = x if (q0 == 0 and q1 == Q) else x[:, q0:q1].contiguous()
            scal4 = ops.gossip_scalars(xs, batch.rowptr, batch.col, sl(q["g0"]), sl(q["g1"]))
            v = {"g1": sl(q["g1"]), "p": sl(q["p"]), "z": sl(q["z"]), "zp": sl(q["zp"]), "r": q["r"],
                 "t": q["t"], "u": pk["ws1"][0], "tp": pk["wsp"][1], "d1": pk["d1"],
                 # the fused kernel takes n-major ([out, in]) weight blocks
                 "w1s": pk["fused_w1s"], "wps": pk["fused_wps"], "w3s": pk["fused_w3s"], "b3": b3,
                 "w5s": pk["fused_w5s"], "b5": b5, "w7": pk["w7"], "b7": pk["b7"]}
            tperm = batch.tile_perm if GOSSIP_TILE_ORDER else None
            if GOSSIP_F16X3:
                v["wstream"], v["winv"] = pk["wstream"], pk["winv"]
                outs.append(ops.gossip_fused_f16(scal4, batch.rowptr, batch.col, N, q1 - q0, v, batch.work_queue,
                                                 tile_perm=tperm,
                                                 out=getattr(batch, "out_buf", None) if (q0 == 0 and q1 == Q) else None))
            else:
                outs.append(ops.gossip_fused(scal4, batch.rowptr, batch.col, N, q1 - q0, v, tile_perm=tperm))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    h1, scal = ops.gossip_layer0(x, batch.rowptr, batch.col, q["g0"], q["g1"], q["p"], q["r"],
                                 q["t"], q["z"])                                  # layer 0
    hh = ops.gossip_gather(h1, batch.rowptr, batch.col, N, Q, q["g1"])           # layer 1 aggregate
    h2 = ops.gemm(hh, pk["wt1"], pk["d1"], a2=h1, act=ops.ACT_RELU, s=scal, ws=pk["ws1"])
    y = ops.gemm(h1, pk["wtp"], q["zp"], a2=h2, act=ops.ACT_LEAKY, slope=0.1, s=scal,
                 ws=pk["wsp"])                                                    # post_mp.0
    del hh
    (w3, b3), (w5, b5) = pk["post"]
    y = ops.gemm(y, w3, b3, act=ops.ACT_RELU)
    y = ops.gemm(y, w5, b5, act=ops.ACT_RELU)
    out = ops.rowdot_add(y, pk["w7"], pk["b7"], add=x.reshape(-1))               # post_mp.7 + x
    return out.view(N, Q)


def gossip_forward_train(gnn: BaseGNN, batch: GossipBatch, query_emb: torch.Tensor) -> torch.Tensor:
    """Differentiable twin of ``gossip_forward`` (returns pred [N, Q] = x + correction).

    Same algebra as the fused kernel as autograd Functions whose forward and
    backward are C-ABI launches.  As in the reference, the layer-0 input is detached
    (gnn_model.py:236-240): ``pre_mp`` and the query embeddings receive no gradient."""
    from . import autograd as AG
    core = gnn.gnn_core
    if core.layer_num != 2 or not core.input_pattern_emb or core.input_dim != 1:
        raise NotImplementedError("gossip training implements the reference configuration only")
    x = batch.x
    N, Q = x.shape
    dev = x.device
    E = query_emb.detach().float().to(dev)
    w_pre, b_pre = core.pre_mp[0].weight[:, 0].detach(), core.pre_mp[0].bias.detach()
    c0, c1 = core.convs[0], core.convs[1]
    C0, cb0, D0, db0 = c0.lin_com.weight, c0.lin_com.bias, c0.lin_update.weight, c0.lin_update.bias
    C1, cb1, D1, db1 = c1.lin_com.weight, c1.lin_com.bias, c1.lin_update.weight, c1.lin_update.bias
    # ---- constants per (node, query): deg_lo, deg_hi, s_lo, s_hi, x (functions of the batch alone: cached on it) ----
    # (keyed on the tensor object -- kept alive by the cache, so its address cannot be reused -- and its version;
    #  the library's in-place writers of x bump the version: ops.scatter_rows)
    cc = batch.__dict__.get("_train_consts")
    if cc is None or cc[0][0] is not x or cc[0][1] != x._version:
        with torch.no_grad():
            ones, zeros = torch.ones(Q, device=dev), torch.zeros(Q, device=dev)
            sa = ops.gossip_scalars(x, batch.rowptr, batch.col, ones, zeros)    # (deg_lo, s_lo, deg_hi, x)
            sb = ops.gossip_scalars(x, batch.rowptr, batch.col, zeros, zeros)   # (deg_hi, s_hi, deg_hi, x)
            deg_lo, s_lo, deg_hi, xr = sa[:, 0], sa[:, 1], sa[:, 2], sa[:, 3]
            s_hi = sb[:, 1]
            one = torch.ones_like(xr)
            C6 = torch.stack([deg_hi, deg_lo - deg_hi, s_hi, s_lo - s_hi, xr, one], 1).contiguous()
            C3 = torch.stack([deg_hi, deg_lo - deg_hi, one], 1).contiguous()
            C2 = torch.stack([xr, one], 1).contiguous()
        cc = batch.__dict__["_train_consts"] = ((x, x._version), C6, C3, C2)
    _, C6, C3, C2 = cc
    # --gossip_dropout (default 0.01, config.py:316): F.dropout behind each layer's relu (gnn_model.py:274) and
    # post_mp.1 = nn.Dropout (:46), in training mode only -- counter-based factors inside GossipTrunk's epilogues
    p_layer = float(core.dropout or 0.0) if gnn.training else 0.0
    p_post = float(gnn.post_mp[1].p or 0.0) if gnn.training else 0.0
    drop = (p_layer, p_post) if (p_layer > 0.0 or p_post > 0.0 or (DROPOUT_AT_ZERO and gnn.training)) else None
    if Q <= 64:
        # The operands folded from the parameters by one kernel each way (autograd.FoldGossip, csrc/train_native.hip;
        # algebra DESIGN.md 4.2), then the whole per-(node, query) pipeline and its backward as one autograd node
        # (autograd.GossipTrunk): the step launches nothing but this library's kernels.
        gl = [c.lin_gate for c in (c0, c1)]
        V0, g1, g1c, wt1, V1, wtp, Vp, w3t, w5t = AG.FoldGossip.apply(
            E.contiguous(), w_pre.contiguous(), b_pre.contiguous(), C0, cb0, D0, db0, C1, cb1, D1, db1,
            gl[0][0].weight, gl[0][0].bias, gl[0][2].weight, gl[0][2].bias,
            gl[1][0].weight, gl[1][0].bias, gl[1][2].weight, gl[1][2].bias,
            gnn.post_mp[0].weight, gnn.post_mp[0].bias, gnn.post_mp[3].weight, gnn.post_mp[5].weight)
        pred = AG.GossipTrunk.apply(batch.rowptr, batch.col, N, Q, C6, C3, C2, x.reshape(-1), g1c,
                                    gnn.post_mp[3].weight.detach(), gnn.post_mp[5].weight.detach(), drop,
                                    V0, g1, wt1, V1, wtp, Vp, w3t, gnn.post_mp[3].bias, w5t, gnn.post_mp[5].bias,
                                    gnn.post_mp[7].weight.view(-1), gnn.post_mp[7].bias)
        return pred.view(N, Q)
    # ---- operands folded from the parameters with differentiable torch ops (more than 64 queries) ---------------------
    g0 = c0._gate_value(E).reshape(-1)
    g1 = c1._gate_value(E).reshape(-1)
    a_q = E @ C0[:, :H].t() + (_mv(C0[:, H:], b_pre) + cb0)
    v = _mv(C0[:, H:], w_pre)
    D0a, D0b, D0c = D0[:, :H], D0[:, H:2 * H], D0[:, 2 * H:]
    p = a_q @ D0a.t()
    r = _mv(D0a, v).expand(Q, H)
    t = _mv(D0c, w_pre).expand(Q, H)
    z = E @ D0b.t() + (_mv(D0c, b_pre) + db0)
    V0 = torch.stack([p, g0[:, None] * p, r, g0[:, None] * r, t, z], 1)          # [Q,6,64]
    D1a, D1b = D1[:, :H], D1[:, H:]
    wt1 = torch.cat([(D1a @ C1).t(), D1b.t()], 0)
    u = _mv(D1a, cb1).expand(Q, H)
    V1 = torch.stack([u, g1[:, None] * u, db1.expand(Q, H)], 1)                     # [Q,3,64]
    P0, p0 = gnn.post_mp[0].weight, gnn.post_mp[0].bias
    wtp = torch.cat([P0[:, 2 * H:3 * H].t(), P0[:, 3 * H:4 * H].t()], 0)
    tp = _mv(P0[:, H:2 * H], w_pre).expand(Q, H)
    zp = E @ P0[:, :H].t() + (_mv(P0[:, H:2 * H], b_pre) + p0)
    Vp = torch.stack([tp, zp], 1)                                                   # [Q,2,64]
    pred = AG.GossipTrunk.apply(batch.rowptr, batch.col, N, Q, C6, C3, C2, x.reshape(-1), (1.0 - g1).detach().contiguous(),
                                gnn.post_mp[3].weight.detach(), gnn.post_mp[5].weight.detach(), drop,
                                V0, g1, wt1, V1, wtp, Vp, gnn.post_mp[3].weight.t(), gnn.post_mp[3].bias,
                                gnn.post_mp[5].weight.t(), gnn.post_mp[5].bias, gnn.post_mp[7].weight.view(-1),
                                gnn.post_mp[7].bias)
    return pred.view(N, Q)
