"""NeighborhoodCountingModel / GossipCountingModel with the reference's API surface
(subgraph_counting/lightning_model.py:37-649) and no Lightning dependency.

``graph_to_count`` / ``train_forward`` run entirely through the HIP kernels (desco_amd.gnn_model);
the 29-iteration Python loops of the reference (lightning_model.py:210-219, 615-625) are replaced
by one batched head kernel / a query axis in the gossip kernels.
"""
from __future__ import annotations

import argparse
import os
import warnings
from typing import Any, Dict, List, Optional, Sequence, Tuple

import networkx as nx
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .optim import Adam
from .batch import GossipBatch, NeighborhoodBatch, QueryBatch
from .data import graph_atlas_plus
from .gnn_model import (BaseGNN, H, QUERY_EDGE_TYPES_TCONV, QUERY_EDGE_TYPES_UNION,
                        QUERY_NODE_TYPES, TARGET_EDGE_TYPES_TCONV, TARGET_EDGE_TYPES_UNION,
                        TARGET_NODE_TYPES)


def gen_queries(query_ids: List[int], queries=None, transform=None, node_feat_len: int = 1,
                hetero=True, device="cpu"):
    """Query graphs by atlas id (lightning_model.py:37-87).  Returns ([(n, edges)], [nx.Graph]);
    the flat pairs replace the list of PyG graphs (``transform`` is applied by QueryBatch).
    ``node_feat_len != 1`` (--use_node_feature): every atlas query is expanded into its
    ``node_feat_len ** n`` labelled copies with one-hot node features (utils.py:258-272), in the
    reference's ``itertools.product`` order; the features are on the returned graphs' ``"feat"``."""
    from .data import add_node_feat_to_networkx
    if queries is None:
        queries_nx = [graph_atlas_plus(q) for q in query_ids]
        if node_feat_len != 1:
            eye = [t for t in np.eye(node_feat_len).tolist()]
            queries_nx = [g for q in queries_nx for g in add_node_feat_to_networkx(q, eye, "feat")]
    else:
        queries_nx = list(queries)
    flat = []
    for g in queries_nx:
        nodes = list(g.nodes)
        idx = {v: i for i, v in enumerate(nodes)}
        flat.append((len(nodes), sorted((min(idx[a], idx[b]), max(idx[a], idx[b]))
                                        for a, b in g.edges())))
    return flat, queries_nx


def query_node_features(queries_nx, input_dim: int) -> Optional[torch.Tensor]:
    """[sum n, input_dim] "feat" rows of the query graphs in node order, or None when no query
    carries features (NetworkxToHetero then fills zeros, transforms.py:380-384)."""
    if not any("feat" in g.nodes[v] for g in queries_nx for v in g.nodes):
        return None
    rows = []
    for g in queries_nx:
        for v in g.nodes:
            f = g.nodes[v].get("feat")
            rows.append(torch.zeros(input_dim) if f is None else
                        torch.as_tensor(f, dtype=torch.float32).reshape(-1))
    return torch.stack(rows)


class _LightningLike(nn.Module):
    """The slice of pl.LightningModule main.py relies on: device, log, hparams, ckpt round trip."""

    def __init__(self):
        super().__init__()
        self.logged: Dict[str, Any] = {}
        self.hparams_dict: Dict[str, Any] = {}

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, name, value, **kwargs):
        # (kept as the device tensor: a float() here would be a host sync -- and a D2H copy -- in every training step;
        #  ``logged_value(name)`` reads it)
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else float(value)

    def logged_value(self, name) -> float:
        return float(self.logged[name])

    def save_hyperparameters(self, **hp):
        self.hparams_dict = hp

    def checkpoint(self) -> Dict[str, Any]:
        """A Lightning-shaped checkpoint dict (keys read by load_from_checkpoint)."""
        return {"state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
                "hyper_parameters": dict(self.hparams_dict)}

    def save_checkpoint(self, path: str):
        torch.save(self.checkpoint(), path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
        # Lightning-free reader with a restricted unpickler: also opens the authors' Lightning 1.6.4
        # files (pytorch_lightning.* classes in them resolve to inert stand-ins, desco_amd/ckpt.py)
        from .ckpt import load_checkpoint
        ckpt = load_checkpoint(checkpoint_path, map_location=map_location or "cpu")
        hp = dict(ckpt["hyper_parameters"])
        hp.update(kwargs)
        args = hp.pop("args")
        model = cls(hp.pop("input_dim"), hp.pop("hidden_dim"), args, **hp)
        model.on_load_checkpoint(ckpt)
        model.load_state_dict(ckpt["state_dict"])
        return model

    def on_load_checkpoint(self, checkpoint):
        return None

    def invalidate_caches(self):
        """Drop every cache derived from the parameters (folded / split weights, head operands,
        query embeddings).  They are keyed on (data_ptr, tensor._version); an update that does not
        bump ``_version`` -- a hipGraph replay of the optimizer step (Trainer(graph_capture=True)),
        a raw ``hipMemcpy`` into a parameter -- needs this call."""
        for m in self.modules():
            if isinstance(m, BaseGNN):
                m._pack_cache = None
        self._head_cache = None
        self._qemb_cache = None


# training: run the query model's trunk on a second HIP stream beside the target batch's (train_forward)
OVERLAP_QUERY_TRUNK = True
# the count head formed straight from the target embeddings at inference (desco_count_head_emb_f16x3_f32): the [B, 256]
# target half of count_model.0 never reaches HBM
HEAD_FROM_EMB = os.environ.get("DESCO_HEAD_FROM_EMB", "1") != "0"


class NeighborhoodCountingModel(_LightningLike):
    def __init__(self, input_dim, hidden_dim, args, **kwargs):
        super().__init__()
        self.emb_with_query = False
        self.query_loader = None          # a QueryBatch (the reference keeps a DataLoader here)
        self.hidden_dim, self.input_dim = hidden_dim, input_dim
        self.kwargs, self.args = kwargs, args
        for k, v in vars(args).items():   # lightning_model.py:113-114
            setattr(self, k, v)
        self.save_hyperparameters(input_dim=input_dim, hidden_dim=hidden_dim, args=args, **kwargs)
        self.emb_model = BaseGNN(input_dim, hidden_dim, hidden_dim, args,
                                 emb_channels=hidden_dim, **kwargs)
        self.emb_model_query = BaseGNN(input_dim, hidden_dim, hidden_dim, args,
                                       emb_channels=hidden_dim, **kwargs)
        self.count_model = nn.Sequential(nn.Linear(2 * hidden_dim, 4 * args.hidden_dim),
                                         nn.LeakyReLU(), nn.Linear(4 * args.hidden_dim, 1))
        self._qemb_cache = None

    # ---- hetero conversion (lightning_model.py:325-421) -----------------------------------------
    def to_hetero_old(self, tconv_target=False, tconv_query=False):
        self.emb_model.gnn_core.to_hetero(
            TARGET_NODE_TYPES, TARGET_EDGE_TYPES_TCONV if tconv_target else TARGET_EDGE_TYPES_UNION)
        self.emb_model_query.gnn_core.to_hetero(
            QUERY_NODE_TYPES, QUERY_EDGE_TYPES_TCONV if tconv_query else QUERY_EDGE_TYPES_UNION)
        self.tconv_target, self.tconv_query = tconv_target, tconv_query
        return self

    def to_hetero(self, order: int = 3, SHMP_target=False, SHMP_query=False):
        if order != 3:
            raise NotImplementedError("order-4 (union_1..11) SHMP is outside the hot path")
        return self.to_hetero_old(tconv_target=SHMP_target, tconv_query=SHMP_query)

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:   # :508-532
        a = checkpoint["hyper_parameters"]["args"]
        use_canonical = getattr(a, "use_canonical", True)
        if a.use_hetero and use_canonical:
            self.to_hetero_old(tconv_target=a.use_tconv, tconv_query=a.use_tconv)
        elif a.use_hetero:
            raise NotImplementedError("to_hetero_wo_canonical (ablation) is outside the hot path")

    # ---- queries ----------------------------------------------------------------------------------
    def set_queries(self, query_ids, queries=None, transform=None, hetero=True, device=None):
        flat, queries_nx = gen_queries(query_ids, queries, transform=transform,
                                       node_feat_len=self.input_dim, hetero=hetero)
        min_len_neighbor = max(nx.diameter(q) for q in queries_nx)
        if self.depth < min_len_neighbor:                                  # :302-308
            warnings.warn("neighborhood diameter {:d} is too small for the queries, the minimum is "
                          "{:d}".format(self.depth, min_len_neighbor))
        self.queries_flat = flat
        self.query_feat = query_node_features(queries_nx, self.input_dim)
        self.query_loader = QueryBatch(flat, device or self.device, self.input_dim, self.query_feat)
        self._qemb_cache = None

    def _queries(self) -> QueryBatch:
        if self.query_loader is None:
            raise RuntimeError("call set_queries() first (main.py:231-233)")
        if self.query_loader.device != self.device and self.device.type == "cuda":
            self.query_loader = QueryBatch(self.queries_flat, self.device, self.input_dim,
                                           getattr(self, "query_feat", None))
        return self.query_loader

    def get_query_emb(self) -> torch.Tensor:                                # :311-316
        """[Q, 64] query embeddings.  The reference recomputes them on every batch (:204-207);
        they only depend on the weights, so they are cached per weight version."""
        qb = self._queries()
        ver = self.emb_model_query._param_version()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.emb_model_query.parameters()):
            return self.emb_model_query(qb)
        if self._qemb_cache is None or self._qemb_cache[0] != (ver, id(qb)):
            with torch.no_grad():
                self._qemb_cache = ((ver, id(qb)), self.emb_model_query(qb))
        return self._qemb_cache[1]

    # ---- forward ---------------------------------------------------------------------------------
    def _head_pack(self):
        """count_model operands in kernel form, cached per weight version (no host sync per call)."""
        ver = tuple((p.data_ptr(), p._version) for p in self.count_model.parameters())
        if getattr(self, "_head_cache", None) is None or self._head_cache[0] != ver:
            with torch.no_grad():
                W1, b1 = self.count_model[0].weight, self.count_model[0].bias       # [256,128]
                self._head_cache = (ver, {
                    "wt_t": W1[:, :H].t().contiguous(), "wt_q": W1[:, H:].t().contiguous(),
                    "w_t_l64": ops.linear64_planes(W1[:, :H]), "w_q_nk": ops.split_bf16_planes(W1[:, H:]),
                    "b1": b1.contiguous(), "w2": self.count_model[2].weight[0].contiguous(),
                    "b2": float(self.count_model[2].bias[0])})
        return self._head_cache[1]

    def _logits(self, batch: NeighborhoodBatch, exp2: bool) -> torch.Tensor:
        emb_q = self.get_query_emb()
        emb_t = self.emb_model(batch)
        hp = self._head_pack()
        from . import gnn_model as GM
        fused_head = (HEAD_FROM_EMB and GM.GEMM_BF16X6 and GM.GEMM_F16X3 and not torch.is_grad_enabled()
                      and emb_q.shape[0] == 29 and hp["w2"].numel() == 256 and emb_t.shape[1] == 64)
        if GM.GEMM_BF16X6:
            T = None if fused_head else ops.linear64(emb_t, hp["w_t_l64"])  # target half
            # query half + bias: a function of (head weights, query embeddings) only -- cached with them at inference
            # (keyed on the tensor OBJECT, which the cache keeps alive: an address could be handed to the next query
            #  set's embeddings by the caching allocator)
            if torch.is_grad_enabled() or hp.get("qh_src") is not emb_q or hp.get("qh_ver") != emb_q._version:
                Qh = ops.gemm_split(emb_q, hp["w_q_nk"], hp["b1"])
                if not torch.is_grad_enabled():
                    hp["qh_src"], hp["qh_ver"], hp["qh"] = emb_q, emb_q._version, Qh
            else:
                Qh = hp["qh"]
        else:
            T = ops.gemm(emb_t, hp["wt_t"])
            Qh = ops.gemm(emb_q, hp["wt_q"], hp["b1"])
        slope = self.count_model[1].negative_slope
        if fused_head:
            # the [B, 256] target half is never written: formed block by block inside the head's launch
            if "w_t_f16" not in hp:
                hp["w_t_f16"] = ops.split_f16_planes(self.count_model[0].weight[:, :emb_t.shape[1]].contiguous())
            return ops.count_head_emb(emb_t, hp["w_t_f16"], Qh, hp["w2"], hp["b2"], slope, exp2,
                                      out=getattr(batch, "out_buf", None) if exp2 else None)
        if Qh.shape[0] <= 32:
            # (InferencePipeline hands every block a slice of ONE persistent result tensor: no torch.cat per pass)
            return ops.count_head(T, Qh, hp["w2"], hp["b2"], slope, exp2,
                                  out=getattr(batch, "out_buf", None) if exp2 else None)
        # more than 32 queries (labelled queries of --use_node_feature: 784 for input_dim 2): the head
        # kernel keeps one accumulator per query in registers, so the query axis goes in groups of 32
        return torch.cat([ops.count_head(T, Qh[q0:q0 + 32], hp["w2"], hp["b2"], slope, exp2)
                          for q0 in range(0, Qh.shape[0], 32)], dim=1)

    def graph_to_count(self, batch) -> torch.Tensor:                         # :198-222
        with torch.no_grad():
            return self._logits(batch.to(self.device), exp2=True)

    def predict_step(self, batch, batch_idx) -> torch.Tensor:                # :195-196
        return self.graph_to_count(batch)

    def graph_to_embed(self, batch) -> torch.Tensor:                         # :224-226
        with torch.no_grad():
            return self.emb_model(batch.to(self.device))

    def criterion(self, count, truth):                                       # :285-289
        return F.smooth_l1_loss(count, truth)

    def train_forward(self, batch, batch_idx=0) -> torch.Tensor:            # :228-254
        """mean_q smooth_l1(logit[:, q], log2(y[:, q] + 1)); forward and backward on the HIP
        kernels (desco_amd.autograd)."""
        from . import autograd as AG
        batch = batch.to(self.device)
        if batch.y is None:
            raise ValueError("train_forward needs batch.y (apply_truth_from_dataset first)")
        from . import distributed as D
        # (not inside a hook-driven data-parallel step: there the gradient buckets are all-reduced from autograd hooks,
        #  which would run on the side stream for the query model's parameters while other gradients of the same bucket
        #  are still being written on the main stream; the replayed data-parallel step reduces after the backward has
        #  joined its streams and keeps the fork)
        # --neigh_dropout > 0: both models' dropout keys are drawn HERE, on one stream, before the query model's pass
        # forks onto its own (two streams advancing one (seed, step) counter would race for their keys)
        kq = kt = None
        if self.training and self.device.type == "cuda" and float(self.emb_model.gnn_core.dropout or 0.0) > 0.0:
            kq, kt = ops.rng_next(self.device), ops.rng_next(self.device)
        if OVERLAP_QUERY_TRUNK and self.device.type == "cuda" and not D.hooks_active():
            # the query model's trunk is ~100 launches on 135 rows: forward (and, through autograd, backward) on a
            # second stream, beside the target batch's launches instead of in front of them (also inside a hipGraph
            # capture: the side stream forks from and joins the capturing stream)
            cur = torch.cuda.current_stream(self.device)
            side = self.__dict__.get("_query_stream")
            if side is None or side.device != self.device:
                side = self.__dict__["_query_stream"] = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                emb_q = self.emb_model_query(self._queries(), drop_key=kq)
            emb_t = self.emb_model(batch, drop_key=kt)
            cur.wait_stream(side)
            emb_q.record_stream(cur)
        else:
            emb_q = self.emb_model_query(self._queries(), drop_key=kq)
            emb_t = self.emb_model(batch, drop_key=kt)
        W1, b1 = self.count_model[0].weight, self.count_model[0].bias
        wt_t, wt_q = AG.SplitT.apply(W1, H)                # K-major halves of count_model.0; one gradient for W1
        T = AG.Linear.apply(emb_t, None, wt_t, None, ops.ACT_NONE, 0.0)
        Qh = AG.Linear.apply(emb_q, None, wt_q, b1, ops.ACT_NONE, 0.0)
        # (views, not selects: a select's backward is a zero fill + copy of torch's)
        w2, b2 = self.count_model[2].weight.view(-1), self.count_model[2].bias.view(())
        slope = self.count_model[1].negative_slope
        if Qh.shape[0] <= 32:
            logits = AG.CountHead.apply(T, Qh, w2, b2, slope)
        else:       # query groups of 32 (see _logits)
            logits = torch.cat([AG.CountHead.apply(T, Qh[q0:q0 + 32].contiguous(), w2, b2, slope)
                                for q0 in range(0, Qh.shape[0], 32)], dim=1)
        # mean over queries of per-query means == mean over all [B, Q] entries; smooth_l1(logit - log2(y + 1)) and its
        # gradient in one kernel pair (criterion below is the same formula in torch, for callers that want it)
        y = batch.y if batch.y.dtype == torch.float32 else batch.y.float()
        return AG.Loss.apply(logits, y, 0)

    def training_step(self, batch, batch_idx):                               # :133-136
        loss = self.train_forward(batch, batch_idx)
        self.log("neighborhood_counting_train_loss", loss, batch_size=batch.num_graphs)
        return loss

    def validation_step(self, batch, batch_idx):                             # :147-154
        with torch.no_grad():
            loss = self.test_forward(batch, batch_idx, train_space=True)
        self.log("neighborhood_counting_val_loss", loss, batch_size=batch.num_graphs, sync_dist=True)
        return loss

    def test_step(self, batch, batch_idx):                                   # :138-145
        with torch.no_grad():
            loss = self.test_forward(batch, batch_idx)
        self.log("neighborhood_counting_test_loss", loss, batch_size=batch.num_graphs, sync_dist=True)
        return loss

    def test_forward(self, batch, batch_idx=0, train_space: bool = False) -> torch.Tensor:   # :256-283
        batch = batch.to(self.device)
        logits = self._logits(batch, exp2=False)
        y = batch.y.to(logits.dtype)
        if train_space:
            return self.criterion(logits, torch.log2(y + 1))
        return self.criterion(F.relu(2 ** (logits - 1)), y)

    def configure_optimizers(self):                                          # :160-173
        optimizer = Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)   # (desco_adam_step_f32)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.5,
                                                           patience=20, min_lr=1e-5)
        return {"optimizer": optimizer, "lr_scheduler": sched,
                "monitor": "neighborhood_counting_val_loss"}


class GossipCountingModel(_LightningLike):
    def __init__(self, input_dim, hidden_dim, args, **kwargs):
        super().__init__()
        self.hidden_dim = hidden_dim
        kwargs["baseline"] = "gossip"                                       # :540
        for k, v in vars(args).items():
            setattr(self, k, v)
        self.save_hyperparameters(input_dim=input_dim, hidden_dim=hidden_dim, args=args, **kwargs)
        self.emb_model = BaseGNN(input_dim, hidden_dim, 1, args, **kwargs)
        self.kwargs = kwargs
        self.query_emb: Optional[torch.Tensor] = None

    def set_query_emb(self, query_emb: torch.Tensor, query_ids=None, queries=None):   # :637-638
        self.query_emb = query_emb.detach()

    def graph_to_count(self, batch: GossipBatch, query_emb=None) -> torch.Tensor:     # :613-628
        qe = self.query_emb if query_emb is None else query_emb
        if qe is None:
            raise RuntimeError("call set_query_emb() first (main.py:334)")
        with torch.no_grad():
            return self.emb_model(batch, query_emb=qe.to(self.device))

    def predict_step(self, batch, batch_idx) -> torch.Tensor:
        return self.graph_to_count(batch)

    def criterion(self, count, truth):                                      # :630-635
        return torch.log2(torch.abs(count - truth) + 1)

    def train_forward(self, batch: GossipBatch, batch_idx=0) -> torch.Tensor:   # :585-608
        """sum over queries and nodes of log2(|neigh_pred + gossip_pred - truth| + 1)."""
        if self.query_emb is None:
            raise RuntimeError("call set_query_emb() first (main.py:334)")
        batch = batch.to(self.device)
        if batch.y is None:
            raise ValueError("train_forward needs batch.y (apply_truth_from_dataset first)")
        pred = self.emb_model(batch, query_emb=self.query_emb.to(self.device))
        # sum log2(|pred - y| + 1) and its gradient in one kernel pair (criterion above is the same formula in torch)
        from . import autograd as AG
        y = batch.y if batch.y.dtype == torch.float32 else batch.y.float()
        return AG.Loss.apply(pred, y, 1)

    def training_step(self, batch, batch_idx):                              # :553-556
        loss = self.train_forward(batch, batch_idx)
        self.log("gossip_counting_train_loss", loss, batch_size=batch.num_graphs)
        return loss

    def validation_step(self, batch, batch_idx):                            # :562-564
        with torch.no_grad():
            loss = self.train_forward(batch, batch_idx)
        self.log("gossip_counting_val_loss", loss, batch_size=batch.num_graphs)
        return loss

    def test_step(self, batch, batch_idx):                                  # :558-560
        with torch.no_grad():
            loss = self.train_forward(batch, batch_idx)
        self.log("gossip_counting_test_loss", loss, batch_size=batch.num_graphs)
        return loss

    def _gate_value(self, query_emb) -> torch.Tensor:                       # :640-649
        assert self.conv_type == "GOSSIP"
        with torch.no_grad():
            return torch.stack([layer._gate_value(query_emb.to(self.device))
                                for layer in self.emb_model.gnn_core.convs], dim=0)

    def configure_optimizers(self):                                         # :570-583
        optimizer = Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)   # (desco_adam_step_f32)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.5,
                                                           patience=20, min_lr=1e-5)
        return {"optimizer": optimizer, "lr_scheduler": sched, "monitor": "gossip_counting_val_loss"}
