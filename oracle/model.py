"""Oracle, float path: pure-torch CPU restatement of the reference's SHMP neighborhood model and
gossip model, IN THE REFERENCE'S FORM (per-edge-type index_select + index_add_ + Linear, 29-iteration
Python loops, query graphs re-embedded on every call).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference has no tests / golden vectors for this
path.  Every function that restates the reference's OWN code is pinned bit-exactly by vectors that
code produced when executed unbound in the build container (tests/golden/float_pieces.npz,
float_flow.npz; the list is in oracle/__init__.py).  PARITY UNPINNED only for the third-party
semantics marked [EXT] below (torch_geometric is not importable here): recalled from PyG 2.2.0, not
executed.

All functions take a plain ``state_dict`` (reference key names) and numpy/torch index
arrays in PyG convention (``edge_index[0]`` = source index inside the source node type,
``edge_index[1]`` = destination index inside the destination node type).
"""
from __future__ import annotations

from collections import deque
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import partition as P


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))


def sage_conv(sd, key, x_src, n_dst, ei, emulate_quirk, bipartite):
    """SAGEConv.forward, gnn_model.py:372-404.

    ``remove_self_loops`` (row != col, no bipartite awareness [EXT]) is applied to every non-empty
    edge type (:389-390): on bipartite types it silently drops edges whose batch-global source and
    destination indices coincide.  ``emulate_quirk=False`` gives the intended model.
    message = x_j (:402-404), aggr = add at edge_index[1] [EXT propagate], then ``lin`` on every
    destination row (:395).
    """
    ei = _t(ei).long()
    if ei.numel() != 0 and (emulate_quirk or not bipartite):
        ei = ei[:, ei[0] != ei[1]]
    agg = torch.zeros(n_dst, x_src.shape[1], dtype=x_src.dtype)
    agg.index_add_(0, ei[1], x_src[ei[0]])
    return _lin(sd, key + ".lin", agg)


def gnn_core_hetero(sd, prefix, feats: Dict[str, torch.Tensor], edge_index: Dict, node_types,
                    edge_types, layer_num, emulate_quirk=True, layer_masks=None):
    """BaseGNNCore.forward (SAGE branch) after pyg.nn.to_hetero(aggr="sum").

    gnn_model.py:230-277; per-type module copies and the pairwise-queue sum over edge types with
    a common destination type are [EXT].  ``layer_masks[l][t]`` [n_t, H]: the factor tensor
    (0 or 1 / (1 - p)) of F.dropout behind layer l's relu (:274) in training mode, given by the caller (see ``post_mp``).
    """
    x = {t: _lin(sd, f"{prefix}.pre_mp.0.{t}", feats[t]) for t in node_types}      # :231
    emb = dict(x)                                                                   # :253
    for l in range(layer_num):
        outs = {t: deque() for t in node_types}
        for (s, r, d) in edge_types:
            key = f"{prefix}.convs.{l}.{s}__{r}__{d}"
            outs[d].append(sage_conv(sd, key, x[s], x[d].shape[0], edge_index[(s, r, d)],
                                     emulate_quirk, bipartite=(s != d)))            # :262
        new_x = {}
        for t in node_types:
            q = outs[t]
            while len(q) >= 2:                       # [EXT] to_hetero pairwise torch.add queue
                a, b = q.popleft(), q.popleft()
                q.append(a + b)
            x_neigh = q[0]
            h = _lin(sd, f"{prefix}.updates.{l}.{t}", torch.cat((x_neigh, x[t]), dim=1))  # :264
            new_x[t] = F.relu(h)                     # :273 ; dropout p=0 (:274) is the identity
            if layer_masks is not None:
                new_x[t] = new_x[t] * layer_masks[l][t]                                 # :274 (training, p > 0)
        x = new_x
        emb = {t: torch.cat((emb[t], x[t]), dim=1) for t in node_types}             # :275
    return emb


def post_mp(sd, prefix, emb, mask=None):
    """BaseGNN.post_mp, gnn_model.py:44-53 (Dropout is the identity at inference / p=0).  ``mask``: the factor tensor
    (0 or 1 / (1 - p) per element) of post_mp.1 = nn.Dropout in training mode, given by the caller -- torch draws it
    from its own generator, the HIP path from a counter (oracle/dropout.py), so parity tests inject the HIP mask."""
    h = _lin(sd, f"{prefix}.post_mp.0", emb)
    if mask is not None:
        h = h * mask                                                                 # :46  (x * mask / (1 - p))
    h = F.leaky_relu(h, 0.1)
    h = F.relu(_lin(sd, f"{prefix}.post_mp.3", h))
    h = F.relu(_lin(sd, f"{prefix}.post_mp.5", h))
    return _lin(sd, f"{prefix}.post_mp.7", h)


def base_gnn_hetero(sd, prefix, batch: Dict, node_types, edge_types, layer_num, input_dim=1,
                    feats=None, emulate_quirk=True, masks=None):
    """BaseGNN.forward, hetero path, gnn_model.py:58-109.  ``batch`` = oracle.partition.collate().
    ``masks`` = (layer_masks, post_mask): the dropout factors of a training pass (gnn_core_hetero, post_mp)."""
    if feats is None:       # ZeroNodeFeat / NetworkxToHetero zeros (transforms.py:380-384)
        feats = {t: torch.zeros(batch["num_nodes"][t], input_dim) for t in node_types}
    emb = gnn_core_hetero(sd, f"{prefix}.gnn_core", feats, batch["edge_index"], node_types,
                          edge_types, layer_num, emulate_quirk,
                          None if masks is None else masks[0])                       # :66
    if "canonical" in emb:                                                          # :69-73
        emb["canonical"] = F.leaky_relu(_lin(sd, f"{prefix}.anchor_mlp.0", emb["canonical"]), 0.1)
    allemb = torch.cat([emb[t] for t in node_types], dim=0)                          # :88-89
    bvec = torch.cat([_t(batch["batch"][t]).long() for t in node_types])
    pooled = torch.zeros(batch["num_graphs"], allemb.shape[1], dtype=allemb.dtype)
    pooled.index_add_(0, bvec, allemb)                                              # :107 global_add_pool
    return post_mp(sd, prefix, pooled, None if masks is None else masks[1])         # :108


def neighborhood_embed_queries(sd, qbatch, layer_num, input_dim=1, qfeats=None, masks=None):
    """emb_model_query on the query batch (lightning_model.py:204-207).  ``qfeats``
    {"union_node": [sum n, input_dim]}: labelled queries (--use_node_feature); None = zeros."""
    return base_gnn_hetero(sd, "emb_model_query", qbatch, ("union_node",), P.QUERY_EDGE_TYPES,
                           layer_num, input_dim, qfeats, masks=masks)


def head_logits(sd, emb_t, emb_q):
    """The query loop of graph_to_count / train_forward (lightning_model.py:210-219) around
    embed_to_count (:176-193): per query ``count_model(cat(emb_target, query_emb.expand_as(emb_target)))``,
    columns concatenated.  PINNED by tests/golden/float_flow.npz (nm_*)."""
    outs = []
    for q in range(emb_q.shape[0]):
        e = torch.cat((emb_t, emb_q[q].expand_as(emb_t)), dim=-1)
        h = F.leaky_relu(_lin(sd, "count_model.0", e))        # nn.LeakyReLU() default slope 0.01
        outs.append(_lin(sd, "count_model.2", h))
    return torch.cat(outs, dim=-1)


def count_from_logits(logits):
    """lightning_model.py:221: ``2**pred - 1``."""
    return 2 ** logits - 1


def train_loss_from_logits(logits, y):
    """train_forward, lightning_model.py:239-253 + criterion :285-289: per query
    smooth_l1(out[:, q], log2(y[:, q] + 1)) (mean over the batch), mean over the queries."""
    losses = [F.smooth_l1_loss(logits[:, q:q + 1], torch.log2(y[:, q].view(-1, 1) + 1))
              for q in range(logits.shape[1])]
    return torch.mean(torch.stack(losses))


def eval_loss_from_logits(logits, y):
    """test_forward, lightning_model.py:267-282: smooth_l1(relu(2**(out-1)), y) per query, mean."""
    losses = [F.smooth_l1_loss(F.relu(2 ** (logits[:, q:q + 1] - 1)), y[:, q].view(-1, 1))
              for q in range(logits.shape[1])]
    return torch.mean(torch.stack(losses))


def neighborhood_logits(sd, batch, qbatch, layer_num=8, input_dim=1, feats=None,
                        emulate_quirk=True, qfeats=None, masks_t=None, masks_q=None):
    """The [B,Q] pre-exponent outputs of graph_to_count / train_forward.

    lightning_model.py:198-219: the queries are re-embedded on every call, then the head loop.
    ``masks_t`` / ``masks_q``: dropout factors of the target / query model's training pass (base_gnn_hetero).
    """
    emb_q = neighborhood_embed_queries(sd, qbatch, layer_num, input_dim, qfeats, masks_q)
    emb_t = base_gnn_hetero(sd, "emb_model", batch, P.NODE_TYPES, P.EDGE_TYPES, layer_num,
                            input_dim, feats, emulate_quirk, masks_t)
    return head_logits(sd, emb_t, emb_q), emb_q


def neighborhood_graph_to_count(sd, batch, qbatch, **kw):
    """graph_to_count, lightning_model.py:198-222: ``2**pred - 1``."""
    logits, _ = neighborhood_logits(sd, batch, qbatch, **kw)
    return count_from_logits(logits)


def neighborhood_loss(sd, batch, qbatch, y, **kw):
    """train_forward, lightning_model.py:228-254 + criterion :285-289."""
    logits, _ = neighborhood_logits(sd, batch, qbatch, **kw)
    return train_loss_from_logits(logits, y)


def neighborhood_test_loss(sd, batch, qbatch, y, **kw):
    """test_forward, lightning_model.py:256-283."""
    logits, _ = neighborhood_logits(sd, batch, qbatch, **kw)
    return eval_loss_from_logits(logits, y)


# ------------------------------------------------------------------------------------------
# gossip
# ------------------------------------------------------------------------------------------
def gossip_gate(sd, key, query_emb):
    """GossipConv.lin_gate, gnn_model.py:294-301: Linear, Sigmoid, Linear, Sigmoid, LeakyReLU()."""
    g = torch.sigmoid(_lin(sd, key + ".lin_gate.0", query_emb))
    g = torch.sigmoid(_lin(sd, key + ".lin_gate.2", g))
    return F.leaky_relu(g)


def gossip_single_query(sd, x_col, edge_index, query_emb, layer_num=2, layer_masks=None, post_mask=None):
    """BaseGNN.forward (baseline == "gossip") for ONE query, gnn_model.py:58-109, 230-277, 303-350.

    ``x_col`` [N,1] neighborhood counts of this query, ``query_emb`` [1,H].  Returns [N,1].
    Training mode with dropout (--gossip_dropout, default 0.01, config.py:316): ``layer_masks[l]`` [N,H] is the factor
    tensor of F.dropout behind layer l's relu (:274), ``post_mask`` [N,H] that of post_mp.1 (:46).
    """
    N = x_col.shape[0]
    x = _lin(sd, "emb_model.gnn_core.pre_mp.0", x_col)                               # :231
    x = torch.cat((query_emb.expand(N, -1), x), dim=-1).clone().detach()            # :236-240
    ei, dirw = P.gossip_edge_index(N, np.asarray(edge_index))                       # :246-248
    ei, dirw = torch.as_tensor(ei).long(), torch.as_tensor(dirw)
    emb = x
    for l in range(layer_num):
        key = f"emb_model.gnn_core.convs.{l}"
        gate = gossip_gate(sd, key, query_emb)                                      # :340
        msg = _lin(sd, key + ".lin_com", x[ei[0]])                                  # :341 (x_j)
        msg[dirw] *= gate                                                           # :342
        msg[~dirw] *= 1 - gate                                                      # :343
        aggr = torch.zeros(N, msg.shape[1], dtype=msg.dtype).index_add_(0, ei[1], msg)               # aggr="add"
        x = _lin(sd, key + ".lin_update", torch.cat((aggr, x), dim=-1))             # :347-348
        x = F.relu(x)                                                               # :273
        if layer_masks is not None:
            x = x * layer_masks[l]                                                  # :274 (F.dropout, training)
        emb = torch.cat((emb, x), dim=1)                                            # :275
    return post_mp(sd, "emb_model", emb, post_mask)                                 # :102-103


def gossip_query_loop(emb_fn, x, query_emb):
    """GossipCountingModel.graph_to_count, lightning_model.py:613-628: per query
    ``neigh_pred + emb_model(batch, query_emb[q])`` with ``batch.node_feature = x[:, q]``, columns
    concatenated.  ``emb_fn(x_col [N,1], query_emb [1,H]) -> [N,1]``.  PINNED by float_flow.npz (gm_*)."""
    outs = []
    for q in range(query_emb.shape[0]):
        x_col = x[:, q].view(-1, 1)
        outs.append(x_col + emb_fn(x_col, query_emb[q, :].view(1, -1)))
    return torch.cat(outs, dim=-1)


def gossip_loss_from_pred(pred, y):
    """train_forward + criterion, lightning_model.py:585-608, 630-635: log2(|pred - y| + 1) per node
    and query, stacked per query and summed (sum, not mean)."""
    return torch.sum(torch.stack([torch.log2(torch.abs(pred[:, q:q + 1] - y[:, q].view(-1, 1)) + 1)
                                  for q in range(pred.shape[1])]))


def gossip_graph_to_count(sd, x, edge_index, query_emb, layer_num=2, masks=None):
    """GossipCountingModel.graph_to_count, lightning_model.py:613-628: 29 sequential passes.
    ``masks``: None, or (layer_masks, post_mask) with layer_masks[l] and post_mask of shape [N, Q, H] -- the dropout
    factors of query q's pass are the [:, q, :] slices."""
    if masks is None:
        return gossip_query_loop(
            lambda x_col, qe: gossip_single_query(sd, x_col, edge_index, qe, layer_num), x, query_emb)
    lm, pm = masks
    outs = []
    for q in range(query_emb.shape[0]):                                             # gossip_query_loop with the q-th masks
        x_col = x[:, q].view(-1, 1)
        outs.append(x_col + gossip_single_query(sd, x_col, edge_index, query_emb[q, :].view(1, -1), layer_num,
                                                [m[:, q, :] for m in lm], pm[:, q, :]))
    return torch.cat(outs, dim=-1)


def gossip_loss(sd, x, y, edge_index, query_emb, layer_num=2, masks=None):
    """train_forward + criterion, lightning_model.py:585-608, 630-635 (sum, not mean)."""
    return gossip_loss_from_pred(gossip_graph_to_count(sd, x, edge_index, query_emb, layer_num, masks), y)


def gossip_gate_values(sd, query_emb, layer_num=2):
    """GossipCountingModel._gate_value, lightning_model.py:640-649 -> [L,Q,1]."""
    return torch.stack([gossip_gate(sd, f"emb_model.gnn_core.convs.{l}", query_emb)
                        for l in range(layer_num)], dim=0)


# ------------------------------------------------------------------------------------------
# dataset-level helpers (A11 / A15)
# ------------------------------------------------------------------------------------------
def apply_neighborhood_count(count, indicator):
    """GossipDataset.apply_neighborhood_count, workload.py:107-112."""
    x = torch.zeros(len(indicator), count.shape[1])
    x[torch.as_tensor(np.asarray(indicator, dtype=bool))] = count.detach()
    return x


def apply_truth(truth, indicator):
    """NeighborhoodDataset.apply_truth_from_dataset, workload.py:296-301."""
    return truth[torch.as_tensor(np.asarray(indicator, dtype=bool)), :]


def aggregate_by_index(count, graph_id, num_graphs):
    """NeighborhoodDataset.aggregate_neighborhood_count, workload.py:303-324 (index_add_)."""
    out = torch.zeros(num_graphs, count.shape[1])
    out.index_add_(0, torch.as_tensor(np.asarray(graph_id)).long(), count.float())
    return out


def aggregate_by_ptr(count, ptr):
    """GossipDataset.aggregate_neighborhood_count, workload.py:136-148 (segment_csr sum [EXT])."""
    ptr = np.asarray(ptr)
    return torch.stack([count[ptr[i]:ptr[i + 1]].sum(dim=0) for i in range(len(ptr) - 1)])


def prebuild_reference_inputs(graphs, queries, depth=4, neigh_batch=512, gossip_batch=256):
    """Everything of ``reference_pipeline`` that is DATA PREPARATION in the reference (dataset
    ``process`` + transforms + DataLoader collate, workload.py:243-294, transforms.py:180-255):
    collated neighborhood batches, the query batch, per-gossip-batch edge arrays.  Built once so
    that timing ``run_reference_prebuilt`` measures the model path only ("data already
    in memory as CSR tensors")."""
    index, indicator, neighs = P.neighborhood_dataset(graphs, depth)
    qbatch = P.query_batch(queries)
    batches = [P.neighborhood_batch(neighs[b0:b0 + neigh_batch])
               for b0 in range(0, len(neighs), neigh_batch)]
    for b in batches:       # index arrays as tensors once (the DataLoader hands out tensors)
        b["edge_index"] = {k: _t(v).long() for k, v in b["edge_index"].items()}
    ptr = np.concatenate([[0], np.cumsum([n for n, _ in graphs])])
    gossip = []
    for g0 in range(0, len(graphs), gossip_batch):
        g1 = min(g0 + gossip_batch, len(graphs))
        n0, n1 = ptr[g0], ptr[g1]
        es = [np.asarray(sorted(e), dtype=np.int64).reshape(-1, 2) + (ptr[g] - n0)
              for g, (_, e) in zip(range(g0, g1), graphs[g0:g1])]
        und = np.concatenate(es) if es else np.zeros((0, 2), dtype=np.int64)
        ei = np.concatenate([und, und[:, ::-1]]).T          # to_networkx(to_undirected) both dirs
        gossip.append((int(n0), int(n1), ei))
    return {"index": index, "indicator": indicator, "batches": batches, "qbatch": qbatch,
            "ptr": ptr, "gossip": gossip, "num_graphs": len(graphs), "num_queries": len(queries)}


def run_reference_prebuilt(sd_neigh, sd_gossip, pre, layer_num=8, gossip_layers=2,
                           emulate_quirk=True):
    """The model path of main.py:296-302, 417-423 in the reference's form on prebuilt inputs:
    neighborhood counts per batch (queries re-embedded on every call), scatter to nodes, 29
    sequential gossip passes per batch, per-graph aggregation."""
    Q = pre["num_queries"]
    counts = [neighborhood_graph_to_count(sd_neigh, b, pre["qbatch"], layer_num=layer_num,
                                          emulate_quirk=emulate_quirk) for b in pre["batches"]]
    neigh_count = torch.cat(counts) if counts else torch.zeros(0, Q)
    x = apply_neighborhood_count(neigh_count, pre["indicator"])
    emb_q = neighborhood_embed_queries(sd_neigh, pre["qbatch"], layer_num)
    outs = [gossip_graph_to_count(sd_gossip, x[n0:n1], ei, emb_q, gossip_layers)
            for n0, n1, ei in pre["gossip"]]
    node_count = torch.cat(outs) if outs else torch.zeros(0, Q)
    index = pre["index"]
    return {
        "index": index, "indicator": pre["indicator"], "neigh_count": neigh_count, "x": x,
        "query_emb": emb_q, "node_count": node_count,
        "graph_neigh_count": aggregate_by_index(neigh_count, index[:, 0], pre["num_graphs"]),
        "graph_gossip_count": aggregate_by_ptr(node_count, pre["ptr"]),
    }


def reference_pipeline(sd_neigh, sd_gossip, graphs, queries, depth=4, neigh_batch=512,
                       gossip_batch=256, layer_num=8, gossip_layers=2, emulate_quirk=True):
    """End-to-end inference in the reference's form (main.py:296-302, 417-423): neighborhood
    counts in batches of ``neigh_batch`` neighborhoods, scatter to nodes, gossip in batches of
    ``gossip_batch`` graphs, per-graph aggregation.  Returns dict of tensors.
    """
    pre = prebuild_reference_inputs(graphs, queries, depth, neigh_batch, gossip_batch)
    return run_reference_prebuilt(sd_neigh, sd_gossip, pre, layer_num, gossip_layers, emulate_quirk)
