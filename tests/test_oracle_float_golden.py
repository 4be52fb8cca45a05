"""Pieces of the float path pinned by vectors the reference's OWN code produced (unbound methods
executed with a stand-in ``self``; tests/golden/make_golden.py: float_pieces): the count head,
both criteria, GossipConv message / update / gate.  The oracle's restatement must reproduce them."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import model as OM


def _load():
    z = np.load(os.path.join(GOLDEN, "float_pieces.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_count_head_matches_reference_embed_to_count():
    g = _load()
    sd = {"count_model.0.weight": g["head_w0"], "count_model.0.bias": g["head_b0"],
          "count_model.2.weight": g["head_w2"], "count_model.2.bias": g["head_b2"]}
    cols = []
    for q in range(g["head_emb_q"].shape[0]):       # the oracle's head loop (oracle/model.py neighborhood_logits)
        e = torch.cat((g["head_emb_t"], g["head_emb_q"][q].expand_as(g["head_emb_t"])), dim=-1)
        cols.append(OM._lin(sd, "count_model.2", F.leaky_relu(OM._lin(sd, "count_model.0", e))))
    torch.testing.assert_close(torch.cat(cols, -1), g["head_out"], rtol=0, atol=0)
    # the separable form the HIP head kernel uses is the same function
    W1 = g["head_w0"]
    T = g["head_emb_t"] @ W1[:, :64].T
    Qh = g["head_emb_q"] @ W1[:, 64:].T + g["head_b0"]
    sep = F.leaky_relu(T[:, None, :] + Qh[None]) @ g["head_w2"][0] + g["head_b2"]
    torch.testing.assert_close(sep, g["head_out"], rtol=1e-5, atol=1e-5)


def test_criteria_match_reference():
    g = _load()
    torch.testing.assert_close(F.smooth_l1_loss(g["crit_count"], g["crit_truth"]), g["crit_neigh"], rtol=0, atol=0)
    torch.testing.assert_close(torch.log2(torch.abs(g["crit_count"] - g["crit_truth"]) + 1), g["crit_gossip"],
                               rtol=0, atol=0)


def test_gossip_conv_matches_reference_message_update_gate():
    g = _load()
    for name in ("g0", "g1"):
        key = "emb_model.gnn_core.convs.0"
        sd = {f"{key}.lin_com.weight": g[f"{name}_com_w"], f"{key}.lin_com.bias": g[f"{name}_com_b"],
              f"{key}.lin_update.weight": g[f"{name}_upd_w"], f"{key}.lin_update.bias": g[f"{name}_upd_b"],
              f"{key}.lin_gate.0.weight": g[f"{name}_gate0_w"], f"{key}.lin_gate.0.bias": g[f"{name}_gate0_b"],
              f"{key}.lin_gate.2.weight": g[f"{name}_gate2_w"], f"{key}.lin_gate.2.bias": g[f"{name}_gate2_b"]}
        x, src, dst, qe = g[f"{name}_x"], g[f"{name}_src"].long(), g[f"{name}_dst"].long(), g[f"{name}_qe"]
        gate = OM.gossip_gate(sd, key, qe)
        torch.testing.assert_close(gate, g[f"{name}_gate"], rtol=0, atol=0)
        # the oracle's layer body (oracle/model.py gossip_single_query) on the same directed edges
        dirw = src < dst
        msg = OM._lin(sd, key + ".lin_com", x[src])
        msg[dirw] *= gate
        msg[~dirw] *= 1 - gate
        torch.testing.assert_close(msg, g[f"{name}_msg"], rtol=0, atol=0)
        aggr = torch.zeros(x.shape[0], 64).index_add_(0, dst, msg)
        upd = OM._lin(sd, key + ".lin_update", torch.cat((aggr, x), dim=-1))
        torch.testing.assert_close(upd, g[f"{name}_upd"], rtol=0, atol=0)
