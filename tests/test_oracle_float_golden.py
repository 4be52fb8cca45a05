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


# ---------------------------------------------------------------------------------------------
# float_flow.npz: the reference-owned CONTROL FLOW (make_golden.py: float_flow), rtol = 0
# ---------------------------------------------------------------------------------------------
def _flow():
    z = np.load(os.path.join(GOLDEN, "float_flow.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_core_sage_loop_matches_reference_basegnncore_forward():
    """gnn_model.py:230-277 executed by the reference itself (SAGE branch, one node type, one edge
    type, SAGEConv stand-in = index_add_ + lin) == the oracle's gnn_core_hetero on the same weights."""
    g = _flow()
    L = 8
    sd = {"c.pre_mp.0.n.weight": g["core_pre_w"], "c.pre_mp.0.n.bias": g["core_pre_b"]}
    for l in range(L):
        sd[f"c.convs.{l}.n__e__n.lin.weight"] = g[f"core_lin_w{l}"]
        sd[f"c.convs.{l}.n__e__n.lin.bias"] = g[f"core_lin_b{l}"]
        sd[f"c.updates.{l}.n.weight"] = g[f"core_upd_w{l}"]
        sd[f"c.updates.{l}.n.bias"] = g[f"core_upd_b{l}"]
    emb = OM.gnn_core_hetero(sd, "c", {"n": g["core_x"]}, {("n", "e", "n"): g["core_ei"]}, ("n",),
                             (("n", "e", "n"),), L)
    torch.testing.assert_close(emb["n"], g["core_emb"], rtol=0, atol=0)


def test_gossip_post_mp_matches_reference_basegnn_forward():
    """gnn_model.py:58-109, baseline == "gossip": no anchor, no pooling, post_mp on the core output."""
    g = _flow()
    sd = {}
    for i in (0, 3, 5, 7):
        sd[f"emb_model.post_mp.{i}.weight"] = g[f"gpost_w{i}"]
        sd[f"emb_model.post_mp.{i}.bias"] = g[f"gpost_b{i}"]
    torch.testing.assert_close(OM.post_mp(sd, "emb_model", g["gpost_in"]), g["gpost_out"], rtol=0, atol=0)


def test_neighborhood_model_flows_match_reference():
    """lightning_model.py:198-222 / 228-254 / 256-283 executed by the reference with seeded embeddings."""
    g = _flow()
    sd = {"count_model.0.weight": g["nm_w0"], "count_model.0.bias": g["nm_b0"],
          "count_model.2.weight": g["nm_w2"], "count_model.2.bias": g["nm_b2"]}
    logits = OM.head_logits(sd, g["nm_emb_t"], g["nm_emb_q"])
    torch.testing.assert_close(OM.count_from_logits(logits), g["nm_count"], rtol=0, atol=0)
    torch.testing.assert_close(OM.train_loss_from_logits(logits, g["nm_y"]), g["nm_train_loss"], rtol=0, atol=0)
    torch.testing.assert_close(OM.eval_loss_from_logits(logits, g["nm_y"]), g["nm_test_loss"], rtol=0, atol=0)


def test_gossip_model_flows_match_reference():
    """lightning_model.py:613-628 / 585-608 executed by the reference with a seeded emb_model stand-in."""
    g = _flow()

    def emb_fn(x_col, qe):
        return F.linear(torch.cat((x_col, qe.expand(x_col.shape[0], -1)), dim=-1), g["gm_corr_w"], g["gm_corr_b"])
    pred = OM.gossip_query_loop(emb_fn, g["gm_x"], g["gm_q"])
    torch.testing.assert_close(pred, g["gm_count"], rtol=0, atol=0)
    torch.testing.assert_close(OM.gossip_loss_from_pred(pred, g["gm_y"]), g["gm_train_loss"], rtol=0, atol=0)


def test_dataset_helpers_match_reference():
    """workload.py:107-112, 303-324, 296-301 executed by the reference."""
    g = _flow()
    ind = g["ds_indicator"].numpy()
    torch.testing.assert_close(OM.apply_neighborhood_count(g["ds_count"], ind), g["ds_x"], rtol=0, atol=0)
    agg = OM.aggregate_by_index(g["ds_count"], g["ds_index"][:, 0].numpy(), 7)
    torch.testing.assert_close(agg, g["ds_agg"], rtol=0, atol=0)
    torch.testing.assert_close(OM.apply_truth(g["ds_truth"], ind), g["ds_y"], rtol=0, atol=0)
