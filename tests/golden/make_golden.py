#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own functions.

Run ONLY in the build container (needs /root/reference, read-only):
    cd tests/golden && python -B make_golden.py
The outputs (*.json) are data: seeded inputs and the reference's outputs on them.  No reference
source is copied.  The GPU box never runs this script (it has no /root/reference).

Reference functions executed (unmodified, via _ref_import's inert third-party stubs):
  data.py:37-58    gen_query_ids          data.py:61-88   SymmetricFactor / GenVMap
  data.py:329-338  k_neigh                data.py:375-396 get_neigh_hetero
  workload.py:327-348 MatchSubgraphWorker workload.py:1128-1671 graph_atlas_plus
  analysis.py:22-83 norm_mse / mse / mae  config.py:185-400 parse_* defaults
"""
import argparse
import io
import json
import contextlib
import os

import networkx as nx
import numpy as np

import _ref_import

HERE = os.path.dirname(os.path.abspath(__file__))
DEPTH = 4


def seeded_graphs():
    """~24 small graphs covering tree / ring / dense / star / disconnected-after-filter cases."""
    gs = []
    # SURVEY Appendix B toy case
    gs.append(("toy_appendix_b", 8, [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (3, 4), (4, 5), (5, 6), (6, 7), (7, 2)]))
    gs.append(("path12", 12, [(i, i + 1) for i in range(11)]))
    gs.append(("ring9", 9, [(i, (i + 1) % 9) for i in range(9)]))
    gs.append(("star_hub_low", 10, [(0, i) for i in range(1, 10)]))
    gs.append(("star_hub_high", 10, [(9, i) for i in range(0, 9)]))
    gs.append(("k6", 6, [(i, j) for i in range(6) for j in range(i + 1, 6)]))
    gs.append(("two_components", 11, [(0, 1), (1, 2), (2, 0), (3, 4), (4, 5), (5, 6), (6, 3), (7, 8), (9, 10)]))
    gs.append(("isolated_nodes", 7, [(1, 2), (2, 5)]))
    # path whose low-id part is only reachable through a higher id (disconnected after the id filter)
    gs.append(("filter_disconnects", 8, [(0, 7), (7, 1), (1, 6), (6, 2), (2, 5), (5, 3), (3, 4)]))
    # long ring: retained nodes can be > DEPTH hops from the canonical node inside the neighborhood
    gs.append(("ring20_perm", 20, [((7 * i) % 20, (7 * (i + 1)) % 20) for i in range(20)]))
    rng = np.random.default_rng(20240817)
    for k, (n, m) in enumerate([(15, 20), (25, 30), (30, 60), (40, 44), (41, 45), (60, 70), (60, 150), (35, 34)]):
        g = nx.gnm_random_graph(n, m, seed=int(rng.integers(1 << 30)))
        gs.append((f"gnm_{n}_{m}", n, sorted(g.edges())))
    for k, (n, mm) in enumerate([(30, 2), (50, 3)]):
        g = nx.barabasi_albert_graph(n, mm, seed=int(rng.integers(1 << 30)))
        perm = rng.permutation(n)
        gs.append((f"ba_{n}_{mm}_relabel", n, sorted((int(min(perm[a], perm[b])), int(max(perm[a], perm[b]))) for a, b in g.edges())))
    g = nx.random_labeled_tree(45, seed=7)
    gs.append(("tree45", 45, sorted(g.edges())))
    g = nx.grid_2d_graph(5, 6)
    g = nx.convert_node_labels_to_integers(g)
    gs.append(("grid5x6", 30, sorted(g.edges())))
    # larger graphs: exercise the reference's hash-ordered node iteration (order-independent checks)
    g = nx.gnm_random_graph(200, 260, seed=11)
    gs.append(("gnm_200_260", 200, sorted(g.edges())))
    g = nx.connected_watts_strogatz_graph(120, 4, 0.2, seed=5)
    gs.append(("ws_120", 120, sorted(g.edges())))
    return gs


def to_nx(n, edges):
    G = nx.Graph()
    G.add_nodes_from(range(n))          # pyg.utils.to_networkx adds nodes 0..n-1 in order
    G.add_edges_from(edges)
    return G


def main():
    ref_data, ref_workload, ref_analysis, ref_config = _ref_import.load()

    # ---- (ii) queries -------------------------------------------------------------------
    query_ids = ref_data.gen_query_ids([3, 4, 5])
    queries = []
    for qid in query_ids:
        q = ref_workload.graph_atlas_plus(qid)
        queries.append({
            "atlas_id": int(qid), "n": q.number_of_nodes(),
            "edges": sorted((int(min(a, b)), int(max(a, b))) for a, b in q.edges()),
            "diameter": int(nx.diameter(q)),
            "symmetry_factor": int(ref_data.SymmetricFactor(q)),
        })
    with open(os.path.join(HERE, "queries.json"), "w") as f:
        json.dump({"query_ids": [int(i) for i in query_ids], "queries": queries}, f)

    # ---- (i) canonical partition ----------------------------------------------------------
    out = {"depth": DEPTH, "graphs": []}
    for name, n, edges in seeded_graphs():
        G = to_nx(n, edges)
        index, indicator, neighs = [], [], []
        ascending, canon_last = True, True
        for node in G.nodes:
            ng = ref_data.get_neigh_hetero(G, node, DEPTH)
            order = [int(v) for v in ng.nodes]
            if len(ng.edges) == 0:                      # workload.py:252-256
                indicator.append(False)
                continue
            indicator.append(True)
            index.append(int(node))
            ascending &= order == sorted(order)
            canon_last &= order[-1] == node
            types = {int(v): ng.nodes[v]["type"] for v in ng.nodes}
            assert types[node] == "canonical" and sum(t == "canonical" for t in types.values()) == 1
            neighs.append({
                "canonical": int(node),
                "nodes": sorted(order),
                "edges": sorted((int(min(a, b)), int(max(a, b))) for a, b in ng.edges()),
                "ref_node_order": order,
            })
        out["graphs"].append({
            "name": name, "n": n, "edges": [[int(a), int(b)] for a, b in edges],
            "indicator": indicator, "index_nodes": index, "neighs": neighs,
            "ref_order_is_ascending": bool(ascending), "ref_canonical_is_last": bool(canon_last),
        })
    with open(os.path.join(HERE, "partition_golden.json"), "w") as f:
        json.dump(out, f)

    # ---- (iii) canonical ground-truth counts ---------------------------------------------------
    nx_queries = [ref_workload.graph_atlas_plus(q) for q in query_ids]
    sym = [q["symmetry_factor"] for q in queries]
    truth = []
    for name, n, edges in seeded_graphs():
        if n > 41:
            continue
        G = to_nx(n, edges)
        cnt = np.zeros((n, len(query_ids)), dtype=np.int64)
        for qi, q in enumerate(nx_queries):
            _, _, items = ref_workload.MatchSubgraphWorker((0, G, qi, q, None))
            for node, c in items:
                assert c % sym[qi] == 0
                cnt[node, qi] = c // sym[qi]
        truth.append({"name": name, "count": cnt.tolist()})
    with open(os.path.join(HERE, "canonical_counts.json"), "w") as f:
        json.dump(truth, f)

    # ---- (iv) metrics --------------------------------------------------------------------------
    rng = np.random.default_rng(3)
    pred = rng.gamma(2.0, 5.0, size=(37, 29)).astype(np.float32)
    tru = np.round(rng.gamma(2.0, 5.0, size=(37, 29))).astype(np.float32)
    groups = [[0, 1], list(range(2, 8)), list(range(8, 29))]
    with contextlib.redirect_stdout(io.StringIO()):
        m = {
            "norm_mse": ref_analysis.norm_mse(pred, tru, groups),
            "mse": ref_analysis.mse(pred, tru, groups),
            "mae": [float(v) for v in ref_analysis.mae(pred, tru, groups)],
            "norm_mse_all": ref_analysis.norm_mse(pred, tru),
        }
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump({"pred": pred.tolist(), "truth": tru.tolist(), "groups": groups, **m}, f)

    # ---- (v) config defaults -------------------------------------------------------------------
    parser = argparse.ArgumentParser()
    ref_config.parse_neighborhood(parser)
    ref_config.parse_gossip(parser)
    ref_config.parse_optimizer(parser)
    ns = parser.parse_args([])

    def jsonable(v):
        return v if isinstance(v, (int, float, str, bool, type(None), list)) else repr(v)
    with open(os.path.join(HERE, "config_defaults.json"), "w") as f:
        json.dump({k: jsonable(v) for k, v in sorted(vars(ns).items())}, f, indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()


def float_pieces():
    """Golden vectors for the pieces of the FLOAT path that the reference's own code can execute
    without PyG: methods that are pure torch once ``self`` carries the right nn.Modules are called
    UNBOUND with a stand-in ``self`` (no reference source is copied or modified):
      lightning_model.py:176-193  NeighborhoodCountingModel.embed_to_count (concat + count_model)
      lightning_model.py:285-289  NeighborhoodCountingModel.criterion
      lightning_model.py:630-635  GossipCountingModel.criterion
      gnn_model.py:335-344        GossipConv.message  (lin_com, direction-gated scaling)
      gnn_model.py:346-350        GossipConv.update   (cat + lin_update)
      gnn_model.py:357-359        GossipConv._gate_value
    MessagePassing.propagate / to_hetero / global_add_pool stay [EXT] (not executable here)."""
    import types
    import torch
    import torch.nn as nn
    _ref_import.install()
    import subgraph_counting.gnn_model as ref_gnn
    import subgraph_counting.lightning_model as ref_lm

    torch.manual_seed(1234)
    out = {}
    # ---- count head -------------------------------------------------------------------------
    count_model = nn.Sequential(nn.Linear(128, 256), nn.LeakyReLU(), nn.Linear(256, 1))
    fake = types.SimpleNamespace(kwargs={}, count_model=count_model)
    emb_t = torch.randn(37, 64)
    emb_q = torch.randn(5, 64)
    cols = []
    with torch.no_grad():
        for q in range(5):      # the loop of lightning_model.py:210-219
            cols.append(ref_lm.NeighborhoodCountingModel.embed_to_count(
                fake, (emb_t, emb_q[q].expand_as(emb_t))))
    out.update(head_w0=count_model[0].weight, head_b0=count_model[0].bias, head_w2=count_model[2].weight,
               head_b2=count_model[2].bias, head_emb_t=emb_t, head_emb_q=emb_q, head_out=torch.cat(cols, -1))
    # ---- criteria -----------------------------------------------------------------------------
    cnt, tru = torch.randn(41, 1) * 3, torch.rand(41, 1) * 6
    out.update(crit_count=cnt, crit_truth=tru,
               crit_neigh=ref_lm.NeighborhoodCountingModel.criterion(fake, cnt, tru),
               crit_gossip=ref_lm.GossipCountingModel.criterion(fake, cnt, tru))
    # ---- GossipConv message / update / gate ---------------------------------------------------
    for name, cin in (("g0", 128), ("g1", 64)):
        conv = types.SimpleNamespace(
            lin_com=nn.Linear(cin, 64), lin_update=nn.Linear(64 + cin, 64),
            lin_gate=nn.Sequential(nn.Linear(64, 64), nn.Sigmoid(), nn.Linear(64, 1), nn.Sigmoid(),
                                   nn.LeakyReLU()))
        n, e = 23, 57
        x = torch.randn(n, cin)
        src, dst = torch.randint(0, n, (e,)), torch.randint(0, n, (e,))
        keep = src != dst
        src, dst = src[keep], dst[keep]
        edge_weight = src < dst                                  # gnn_model.py:248
        qe = torch.randn(1, 64)
        with torch.no_grad():
            msg = ref_gnn.GossipConv.message(conv, x[dst], x[src], edge_weight, qe)
            aggr = torch.zeros(n, 64).index_add_(0, dst, msg)     # aggr="add" at edge_index[1] [EXT]
            upd = ref_gnn.GossipConv.update(conv, aggr, x, None)
            gate = ref_gnn.GossipConv._gate_value(conv, qe)
        out.update({f"{name}_com_w": conv.lin_com.weight, f"{name}_com_b": conv.lin_com.bias,
                    f"{name}_upd_w": conv.lin_update.weight, f"{name}_upd_b": conv.lin_update.bias,
                    f"{name}_gate0_w": conv.lin_gate[0].weight, f"{name}_gate0_b": conv.lin_gate[0].bias,
                    f"{name}_gate2_w": conv.lin_gate[2].weight, f"{name}_gate2_b": conv.lin_gate[2].bias,
                    f"{name}_x": x, f"{name}_src": src, f"{name}_dst": dst, f"{name}_qe": qe,
                    f"{name}_msg": msg, f"{name}_upd": upd, f"{name}_gate": gate})
    np.savez_compressed(os.path.join(HERE, "float_pieces.npz"),
                        **{k: v.detach().numpy() for k, v in out.items()})
    print("float_pieces.npz written")


if __name__ == "__main__":
    float_pieces()


def float_flow():
    """Golden vectors for the reference-owned float CONTROL FLOW that is executable without PyG
    (missing #1).  Every function below is the reference's own code, called UNBOUND with
    a stand-in ``self`` whose sub-modules are plain torch modules or seeded callables (no reference
    source is copied or modified; the stand-ins replace only what PyG / Lightning hold):
      gnn_model.py:230-277      BaseGNNCore.forward, SAGE branch: pre_mp -> convs[i] -> updates[i](cat) ->
                                relu -> dropout -> running cat.  ``convs[i]`` = index_add_ + Linear closure
                                (the SAGEConv stand-in; PyG's propagate stays [EXT])
      gnn_model.py:58-109       BaseGNN.forward, baseline == "gossip" path: core -> (no anchor) -> post_mp
      lightning_model.py:198-222  NeighborhoodCountingModel.graph_to_count  (query loop, 2**pred - 1)
      lightning_model.py:228-254  .train_forward  (log2(y+1), smooth_l1 per query, mean)
      lightning_model.py:256-283  .test_forward   (relu(2**(pred-1)) vs y)
      lightning_model.py:613-628  GossipCountingModel.graph_to_count  (neigh_pred + gossip_pred per query)
      lightning_model.py:585-608  .train_forward  (sum over queries of log2(|pred-y|+1))
      workload.py:107-112       GossipDataset.apply_neighborhood_count (zeros + masked row scatter)
      workload.py:303-324       NeighborhoodDataset.aggregate_neighborhood_count (index_add_)
      workload.py:296-301       NeighborhoodDataset.apply_truth_from_dataset (indicator row select)
    """
    import types
    import torch
    import torch.nn as nn
    _ref_import.install()
    import subgraph_counting.gnn_model as ref_gnn
    import subgraph_counting.lightning_model as ref_lm
    import subgraph_counting.workload as ref_wl

    torch.manual_seed(4321)
    out = {}
    H, L = 64, 8
    # ---- BaseGNNCore.forward, SAGE branch (homogeneous) -----------------------------------------
    n, e = 19, 46
    src, dst = torch.randint(0, n, (e,)), torch.randint(0, n, (e,))
    keep = src != dst
    ei = torch.stack((src[keep], dst[keep]))
    pre_mp = nn.Sequential(nn.Linear(1, H))
    lins = nn.ModuleList([nn.Linear(H, H) for _ in range(L)])
    updates = nn.ModuleList([nn.Linear(2 * H, H) for _ in range(L)])

    def conv(l):    # SAGEConv stand-in: aggr="add" at edge_index[1] of x_j = x[edge_index[0]], then lin
        return lambda x, edge_index: lins[l](
            torch.zeros(x.shape[0], H).index_add_(0, edge_index[1], x[edge_index[0]]))
    core = types.SimpleNamespace(pre_mp=pre_mp, input_pattern_emb=False, conv_type="SAGE",
                                 convs=[conv(l) for l in range(L)], updates=updates, dropout=0.0,
                                 training=False)
    x = torch.randn(n, 1)
    with torch.no_grad():
        emb = ref_gnn.BaseGNNCore.forward(core, x, ei)
    assert emb.shape == (n, H * (L + 1))
    out.update(core_x=x, core_ei=ei, core_emb=emb, core_pre_w=pre_mp[0].weight, core_pre_b=pre_mp[0].bias)
    for l in range(L):
        out.update({f"core_lin_w{l}": lins[l].weight, f"core_lin_b{l}": lins[l].bias,
                    f"core_upd_w{l}": updates[l].weight, f"core_upd_b{l}": updates[l].bias})
    # ---- BaseGNN.forward, gossip path: post_mp over the core output, no anchor, no pooling ----------
    post = nn.Sequential(nn.Linear(256, H), nn.Dropout(0.0), nn.LeakyReLU(0.1), nn.Linear(H, H), nn.ReLU(),
                         nn.Linear(H, 256), nn.ReLU(), nn.Linear(256, 1)).eval()
    core_out = torch.randn(n, 256)
    gnn = types.SimpleNamespace(use_hetero=False, kwargs={"baseline": "gossip"}, post_mp=post,
                                anchor_mlp=None,
                                gnn_core=types.SimpleNamespace(forward=lambda x, ei, query_emb=None: core_out))
    data = types.SimpleNamespace(node_feature=torch.ones(n, 1), edge_index=ei,
                                 batch=torch.zeros(n, dtype=torch.long))
    with torch.no_grad():
        gpost = ref_gnn.BaseGNN.forward(gnn, data, query_emb=torch.randn(1, H))
    out.update(gpost_in=core_out, gpost_out=gpost,
               **{f"gpost_w{i}": post[i].weight for i in (0, 3, 5, 7)},
               **{f"gpost_b{i}": post[i].bias for i in (0, 3, 5, 7)})
    # ---- NeighborhoodCountingModel: graph_to_count / train_forward / test_forward --------------------
    B, Q = 37, 5
    count_model = nn.Sequential(nn.Linear(2 * H, 4 * H), nn.LeakyReLU(), nn.Linear(4 * H, 1))
    ET, EQ = torch.randn(B, H), torch.randn(Q, H)

    class _QB:      # a query batch: only .to(device) is used (lightning_model.py:205)
        def to(self, device):
            return self
    nm = types.SimpleNamespace(kwargs={}, count_model=count_model, device="cpu", query_loader=[_QB()],
                               emb_model_query=lambda qb: EQ, emb_model=lambda b: ET)
    nm.embed_to_count = types.MethodType(ref_lm.NeighborhoodCountingModel.embed_to_count, nm)
    nm.criterion = types.MethodType(ref_lm.NeighborhoodCountingModel.criterion, nm)
    y = torch.floor(torch.rand(B, Q) * 9)
    batch = types.SimpleNamespace(y=y)
    with torch.no_grad():
        out.update(nm_count=ref_lm.NeighborhoodCountingModel.graph_to_count(nm, batch),
                   nm_train_loss=ref_lm.NeighborhoodCountingModel.train_forward(nm, batch, 0),
                   nm_test_loss=ref_lm.NeighborhoodCountingModel.test_forward(nm, batch, 0))
    out.update(nm_emb_t=ET, nm_emb_q=EQ, nm_y=y, nm_w0=count_model[0].weight, nm_b0=count_model[0].bias,
               nm_w2=count_model[2].weight, nm_b2=count_model[2].bias)
    # ---- GossipCountingModel: graph_to_count / train_forward -------------------------------------------
    N = 29
    corr = nn.Linear(1 + H, 1)      # emb_model stand-in: any function of (batch.node_feature, query_emb)

    def emb_model(b, query_emb=None):
        return corr(torch.cat((b.node_feature, query_emb.expand(b.node_feature.shape[0], -1)), dim=-1))
    gx, gy, gq = torch.rand(N, Q) * 7, torch.floor(torch.rand(N, Q) * 9), torch.randn(Q, H)
    gm = types.SimpleNamespace(query_emb=gq, device="cpu", emb_model=emb_model)
    gm.criterion = types.MethodType(ref_lm.GossipCountingModel.criterion, gm)
    gb = types.SimpleNamespace(x=gx, y=gy)
    with torch.no_grad():
        out.update(gm_count=ref_lm.GossipCountingModel.graph_to_count(gm, gb),
                   gm_train_loss=ref_lm.GossipCountingModel.train_forward(gm, gb, 0))
    out.update(gm_x=gx, gm_y=gy, gm_q=gq, gm_corr_w=corr.weight, gm_corr_b=corr.bias)
    # ---- dataset helpers -----------------------------------------------------------------------------
    indicator = torch.rand(41) < 0.7
    cnt = torch.rand(int(indicator.sum()), Q) * 5
    gd = types.SimpleNamespace(data=types.SimpleNamespace(), slices={"y": torch.tensor([0, 20, 41])})
    ref_wl.GossipDataset.apply_neighborhood_count(gd, cnt, indicator)
    gids = torch.sort(torch.randint(0, 6, (cnt.shape[0],))).values
    index = np.stack([gids.numpy(), np.arange(cnt.shape[0])], axis=1)
    nd = types.SimpleNamespace(nx_neighs_index=index, dataset=list(range(7)), data=types.SimpleNamespace(),
                               nx_neighs_indicator=indicator)
    agg = ref_wl.NeighborhoodDataset.aggregate_neighborhood_count(nd, cnt)
    truth = torch.floor(torch.rand(41, Q) * 4)
    ref_wl.NeighborhoodDataset.apply_truth_from_dataset(nd, truth)
    out.update(ds_indicator=indicator, ds_count=cnt, ds_x=gd.data.x, ds_index=torch.from_numpy(index),
               ds_agg=agg, ds_truth=truth, ds_y=nd.data.y)
    np.savez_compressed(os.path.join(HERE, "float_flow.npz"),
                        **{k: v.detach().numpy() for k, v in out.items()})
    print("float_flow.npz written")


if __name__ == "__main__":
    float_flow()
