"""The float oracle (reference-form restatement) cross-checked against independent dense-matrix
formulas on tiny graphs.  (Parity with the reference itself is UNPINNED, see oracle/__init__.py.)"""
import numpy as np
import torch

from helpers import cpu_sd, golden_graphs, make_models, standard_queries
from oracle import model as OM
from oracle import partition as OP


def _dense_adj(n_dst, n_src, ei):
    A = torch.zeros(n_dst, n_src, dtype=torch.double)
    for s, d in np.asarray(ei).T.tolist():
        A[d, s] += 1
    return A


def test_shmp_layer_dense_formula():
    nm, _ = make_models(seed=1)
    sd = {k: v.double() for k, v in cpu_sd(nm).items()}
    graphs = golden_graphs(max_n=12)[:3]
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    b = OP.neighborhood_batch(neighs)
    n = b["num_nodes"]
    feats = {t: torch.zeros(n[t], 1, dtype=torch.double) for t in OP.NODE_TYPES}
    emb = OM.gnn_core_hetero(sd, "emb_model.gnn_core", feats, b["edge_index"], OP.NODE_TYPES,
                             OP.EDGE_TYPES, 2, emulate_quirk=False)
    # dense: x_d' = relu(U [sum_t A_t x_s W_t^T + b_t | x_d] + c)
    p = "emb_model.gnn_core"
    x = {t: feats[t] @ sd[f"{p}.pre_mp.0.{t}.weight"].T + sd[f"{p}.pre_mp.0.{t}.bias"] for t in OP.NODE_TYPES}
    cat = dict(x)
    for l in range(2):
        nx_ = {}
        for d in OP.NODE_TYPES:
            acc = 0
            for (s, r, dd) in OP.EDGE_TYPES:
                if dd != d:
                    continue
                A = _dense_adj(n[d], n[s], b["edge_index"][(s, r, dd)])
                k = f"{p}.convs.{l}.{s}__{r}__{dd}.lin"
                acc = acc + (A @ x[s]) @ sd[k + ".weight"].T + sd[k + ".bias"]
            U, c = sd[f"{p}.updates.{l}.{d}.weight"], sd[f"{p}.updates.{l}.{d}.bias"]
            nx_[d] = torch.relu(torch.cat([acc, x[d]], 1) @ U.T + c)
        x = nx_
        cat = {t: torch.cat([cat[t], x[t]], 1) for t in OP.NODE_TYPES}
    for t in OP.NODE_TYPES:
        torch.testing.assert_close(emb[t], cat[t], rtol=1e-10, atol=1e-10)


def test_quirk_only_touches_affected_neighborhoods():
    nm, _ = make_models(seed=2)
    sd = cpu_sd(nm)
    qids, queries = standard_queries()
    graphs = golden_graphs(max_n=12)[:3]
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    b, qb = OP.neighborhood_batch(neighs), OP.query_batch(queries)
    on, _ = OM.neighborhood_logits(sd, b, qb, emulate_quirk=True)
    off, _ = OM.neighborhood_logits(sd, b, qb, emulate_quirk=False)
    changed = (on - off).abs().max(dim=1).values > 1e-7
    # graph 0 of the batch is [0,1] + edge (0,1): count node 0 == canonical index 0 -> affected;
    # every later graph g needs count index g inside its own range, impossible once any earlier
    # neighborhood has >1 count node
    assert changed[0] and changed.sum() <= 2


def test_gossip_dense_formula():
    _, gm = make_models(seed=3)
    sd = {k: v.double() for k, v in cpu_sd(gm).items()}
    n = 7
    und = [(0, 1), (1, 2), (2, 0), (2, 3), (4, 5), (5, 6)]
    ei = np.array(und + [(b, a) for a, b in und]).T
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, 1, generator=g, dtype=torch.double) * 10
    qe = torch.randn(1, 64, generator=g, dtype=torch.double)
    out = OM.gossip_single_query(sd, x, ei, qe, 2)
    # dense: gated adjacency G[i,j] = gate if j<i else 1-gate for neighbours
    A = _dense_adj(n, n, ei)
    lower = torch.tril(torch.ones(n, n, dtype=torch.double), -1)
    p = "emb_model.gnn_core"
    h = torch.cat([qe.expand(n, -1), x @ sd[f"{p}.pre_mp.0.weight"].T + sd[f"{p}.pre_mp.0.bias"]], 1)
    emb = h
    for l in range(2):
        k = f"{p}.convs.{l}"
        gate = OM.gossip_gate(sd, k, qe).item()
        Gm = A * (lower * gate + (1 - lower) * (1 - gate))
        msg = h @ sd[k + ".lin_com.weight"].T + sd[k + ".lin_com.bias"]
        h = torch.relu(torch.cat([Gm @ msg, h], 1) @ sd[k + ".lin_update.weight"].T + sd[k + ".lin_update.bias"])
        emb = torch.cat([emb, h], 1)
    ref = OM.post_mp(sd, "emb_model", emb)
    torch.testing.assert_close(out, ref, rtol=1e-10, atol=1e-10)


def test_reference_pipeline_shapes_and_aggregation():
    nm, gm = make_models(seed=0)
    qids, queries = standard_queries()
    graphs = golden_graphs(max_n=12)[:4]
    r = OM.reference_pipeline(cpu_sd(nm), cpu_sd(gm), graphs, queries)
    N = sum(n for n, _ in graphs)
    assert r["x"].shape == (N, 29) and r["node_count"].shape == (N, 29)
    assert r["graph_gossip_count"].shape == (4, 29)
    assert torch.equal(r["x"][torch.from_numpy(~r["indicator"])], torch.zeros((~r["indicator"]).sum(), 29))
    torch.testing.assert_close(r["graph_neigh_count"].sum(0), r["neigh_count"].sum(0), rtol=1e-5, atol=1e-5)
