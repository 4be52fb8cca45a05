"""--use_node_feature (config.py:226-230, main.py:51-64, utils.py:258-272): labelled
query expansion, node features through the datasets into the batches, input_dim > 1 through the
kernels, more than 32 / 64 queries through the head / gossip launches, labelled ground truth."""
import itertools

import networkx as nx
import numpy as np
import pytest
import torch

from desco_amd.data import add_node_feat_to_networkx, graph_atlas_plus
from desco_amd.graphs import GraphSet
from helpers import golden_graphs, gossip_args, neigh_args

F = 2
QIDS = [6, 7, 13, 14, 15, 16]          # 2 x 2**3 + 4 x 2**4 = 80 labelled queries


def _featured_graphs():
    graphs = golden_graphs(max_n=30)[:8]
    rng = np.random.default_rng(11)
    feats = [np.eye(F, dtype=np.float32)[rng.integers(F, size=n)] for n, _ in graphs]
    return graphs, feats, GraphSet.from_edge_lists(graphs, node_feat=feats)


def test_query_expansion_order_and_features():
    from desco_amd.lightning_model import gen_queries, query_node_features
    flat, qs = gen_queries([6, 7], node_feat_len=F)
    assert len(qs) == 2 * F ** 3 and all(n == 3 for n, _ in flat)
    eye = np.eye(F).tolist()
    want = list(itertools.product(eye, repeat=3))
    for g, w in zip(qs[:8], want):                       # itertools.product order, node k <- k-th entry
        assert [g.nodes[k]["feat"] for k in range(3)] == list(w)
    nf = query_node_features(qs, F)
    assert nf.shape == (16 * 3, F) and torch.equal(nf[:3], torch.tensor(want[0]))
    assert query_node_features([graph_atlas_plus(6)], 1) is None
    assert len(add_node_feat_to_networkx(nx.path_graph(2), eye)) == F ** 2


def test_graphset_features_follow_subset_replicate_and_relabel():
    from desco_amd.data import relabel
    graphs, feats, gs = _featured_graphs()
    sub = gs.subset(2, 5)
    assert np.array_equal(sub.node_feat, np.concatenate(feats[2:5]))
    assert np.array_equal(gs.replicate(2).node_feat, np.concatenate(feats + feats))
    r = relabel(gs, "decreasing_degree")
    deg = np.diff(r.rowptr)
    for g in range(r.num_graphs):                        # degrees non-increasing, features moved along
        a, b = r.graph_ptr[g], r.graph_ptr[g + 1]
        assert (np.diff(deg[a:b]) <= 0).all()
        assert sorted(map(tuple, r.node_feat[a:b])) == sorted(map(tuple, feats[g]))


def test_labelled_ground_truth_sums_to_unlabelled():
    """Summing the labelled canonical counts over all labelings of a query (weighted by the symmetry
    factors) gives the unlabelled count -- ties the VF2 path to the native enumerator."""
    from desco_amd.groundtruth import canonical_counts, canonical_counts_labelled
    graphs, feats, gs = _featured_graphs()
    small = gs.subset(0, 3)
    tri, path = graph_atlas_plus(7), graph_atlas_plus(6)
    eye = np.eye(F).tolist()
    for q in (tri, path):
        lab = add_node_feat_to_networkx(q, eye, "feat")
        c_lab = canonical_counts_labelled(small, lab)
        c = canonical_counts(small, [q], backend="host")
        match = lambda a, b: a["feat"] == b["feat"]     # noqa: E731
        sym_lab = torch.tensor([float(sum(1 for _ in nx.algorithms.isomorphism.GraphMatcher(
            g, g, node_match=match).subgraph_isomorphisms_iter())) for g in lab], dtype=torch.double)
        sym = float(sum(1 for _ in nx.algorithms.isomorphism.GraphMatcher(q, q).subgraph_isomorphisms_iter()))
        # every unlabelled embedding is a labelled embedding of exactly one labeling
        total = (c_lab * sym_lab).sum(dim=1) / sym
        torch.testing.assert_close(total, c[:, 0].double(), rtol=0, atol=1e-9)


def test_workload_requires_matching_features():
    from desco_amd.workload import Workload
    graphs, feats, gs = _featured_graphs()
    with pytest.raises(ValueError):
        Workload(GraphSet.from_edge_lists(graphs), root=None, node_feat_len=F)
    with pytest.raises(ValueError):
        Workload(gs, root=None, node_feat_len=F + 1)


@pytest.mark.gpu
def test_node_feature_model_vs_oracle():
    from desco_amd.batch import GossipBatch
    from desco_amd.lightning_model import GossipCountingModel, NeighborhoodCountingModel
    from desco_amd.workload import Workload
    from helpers import assert_logits_close, cpu_sd, report
    from oracle import model as OM
    from oracle import partition as OP
    dev = "cuda"
    graphs, feats, gs = _featured_graphs()
    torch.manual_seed(0)
    nm = NeighborhoodCountingModel(F, 64, neigh_args(input_dim=F)).to_hetero_old(True, True)
    gm = GossipCountingModel(1, 64, gossip_args(), emb_channels=64, input_pattern_emb=True)
    with torch.no_grad():
        for m, gain in ((nm, 1.3), (gm, 1.4)):
            for p in m.parameters():
                if p.dim() == 2:
                    p.mul_(gain)
    nm, gm = nm.to(dev), gm.to(dev)
    nm.set_queries(QIDS)
    assert len(nm.queries_flat) == 80 and nm.query_loader.node_feature.shape == (2 * 8 * 3 + 4 * 16 * 4, F)
    w = Workload(gs, root=None, node_feat_len=F)
    w.generate_pipeline_datasets(depth_neigh=4)
    nd = w.neighborhood_dataset
    batch = nd.batch(0, len(nd), dev)
    assert batch.node_feature.shape == (batch.num_rows, F)
    # oracle inputs: per neighborhood the count nodes (ascending id) then the canonical node
    idx, ind, neighs = OP.neighborhood_dataset(graphs, 4)
    gfeat = [torch.from_numpy(f) for f in feats]
    cf = torch.cat([gfeat[g][nodes[:-1]] for (g, _), (nodes, _) in zip(idx, neighs)])
    kf = torch.stack([gfeat[g][nodes[-1]] for (g, _), (nodes, _) in zip(idx, neighs)])
    qs = [(n, e) for n, e in nm.queries_flat]
    qfeat = {"union_node": nm.query_feat}
    ref, emb_q = OM.neighborhood_logits(cpu_sd(nm), OP.neighborhood_batch(neighs), OP.query_batch(qs),
                                        input_dim=F, feats={"count": cf, "canonical": kf},
                                        emulate_quirk=False, qfeats=qfeat)
    with torch.no_grad():
        got = nm._logits(batch, exp2=False)                        # 80 queries: three head launches
    report("node-feature logits", got, ref)
    assert_logits_close("node-feature logits", got, ref)
    assert float(ref.std()) > 1e-3 and float((ref[:, 0] - ref[:, 1]).abs().max()) > 1e-4   # labels matter
    assert_logits_close("node-feature query_emb", nm.get_query_emb(), emb_q)
    # gossip over 80 query columns: two column groups of the scalars / fused launches
    g = torch.Generator().manual_seed(1)
    x = torch.rand(gs.num_nodes, 80, generator=g) * 20
    gm.set_query_emb(nm.get_query_emb())
    gb = GossipBatch(gs, dev, x=x)
    gref = OM.gossip_graph_to_count(cpu_sd(gm), x, gb.edge_index.numpy(), emb_q, 2) - x
    ggot = gm.graph_to_count(gb).cpu() - x
    report("node-feature gossip_corr", ggot, gref)
    assert_logits_close("node-feature gossip_corr", ggot, gref)
    # training step with features: loss vs oracle
    y = torch.floor(torch.rand(len(nd), 80, generator=g) ** 3 * 40)
    nd.y = y
    loss = nm.train_forward(nd.batch(0, len(nd), dev), 0)
    ref_loss = OM.neighborhood_loss(cpu_sd(nm), OP.neighborhood_batch(neighs), OP.query_batch(qs), y,
                                    input_dim=F, feats={"count": cf, "canonical": kf}, emulate_quirk=False,
                                    qfeats=qfeat)
    torch.testing.assert_close(loss.detach().cpu(), ref_loss, rtol=1e-4, atol=1e-5)
    loss.backward()
    assert nm.emb_model.gnn_core.pre_mp[0]["count"].weight.grad.abs().max() > 0
