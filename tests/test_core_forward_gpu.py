"""``BaseGNNCore.forward(x, edge_index, query_emb=None)`` with the reference's arguments (gnn_model.py:230-277):
the op-by-op form on this library's kernels against the oracle's restatement of the same function, for the hetero
SAGE core (target metadata with tconv, query metadata) and for the gossip core (one query)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as OM  # noqa: E402
from oracle import partition as OP  # noqa: E402

from helpers import assert_logits_close, cpu_sd, golden_graphs, make_models, standard_queries  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def models():
    nm, gm = make_models(seed=0)
    qids, queries = standard_queries()
    nm, gm = nm.to(DEV), gm.to(DEV)
    nm.set_queries(qids)
    return nm, gm, queries


def test_sage_core_forward_on_a_hetero_batch(models):
    nm, _, queries = models
    core = nm.emb_model.gnn_core.eval()
    graphs = golden_graphs(max_n=22)[:3]
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    batch = OP.neighborhood_batch(neighs[:48])
    feats = {t: torch.zeros(batch["num_nodes"][t], 1) for t in ("count", "canonical")}
    ref = OM.gnn_core_hetero(cpu_sd(nm), "emb_model.gnn_core", feats, batch["edge_index"], ("count", "canonical"),
                             OP.EDGE_TYPES, 8, emulate_quirk=True)
    got = core({t: f.to(DEV) for t, f in feats.items()},
               {et: torch.as_tensor(ei).to(DEV) for et, ei in batch["edge_index"].items()})
    assert set(got) == {"count", "canonical"}
    for t in ("count", "canonical"):
        assert got[t].shape == (batch["num_nodes"][t], 64 * 9)
        assert_logits_close(f"BaseGNNCore.forward [{t}]", got[t], ref[t])


def test_sage_core_forward_on_the_query_batch(models):
    nm, _, queries = models
    core = nm.emb_model_query.gnn_core.eval()
    qb = OP.query_batch(queries)
    feats = {"union_node": torch.zeros(qb["num_nodes"]["union_node"], 1)}
    ref = OM.gnn_core_hetero(cpu_sd(nm), "emb_model_query.gnn_core", feats, qb["edge_index"], ("union_node",),
                             OP.QUERY_EDGE_TYPES, 8)
    got = core({"union_node": feats["union_node"].to(DEV)},
               {et: torch.as_tensor(ei).to(DEV) for et, ei in qb["edge_index"].items()})
    assert_logits_close("BaseGNNCore.forward [union_node]", got["union_node"], ref["union_node"])


def test_gossip_core_forward_for_one_query(models):
    nm, gm, queries = models
    core = gm.emb_model.gnn_core.eval()
    n, edges = golden_graphs(max_n=41)[5]
    ei = np.array(sorted(edges)).T
    g = torch.Generator().manual_seed(3)
    x = torch.rand(n, 1, generator=g) * 10
    qe = nm.get_query_emb().detach()[7:8]
    # the oracle's gossip_single_query up to post_mp: rebuild emb the same way
    sd = cpu_sd(gm)
    xx = OM._lin(sd, "emb_model.gnn_core.pre_mp.0", x)
    xx = torch.cat((qe.cpu().expand(n, -1), xx), dim=-1)
    e2, dirw = OP.gossip_edge_index(n, ei)
    e2, dirw = torch.as_tensor(e2).long(), torch.as_tensor(dirw)
    emb = xx
    for l in range(2):
        key = f"emb_model.gnn_core.convs.{l}"
        gate = OM.gossip_gate(sd, key, qe.cpu())
        msg = OM._lin(sd, key + ".lin_com", xx[e2[0]])
        msg[dirw] *= gate
        msg[~dirw] *= 1 - gate
        aggr = torch.zeros(n, 64).index_add_(0, e2[1], msg)
        xx = torch.relu(OM._lin(sd, key + ".lin_update", torch.cat((aggr, xx), dim=-1)))
        emb = torch.cat((emb, xx), dim=1)
    got = core(x.to(DEV), torch.as_tensor(ei).to(DEV), query_emb=qe)
    assert got.shape == (n, 64 * 4)
    assert_logits_close("BaseGNNCore.forward [gossip, one query]", got, emb)
    with pytest.raises(AssertionError):
        core(x.to(DEV), torch.as_tensor(ei).to(DEV))
