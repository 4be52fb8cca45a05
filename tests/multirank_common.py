"""Inputs shared by tests/test_multirank_gpu.py (1-rank side) and tests/_multirank_worker.py (the
2-rank side): the same seeded models, graphs and labels in every process."""
import numpy as np
import torch

from helpers import golden_graphs, make_models, standard_queries


def mixed_graphs():
    """Small molecule-like, clique-union, G(n,m) and Syn_1827-shaped graphs (46 graphs)."""
    from desco_amd import synthetic
    g = golden_graphs(max_n=60)
    m = synthetic.msrc_imdb_mixed(3, 6).edge_lists()
    s = synthetic.syn_1827_shaped(60).edge_lists()
    return g + m + [s[i] for i in (20, 30, 40, 44, 48)]


def models(device, gains=(0.8, 1.2), seed=0, gossip_dropout=0.0):
    # (narrower weights than the molecule-sized tests: the dense graphs sum over more neighbours per
    #  row, and 2**logit must stay finite for the count comparison)
    nm, gm = make_models(seed=seed, gains=gains)
    if gossip_dropout:          # --gossip_dropout (reference default 0.01): same weights, dropout on in training mode
        gm.dropout = gm.emb_model.dropout = gm.emb_model.gnn_core.dropout = gossip_dropout
        gm.emb_model.post_mp[1].p = gossip_dropout
    qids, queries = standard_queries()
    nm, gm = nm.to(device), gm.to(device)
    nm.set_queries(qids)
    return nm, gm, qids, queries


def neigh_labels(num_neigh, num_q, seed=4):
    g = torch.Generator().manual_seed(seed)
    return torch.floor(torch.rand(num_neigh, num_q, generator=g) ** 3 * 40)


def gossip_inputs(num_nodes, num_q, seed=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(num_nodes, num_q, generator=g) * 20
    y = torch.floor(torch.rand(num_nodes, num_q, generator=g) * 25)
    return x, y


TRAIN_GRAPHS = 16          # graphs of the training checks
NEIGH_BATCH = 96           # neighborhoods per DataLoader batch there (ragged last batch)
GOSSIP_BATCH = 5           # graphs per gossip batch (16 graphs -> 4 batches, last one short)
CHUNKS = 4                 # placement-independent mode of InferencePipeline (rank-count-independent chunks)
