"""Shared helpers for the parity tests."""
import argparse
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def neigh_args(**over):
    a = dict(layer_num=8, conv_type="SAGE", use_hetero=True, dropout=0.0, depth=4, lr=1e-4,
             weight_decay=0.0, use_tconv=True, hidden_dim=64, input_dim=1, batch_size=512)
    a.update(over)
    return argparse.Namespace(**a)


def gossip_args(**over):
    a = dict(layer_num=2, conv_type="GOSSIP", use_hetero=False, dropout=0.0, lr=1e-3,
             weight_decay=0.0, hidden_dim=64, batch_size=256)
    a.update(over)
    return argparse.Namespace(**a)


def make_models(seed=0, scale_bias=True, gains=(1.3, 1.4)):
    """Seeded NeighborhoodCountingModel + GossipCountingModel (reference key names).
    ``gains``: widening of the (neighborhood, gossip) weight matrices (dense Syn-shaped batches sum
    over many more neighbours per row and want a smaller one to stay O(1))."""
    from desco_amd.lightning_model import GossipCountingModel, NeighborhoodCountingModel
    torch.manual_seed(seed)
    nm = NeighborhoodCountingModel(1, 64, neigh_args()).to_hetero_old(True, True)
    gm = GossipCountingModel(1, 64, gossip_args(), emb_channels=64, input_pattern_emb=True)
    if scale_bias:
        # default nn.Linear init makes 8 relu layers collapse to ~constant outputs; widen the
        # weights so that parity tests see structure-dependent, O(1) values
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for m, gain in ((nm, gains[0]), (gm, gains[1])):
                for name, p in m.named_parameters():
                    if p.dim() == 2:
                        p.mul_(gain)
                    else:
                        p.add_(0.1 * torch.randn(p.shape, generator=g))
    return nm, gm


def golden_graphs(max_n=10 ** 9):
    with open(os.path.join(GOLDEN, "partition_golden.json")) as f:
        d = json.load(f)
    return [(g["n"], [tuple(e) for e in g["edges"]]) for g in d["graphs"] if g["n"] <= max_n]


def standard_queries():
    with open(os.path.join(GOLDEN, "queries.json")) as f:
        d = json.load(f)
    return d["query_ids"], [(q["n"], [tuple(e) for e in q["edges"]]) for q in d["queries"]]


def cpu_sd(model):
    return {k: v.detach().cpu().float() for k, v in model.state_dict().items()}


def report(name, got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    d = (got - ref).abs()
    rel = d / ref.abs().clamp_min(1e-6)
    print(f"[parity] {name}: max|d|={d.max().item():.3e} max rel={rel.max().item():.3e} "
          f"ref max={ref.abs().max().item():.3e}")
    return d.max().item()
