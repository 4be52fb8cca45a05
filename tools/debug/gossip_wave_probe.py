#!/usr/bin/env python3
"""Localise differences between the wave-autonomous fp16 gossip kernel and the block form: same operands, one weight
block zeroed at a time; prints where the largest differences sit (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from desco_amd import ops
from desco_amd.batch import GossipBatch
from desco_amd.graphs import GraphSet
from helpers import golden_graphs

dev = "cuda"
torch.manual_seed(0)
gs = GraphSet.from_edge_lists(golden_graphs(max_n=60))
Q = 29
x = torch.rand(gs.num_nodes, Q) * 30
batch = GossipBatch(gs, dev, x=x)
N = gs.num_nodes
g0 = torch.rand(Q, device=dev) * 0.8 + 0.1
g1 = torch.rand(Q, device=dev) * 0.8 + 0.1
scal = ops.gossip_scalars(batch.x, batch.rowptr, batch.col, g0, g1)
r = lambda *s: (torch.randn(*s, device=dev) * 0.2).contiguous()
base = dict(g1=g1, p=r(Q, 64), z=r(Q, 64), zp=r(Q, 64), r=r(64), t=r(64), u=r(64), tp=r(64), d1=r(64),
            b3=r(64), b5=r(256), w7=r(256), b7=0.3)
W = dict(w1=r(64, 128), wp=r(64, 128), w3=r(64, 64), w5=r(256, 64))
deg = (batch.rowptr[1:] - batch.rowptr[:-1]).cpu().numpy()
print("nodes", N, "max degree", deg.max(), "groups", (N + 15) // 16)


def run(Wm, v):
    v16 = dict(v)
    v16["wstream"], v16["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(Wm[k].contiguous()) for k in ("w1", "wp", "w3", "w5")])
    a = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, tile_perm=batch.tile_perm)
    b = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, wave_form=True)
    b2 = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, wave_form=True)
    d = (a - b).abs().cpu().numpy()
    bad = np.argwhere(d > 1e-3)
    return d.max(), float((b - b2).abs().max()), len(bad), sorted(set(bad[:, 0].tolist()))[:12], sorted(set(bad[:, 1].tolist()))[:12], \
        [int(deg[i]) for i in sorted(set(bad[:, 0].tolist()))[:12]]


def variant(name, **chg):
    Wm = {k: W[k].clone() for k in W}
    v = dict(base)
    for k, f in chg.items():
        if k in Wm:
            Wm[k] = f(Wm[k])
        else:
            v[k] = f(v[k])
    print(f"{name:36s} max {run(Wm, v)}", flush=True)


variant("baseline")
variant("w1[:, :64] = 0 (no hh)", w1=lambda w: torch.cat([w[:, :64] * 0, w[:, 64:]], 1))
variant("w1[:, 64:] = 0 (no h1 in layer 1)", w1=lambda w: torch.cat([w[:, :64], w[:, 64:] * 0], 1))
variant("w1 = 0", w1=lambda w: w * 0)
variant("wp[:, 64:] = 0 (no h2 in y1)", wp=lambda w: torch.cat([w[:, :64], w[:, 64:] * 0], 1))
variant("wp[:, :64] = 0 (no h1 in y1)", wp=lambda w: torch.cat([w[:, :64] * 0, w[:, 64:]], 1))
variant("wp = 0", wp=lambda w: w * 0)
variant("zp = 0", zp=lambda t: t * 0)
variant("w3 = 0", w3=lambda w: w * 0)
