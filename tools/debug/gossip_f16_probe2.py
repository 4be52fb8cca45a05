#!/usr/bin/env python3
"""Run-to-run repeatability of the fp16 gossip kernel with parts of the network switched off (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
Z = lambda w, lo, hi: torch.cat([w[:, :lo], w[:, lo:hi] * 0, w[:, hi:]], 1)


def rep(name, Wm, runs=12):
    v16 = dict(base)
    v16["wstream"], v16["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(Wm[k].contiguous()) for k in ("w1", "wp", "w3", "w5")])
    outs = [ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, tile_perm=None) for _ in range(runs)]
    ref = torch.stack(outs).median(0).values
    d = torch.stack([(o - ref).abs().amax(1) for o in outs])          # [runs, N]
    bad = torch.nonzero(d > 1e-4)
    print(f"{name:46s} runs differing from the median: {int((d.amax(1) > 1e-4).sum())}/{runs}; bad (run, node%128): "
          f"{[(int(a), int(b) % 128) for a, b in bad[:10]]}", flush=True)


full = {k: W[k].clone() for k in W}
rep("full", full)
a = dict(full); a["w1"] = W["w1"] * 0; a["wp"] = Z(W["wp"], 64, 128)
rep("only h1 -> y1 (block 2)", a)
b = dict(full); b["w1"] = Z(W["w1"], 64, 128); b["wp"] = Z(W["wp"], 0, 64)
rep("only hh -> h2 -> y1 (blocks 0, 3)", b)
c = dict(full); c["w1"] = Z(W["w1"], 0, 64); c["wp"] = Z(W["wp"], 0, 64)
rep("only h1 -> h2 -> y1 (blocks 1, 3)", c)
d = dict(full); d["wp"] = W["wp"] * 0
rep("wp = 0 (blocks 4..8 only matter)", d)
e = dict(full); e["w1"] = W["w1"] * 0; e["wp"] = Z(W["wp"], 0, 64)
rep("only const h2 -> y1 (block 3)", e)
