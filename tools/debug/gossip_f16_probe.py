#!/usr/bin/env python3
"""Localise differences between the fp16 three-product gossip kernel and the bf16x6 one: same operands, one weight
block zeroed / simplified at a time (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from desco_amd import ops
from desco_amd.batch import GossipBatch
from desco_amd.graphs import GraphSet
from helpers import golden_graphs

dev = "cuda"
torch.manual_seed(0)
graphs = golden_graphs(max_n=60)
gs = GraphSet.from_edge_lists(graphs)
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


def run(Wm, v):
    v6 = dict(v)
    for k in Wm:
        v6[k + "s"] = ops.split_bf16_planes(Wm[k].contiguous())
    a = ops.gossip_fused(scal, batch.rowptr, batch.col, N, Q, v6, tile_perm=batch.tile_perm)
    v16 = dict(v)
    v16["wstream"], v16["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(Wm[k].contiguous()) for k in ("w1", "wp", "w3", "w5")])
    b = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, tile_perm=batch.tile_perm)
    d = (a - b).abs()
    return d.max().item(), (a - batch.x).abs().max().item(), int(d.argmax() // Q), int(d.argmax() % Q)


def variant(name, **chg):
    Wm = {k: W[k].clone() for k in W}
    v = dict(base)
    for k, f in chg.items():
        if k in Wm:
            Wm[k] = f(Wm[k])
        else:
            v[k] = f(v[k])
    print(f"{name:40s} maxdiff {run(Wm, v)}", flush=True)


variant("baseline")
variant("w1[:, :64] = 0 (no hh)", w1=lambda w: torch.cat([w[:, :64] * 0, w[:, 64:]], 1))
variant("w1[:, 64:] = 0 (no h1 in layer 1)", w1=lambda w: torch.cat([w[:, :64], w[:, 64:] * 0], 1))
variant("w1 = 0", w1=lambda w: w * 0)
variant("wp[:, 64:] = 0 (no h2 in y1)", wp=lambda w: torch.cat([w[:, :64], w[:, 64:] * 0], 1))
variant("wp[:, :64] = 0 (no h1 in y1)", wp=lambda w: torch.cat([w[:, :64] * 0, w[:, 64:]], 1))
variant("wp = 0", wp=lambda w: w * 0)
variant("w3 = 0", w3=lambda w: w * 0)
variant("w5 = 0", w5=lambda w: w * 0)
variant("w3 = I", w3=lambda w: torch.eye(64, device=dev))
variant("w5 rows 64.. = 0", w5=lambda w: torch.cat([w[:64], w[64:] * 0]))
variant("u = tp = 0", u=lambda t: t * 0, tp=lambda t: t * 0)

# which nodes differ?
Wm = {k: W[k].clone() for k in W}
Wm["wp"] = torch.cat([Wm["wp"][:, :64], Wm["wp"][:, 64:] * 0], 1)
v6 = dict(base)
for k in Wm:
    v6[k + "s"] = ops.split_bf16_planes(Wm[k].contiguous())
for tp_ in (batch.tile_perm, None):
    a = ops.gossip_fused(scal, batch.rowptr, batch.col, N, Q, v6, tile_perm=tp_)
    v16 = dict(base)
    v16["wstream"], v16["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(Wm[k].contiguous()) for k in ("w1", "wp", "w3", "w5")])
    b = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, tile_perm=tp_)
    b2 = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, tile_perm=tp_)
    d = (a - b).abs().amax(1).cpu().numpy()
    bad = np.nonzero(d > 1e-3)[0]
    deg = np.diff(gs.rowptr)
    reps = [ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v16, batch.work_queue, tile_perm=tp_) for _ in range(6)]
    rd = max((r_ - b).abs().max().item() for r_ in reps)
    print("tile_perm" if tp_ is not None else "no tile_perm", "N", N, "bad nodes", len(bad), "repeatable", bool(torch.equal(b, b2)), "max run-to-run diff over 6 more runs", rd)
    print(" bad idx % 128:", (bad % 128)[:40], "deg:", deg[bad][:40])
    if tp_ is not None:
        perm = tp_.cpu().numpy().reshape(-1, 128)
        slot_of = np.zeros_like(perm)
        for t in range(perm.shape[0]):
            slot_of[t, perm[t]] = np.arange(128)
        print(" slots of bad nodes:", [int(slot_of[i // 128, i % 128]) for i in bad[:40]])

    big = torch.stack([(r_ > 5e8).sum() for r_ in reps + [b, b2]]).cpu().numpy()
    big2 = torch.stack([(r_ > 1.5e9).sum() for r_ in reps + [b, b2]]).cpu().numpy()
    print(" entries flagged (early read of own h1 rows != read after a barrier):", big - big2, " hh rows:", big2)
    bb = reps[0]
    idx = torch.nonzero(bb > 5e8).cpu().numpy()[:12]
    print("  flagged (node, query):", idx.tolist())
