#!/usr/bin/env python3
"""Does the wave-form gossip kernel write every output element exactly?  (developer tool)"""
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
v = dict(g1=g1, p=r(Q, 64), z=r(Q, 64), zp=r(Q, 64), r=r(64), t=r(64), u=r(64), tp=r(64), d1=r(64),
         b3=r(64), b5=r(256), w7=r(256), b7=0.3)
W = dict(w1=r(64, 128), wp=r(64, 128), w3=r(64, 64), w5=r(256, 64))
v["wstream"], v["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(W[k].contiguous()) for k in ("w1", "wp", "w3", "w5")])
a = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v, batch.work_queue, tile_perm=batch.tile_perm)
print("queue after block form:", batch.work_queue.tolist())
for rep in range(3):
    out = torch.full((N, Q), -12345.0, device=dev)
    b = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, N, Q, v, batch.work_queue, wave_form=True, out=out)
    torch.cuda.synchronize()
    left = (b == -12345.0)
    d = (a - b).abs()
    d[left] = 0
    bad = (d > 1e-3).nonzero().cpu().numpy()
    print(f"rep {rep}: unwritten {int(left.sum())} of {N * Q}; wrong among written {len(bad)}; max {float(d.max()):.3e}; queue {batch.work_queue.tolist()}")
    if int(left.sum()):
        lw = left.nonzero().cpu().numpy()
        print("   unwritten groups:", sorted(set((lw[:, 0] // 16).tolist()))[:20], "q:", sorted(set(lw[:, 1].tolist())))
    if len(bad):
        print("   wrong groups:", sorted(set((bad[:, 0] // 16).tolist()))[:20], "q:", sorted(set(bad[:, 1].tolist())))
    import collections
    per = collections.defaultdict(list)
    for n_, q_ in bad:
        per[(int(n_) // 16, int(q_))].append(int(n_) % 16)
    for k_ in list(per)[:12]:
        print("   group, q", k_, "lanes", sorted(per[k_]), "deg", [int((batch.rowptr[k_[0] * 16 + l + 1] - batch.rowptr[k_[0] * 16 + l])) for l in sorted(per[k_])],
              "err", [round(float(d[k_[0] * 16 + l, k_[1]]), 3) for l in sorted(per[k_])])
