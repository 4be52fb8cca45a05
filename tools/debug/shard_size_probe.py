#!/usr/bin/env python3
"""Times the inference pass of syn_1827.subset(0, n) for a range of n and prints the per-kernel times of the slowest: the
strong-scaling sweep of round 5 hit one shard size (1581 graphs) that took 22.8 ms where its neighbours took 9 (developer
tool).  usage (GPU box): python tools/debug/shard_size_probe.py 1576 1588"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from desco_amd import ops, synthetic
from desco_amd.data import STANDARD_QUERY_IDS
from desco_amd.pipeline import InferencePipeline

lo, hi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
nm, gm = bench.build_models(dev, gains=(0.8, 1.2))
nm.set_queries(STANDARD_QUERY_IDS)
gm.set_query_emb(nm.get_query_emb().detach())
gs = synthetic.WORKLOADS["syn_1827"]()
res = {}
for n in range(lo, hi + 1):
    ps = InferencePipeline(nm, gm, gs.subset(0, n), depth=4, device=dev, rank=0, world=1)
    ps.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ps.run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    ops.profile_reset() if hasattr(ops, "profile_reset") else None
    res[n] = ms
    print(n, round(ms, 3), "rows", ps.partition.num_rows, "neigh", ps.partition.num_neigh, "nodes", ps.graphs.num_nodes,
          "blocks", len(ps.neigh_batches), len(ps.gossip_batches), flush=True)
    del ps
    torch.cuda.empty_cache()
worst = max(res, key=res.get)
print("slowest", worst, res[worst])
ps = InferencePipeline(nm, gm, gs.subset(0, worst), depth=4, device=dev, rank=0, world=1)
ps.run()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    ps.run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
