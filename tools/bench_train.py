#!/usr/bin/env python3
"""Training-step throughput of the neighborhood model (config C3 shape: Syn_1827-shaped
neighborhoods, batch 512, fp32) and of the gossip model (batch 256 graphs) -- developer tool.
Forward and backward run on the C-ABI kernels via desco_amd.autograd; Adam is desco_adam_step_f32 (desco_amd.optim)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from desco_amd import synthetic
from desco_amd.batch import GossipBatch, NeighborhoodBatch
from desco_amd.data import STANDARD_QUERY_IDS
from desco_amd.partition import build_partition


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn_1827")
    ap.add_argument("--graphs", type=int, default=200)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--batch", type=int, default=512, help="neighborhoods per training step")
    ap.add_argument("--graph", action="store_true",
                    help="capture each batch's step (forward, backward, Adam) in a hipGraph and time replays "
                         "(what Trainer(graph_capture=True) does)")
    ap.add_argument("--profile", action="store_true", help="per-kernel HIP-event breakdown of the timed steps")
    ap.add_argument("--json", default=None, help="also write the neighborhood-step result (with the dominant "
                    "kernel's roofline when --profile) to this file")
    args = ap.parse_args()
    from desco_amd import autograd as AG
    AG.set_precision(args.precision)
    dev = torch.device("cuda", 0)
    nm, gm = bench.build_models(dev)
    nm.set_queries(STANDARD_QUERY_IDS)
    gs = synthetic.WORKLOADS[args.workload]()
    gs = gs.subset(0, min(args.graphs, gs.num_graphs))
    part = build_partition(gs, 4)
    g = torch.Generator().manual_seed(0)
    Q = len(STANDARD_QUERY_IDS)
    batches = []
    for b0 in range(0, part.num_neigh, args.batch):
        p = part.slice(b0, b0 + args.batch)
        y = torch.floor(torch.rand(p.num_neigh, Q, generator=g) ** 3 * 50)     # surrogate labels
        batches.append(NeighborhoodBatch(p, dev, y=y))
    batches = batches[:args.steps + 2]
    opt = nm.configure_optimizers()["optimizer"]
    def step(b):
        opt.zero_grad(set_to_none=True)
        loss = nm.training_step(b, 0)
        loss.backward()
        opt.step()
        return loss
    if args.graph:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        graphs = {}
        with torch.cuda.stream(side):
            for b in batches:
                opt.zero_grad(set_to_none=True)
                nm.train_forward(b, 0).backward()
                opt.step()
            for b in batches:
                opt.zero_grad(set_to_none=True)
                cg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cg, stream=side):
                    nm.train_forward(b, 0).backward()
                    opt.step()
                graphs[id(b)] = cg
        torch.cuda.current_stream().wait_stream(side)
        def step(b):            # noqa: F811
            graphs[id(b)].replay()
    for b in batches:           # first epoch untimed: builds the per-batch backward indices, warms caches
        step(b)
    torch.cuda.synchronize()
    from desco_amd import ops
    ops.PROFILER.enabled = args.profile
    ops.PROFILER.reset()
    t0 = time.perf_counter()
    n = rows = 0
    for b in batches[2:]:
        step(b)
        n += b.num_graphs
        rows += b.num_rows
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.PROFILER.enabled = False
    if args.profile:
        summ = ops.PROFILER.summary()
        tot = sum(v["ms"] for v in summ.values())
        print(f"profiled kernel time {tot / (len(batches) - 2):.1f} ms/step:")
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:12]:
            print(f"   {k:32s} {v['calls'] // (len(batches) - 2):4d} launches/step {v['ms'] / (len(batches) - 2):7.2f} ms/step")
    if args.json:
        import json
        res = {"metric": "neighborhoods/s (neighborhood-model training step: forward, backward, Adam)",
               "value": n / dt, "unit": "neighborhoods/s", "ms_per_step": 1e3 * dt / (len(batches) - 2),
               "rows_per_s": rows / dt, "steps": len(batches) - 2, "dtype": "f32" if args.precision == "fp32" else "bf16 products, fp32 accumulate",
               "data": "synthetic",
               "config": {"workload": f"{args.workload}-shaped synthetic, first {gs.num_graphs} graphs, batch {args.batch} "
                                      f"neighborhoods, 29 queries, surrogate labels, Adam (desco_adam_step_f32)",
                          "launch_mode": "hipGraph replay per batch" if args.graph else "eager launches",
                          "rows_per_step": rows / (len(batches) - 2)}}
        if args.profile:
            tot = sum(v["ms"] for v in summ.values())
            name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            peak = {"gemm_f32_kernel": bench.PEAK_F32_MFMA_TFLOPS, "linear_bwd_w_kernel": bench.PEAK_F32_MFMA_TFLOPS,
                    "gemm_tn_partial_kernel": bench.PEAK_F32_MFMA_TFLOPS,
                    "gemm_bf16_kernel": bench.PEAK_BF16_MFMA_TFLOPS}.get(name)
            if peak:
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                res["roofline"] = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                   "frac": ach / peak, "launches": d["calls"], "share_of_kernel_time": d["ms"] / tot,
                                   "traffic": None}
            else:
                ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                res["roofline"] = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": bench.PEAK_HBM_GBS,
                                   "unit": "GB/s", "frac": ach / bench.PEAK_HBM_GBS, "launches": d["calls"],
                                   "share_of_kernel_time": d["ms"] / tot, "traffic": None}
            res["kernels"] = {k: {"launches_per_step": v["calls"] // (len(batches) - 2),
                                  "ms_per_step": round(v["ms"] / (len(batches) - 2), 3)}
                              for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)
    mode = ", hipGraph replay" if args.graph else ""
    print(f"neighborhood training ({args.workload}-shaped, batch {args.batch}, {args.precision}{mode}): {len(batches) - 2} steps, "
          f"{n / dt:.0f} neighborhoods/s, {rows / dt / 1e6:.2f} M rows/s, {1e3 * dt / (len(batches) - 2):.1f} ms/step")
    # gossip
    x = torch.rand(gs.num_nodes, Q, generator=g) * 20
    yg = torch.floor(torch.rand(gs.num_nodes, Q, generator=g) * 25)
    gm.set_query_emb(nm.get_query_emb())
    gb = [GossipBatch(gs.subset(g0, min(g0 + 256, gs.num_graphs)), dev) for g0 in range(0, gs.num_graphs, 256)]
    off = 0
    for b in gb:
        b.x, b.y = x[off:off + b.num_nodes].to(dev), yg[off:off + b.num_nodes].to(dev)
        off += b.num_nodes
    gopt = gm.configure_optimizers()["optimizer"]
    def gstep(b):
        gopt.zero_grad(set_to_none=True)
        loss = gm.training_step(b, 0)
        loss.backward()
        gopt.step()
    gstep(gb[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nodes = 0
    for _ in range(3):
        for b in gb:
            gstep(b)
            nodes += b.num_nodes
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"gossip training (batch 256 graphs, 29 queries, fp32): {nodes / dt:.0f} nodes/s, "
          f"{1e3 * dt / (3 * len(gb)):.1f} ms/step")


if __name__ == "__main__":
    main()
