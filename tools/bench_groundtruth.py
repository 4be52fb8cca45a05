#!/usr/bin/env python3
"""Canonical ground-truth counts: device enumerator vs the host OpenMP one --
developer tool.  Prints seconds and matched-subgraph rates; the results are compared bit for bit."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch

from desco_amd import synthetic
from desco_amd.groundtruth import canonical_counts, canonical_counts_device
from helpers import standard_queries


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn_1827")
    ap.add_argument("--replicas", type=int, default=1)
    args = ap.parse_args()
    _, queries = standard_queries()
    gs = synthetic.WORKLOADS[args.workload]().replicate(args.replicas)
    canonical_counts_device(gs.subset(0, 2), queries)            # load the library, warm the context
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev = canonical_counts_device(gs, queries)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = canonical_counts(gs, queries, backend="host")
    t_host = time.perf_counter() - t0
    same = torch.equal(dev.cpu().double(), host)
    m = float(host.sum())
    print(f"{args.workload} x{args.replicas}: {gs.num_nodes} nodes, {gs.num_directed_edges // 2} edges, "
          f"{m:.3e} matched subgraphs (29 queries)")
    print(f"  device: {t_dev:.3f} s incl. upload ({m / t_dev:.3e} matches/s)   host ({os.cpu_count()} logical "
          f"cores, OpenMP): {t_host:.3f} s   speed-up {t_host / t_dev:.1f}x   identical: {same}")


if __name__ == "__main__":
    main()
