#!/usr/bin/env python3
"""Stand-alone CSR gather kernel (training path / un-fused form) on the benchmark's partition:
algorithmic HBM bytes per launch (X once + indices + the 4-slot aggregate) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desco_amd import ops, synthetic
from desco_amd.batch import NeighborhoodBatch
from desco_amd.partition import build_partition_device
from tools.bench_kernels import timeit

for wl, rep in (("cox2", 32), ("syn_1827", 1)):
    gs = synthetic.WORKLOADS[wl]().replicate(rep)
    b = NeighborhoodBatch(build_partition_device(gs, 4), "cuda")
    N, E = b.num_rows, b.vcol.numel()
    x = torch.randn(N, 64, device="cuda")
    out = torch.empty(N, 256, device="cuda")
    ms = timeit(lambda: ops.csr_gather_sum(x, b.vrowptr, b.vcol, N, 4, out=out))
    by = 256.0 * N + 4.0 * (E + 4 * N + 1) + 1024.0 * N
    ti = b.train_index()
    d = torch.randn(N * 4, 64, device="cuda")
    dx = torch.empty(N, 64, device="cuda")
    ms_t = timeit(lambda: ops.csr_gather_sum(d, ti["t_rowptr"], ti["t_col"], N, 1, out=dx))
    by_t = 256.0 * E + 4.0 * (E + N + 1) + 256.0 * N        # every referenced virtual row once
    print(f"{wl} x{rep}: N={N} E={E} | forward gather {ms:.3f} ms {by / ms / 1e6:.0f} GB/s ({by / ms / 8e7:.1f} % of 8 TB/s) | "
          f"transposed (backward) gather {ms_t:.3f} ms {by_t / ms_t / 1e6:.0f} GB/s", flush=True)
