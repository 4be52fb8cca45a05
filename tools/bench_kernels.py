#!/usr/bin/env python3
"""Micro-benchmarks of single C-ABI kernels at the shapes the COX2 x64 workload produces
(developer tool; interleaved rounds in one process, median of N)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from desco_amd import ops

DEV = "cuda"


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def gemm_case(m, k1, k2, n, lda1=None, check=False):
    a1 = torch.randn(m, lda1 or k1, device=DEV)[:, :k1]
    a2 = torch.randn(m, k2, device=DEV) if k2 else None
    wt = torch.randn(k1 + k2, n, device=DEV) / (k1 + k2) ** 0.5
    b = torch.randn(n, device=DEV)
    out = torch.empty(m, n, device=DEV)
    ms = timeit(lambda: ops.gemm(a1, wt, b, a2=a2, act=ops.ACT_RELU, out=out))
    fl = 2.0 * m * (k1 + k2) * n
    by = 4.0 * (m * (k1 + k2) + m * n)
    msg = f"gemm m={m} k={k1}+{k2} n={n}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF/s  {by / ms / 1e6:.0f} GB/s"
    w_nk = ops.split_bf16_planes(wt.t().contiguous())
    ms2 = timeit(lambda: ops.gemm_split(a1, w_nk, b, a2=a2, act=ops.ACT_RELU, out=out))
    msg += f" | bf16x6: {ms2:.3f} ms {fl / ms2 / 1e9:.1f} TF/s(fp32-equiv)"
    if check:
        A = a1 if a2 is None else torch.cat([a1, a2], 1)
        ref64 = torch.relu(A[:4096].double() @ wt.double() + b.double())
        msg += f" maxdiff(bf16x6) {(out[:4096].double() - ref64).abs().max().item():.2e}"
        ops.gemm(a1, wt, b, a2=a2, act=ops.ACT_RELU, out=out)
        ref = torch.relu(A[:4096].double() @ wt.double() + b.double())
        msg += f"  maxdiff {(out[:4096].double() - ref).abs().max().item():.2e}"
    print(msg, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only-anchor", action="store_true", help="just the 576x576 case (for counter passes)")
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.only_anchor:
        gemm_case(750_000, 576, 0, 576)
        return
    gemm_case(5_240_000, 256, 64, 64, check=args.check)          # SHMP count rows
    gemm_case(750_000, 128, 64, 64, lda1=256, check=args.check)  # SHMP canonical rows
    gemm_case(750_000, 576, 0, 576, check=args.check)            # anchor
    gemm_case(4_000_000, 64, 64, 64, check=args.check)           # gossip h2 / post_mp.0
    gemm_case(4_000_000, 64, 0, 64, check=args.check)            # post_mp.3
    gemm_case(4_000_000, 64, 0, 256, check=args.check)           # post_mp.5


if __name__ == "__main__":
    main()
