#!/usr/bin/env python3
"""bf16x6 GEMM (desco_gemm_bf16x6_f32) alone at the anchor shape of the COX2 x64 workload
(m = 1.23 M rows, k = 512, n = 576) and at the 64-wide post-MLP shapes; developer tool.
  DESCO_LIB=<other build> python tools/bench_gemm_split.py [--check]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from desco_amd import ops


def timeit(fn, iters=7, warmup=2):
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


def case(m, k, n, check, tag):
    torch.manual_seed(1)
    a = torch.randn(m, k, device="cuda")
    wt = torch.randn(k, n, device="cuda") / k ** 0.5
    b = torch.randn(n, device="cuda")
    out = torch.empty(m, n, device="cuda")
    w_nk = ops.split_bf16_planes(wt.t().contiguous())
    ms = timeit(lambda: ops.gemm_split(a, w_nk, b, act=ops.ACT_LEAKY, slope=0.1, out=out))
    fl = 2.0 * m * k * n
    msg = f"{tag} m={m} k={k} n={n}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF/s fp32-equivalent ({6 * fl / ms / 1e9 / 2500:.3f} of 2.5 PF x6)"
    if check:
        rows = torch.cat([torch.arange(0, 2048), torch.arange(m - 2048, m)]).cuda()
        ref = torch.nn.functional.leaky_relu(a[rows].double() @ wt.double() + b.double(), 0.1)
        msg += f" maxdiff {(out[rows].double() - ref).abs().max().item():.2e}"
    print(msg, flush=True)
    # the three-product fp16 form (csrc/gemm_f16x3.hip): row-scale pre-pass + GEMM, and the GEMM alone
    w16 = ops.split_f16_planes(wt.t().contiguous())
    rs = ops.row_scale_f16(a)
    ms_rs = timeit(lambda: ops.row_scale_f16(a))
    ms16 = timeit(lambda: ops.gemm_f16x3(a, w16, b, act=ops.ACT_LEAKY, slope=0.1, out=out, row_scale=rs))
    msg = (f"{tag} m={m} k={k} n={n}: f16x3 {ms16:.3f} ms + row scale {ms_rs:.3f} ms = {fl / (ms16 + ms_rs) / 1e9:.1f} TF/s "
           f"fp32-equivalent ({3 * fl / ms16 / 1e9 / 2500:.3f} of 2.5 PF x3 for the GEMM alone)")
    if check:
        msg += f" maxdiff {(out[rows].double() - ref).abs().max().item():.2e}"
    print(msg, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("DESCO_LIB", "default")))
    args = ap.parse_args()
    case(1_230_000, 512, 576, args.check, args.tag)
    if args.all:
        case(1_230_000, 576, 64, args.check, args.tag)
        case(4_000_000, 64, 256, args.check, args.tag)
        case(1_230_000, 576, 128, args.check, args.tag)


if __name__ == "__main__":
    main()
