#!/usr/bin/env python3
"""Micro-benchmark of the fused SHMP layer kernel on synthetic CSR (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desco_amd import ops
from tools.bench_kernels import timeit
DEV = "cuda"

def case(n, deg01, tab, sm=2, S=4, local=True):
    g = torch.Generator().manual_seed(0)
    cnt = torch.zeros(n * S, dtype=torch.long)
    for s in range(sm):
        cnt[s::S] = deg01
    if tab:
        cnt[sm::S] = (torch.rand(n, generator=g) < 0.45).long()
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(cnt, 0)]).to(torch.int32)
    E = int(ptr[-1])
    dst = torch.repeat_interleave(torch.arange(n * S), cnt) // S
    col = (dst + torch.randint(-8, 9, (E,), generator=g)).clamp(0, n - 1) if local else torch.randint(0, n, (E,), generator=g)
    nb = n // 8
    if tab:   # table slot sources index the table rows [0, nb)
        is_tab = (torch.repeat_interleave(torch.arange(n * S), cnt) % S) >= sm
        col = torch.where(is_tab, (dst // 8).clamp(0, nb - 1), col)
    x = torch.randn(n, 64, device=DEV)
    wt = torch.randn((sm + 1) * 64, 64, device=DEV) / 12
    bias = torch.randn(64, device=DEV)
    out = torch.empty(n, 64, device=DEV)
    ytab = torch.randn(nb, 128, device=DEV) if tab else None
    ptr, col = ptr.to(DEV), col.to(torch.int32).to(DEV)
    fl = 2.0 * n * (sm + 1) * 4096
    for x6 in (False, True):
        if x6 and sm > 2:
            continue
        w = ops.split_bf16_planes(wt.t().contiguous()) if x6 else wt
        ms = timeit(lambda: ops.shmp_layer(x, ptr, col, 0, n, S, sm, w, bias, out, ytab=ytab, ytab_row0=0))
        print(f"shmp{'-x6' if x6 else '   '} n={n} sm={sm} deg/slot={deg01} table={tab} local={local}: {ms:.3f} ms "
              f"{fl/ms/1e9:.1f} TF/s  {512.0*n/ms/1e6:.0f} GB/s(x+out)", flush=True)

if __name__ == "__main__":
    n = 4_000_000
    case(n, 0, False)
    case(n, 1, False)
    case(n, 1, True)
    case(n, 2, True)
    case(n, 1, True, local=False)
    case(n, 0, False, sm=0)
    case(n, 1, False, sm=3)
