#!/usr/bin/env python3
"""Bitwise comparison of the variants of gossip_f16_var.hip (build_variants.sh) on one GPU box (developer tool).

Every variant computes the same IEEE arithmetic (packed and scalar fp32 FMA round identically; contraction is the same
in all of them up to the compiler's choices, which are deterministic), so a variant whose output CHANGES FROM LAUNCH TO
LAUNCH is wrong by itself, and the number of (node, query) results that differ from the no-packed reference is the
size of the effect.  Prints one line per variant:  name | results differing from the reference (first launch) |
results differing between launches (worst of `runs`) | ms per launch.
usage: python tools/debug/gf16_hazard/probe_variants.py [--nodes N] [--runs R] [--only a,b,c] [--tile-perm]"""
import argparse
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from desco_amd import ops

HERE = os.path.dirname(os.path.abspath(__file__))
vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float


def load(path):
    """the launcher of a variant as fn(common args..., out, tile_perm, queue, stream); the old_* builds (commit 0d06b19)
    have the entry point of that commit, without a tile order"""
    h = ctypes.CDLL(path)
    if os.path.basename(path).startswith("libgf16_old_"):
        old = h.desco_gossip_wave_f16x3_f32
        old.restype = ctypes.c_int
        old.argtypes = [vp, vp, vp, i64, i32] + [vp] * 14 + [f32, vp, vp, vp]
        return lambda *a: old(*a[:-3], a[-2], a[-1])
    fn = h.desco_gossip_fused_f16x3_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [vp, vp, vp, i64, i32] + [vp] * 14 + [f32, vp, vp, vp, vp]
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=400_000)
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--only", default="")
    ap.add_argument("--tile-perm", action="store_true")
    ap.add_argument("--deg", type=float, default=2.1, help="mean degree of the random forest-plus-rings graph")
    ap.add_argument("--diagnose", default="", help="bad:good -- where the results of variant `bad` differ from `good` by > 1e-3")
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    N, Q = a.nodes, 29
    # COX2-like sparse graphs: a random forest over blocks of 41 nodes plus a few ring-closing edges
    g = torch.Generator().manual_seed(1)
    blk = 41
    ids = torch.arange(N)
    par = (ids // blk) * blk + (torch.rand(N, generator=g) * (ids % blk).clamp(min=1)).long()
    keep = (ids % blk) != 0
    src, dst = ids[keep], par[keep]
    extra = int(N * max(a.deg - 2.0, 0.0) / 2)
    es = torch.randint(0, N, (extra,), generator=g)
    ed = (es // blk) * blk + torch.randint(0, blk, (extra,), generator=g)
    ok = (es != ed) & (ed < N)
    src, dst = torch.cat([src, es[ok]]), torch.cat([dst, ed[ok]])
    lo, hi = torch.minimum(src, dst), torch.maximum(src, dst)
    und = torch.unique(lo * N + hi)
    lo, hi = und // N, und % N
    s2, d2 = torch.cat([lo, hi]), torch.cat([hi, lo])
    order = torch.argsort(s2 * N + d2)
    s2, d2 = s2[order], d2[order]
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(s2, minlength=N), 0)
    rowptr = rowptr.to(torch.int32).to(dev)
    col = d2.to(torch.int32).to(dev)
    x = (torch.rand(N, Q) * 30).to(dev)
    g0 = torch.rand(Q, device=dev) * 0.8 + 0.1
    g1 = torch.rand(Q, device=dev) * 0.8 + 0.1
    scal = ops.gossip_scalars(x, rowptr, col, g0, g1)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.2).contiguous()
    v = dict(g1=g1, p=r(Q, 64), z=r(Q, 64), zp=r(Q, 64), r=r(64), t=r(64), u=r(64), tp=r(64), d1=r(64), b3=r(64),
             b5=r(256), w7=r(256))
    W = dict(w1=r(64, 128), wp=r(64, 128), w3=r(64, 64), w5=r(256, 64))
    v["wstream"], v["winv"] = ops.gossip_f16_stream(*[ops.split_f16_planes(W[k]) for k in ("w1", "wp", "w3", "w5")])
    tperm = ops.gossip_tile_order(rowptr, N) if a.tile_perm else None
    queue = torch.zeros(2, dtype=torch.int64, device=dev)
    names = ("g1", "p", "z", "zp", "r", "t", "u", "tp", "d1", "wstream", "winv", "b3", "b5", "w7")
    ptrs = [v[n].data_ptr() for n in names]
    torch.cuda.synchronize()

    def launch(fn, out):
        rc = fn(scal.data_ptr(), rowptr.data_ptr(), col.data_ptr(), N, Q, *ptrs, 0.3, out.data_ptr(),
                None if tperm is None else tperm.data_ptr(), queue.data_ptr(), None)
        if rc:
            raise RuntimeError(f"launch failed rc={rc}")

    libs = sorted(glob.glob(os.path.join(HERE, "_build", "libgf16_*.so")))
    libs = {os.path.basename(p)[len("libgf16_"):-3]: p for p in libs}
    order_ = ["nopk"] + [n for n in libs if n != "nopk"]
    if a.only:
        order_ = ["nopk"] + [n for n in a.only.split(",") if n != "nopk"]
    ref = None
    deg = (rowptr[1:] - rowptr[:-1]).float()
    if a.diagnose:
        bad_n, good_n = a.diagnose.split(":")
        good = torch.empty(N, Q, device=dev)
        launch(load(libs[good_n]), good)
        torch.cuda.synchronize()
        degi = deg.long()
        gmax = torch.nn.functional.pad(degi, (0, (-N) % 16)).view(-1, 16).amax(1).repeat_interleave(16)[:N]
        for rep in range(3):
            bad = torch.empty(N, Q, device=dev)
            launch(load(libs[bad_n]), bad)
            torch.cuda.synchronize()
            wrong = (bad - good).abs() > 1e-3
            nz = wrong.nonzero()
            print(f"--- launch {rep}: {int(wrong.sum())} wrong results in {int(wrong.any(1).sum())} nodes, "
                  f"{int(wrong.view(-1).numel())} total")
            print("by query      :", wrong.sum(0).tolist())
            print("by node % 16  :", torch.bincount(nz[:, 0] % 16, minlength=16).tolist())
            print("by degree     :", torch.bincount(degi[nz[:, 0]], minlength=13).tolist(), " (all nodes:",
                  torch.bincount(degi, minlength=13).tolist(), ")")
            print("by group max degree - own degree:", torch.bincount(gmax[nz[:, 0]] - degi[nz[:, 0]], minlength=8).tolist())
            # events: (group, query) cells and how many of their 16 nodes are wrong
            cell = (nz[:, 0] // 16) * Q + nz[:, 1]
            per_cell = torch.bincount(torch.unique(cell, return_counts=True)[1], minlength=17)
            print("wrong nodes per (16-node group, query) event:", per_cell.tolist())
            grp_of_blockwave = (nz[:, 0] // 16)
            print("group index mod 8 / mod 32:", torch.bincount(grp_of_blockwave % 8, minlength=8).tolist(),
                  torch.bincount(grp_of_blockwave % 32, minlength=32).tolist())
            d = (bad - good)[wrong]
            print("error magnitude quantiles:", [round(float(v), 4) for v in torch.quantile(d.abs().float().cpu()[:1000000], torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0]))])
            for i in range(min(6, nz.shape[0])):
                n_, q_ = int(nz[i * 97 % nz.shape[0], 0]), int(nz[i * 97 % nz.shape[0], 1])
                print(f"   node {n_} (deg {int(degi[n_])}, group max {int(gmax[n_])}) query {q_}: bad {float(bad[n_, q_]):.5f} good {float(good[n_, q_]):.5f} x {float(x[n_, q_]):.5f}")
        return
    print(f"nodes {N}  queries {Q}  edges {col.numel()}  mean degree {deg.mean().item():.2f}  max {int(deg.max())}  "
          f"tile_perm {'on' if a.tile_perm else 'off'}  results per launch {N * Q}", flush=True)
    for name in order_:
        fn = load(libs[name])
        outs = []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o = torch.empty(N, Q, device=dev)
        launch(fn, o)                                   # warm-up (attribute set, code object load)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(a.runs):
            o = torch.full((N, Q), float("nan"), device=dev)
            launch(fn, o)
            outs.append(o)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.runs
        first = outs[0].view(torch.int32)
        if ref is None:
            ref = first
        vs_ref = int((first != ref).sum())
        maxerr = float((outs[0] - ref.view(torch.float32)).abs().max())
        between = max(int((o.view(torch.int32) != first).sum()) for o in outs[1:])
        nan = int(torch.isnan(outs[0]).sum())
        print(f"{name:18s} differs from nopk: {vs_ref:9d} (max |d| {maxerr:9.3e})   differs between launches: {between:9d}   "
              f"nan {nan}   {ms:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
