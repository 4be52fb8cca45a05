#!/usr/bin/env python3
"""Per-block finish-time spread of shmp_layer16 launches (needs the -DSH16_TAIL build: DESCO_LIB=tools/debug/_ab/libTAIL.so).
   python tools/debug/tail_probe.py --workload syn_1827 --replicas 1"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from desco_amd import _lib, ops, synthetic
from desco_amd.pipeline import InferencePipeline
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import make_models, standard_queries  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="syn_1827")
    ap.add_argument("--replicas", type=int, default=1)
    ap.add_argument("--kernel", default="shmp", choices=["shmp", "gossip"])
    args = ap.parse_args()
    L = _lib.lib()
    probe = L.desco_debug_sh16_tail if args.kernel == "shmp" else L.desco_debug_gf_tail
    probe.restype = ctypes.c_int
    probe.argtypes = [ctypes.c_void_p, ctypes.c_int]
    nm, gm = make_models(seed=0, gains=(0.8, 1.2))
    qids, _ = standard_queries()
    nm, gm = nm.cuda(), gm.cuda()
    nm.set_queries(qids)
    gs = synthetic.WORKLOADS[args.workload]().replicate(args.replicas)
    pipe = InferencePipeline(nm, gm, gs, depth=4, device="cuda")
    pipe.run()
    target = "shmp_layer" if args.kernel == "shmp" else "gossip_fused"
    orig = getattr(ops, target)
    buf = np.zeros((1024, 2), np.uint64)
    rows_seen = []

    def probed(x, vrowptr, vcol, row0, num_rows, *a, **k):
        torch.cuda.synchronize()
        probe(None, 1)
        out = orig(x, vrowptr, vcol, row0, num_rows, *a, **k)
        torch.cuda.synchronize()
        probe(buf.ctypes.data, 0)
        if args.kernel == "gossip":
            num_rows = row0 * num_rows          # (scal, rowptr, col, num_nodes, num_q, ...): node x query rows
        if num_rows > 1_000_000:
            b = buf[buf[:, 1] > 0].astype(np.int64)
            t0 = b[:, 0].min()
            end = (b[:, 1] - t0) / 100.0        # microseconds
            rows_seen.append((num_rows, len(b), end.min(), np.median(end), end.mean(), end.max()))
        return out

    setattr(ops, target, probed)
    pipe.run()
    setattr(ops, target, orig)
    for r in rows_seen:
        print("rows %9d blocks %4d: block finish min %.0f median %.0f mean %.0f max %.0f us -> tail (max/mean) %.3f" % (*r, r[5] / r[4]))


if __name__ == "__main__":
    main()
