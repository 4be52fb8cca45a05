#!/usr/bin/env python3
"""A/B of the SHMP stage: neighborhood-resident multi-layer kernel vs the layer-by-layer kernels.
    python tools/bench_resident.py [workload] [replicas]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import desco_amd.gnn_model as GM  # noqa: E402
from desco_amd import ops, synthetic  # noqa: E402
from desco_amd.batch import NeighborhoodBatch  # noqa: E402
from desco_amd.partition import build_partition_device  # noqa: E402
from helpers import make_models, standard_queries  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "syn_1827"
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = "cuda"
    nm, _ = make_models(seed=0, gains=(0.8, 1.2))
    qids, _ = standard_queries()
    nm = nm.to(dev)
    nm.set_queries(qids)
    gs = synthetic.WORKLOADS[wl]()
    if rep > 1:
        gs = gs.replicate(rep)
    part = build_partition_device(gs, 4, dev)
    batch = NeighborhoodBatch(part, dev)
    GM.RESIDENT_MIN_ROWS = int(os.environ.get("RES_MIN_ROWS", GM.RESIDENT_MIN_ROWS))
    plan = batch.resident_plan(GM.RESIDENT_MIN_ROWS)
    print(f"{wl} x{rep}: {part.num_neigh} neighborhoods, {part.num_rows} rows; min rows {GM.RESIDENT_MIN_ROWS}: "
          f"{plan['num_packs']} packs, {plan.get('rows', 0)} packed rows ({plan.get('tile_rows', 0)} with padding), "
          f"other neighborhoods {0 if plan['rest_index'] is None else len(plan['rest_index'])}", flush=True)
    out = {}
    for mode in (False, True):
        GM.RESIDENT_SHMP = mode
        with torch.no_grad():
            for _ in range(2):
                o = nm._logits(batch, exp2=False)
            torch.cuda.synchronize()
            ops.PROFILER.enabled = True
            ops.PROFILER.reset()
            t0 = time.perf_counter()
            for _ in range(3):
                o = nm._logits(batch, exp2=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            ops.PROFILER.enabled = False
        out[mode] = o
        print(f"resident={mode}: {dt * 1e3:.2f} ms per pass", flush=True)
        for k, v in sorted(ops.PROFILER.summary().items(), key=lambda kv: -kv[1]["ms"])[:6]:
            print(f"    {k:40s} {v['calls'] // 3:4d} calls {v['ms'] / 3:9.3f} ms  {v['flops'] / max(v['ms'], 1e-9) / 1e9:8.1f} TF/s")
    if os.environ.get("RES_PROF"):
        import ctypes
        from desco_amd import _lib
        buf = (ctypes.c_ulonglong * 32)()
        _lib.lib().desco_debug_resident_prof(buf, 1)
        names = ["prologue", "heads+closed", "w0 table", "-", "count steps", "wait A", "epilogue",
                 "wait B", "pool out", "bar+pool(1)"]
        tot0, tot1 = sum(buf[:12]), sum(buf[16:28])
        for i, nm_ in enumerate(names):
            print(f"  phase {nm_:14s} wave0 {100.0 * buf[i] / max(tot0, 1):5.1f} %   waves1-7 {100.0 * buf[16 + i] / max(tot1, 1):5.1f} %")
        print(f"  phase totals (cycles over all packs/passes): wave0 {tot0:.3e}, waves1-7 {tot1:.3e} (7 waves)")
    d = (out[True] - out[False]).abs().max().item()
    print(f"max |resident - layerwise| logits: {d:.3e} (max |logit| {out[False].abs().max().item():.3e})")


if __name__ == "__main__":
    main()
