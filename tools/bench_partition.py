#!/usr/bin/env python3
"""Canonical-partition build time: host C++ (OpenMP) vs device builder (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from desco_amd import synthetic
from desco_amd.partition import build_partition, build_partition_device

for wl, rep in (("cox2", 64), ("msrc_imdb", 8), ("syn_1827", 2)):
    gs = synthetic.WORKLOADS[wl]().replicate(rep)
    t0 = time.perf_counter(); h = build_partition(gs, 4); th = time.perf_counter() - t0
    build_partition_device(gs.subset(0, 8), 4)          # warm-up (module load)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); d = build_partition_device(gs, 4); torch.cuda.synchronize(); td = time.perf_counter() - t0
    same = all((getattr(h, f) == getattr(d, f)).all() for f in ("count_ptr", "vrowptr", "vcol", "count_orig"))
    print(f"{wl} x{rep}: {gs.num_graphs} graphs, {gs.num_nodes} nodes, {h.num_neigh} neighborhoods, {h.num_rows} rows, "
          f"{h.num_edges} edges | host {th:.3f} s ({os.cpu_count()} logical cores) | device {td:.3f} s incl. upload+download | identical {same}", flush=True)
