#!/usr/bin/env python3
"""After tools/train_convergence.sh: is a bad held-out metric the MODEL or the KERNELS?  (developer tool)
1. per-graph error of the neighborhood predictions main.py left in /tmp/desco_results (which test graphs carry it);
2. HIP logits of the TRAINED neighborhood model against the CPU oracle's, on the worst graphs and a few others.
usage (GPU box, same call as the training): python tools/debug/trained_model_parity.py /tmp/desco_ckn /tmp/desco_results"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pandas as pd
import torch

from desco_amd.batch import NeighborhoodBatch
from desco_amd.data import load_data
from desco_amd.graphs import GraphSet
from desco_amd.lightning_model import NeighborhoodCountingModel
from desco_amd.partition import build_partition
from helpers import cpu_sd, standard_queries
from oracle import model as OM
from oracle import partition as OP

ckdir, resdir = sys.argv[1], sys.argv[2]
ds = "Syn_1827_test"
pred = pd.read_csv(os.path.join(resdir, f"graphlet_count_{ds}.csv"), index_col=0).to_numpy()
truth = pd.read_csv(os.path.join(resdir, f"graphlet_truth_{ds}.csv"), index_col=0).to_numpy()
err = ((pred - truth) ** 2).sum(1)
order = np.argsort(-err)
print("graphs", len(err), "total squared error %.3e" % err.sum())
for g in order[:8]:
    print(f"  graph {g}: share of squared error {err[g] / err.sum():.4f}  truth max {truth[g].max():.3e}  pred max {pred[g].max():.3e}")
print("  median per-graph |pred - truth| / (1 + truth), worst query:", np.median((np.abs(pred - truth) / (1 + truth)).max(1)))
from desco_amd.analysis import mae, norm_mse
sizes = sorted({n for n, _ in standard_queries()[1]})
groupby = [[i for i, (n, _) in enumerate(standard_queries()[1]) if n == s] for s in sizes]
for drop in (0, 1, 5):
    keep = np.ones(len(err), bool)
    keep[order[:drop]] = False
    print(f"  norm-MSE by query size {sizes} without the {drop} worst graphs:",
          [float("%.4g" % v) for v in norm_mse(pred[keep], truth[keep], groupby)],
          " MAE:", [float("%.4g" % v) for v in mae(pred[keep], truth[keep], groupby)])

ck = sorted(glob.glob(os.path.join(ckdir, "*best.ckpt")))[-1]
nm = NeighborhoodCountingModel.load_from_checkpoint(ck).to("cuda")
qids, queries = standard_queries()
nm.set_queries(qids)
gs = load_data(ds, root_folder="/tmp/desco_data")
sel = list(order[:3]) + list(order[len(order) // 2:len(order) // 2 + 3])
for g in sel:
    sub = gs.subset(int(g), int(g) + 1)
    n = sub.num_nodes
    e = [(int(i), int(j)) for i in range(n) for j in sub.col[sub.rowptr[i]:sub.rowptr[i + 1]] if i < int(j)]
    part = build_partition(sub, 4)
    with torch.no_grad():
        got = nm._logits(NeighborhoodBatch(part, "cuda"), exp2=False).cpu()
    _, _, neighs = OP.neighborhood_dataset([(n, e)], 4)
    ref = OM.neighborhood_logits(cpu_sd(nm), OP.neighborhood_batch(neighs), OP.query_batch(queries), emulate_quirk=False)[0]
    d = ((got - ref).abs() / (1 + ref.abs())).max().item()
    print(f"graph {g}: nodes {n}, neighborhoods {part.num_neigh}, logits max {ref.max().item():.2f} (HIP {got.max().item():.2f}), "
          f"HIP vs oracle max |d| / (1 + |ref|) = {d:.2e}")
