#!/usr/bin/env python3
"""Which held-out graphs carry the squared error of a main.py run (reads <output_dir>/graphlet_{count,truth}_<ds>.csv)."""
import sys
import numpy as np
import pandas as pd

out, ds = sys.argv[1], sys.argv[2]
pred = pd.read_csv(f"{out}/graphlet_count_{ds}.csv", index_col=0).to_numpy(dtype=np.float64)
truth = pd.read_csv(f"{out}/graphlet_truth_{ds}.csv", index_col=0).to_numpy(dtype=np.float64)
groups = {"size 3": range(0, 2), "size 4": range(2, 8), "size 5": range(8, 29)}
for name, cols in groups.items():
    cols = list(cols)
    se = ((pred[:, cols] - truth[:, cols]) ** 2)
    var = truth[:, cols].var(axis=0)
    nm = se.mean(axis=0) / var                       # the reference's metric per query, then its mean over the group
    per_graph = (se / var).mean(axis=1)              # a graph's contribution to the group's norm-MSE x number of graphs
    order = np.argsort(per_graph)[::-1]
    tot = per_graph.sum()
    print(f"{name}: norm-MSE {nm.mean():.4g}; share of the top 1 / 5 / 20 of {len(per_graph)} graphs: "
          f"{per_graph[order[:1]].sum() / tot:.3f} / {per_graph[order[:5]].sum() / tot:.3f} / {per_graph[order[:20]].sum() / tot:.3f}; "
          f"without the top 5: {np.delete(per_graph, order[:5]).sum() / len(per_graph):.4g}; "
          f"top graph: truth {truth[order[0], cols].max():.3g} pred {pred[order[0], cols].max():.3g}")
