"""Bisect which part of a training step survives hipGraph capture: each stage in its own process."""
import subprocess, sys, os
STAGES = ["fwd", "fwd_bwd", "fwd_bwd_adam"]
if len(sys.argv) == 1:
    for s in STAGES:
        r = subprocess.run([sys.executable, __file__, s], capture_output=True, text=True)
        print(s, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
        if r.returncode != 0:
            print("\n".join(r.stderr.splitlines()[:12]))
    sys.exit(0)
stage = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from helpers import golden_graphs, make_models, standard_queries          # noqa
from desco_amd.graphs import GraphSet
from desco_amd.partition import build_partition
from desco_amd.batch import NeighborhoodBatch
DEV = "cuda:0"
qids, queries = standard_queries()
nm, _ = make_models(seed=2)
nm = nm.to(DEV); nm.set_queries(qids); nm.train()
part = build_partition(GraphSet.from_edge_lists(golden_graphs(max_n=41)[:12]), 4)
y = torch.floor(torch.rand(part.num_neigh, len(queries)) ** 3 * 40)
batch = NeighborhoodBatch(part, DEV, y=y)
opt = nm.configure_optimizers()["optimizer"]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):          # AccumulateGrad nodes must be born on the capture stream
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = nm.train_forward(batch, 0); loss.backward(); opt.step()
    del loss
torch.cuda.synchronize()
opt.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    loss = nm.train_forward(batch, 0)
    if stage != "fwd":
        loss.backward()
    if stage == "fwd_bwd_adam":
        opt.step()
g.replay(); torch.cuda.synchronize()
print("ok", float(loss))
