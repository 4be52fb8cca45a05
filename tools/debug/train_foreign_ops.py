#!/usr/bin/env python3
"""Which lines of this package launch kernels that are NOT this library's inside a training step (developer tool).

Runs a few steps of the neighborhood and of the gossip training leg under torch.profiler with python stacks and prints,
per leg, every device kernel whose name is not a desco:: kernel with its launches per step and the innermost frame of
desco_amd/ (or bench / torch.optim) that caused it.  The list is the to-do list of "training steps made only of this
library's kernels" (VERDICT r4 item 3); tools/check_pass_is_native.sh --train is the pass/fail form.
usage (GPU box): python tools/debug/train_foreign_ops.py [--steps 3]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile


def legs(device):
    import bench
    from desco_amd import autograd as AG, synthetic
    from desco_amd.batch import GossipBatch, NeighborhoodBatch
    from desco_amd.data import STANDARD_QUERY_IDS, graph_atlas_plus
    from desco_amd.groundtruth import canonical_counts
    from desco_amd.partition import build_partition_device
    nm, gm = bench.build_models(device, gains=(0.8, 1.2))
    nm.set_queries(STANDARD_QUERY_IDS)
    gs = synthetic.WORKLOADS["syn_1827"]().subset(0, 400)
    part = build_partition_device(gs, 4, device)
    queries = [graph_atlas_plus(i) for i in STANDARD_QUERY_IDS]
    truth = canonical_counts(gs, queries, backend="auto").float()
    y_all = truth[torch.from_numpy(part.indicator)]
    nb = [NeighborhoodBatch(part.slice(b0, b0 + 512), device, y=y_all[b0:b0 + 512]) for b0 in (0, 512, 1024)]
    opt_n = nm.configure_optimizers()["optimizer"]

    def step_n(b):
        opt_n.zero_grad(set_to_none=True)
        loss = nm.training_step(b, 0)
        AG.backward(loss)
        opt_n.step()

    gm.set_query_emb(nm.get_query_emb().detach())
    x = truth * 1.05
    gb, off = [], 0
    for g0 in (0, 128, 256):
        sub = gs.subset(g0, g0 + 128)
        gb.append(GossipBatch(sub, device, x=x[off:off + sub.num_nodes], y=truth[off:off + sub.num_nodes]))
        off += sub.num_nodes
    opt_g = gm.configure_optimizers()["optimizer"]

    def step_g(b):
        opt_g.zero_grad(set_to_none=True)
        loss = gm.training_step(b, 0)
        AG.backward(loss)
        opt_g.step()

    return {"neighborhood": (step_n, nb), "gossip": (step_g, gb)}


class _AtenLog(torch.utils._python_dispatch.TorchDispatchMode):
    """every aten op that reaches the dispatcher, with the innermost frame of this repo that called it (forward ops) or
    the autograd node that runs it (backward ops: the engine calls them from C++, the python stack is the Function's)"""

    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        import traceback
        name = str(func).replace("aten.", "")
        if name.split(".")[0] not in ("view", "_unsafe_view", "t", "transpose", "slice", "select", "expand", "as_strided",
                                      "unsqueeze", "squeeze", "detach", "alias", "permute", "reshape", "empty",
                                      "empty_like", "empty_strided", "new_empty", "_reshape_alias", "unbind", "split",
                                      "is_same_size", "stride", "sym_size", "lift_fresh", "_local_scalar_dense"):
            fr = [f for f in traceback.extract_stack() if ("desco_amd/" in f.filename or "bench.py" in f.filename)
                  and "train_foreign_ops" not in f.filename]
            where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(autograd engine)"
            self.ops[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, (step, batches) in legs(dev).items():
        for b in batches:
            step(b)
        with _AtenLog() as log:
            step(batches[0])
        print(f"=== {name}: aten ops that launch kernels, one step, by calling line")
        for (op, where), c in sorted(log.ops.items(), key=lambda kv: (kv[0][1], -kv[1])):
            print(f"  {c:4d}  {op:34s} {where}")
    for name, (step, batches) in legs(dev).items():
        for b in batches:
            step(b)                                       # warm-up: backward indices, caches
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for i in range(a.steps):
                step(batches[i % len(batches)])
            torch.cuda.synchronize()
        # map device kernels to the CPU op that launched them through the correlation id
        evs = prof.events()
        foreign = collections.Counter()
        ours = 0
        for e in evs:
            if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
                continue
            for k in e.kernels:
                if "desco" in k.name or k.name.split("(")[0].split("<")[0].strip().endswith("_kernel") and "at::" not in k.name:
                    ours += 1
                    continue
                frame = next((f for f in (e.stack or []) if "desco_amd/" in f or "bench.py" in f or "optim" in f), "?")
                frame = frame.replace(ROOT + "/", "")
                foreign[(k.name.split("<")[0].split("(")[0][:60], e.name, frame[:110])] += 1
        print(f"=== {name}: {ours / a.steps:.0f} launches of this library per step, "
              f"{sum(foreign.values()) / a.steps:.0f} foreign launches per step")
        for (kn, op, fr), c in sorted(foreign.items(), key=lambda kv: -kv[1]):
            print(f"  {c / a.steps:6.1f}/step  {op:32s} {kn:44s} {fr}")


if __name__ == "__main__":
    main()
