"""The N>1 path on the GPU: 2 ranks sharing the one GPU of the test box (DESCO_SHARE_GPU=1: both
ranks on cuda:0, gloo backend with host-staged collectives; the driver's multi-GPU runs use one
GPU per rank and RCCL).  The ranks are separate processes started by desco_amd.distributed.launch,
exactly as bench.py / main.py start them.

  * the REAL two-stage pipeline sharded over 2 ranks == the 1-rank pipeline, per graph, per node
  * one REAL training step: 2-rank bucketed + count-weighted gradients == 1-rank gradients of the
    loss over the union batch (neighborhood: mean loss; gossip: sum loss)
  * Trainer(strategy="ddp").fit / predict == the equivalent 1-rank run
  * bench.py --gpus 2 (weak and strong) starts its own ranks and reports n_gpus = 2
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from desco_amd import distributed as D  # noqa: E402
from desco_amd.graphs import GraphSet  # noqa: E402

import multirank_common as C  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_multirank_worker.py")
DEV = "cuda"


def _run2(mode, tmp_path):
    out = str(tmp_path / mode)
    env = dict(os.environ, DESCO_SHARE_GPU="1")
    rc = D.launch([WORKER, mode, out], 2, env=env, timeout=900)
    assert rc == 0, f"2-rank worker ({mode}) failed with exit code {rc}"
    return [torch.load(f"{out}.rank{r}", weights_only=False) for r in (0, 1)]


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_two_rank_pipeline_equals_one_rank(tmp_path):
    from desco_amd.pipeline import InferencePipeline
    r0, r1 = _run2("pipeline", tmp_path)
    nm, gm, qids, queries = C.models(DEV)
    gs = GraphSet.from_edge_lists(C.mixed_graphs())
    one = InferencePipeline(nm, gm, gs, depth=4, device=DEV, rank=0, world=1).run()
    (a0, b0), (a1, b1) = r0["range"], r1["range"]
    assert a0 == 0 and b0 == a1 and b1 == gs.num_graphs and 0 < b0 < gs.num_graphs     # a real split
    for k in ("graph_neigh_count", "graph_gossip_count", "neigh_count", "node_count", "x"):
        got, ref = r0[k], one[k].cpu()
        assert got.shape == ref.shape, k
        assert torch.isfinite(ref).all() and torch.isfinite(got).all(), k
        exact = torch.equal(got, ref)
        # a row's in-tile summation order depends on where its tile falls in the launch, so shards
        # agree to fp32 rounding (amplified by 2**logit), as slices of one launch do
        # (counts are 2**logit - 1: deviations in the log-space scale 1 + |count|)
        worst = float(((got - ref).abs() / (1.0 + ref.abs())).max())
        print(f"[multirank] {k}: bit-identical={exact}, worst deviation {worst:.2e} (relative to 1 + |count|)")
        assert worst < 1e-4, (k, worst)
    assert float(one["graph_gossip_count"].abs().max()) > 1e-3


def test_two_rank_pipeline_is_bit_identical_to_one_rank_in_chunk_mode(tmp_path):
    """SURVEY 8e, "N ranks == 1 rank": with the dataset cut into rank-count-independent chunks
    (InferencePipeline(chunks=...)) every chunk runs through the same launches whichever rank owns it, so
    the 2-rank result equals the 1-rank result bit for bit -- per graph, per node, per neighborhood."""
    from desco_amd.pipeline import InferencePipeline
    r0, r1 = _run2("pipeline_chunks", tmp_path)
    nm, gm, qids, queries = C.models(DEV)
    gs = GraphSet.from_edge_lists(C.mixed_graphs())
    pipe = InferencePipeline(nm, gm, gs, depth=4, device=DEV, rank=0, world=1, chunks=C.CHUNKS)
    assert len(pipe.neigh_batches) == C.CHUNKS
    one = pipe.run()
    (a0, b0), (a1, b1) = r0["range"], r1["range"]
    assert a0 == 0 and b0 == a1 and b1 == gs.num_graphs and 0 < b0 < gs.num_graphs
    for k in ("graph_neigh_count", "graph_gossip_count", "neigh_count", "node_count", "x"):
        assert torch.equal(r0[k], one[k].cpu()), k
    # and the chunked 1-rank run agrees with the un-chunked one to fp32 rounding
    plain = InferencePipeline(nm, gm, gs, depth=4, device=DEV, rank=0, world=1).run()
    for k in ("graph_gossip_count", "node_count"):
        worst = float(((one[k] - plain[k]).abs() / (1.0 + plain[k].abs())).max())
        assert worst < 1e-4, (k, worst)


def test_more_ranks_than_graphs(tmp_path):
    from desco_amd.pipeline import InferencePipeline
    r0, r1 = _run2("tiny", tmp_path)
    assert sorted([r0["range"], r1["range"]]) in ([(0, 0), (0, 1)], [(0, 1), (1, 1)])
    nm, gm, qids, queries = C.models(DEV)
    gs = GraphSet.from_edge_lists(C.mixed_graphs()[:1])
    one = InferencePipeline(nm, gm, gs, depth=4, device=DEV, rank=0, world=1).run()
    for k in ("graph_gossip_count", "node_count", "neigh_count"):
        torch.testing.assert_close(r0[k], one[k].cpu(), rtol=1e-5, atol=1e-5)


def test_two_rank_gradients_equal_one_rank_union_batch(tmp_path):
    from desco_amd.batch import GossipBatch, NeighborhoodBatch
    from desco_amd.partition import build_partition
    r0, r1 = _run2("grads", tmp_path)
    nm, gm, qids, queries = C.models(DEV)
    gs = GraphSet.from_edge_lists(C.mixed_graphs()[:C.TRAIN_GRAPHS])
    part = build_partition(gs, 4)
    y = C.neigh_labels(part.num_neigh, len(queries))
    nm.zero_grad()
    nm.train_forward(NeighborhoodBatch(part, DEV, y=y), 0).backward()         # mean over the union
    worst = 0.0
    for n, p in nm.named_parameters():
        ref = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        for r in (r0, r1):                                                      # identical on both ranks
            g = r["neigh"][n]
            if float(ref.abs().max()) == 0.0:
                assert float(g.abs().max()) == 0.0, n
                continue
            e = _rel(g, ref)
            worst = max(worst, e)
            assert e < 1e-3, (n, e)
        assert torch.equal(r0["neigh"][n], r1["neigh"][n]), n
    print(f"[multirank] neighborhood: worst relative gradient deviation 2-rank vs 1-rank {worst:.2e}")
    x, yg = C.gossip_inputs(gs.num_nodes, len(queries))
    gm.set_query_emb(nm.get_query_emb())
    gm.zero_grad()
    gm.train_forward(GossipBatch(gs, DEV, x=x, y=yg), 0).backward()            # sum over the union
    worst = 0.0
    for n, p in gm.named_parameters():
        ref = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        g = r0["gossip"][n]
        if float(ref.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, n
            continue
        e = _rel(g, ref)
        worst = max(worst, e)
        assert e < 1e-3, (n, e)
        assert torch.equal(g, r1["gossip"][n]), n
    print(f"[multirank] gossip: worst relative gradient deviation 2-rank vs 1-rank {worst:.2e}")


def test_two_rank_trainer_fit_equals_one_rank_union_steps(tmp_path):
    """Trainer(strategy="ddp"): step k consumes batches (2k, 2k+1), one per rank, count-weighted;
    the 1-rank equivalent steps on the union of the two (batch size 2 x NEIGH_BATCH)."""
    from desco_amd.lightning_data import LightningDataLoader
    from desco_amd.workload import Workload
    r0, r1 = _run2("fit", tmp_path)
    assert r0["history"] == r1["history"] and r0["best"] == r1["best"] and os.path.exists(r0["best"])
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k                 # replicas stay in sync
    torch.testing.assert_close(r0["pred"], r1["pred"], rtol=0, atol=0)
    nm, gm, qids, queries = C.models(DEV)
    gs = GraphSet.from_edge_lists(C.mixed_graphs()[:C.TRAIN_GRAPHS])
    w = Workload(gs, root=None)
    w.generate_pipeline_datasets(depth_neigh=4)
    nd = w.neighborhood_dataset
    nd.y = C.neigh_labels(len(nd), len(queries))
    assert (len(nd) + C.NEIGH_BATCH - 1) // C.NEIGH_BATCH % 2 == 1, "want an odd number of batches"
    opt = nm.configure_optimizers()["optimizer"]
    for _ in range(2):
        for b in nd.batches(2 * C.NEIGH_BATCH, DEV):
            opt.zero_grad(set_to_none=True)
            nm.training_step(b, 0).backward()
            opt.step()
    worst = 0.0
    for k, v in nm.state_dict().items():
        d = float((r0["params"][k] - v.cpu()).abs().max())
        worst = max(worst, d)
    # Adam's m / sqrt(v) amplifies round-off where gradients are tiny: parameters move by up to
    # lr = 1e-4 per step, so 4 steps agree to a fraction of that
    print(f"[multirank] fit: worst |param(2 ranks) - param(1 rank, union batches)| = {worst:.2e}")
    assert worst < 5e-5
    nm.eval()
    pred = torch.cat([nm.predict_step(b, 0) for b in nd.batches(C.NEIGH_BATCH, DEV)]).cpu()
    assert r0["pred"].shape == pred.shape
    x, yg = C.gossip_inputs(gs.num_nodes, len(queries))
    gd = w.gossip_dataset
    gd.x, gd.y = x, yg
    gm.set_query_emb(nm.get_query_emb())
    opt = gm.configure_optimizers()["optimizer"]
    gm.train()
    for g0 in range(0, len(gd), 2 * C.GOSSIP_BATCH):
        opt.zero_grad(set_to_none=True)
        gm.training_step(gd.batch(g0, min(g0 + 2 * C.GOSSIP_BATCH, len(gd)), DEV), 0).backward()
        opt.step()
    worst = max(float((r0["gparams"][k] - v.cpu()).abs().max()) for k, v in gm.state_dict().items())
    print(f"[multirank] gossip fit: worst parameter deviation {worst:.2e}")
    assert worst < 2e-3          # lr 1e-3, 2 steps; query embeddings differ by the 1e-5 of the stage before


def test_two_rank_replayed_ddp_equals_eager_ddp_bit_for_bit(tmp_path):
    """Trainer(strategy="ddp", graph_capture=True): epochs >= 1 run trainer.DDPReplay -- graph A (zero, forward,
    backward, pack into the buckets), the bucket all-reduces, graph B (Adam on the bucket views).  After 3 epochs (2
    steps each; the second step's group is short: rank 1 has no batch and replays the zero fill alone) the parameters
    equal the eager hook-driven DDP run's bit for bit -- neighborhood model, and the gossip model at the reference's
    default dropout 0.01 (whose masks come from the (seed, step) pair a captured launch advances)."""
    r0, r1 = _run2("fit_replay", tmp_path)
    for r in (r0, r1):
        for stage in ("_params", "_gparams"):
            for k in r["eager" + stage]:
                assert torch.equal(r["eager" + stage][k], r["replay" + stage][k]), (stage, k)
        assert r["eager_history"] == r["replay_history"]
        assert r["eager_rng"] == r["replay_rng"] and r["eager_rng"][0] == 77 and r["eager_rng"][1] >= 3
    for k in r0["replay_params"]:
        assert torch.equal(r0["replay_params"][k], r1["replay_params"][k]), k          # replicas in sync
    # the run did train: parameters moved from their initial values
    nm, gm, *_ = C.models("cpu")
    moved = max(float((r0["replay_params"][k] - v).abs().max()) for k, v in nm.state_dict().items())
    assert moved > 1e-5


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_starts_its_own_ranks(scaling):
    env = dict(os.environ, DESCO_SHARE_GPU="1", DESCO_BENCH_GRAD_CHECK="1" if scaling == "weak" else "0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--replicas", "2", "--scaling", scaling, "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == scaling and r["value"] > 0
    # the run proves its own collective path: live backend, ranks reached by an all-reduce of ones,
    # and (nccl always; here on request, through the host-staged gloo path) the bucketed gradient all-reduce check
    col = r["collective"]
    assert col["ranks_seen"] == 2 and col["backend"] == "gloo" and col["shared_gpu_test_mode"] is True
    if scaling == "weak":
        gc = col["grad_allreduce_selftest"]
        assert gc["status"] == "ok" and gc["world"] == 2 and gc["worst_rel_grad_diff_vs_union_batch"] < 1e-4
    total = 467 * 2
    per = r["config"]["graphs_per_gpu"]
    assert (per == total) if scaling == "weak" else (0 < per < total)
