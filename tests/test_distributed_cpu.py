"""World-size-2 gloo tests of the N>1 path (CPU): graph sharding + ordered gather reproduce the
single-process result; the gradient buckets (async all-reduce issued from autograd hooks) and the
count-weighted step packing give the single-process gradient of the union batch; ``Trainer.fit`` /
``Trainer.predict`` in "ddp" mode with an ODD number of batches end in the same state as the
equivalent single-process run."""
import os
import queue as _queue
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from desco_amd import distributed as D
from desco_amd.graphs import GraphSet
from helpers import golden_graphs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_stage(gs: GraphSet) -> torch.Tensor:
    """Stand-in for the per-graph result of the hot path (order-sensitive, graph-local)."""
    out = []
    for n, edges in gs.edge_lists():
        out.append([float(n), float(len(edges)), float(sum(a * 3 + b for a, b in edges))])
    return torch.tensor(out).reshape(-1, 3)


# ---- a tiny Lightning-shaped model + datamodule on CPU tensors ------------------------------------
class _Batch:
    def __init__(self, x, y):
        self.x, self.y, self.num_graphs = x, y, x.shape[0]

    def to(self, device):
        return self


class _Loader:
    def __init__(self, batches):
        self.b = batches

    def __iter__(self):
        return iter(self.b)

    def __len__(self):
        return len(self.b)


class _Data:
    def __init__(self, sizes, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.batches = [_Batch(torch.randn(n, 4, generator=g), torch.randn(n, 3, generator=g)) for n in sizes]

    def train_dataloader(self):
        return _Loader(self.batches)

    val_dataloader = test_dataloader = train_dataloader


class _Model(torch.nn.Module):
    """mean loss (like the neighborhood model) or sum loss (like the gossip model)."""

    def __init__(self, reduce="mean"):
        super().__init__()
        torch.manual_seed(1)
        self.net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
        self.unused = torch.nn.Linear(2, 2)      # never reached by backward (query-side anchor_mlp)
        self.reduce = reduce
        self.saved = []

    def _loss(self, b):
        d = (self.net(b.x) - b.y) ** 2
        return d.mean() if self.reduce == "mean" else d.sum()

    def training_step(self, b, i):
        return self._loss(b)

    train_forward = training_step

    def validation_step(self, b, i):
        return self._loss(b)

    test_step = validation_step

    def predict_step(self, b, i):
        return self.net(b.x)

    def configure_optimizers(self):
        opt = torch.optim.SGD(self.parameters(), lr=0.05)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=20)
        return {"optimizer": opt, "lr_scheduler": sched, "monitor": "val"}

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)


SIZES = [6, 6, 3]        # odd number of batches, ragged last one


def _single_process_reference(reduce, epochs):
    """What the 2-rank run must equal: one process stepping on the UNION of each group of 2 batches
    (mean loss over the union / sum loss over the union)."""
    m = _Model(reduce)
    data = _Data(SIZES)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    for _ in range(epochs):
        for k in range(0, len(SIZES), 2):
            bs = data.batches[k:k + 2]
            u = _Batch(torch.cat([b.x for b in bs]), torch.cat([b.y for b in bs]))
            opt.zero_grad()
            m._loss(u).backward()
            opt.step()
    return m


def _worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from desco_amd.trainer import ModelCheckpoint, Trainer
    D.init_from_env(torch.device("cpu"))
    try:
        res = {}
        gs = GraphSet.from_edge_lists(golden_graphs())
        shard, (lo, hi) = D.shard_graphs(gs, rank, world)
        res["full"] = D.gather_rows(_fake_stage(shard), dst=0)
        res["full_all"] = D.allgather_rows(_fake_stage(shard))
        res["range"] = (lo, hi)
        # agree=True: rank 0 decides the cuts.  Rank 1 is given other cost weights (the stand-in for "fell back to the
        # bound" / "ran short of memory"): alone it cuts elsewhere, in agreement it uses rank 0's cuts
        if rank == 1:
            D.COST_WEIGHTS = (5.0, 0.0, 300.0)
        res["cuts_alone"] = D.shard_cuts(gs, 3)
        res["cuts_agreed"] = D.shard_cuts(gs, 3, agree=True)
        D.COST_WEIGHTS = (0.5, 24.0, 12.0)
        sub = gs if rank == 0 else gs.subset(0, gs.num_graphs - 1)      # ranks holding different datasets: an error
        try:
            D.shard_cuts(sub, 3, agree=True)
            res["cuts_mismatch"] = "no error"
        except RuntimeError as e:
            res["cuts_mismatch"] = str(e)
        # one-shot gradient all-reduce
        torch.manual_seed(0)
        lin = torch.nn.Linear(4, 3)
        unused = torch.nn.Linear(2, 2)
        x = torch.full((5, 4), float(rank + 1))
        lin(x).sum().backward()
        D.allreduce_grads(list(lin.parameters()) + list(unused.parameters()), mode="mean")
        res["wgrad"], res["ugrad"] = lin.weight.grad.clone(), unused.weight.grad.clone()
        # bucketed, hook-driven all-reduce: one weighted step on batches (0, 1) and one on (2, None)
        m = _Model("mean")
        data = _Data(SIZES)
        bk = D.GradBuckets(list(m.parameters()), num_buckets=3)
        grads, issued = [], []
        for group in D.step_groups(SIZES, world):
            bk.zero()
            i = group[rank]
            if i is not None:
                (m._loss(data.batches[i]) * D.mean_loss_weight(SIZES, group, rank)).backward()
            issued.append(bk._next)              # buckets already issued from the hooks
            bk.finish()
            grads.append([p.grad.clone() for p in m.parameters()])
        res["bucket_layout"] = (bk._bucket_of[id(m.unused.weight)], bk._bucket_of[id(m.unused.bias)],
                                len(bk.buckets), issued)
        bk.close()
        res["bucket_grads"] = grads
        # DDPReplay's step sequence (on a CPU device it runs eagerly: fresh gradients packed into the buckets, all
        # buckets reduced after the backward, optimizer on the bucket views) == the hook-driven step, bit for bit
        from desco_amd.trainer import DDPReplay
        finals = []
        for replay in (False, True):
            m = _Model("mean")
            data = _Data(SIZES)
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            bk = D.GradBuckets(list(m.parameters()), num_buckets=3)
            ddp = DDPReplay(m, opt, bk, torch.device("cpu"))
            for epoch in range(3):
                for k, group in enumerate(D.step_groups(SIZES, world)):
                    i = group[rank]
                    b = None if i is None else data.batches[i]
                    w = D.mean_loss_weight(SIZES, group, rank)
                    if replay and epoch >= 1:
                        ddp.step(k, i, b, w)
                        continue
                    bk.zero()
                    if b is not None:
                        (m._loss(b) * w).backward()
                    bk.finish()
                    opt.step()
            bk.close()
            finals.append([p.detach().clone() for p in m.parameters()])
        res["ddp_replay_equal"] = all(torch.equal(a, b) for a, b in zip(*finals))
        # Trainer.fit / predict in ddp mode
        for reduce in ("mean", "sum"):
            m = _Model(reduce)
            if rank == 1:       # replicas must be re-synchronised from rank 0
                with torch.no_grad():
                    for p in m.parameters():
                        p.add_(1.0)
            ck = ModelCheckpoint(monitor="val")
            tr = Trainer(max_epochs=2, accelerator="cpu", devices=[0, 1], strategy="ddp",
                         default_root_dir=os.path.join(tmp, reduce), callbacks=[ck], grad_reduce=reduce)
            tr.fit(m, _Data(SIZES))
            res[f"fit_{reduce}"] = [p.detach().clone() for p in m.parameters()]
            res[f"ckpt_{reduce}"] = (ck.best_model_path, os.path.exists(ck.best_model_path))
            res[f"hist_{reduce}"] = tr.history
            res[f"pred_{reduce}"] = torch.cat(tr.predict(m, _Data(SIZES).train_dataloader()))
            res[f"test_{reduce}"] = tr.test(m, _Data(SIZES))[0]["test_loss"]
        torch.save(res, os.path.join(tmp, f"res_{rank}.pt"))     # (tensors through a queue need the
        q.put(rank)                                               #  sender alive while they are read)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_world(world, tmp):
    """Spawn the ranks; a lost rendezvous (the probed port taken in between, a slow spawn) is retried
    on a fresh port -- the assertions on the results are made once, by the caller."""
    last = None
    for _attempt in range(3):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, tmp)) for r in range(world)]
        for p in procs:
            p.start()
        res = {}
        try:
            for _ in range(world):
                r = q.get(timeout=240)
                res[r] = torch.load(os.path.join(tmp, f"res_{r}.pt"), weights_only=False)
        except _queue.Empty:
            res = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
                p.join()
        last = [p.exitcode for p in procs]
        if res is not None and all(c == 0 for c in last):
            return res
    raise AssertionError(f"gloo world of {world} did not complete (exit codes {last})")


@pytest.fixture(scope="module")
def world2(tmp_path_factory):
    return _run_world(2, str(tmp_path_factory.mktemp("ddp")))


def test_sharded_inference_gather_and_grad_allreduce(world2):
    r0 = world2[0]
    gs = GraphSet.from_edge_lists(golden_graphs())
    assert torch.equal(r0["full"], _fake_stage(gs))                      # 1-vs-N rank equality
    assert world2[1]["full"] is None
    assert torch.equal(world2[1]["full_all"], _fake_stage(gs))           # all-gather: on every rank
    lo, hi = r0["range"]
    assert lo == 0 and 0 < hi < gs.num_graphs
    # mean of per-rank grads: rank r contributes 5*(r+1) per weight entry
    assert torch.allclose(r0["wgrad"], torch.full((3, 4), 5 * (1 + 2) / 2))
    assert torch.equal(r0["ugrad"], torch.zeros(2, 2))                   # unused params -> zeros


def test_ddp_replay_sequence_equals_the_hook_driven_step(world2):
    assert world2[0]["ddp_replay_equal"] and world2[1]["ddp_replay_equal"]


def test_shard_cuts_are_decided_by_rank_0_for_all(world2):
    """nothing checked that ranks agree on the shard cuts."""
    a0, a1 = world2[0]["cuts_alone"], world2[1]["cuts_alone"]
    assert a0 != a1                                        # the divergent rank would have cut elsewhere
    assert world2[0]["cuts_agreed"] == world2[1]["cuts_agreed"] == a0
    assert world2[0]["cuts_mismatch"] == "no error" and "different datasets" in world2[1]["cuts_mismatch"]


def test_graph_costs_reraises_everything_but_the_workspace_error(monkeypatch):
    """Only "largest graph does not fit the LDS workspace" may fall back to the cost bound."""
    import desco_amd.partition as P
    gs = GraphSet.from_edge_lists(golden_graphs(max_n=20))

    def boom(msg):
        def f(*a, **k):
            raise RuntimeError(msg)
        return f
    monkeypatch.setattr(P, "build_partition_device", boom("libdesco_hip x failed (code -1): desco_partition_dev: largest "
                                                          "graph does not fit the LDS workspace (n = 9999)"))
    assert np.array_equal(D.graph_costs(gs, 29, "cuda:0"), D.graph_costs(gs, 29))
    monkeypatch.setattr(P, "build_partition_device", boom("HIP error: out of memory"))
    with pytest.raises(RuntimeError, match="out of memory"):
        D.graph_costs(gs, 29, "cuda:0")


def test_bucketed_weighted_gradients_equal_union_batch_gradient(world2):
    """sum-all-reduce of count-weighted per-rank mean-loss gradients == gradient of the mean loss over
    the union of the group's batches, incl. the short last group (rank 1 has no batch there) and a
    parameter that never receives a gradient."""
    data = _Data(SIZES)
    for step, k in enumerate(range(0, len(SIZES), 2)):
        m = _Model("mean")
        bs = data.batches[k:k + 2]
        u = _Batch(torch.cat([b.x for b in bs]), torch.cat([b.y for b in bs]))
        m._loss(u).backward()
        want = [p.grad if p.grad is not None else torch.zeros_like(p) for p in m.parameters()]
        for r in (0, 1):
            got = world2[r]["bucket_grads"][step]
            for g, w in zip(got, want):
                torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6)


def test_unused_parameters_move_to_the_last_bucket_after_step_0(world2):
    """a never-used parameter in an early bucket blocked every hook-issued all-reduce
    until finish().  Step 0 finds it (flags agreed over the ranks), step 1 issues every other bucket
    from the hooks (rank 0 has the batch of the short last group)."""
    for r in (0, 1):
        bw, bb, nb, issued = world2[r]["bucket_layout"]
        assert bw == bb == nb - 1, (bw, bb, nb)
    issued0 = world2[0]["bucket_layout"][3]
    assert issued0[0] == 0                       # step 0: bucket 0 still held the unused parameters
    assert issued0[1] == world2[0]["bucket_layout"][2] - 1      # step 1: all but the unused bucket


_RANK_SCRIPT = """
import os, sys, time
sys.path.insert(0, {root!r})
import torch
from desco_amd import distributed as D
mode = sys.argv[1]
r = int(os.environ["RANK"])
if mode == "crash":
    if r == 1:
        sys.exit(3)
    time.sleep(120)
elif mode == "rank0_first":
    os.environ["DESCO_PG_TIMEOUT_S"] = "2"          # shorter than rank 0's build below
    D.init_from_env(torch.device("cpu"))
    order = []
    def build():
        if D.rank() == 0:
            time.sleep(5)
        open(os.path.join(sys.argv[2], "built_%d" % D.rank()), "w").write(str(time.time()))
        return D.rank()
    assert D.rank0_first(build) == r
    import torch.distributed as dist
    t = torch.ones(1)
    dist.all_reduce(t)
    assert t.item() == 2
    dist.destroy_process_group()
elif mode == "rank0_fails":
    D.init_from_env(torch.device("cpu"))
    def build():
        if D.rank() == 0:
            raise ValueError("no data")
        return 1
    D.rank0_first(build)
"""


def _script(tmp_path):
    p = tmp_path / "rank_script.py"
    p.write_text(_RANK_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    return str(p)


def test_launch_notices_a_crashed_rank_while_others_block(tmp_path):
    """launch() waited for rank 0 first and never saw rank 1 die."""
    import time
    t0 = time.time()
    rc = D.launch([_script(tmp_path), "crash"], 2)
    assert rc == 3 and time.time() - t0 < 60


def test_rank0_first_waits_on_the_store_not_in_a_collective(tmp_path):
    """a long cache build on rank 0 must not trip the process group's watchdog: the other
    ranks wait on the store (PG timeout 2 s here, build 5 s) and build after rank 0."""
    rc = D.launch([_script(tmp_path), "rank0_first", str(tmp_path)], 2, timeout=120)
    assert rc == 0
    t = [float(open(tmp_path / f"built_{r}").read()) for r in (0, 1)]
    assert t[1] >= t[0]


def test_rank0_first_passes_a_failure_on(tmp_path):
    rc = D.launch([_script(tmp_path), "rank0_fails"], 2, timeout=120)
    assert rc != 0


def test_local_device_honours_launch_devices(monkeypatch):
    monkeypatch.setenv("DESCO_DEVICES", "5,7")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.delenv("DESCO_SHARE_GPU", raising=False)
    assert D.local_device() == torch.device("cuda", 7)
    assert D.local_device([2, 3]) == torch.device("cuda", 3)     # an explicit list wins


@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_trainer_ddp_fit_matches_single_process_union_steps(world2, reduce):
    ref = _single_process_reference(reduce, epochs=2)
    for r in (0, 1):
        for g, w in zip(world2[r][f"fit_{reduce}"], ref.parameters()):
            torch.testing.assert_close(g, w.detach(), rtol=1e-5, atol=1e-6)
    # the best checkpoint path is known on every rank and the file exists (rank 0 wrote it)
    p0, ok0 = world2[0][f"ckpt_{reduce}"]
    p1, ok1 = world2[1][f"ckpt_{reduce}"]
    assert p0 == p1 and p0 and ok0 and ok1
    assert world2[0][f"hist_{reduce}"] == world2[1][f"hist_{reduce}"]     # synchronised val loss


@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_trainer_ddp_predict_and_test_are_sharded_and_complete(world2, reduce):
    ref = _single_process_reference(reduce, epochs=2)
    data = _Data(SIZES)
    want = torch.cat([ref.predict_step(b, 0) for b in data.batches]).detach()
    for r in (0, 1):
        torch.testing.assert_close(world2[r][f"pred_{reduce}"], want, rtol=1e-5, atol=1e-6)
    tot = sum(float(ref._loss(b)) * b.num_graphs for b in data.batches) / sum(SIZES)
    assert abs(world2[0][f"test_{reduce}"] - tot) < 1e-4 * max(1.0, abs(tot))
    assert world2[0][f"test_{reduce}"] == world2[1][f"test_{reduce}"]


def test_trainer_refuses_many_devices_without_ranks():
    from desco_amd.trainer import Trainer
    with pytest.raises(RuntimeError, match="processes"):
        Trainer(accelerator="cpu", devices=[0, 1], strategy="ddp")


def test_step_groups_and_weights():
    g = D.step_groups([512, 512, 512, 100, 7], 2)
    assert g == [[0, 1], [2, 3], [4, None]]
    assert D.mean_loss_weight([512, 512, 512, 100, 7], g[0], 0) == 0.5
    assert abs(D.mean_loss_weight([512, 512, 512, 100, 7], g[1], 1) - 100 / 612) < 1e-12
    assert D.mean_loss_weight([512, 512, 512, 100, 7], g[2], 0) == 1.0
    assert D.mean_loss_weight([512, 512, 512, 100, 7], g[2], 1) == 0.0
    assert D.step_groups([5, 5], 1) == [[0], [1]]


def test_contiguous_shards_balance_and_cover():
    costs = np.array([5, 1, 1, 1, 8, 2, 2, 4, 4, 4], dtype=float)
    for w in (1, 2, 3, 4, 8, 16):
        sh = D.contiguous_shards(costs, w)
        assert len(sh) == w and sh[0][0] == 0 and sh[-1][1] == len(costs)
        assert all(a[1] == b[0] for a, b in zip(sh[:-1], sh[1:]))
    sh = D.contiguous_shards(costs, 2)
    loads = [costs[a:b].sum() for a, b in sh]
    assert max(loads) <= 0.75 * costs.sum()


def _true_cost(gs, q=29):
    """D.graph_costs' formula on the exact counts of the host partition builder (what the device builder gives a GPU)"""
    from desco_amd.partition import build_partition
    part = build_partition(gs, 4)
    G, B, ng = gs.num_graphs, part.num_neigh, part.neigh_index[:, 0]
    rows = np.bincount(ng, weights=np.diff(part.count_ptr).astype(np.float64) + 1.0, minlength=G)
    neigh = np.bincount(ng, minlength=G).astype(np.float64)
    per_row = np.diff(part.vrowptr.astype(np.int64)).reshape(-1, 4).sum(1).astype(np.float64)
    owner = np.concatenate([np.repeat(np.arange(B), np.diff(part.count_ptr)), np.arange(B)])
    edges = np.bincount(ng[owner], weights=per_row, minlength=G)
    w_e, w_n, w_v = D.COST_WEIGHTS
    return rows + w_e * edges + w_n * neigh + w_v * (q / 29.0) * np.diff(gs.graph_ptr)


@pytest.mark.parametrize("workload,replicas", [("cox2", 8), ("syn_1827", 1), ("msrc_imdb", 2)])
def test_shard_costs_are_balanced_on_the_bench_workloads(workload, replicas):
    """VERDICT r4 item 8a.  The cost of a graph is its neighborhood rows, edges and neighborhoods (weights from MI355X
    shard timings, bench.py secondary.*.strong_scaling_8); on a GPU D.graph_costs counts them exactly with the device
    partition builder.  Here, with the host builder as the ruler: shards cut on the exact costs differ by at most
    5 % between the heaviest rank and the mean for 2, 4 and 8 ranks, and the device-free proxy (what CPU-side planning
    falls back to) stays within 12 % -- on Syn_1827, where round 4's degree proxy was off by 2x (8 shards between 2.7 and
    16.5 ms), as on the molecule and social shapes."""
    from desco_amd import synthetic
    gs = synthetic.WORKLOADS[workload]().replicate(replicas)
    true = _true_cost(gs)
    proxy = D.graph_costs(gs, 29)
    for world in (2, 4, 8):
        for name, c, tol in (("exact", true, 1.05), ("proxy", proxy, 1.12)):
            loads = np.array([true[a:b].sum() for a, b in D.contiguous_shards(c, world)])
            imbalance = loads.max() / loads.mean()
            print(f"[shards] {workload} x{replicas}, {world} ranks, cut on the {name} costs: max / mean true cost {imbalance:.4f}")
            assert loads.min() > 0 and imbalance <= tol, (workload, world, name, imbalance)
