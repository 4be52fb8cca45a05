"""Minimal stand-in for the slice of ``pytorch_lightning.Trainer`` that main.py uses
(main.py:205-213, 242-273, 296-301, 338-347, 370-379): fit / test / predict, ReduceLROnPlateau on
the monitored validation loss, best/last checkpointing, and data-parallel training / prediction
with one process per GPU (``strategy="ddp"``: gradient buckets all-reduced over RCCL while backward
is still running -- desco_amd.distributed)."""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import distributed as D


class ModelCheckpoint:
    """monitor / mode / save_top_k=1 / save_last of main.py:199-204."""

    def __init__(self, monitor: str, mode: str = "min", save_top_k: int = 1, save_last: bool = True):
        assert mode == "min" and save_top_k == 1
        self.monitor, self.save_last = monitor, save_last
        self.best_model_path = ""
        self.best_score: Optional[float] = None


class DDPReplay:
    """A data-parallel optimisation step at replay speed (main.py:242-255 of the reference hands the devices to a "ddp"
    Trainer).  A collective cannot be issued from inside a captured backward, so the step is cut in two graphs around it:

        graph A[k]  zero the gradient buckets, forward, backward, pack the gradients into the buckets   (per batch k)
        all-reduce  every bucket over RCCL, asynchronously in index order, then wait                     (not captured)
        graph B     the optimizer step reading the bucket views                                          (one for all k)

    Gradients are produced into fresh tensors (p.grad = None at capture: autograd stores instead of accumulating -- the
    hook-driven eager form pays one add launch per parameter for its views) and packed by copy2d launches of 24 tensors
    each; a rank without a batch in a step replays the zero fill alone and still joins the collectives.  The arithmetic
    is the eager DDP step's (same seed weight, same bucket layout and order), so parameters agree bit for bit
    (tests/test_multirank_gpu.py).  On a CPU device the same sequence runs eagerly (gloo tests)."""

    def __init__(self, model, opt, buckets: "D.GradBuckets", device: torch.device):
        self.model, self.opt, self.buckets, self.device = model, opt, buckets, device
        self.use_graphs = device.type == "cuda"
        self.graphs = {}
        self.adam = None
        self.layout = None          # the bucket layout the graphs were captured for

    def _forward_backward(self, i, batch, w):
        from . import autograd as AG
        self.buckets.fill_zero()
        if batch is not None:
            for p in self.buckets.params:
                p.grad = None
            AG.backward(self.model.train_forward(batch, i), w)
            self.buckets.pack_from_grads()

    def step(self, k, i, batch, w, stream=None):
        """step k of the epoch (batch index i, weight w of this rank's loss); call it on the training stream."""
        bk = self.buckets
        if self.layout is not None and self.layout is not bk.buckets:
            raise RuntimeError("the gradient buckets were laid out again after a step was captured")
        if not self.use_graphs:
            self._forward_backward(i, batch, w)
        else:
            g = self.graphs.get(k)
            if g is None:
                g = torch.cuda.CUDAGraph()
                # (capture_error_mode "thread_local": RCCL's watchdog thread polls its work events with hipEventQuery
                #  while this thread captures -- in the default "global" mode that call invalidates the capture)
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    self._forward_backward(i, batch, w)        # (the capture does not execute the step)
                self.graphs[k] = g
                self.layout = bk.buckets
            g.replay()
        bk.allreduce_all()
        if not self.use_graphs:
            bk.attach_views()
            self.opt.step()
            return
        if self.adam is None:
            bk.attach_views()
            self.adam = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.adam, stream=stream, capture_error_mode="thread_local"):
                self.opt.step()
        self.adam.replay()


class Trainer:
    def __init__(self, max_epochs: int = 1, accelerator: str = "gpu", devices=None,
                 default_root_dir: str = ".", callbacks=None, strategy: Optional[str] = None,
                 grad_reduce: str = "mean", precision: str = "fp32", graph_capture: bool = False,
                 num_buckets: int = 4, verbose: bool = False, **unused):
        # precision: "fp32" | "bf16" (Lightning's "32" / "bf16-mixed" spellings accepted): matrix
        # products of the training step in fp32 or bf16 MFMA (desco_amd.autograd.set_precision)
        self.precision = precision
        # graph_capture: after one eager epoch, every training batch's step (forward, backward, Adam)
        # is captured in a hipGraph and replayed in later epochs -- the ~400 launches of a step are
        # host-bound otherwise.  Data parallel: two graphs around the gradient all-reduces (DDPReplay).
        self.graph_capture = graph_capture
        self.max_epochs = max_epochs
        self.root = default_root_dir
        self.callbacks = callbacks or []
        self.strategy = strategy
        self.grad_reduce = grad_reduce      # "mean" (neighborhood loss) or "sum" (gossip loss)
        self.num_buckets = num_buckets
        self.verbose = verbose              # print the monitored validation loss per epoch (rank 0)
        self.device = D.local_device(devices, accelerator)
        if self.device.type == "cuda":
            # the C ABI launches on the CURRENT device / stream: bind this process to its GPU
            torch.cuda.set_device(self.device)
        ndev = len(devices) if isinstance(devices, (list, tuple)) else 1
        if strategy == "ddp" or ndev > 1:
            _, w, _ = D.env_world()
            if w == 1 and ndev > 1 and not D.is_initialized():
                raise RuntimeError(
                    f"Trainer(devices={list(devices)}): {ndev} devices need {ndev} processes (one per GPU) "
                    "but WORLD_SIZE is 1 -- start them with `main.py --gpu 0 1 ..` (which spawns the "
                    "ranks), desco_amd.distributed.launch, or `python -m torch.distributed.run`")
            D.init_from_env(self.device)
        self.history: List[dict] = []

    # ---- helpers ----------------------------------------------------------------------------
    def _rank0(self):
        return D.rank() == 0

    def _shard(self, batches):
        """Round-robin batches over ranks (validation / test: one all-reduce of (sum, count) at the
        end, so unequal per-rank batch counts are harmless)."""
        r, w = D.rank(), D.world_size()
        return batches if w == 1 else [b for i, b in enumerate(batches) if i % w == r]

    def _mean_loss(self, model, loader, step_name) -> float:
        tot, cnt = 0.0, 0
        with torch.no_grad():
            for i, batch in enumerate(self._shard(list(loader))):
                loss = getattr(model, step_name)(batch.to(self.device), i)
                tot += float(loss) * batch.num_graphs
                cnt += batch.num_graphs
        if D.world_size() > 1:                                       # sync_dist=True
            t = torch.tensor([tot, cnt], dtype=torch.float64,
                             device=self.device if self.device.type == "cuda" else "cpu")
            D.all_reduce_(t, "sum")
            tot, cnt = float(t[0]), float(t[1])
        return tot / max(cnt, 1)

    def _ddp_epoch(self, model, opt, steps, buckets, ddp, side, capture: bool):
        """One data-parallel epoch on the side stream: the hook-driven eager step (epoch 0: it also settles the bucket
        layout), then DDPReplay's two graphs around the bucket all-reduces."""
        from . import autograd as AG
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for k, (i, batch, w) in enumerate(steps):
                if capture:
                    ddp.step(k, i, batch, w, stream=side)
                    continue
                buckets.zero()
                if batch is not None:
                    AG.backward(model.training_step(batch, i), w)     # bucket all-reduces start from the hooks
                buckets.finish()
                opt.step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        if capture:
            model.invalidate_caches()

    def _graph_epoch(self, model, opt, batches, graphs, side, capture: bool):
        """One epoch on the side stream: eager (epoch 0), then capture-once / replay per batch."""
        from . import autograd as AG
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for i, batch in enumerate(batches):
                if not capture:
                    opt.zero_grad(set_to_none=True)
                    AG.backward(model.train_forward(batch, i))
                    opt.step()
                    continue
                if i not in graphs:
                    opt.zero_grad(set_to_none=True)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):      # (the capture does not execute the step)
                        AG.backward(model.train_forward(batch, i))
                        opt.step()
                    graphs[i] = g
                graphs[i].replay()
        torch.cuda.current_stream(self.device).wait_stream(side)
        if capture:
            # a replay updates the parameters without bumping tensor._version, which is what the
            # inference-side caches (folded weights, head operands, query embeddings) are keyed on
            model.invalidate_caches()

    # ---- API ----------------------------------------------------------------------------------
    def fit(self, model, datamodule):
        from . import autograd as AG
        AG.set_precision(self.precision)
        model.to(self.device)
        world, rank = D.world_size(), D.rank()
        multi = D.collectives_on()             # (a world of one rank too under DESCO_FORCE_COLLECTIVES=1)
        if multi:
            D.broadcast_params(model)          # every replica starts from rank 0's weights
        cfg = model.configure_optimizers()
        opt, sched = cfg["optimizer"], cfg["lr_scheduler"]
        ckpt = next((c for c in self.callbacks if isinstance(c, ModelCheckpoint)), None)
        os.makedirs(self.root, exist_ok=True)
        # (dropout does not stand in the way of replay: the masks are functions of a (seed, step) pair in device memory
        #  that a captured launch advances -- ops.rng_next -- so every replay draws the next mask)
        use_graphs = self.graph_capture and self.device.type == "cuda"
        # shuffle=False (main.py:195): the batch stream is the same every epoch, so the device-resident
        # batches (and their backward indices) are built once.  Data parallel: optimisation step k
        # consumes the `world` consecutive batches [k*world, (k+1)*world), one per rank, weighted by
        # their neighborhood counts for the mean loss (distributed.step_groups)
        host_batches = list(datamodule.train_dataloader())
        sizes = [b.num_graphs for b in host_batches]
        groups = D.step_groups(sizes, world)
        steps = []
        for g in groups:
            i = g[rank]
            w = 1.0 if self.grad_reduce == "sum" else D.mean_loss_weight(sizes, g, rank)
            steps.append((i, None if i is None else host_batches[i].to(self.device), w))
        del host_batches
        buckets = D.GradBuckets(list(model.parameters()), self.num_buckets) if multi else None
        ddp = DDPReplay(model, opt, buckets, self.device) if (multi and use_graphs) else None
        graphs = {}
        # capture happens on a side stream, and autograd's AccumulateGrad nodes must have been created
        # on that same stream: with graph_capture the whole training loop runs on it
        side = torch.cuda.Stream(self.device) if use_graphs else None
        for epoch in range(self.max_epochs):
            model.train()
            if use_graphs and multi:
                self._ddp_epoch(model, opt, steps, buckets, ddp, side, capture=epoch >= 1)
            elif use_graphs:
                self._graph_epoch(model, opt, [b for _, b, _ in steps], graphs, side, capture=epoch >= 1)
            elif multi:
                for i, batch, w in steps:
                    buckets.zero()
                    if batch is not None:
                        AG.backward(model.training_step(batch, i), w)   # bucket all-reduces start from the hooks
                    buckets.finish()
                    opt.step()
            else:
                for i, batch, _ in steps:
                    opt.zero_grad(set_to_none=True)
                    loss = model.training_step(batch, i)
                    AG.backward(loss)          # (loss.backward() seeded without torch's ones_like fill)
                    opt.step()
            model.eval()
            val = self._mean_loss(model, datamodule.val_dataloader(), "validation_step")
            sched.step(val)
            if hasattr(opt, "sync_lr"):      # replayed steps read the rate from device memory (optim.Adam)
                opt.sync_lr()
            self.history.append({"epoch": epoch, cfg["monitor"]: val, "lr": float(opt.param_groups[0]["lr"])})
            if self._rank0() and self.verbose:
                print(f"epoch {epoch}: {cfg['monitor']} = {val:.6g}  lr = {float(opt.param_groups[0]['lr']):.3g}",
                      flush=True)
            if ckpt is not None:
                if self._rank0():
                    if ckpt.save_last:
                        model.save_checkpoint(os.path.join(self.root, "last.ckpt"))
                    if ckpt.best_score is None or val < ckpt.best_score:
                        ckpt.best_score = val
                        ckpt.best_model_path = os.path.join(self.root, f"epoch={epoch}-best.ckpt")
                        model.save_checkpoint(ckpt.best_model_path)
                if multi:     # every rank reloads the best checkpoint afterwards (main.py:262-264)
                    ckpt.best_score, ckpt.best_model_path = D.broadcast_object(
                        (ckpt.best_score, ckpt.best_model_path))
        if buckets is not None:
            buckets.close()
            for p in model.parameters():      # detach the views from the (now unused) buckets
                p.grad = None
        if multi:
            D.barrier()                        # the checkpoint files are complete on every rank
        return self

    def test(self, model, datamodule):
        model.to(self.device).eval()
        loss = self._mean_loss(model, datamodule.test_dataloader(), "test_step")
        return [{"test_loss": loss}]

    def predict(self, model, dataloader) -> List[torch.Tensor]:
        """Per-batch predictions.  Data parallel: every rank predicts a contiguous range of the
        batches and the row blocks are all-gathered in order, so each rank returns the full result
        (as ONE tensor in the list; the caller concatenates, main.py:279-301)."""
        model.to(self.device).eval()
        world, rank = D.world_size(), D.rank()
        with torch.no_grad():
            if world == 1:
                return [model.predict_step(b.to(self.device), i) for i, b in enumerate(dataloader)]
            batches = list(dataloader)
            n = len(batches)
            lo, hi = n * rank // world, n * (rank + 1) // world
            outs = [model.predict_step(batches[i].to(self.device), i) for i in range(lo, hi)]
            local = torch.cat(outs) if outs else None
            # fewer batches than ranks: a rank without a batch contributes an empty block of the
            # common width, which it learns from the others
            width = torch.tensor([local.shape[1] if local is not None else 0], dtype=torch.int64)
            width = width.to(self.device) if self.device.type == "cuda" else width
            D.all_reduce_(width, "max")
            if local is None:
                local = torch.zeros((0, int(width.item())), device=self.device)
            return [D.allgather_rows(local)]
