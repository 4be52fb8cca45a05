"""Minimal stand-in for the slice of ``pytorch_lightning.Trainer`` that main.py uses
(main.py:205-213, 242-273, 296-301, 338-347, 370-379): fit / test / predict, ReduceLROnPlateau on
the monitored validation loss, best/last checkpointing, optional data-parallel training
(one process per GPU, flat-bucket gradient all-reduce over RCCL -- desco_amd.distributed)."""
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


class Trainer:
    def __init__(self, max_epochs: int = 1, accelerator: str = "gpu", devices=None,
                 default_root_dir: str = ".", callbacks=None, strategy: Optional[str] = None,
                 grad_reduce: str = "mean", precision: str = "fp32", graph_capture: bool = False,
                 **unused):
        # precision: "fp32" | "bf16" (Lightning's "32" / "bf16-mixed" spellings accepted): matrix
        # products of the training step in fp32 or bf16 MFMA (desco_amd.autograd.set_precision)
        self.precision = precision
        # graph_capture: after one eager epoch, every training batch's step (forward, backward, Adam)
        # is captured in a hipGraph and replayed in later epochs -- the ~400 launches of a step are
        # host-bound otherwise.  Single process, models without dropout (the neighborhood model).
        self.graph_capture = graph_capture
        self.max_epochs = max_epochs
        self.root = default_root_dir
        self.callbacks = callbacks or []
        self.strategy = strategy
        self.grad_reduce = grad_reduce      # "mean" (neighborhood loss) or "sum" (gossip loss)
        dev = devices[0] if isinstance(devices, (list, tuple)) and devices else 0
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", dev if dev != "auto" else 0)))
        self.history: List[dict] = []

    # ---- helpers ----------------------------------------------------------------------------
    def _rank0(self):
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def _shard(self, batches):
        """Round-robin batches over ranks (what Lightning's DistributedSampler does per item)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return batches
        r, w = dist.get_rank(), dist.get_world_size()
        return [b for i, b in enumerate(batches) if i % w == r]

    def _mean_loss(self, model, loader, step_name) -> float:
        import torch.distributed as dist
        tot, cnt = 0.0, 0
        with torch.no_grad():
            for i, batch in enumerate(self._shard(list(loader))):
                loss = getattr(model, step_name)(batch.to(self.device), i)
                tot += float(loss) * batch.num_graphs
                cnt += batch.num_graphs
        if dist.is_available() and dist.is_initialized():          # sync_dist=True
            t = torch.tensor([tot, cnt], device=self.device, dtype=torch.float64)
            dist.all_reduce(t)
            tot, cnt = float(t[0]), float(t[1])
        return tot / max(cnt, 1)

    def _graph_epoch(self, model, opt, batches, graphs, side, capture: bool):
        """One epoch on the side stream: eager (epoch 0), then capture-once / replay per batch."""
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for i, batch in enumerate(batches):
                if not capture:
                    opt.zero_grad(set_to_none=True)
                    model.train_forward(batch, i).backward()
                    opt.step()
                    continue
                if i not in graphs:
                    opt.zero_grad(set_to_none=True)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):      # (the capture does not execute the step)
                        model.train_forward(batch, i).backward()
                        opt.step()
                    graphs[i] = g
                graphs[i].replay()
        torch.cuda.current_stream(self.device).wait_stream(side)

    # ---- API ----------------------------------------------------------------------------------
    def fit(self, model, datamodule):
        from . import autograd as AG
        AG.set_precision(self.precision)
        model.to(self.device)
        cfg = model.configure_optimizers()
        opt, sched = cfg["optimizer"], cfg["lr_scheduler"]
        ckpt = next((c for c in self.callbacks if isinstance(c, ModelCheckpoint)), None)
        os.makedirs(self.root, exist_ok=True)
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        use_graphs = self.graph_capture and not multi and float(getattr(model, "dropout", 0.0) or 0.0) == 0.0
        # shuffle=False (main.py:195): the batch stream is the same every epoch, so the device-resident
        # batches (and their backward indices) are built once
        train_batches = [b.to(self.device) for b in self._shard(list(datamodule.train_dataloader()))]
        graphs = {}
        # capture happens on a side stream, and autograd's AccumulateGrad nodes must have been created
        # on that same stream: with graph_capture the whole training loop runs on it
        side = torch.cuda.Stream(self.device) if use_graphs else None
        if use_graphs:
            for g_ in opt.param_groups:       # Adam state and lr as device tensors: capturable
                g_["capturable"] = True
                g_["lr"] = torch.tensor(float(g_["lr"]), device=self.device)
        for epoch in range(self.max_epochs):
            model.train()
            if use_graphs:
                self._graph_epoch(model, opt, train_batches, graphs, side, capture=epoch >= 1)
            else:
                for i, batch in enumerate(train_batches):
                    opt.zero_grad(set_to_none=True)
                    loss = model.training_step(batch, i)
                    loss.backward()
                    D.allreduce_grads(list(model.parameters()), mode=self.grad_reduce)
                    opt.step()
            model.eval()
            val = self._mean_loss(model, datamodule.val_dataloader(), "validation_step")
            if use_graphs:       # keep the captured graphs' lr tensor; the scheduler works on floats
                lr_t = [g_["lr"] for g_ in opt.param_groups]
                for g_, t_ in zip(opt.param_groups, lr_t):
                    g_["lr"] = float(t_)
                sched.step(val)
                for g_, t_ in zip(opt.param_groups, lr_t):
                    t_.fill_(float(g_["lr"]))
                    g_["lr"] = t_
            else:
                sched.step(val)
            self.history.append({"epoch": epoch, cfg["monitor"]: val, "lr": float(opt.param_groups[0]["lr"])})
            if ckpt is not None and self._rank0():
                if ckpt.save_last:
                    model.save_checkpoint(os.path.join(self.root, "last.ckpt"))
                if ckpt.best_score is None or val < ckpt.best_score:
                    ckpt.best_score = val
                    ckpt.best_model_path = os.path.join(self.root, f"epoch={epoch}-best.ckpt")
                    model.save_checkpoint(ckpt.best_model_path)
        return self

    def test(self, model, datamodule):
        model.to(self.device).eval()
        loss = self._mean_loss(model, datamodule.test_dataloader(), "test_step")
        return [{"test_loss": loss}]

    def predict(self, model, dataloader) -> List[torch.Tensor]:
        model.to(self.device).eval()
        with torch.no_grad():
            return [model.predict_step(b.to(self.device), i) for i, b in enumerate(dataloader)]
