"""The optimizer of both models (reference configure_optimizers, lightning_model.py:160-173, 570-583:
``torch.optim.Adam(self.parameters(), lr=..., weight_decay=...)``) on the library's own launch.

``Adam`` is a ``torch.optim.Optimizer`` (so ``ReduceLROnPlateau`` drives it unchanged, ``zero_grad`` is inherited and
``param_groups[i]["lr"]`` stays a Python float) whose ``step()`` is ``desco_adam_step_f32``: one launch per 128
parameters instead of torch's ~10 multi-tensor launches + one launch per parameter in capturable mode.  All state the
launch reads lives on the device (moments, per-tensor step counts, learning rate), so the same ``step()`` is what a
hipGraph capture records: eager and replayed steps run identical arithmetic.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
from torch.autograd.graph import increment_version

from . import _lib, ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=float(lr), betas=tuple(betas), eps=float(eps),
                                      weight_decay=float(weight_decay)))
        self._dev_state = {}          # id(group) -> dict

    # ---- device state -------------------------------------------------------------------------
    def _group_state(self, group):
        st = self._dev_state.get(id(group))
        ps = group["params"]
        if st is not None and st["n"] == len(ps):
            return st
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("desco_amd.optim.Adam: parameters must be contiguous fp32 tensors on one device")
        if dev.type != "cuda":
            raise RuntimeError("desco_amd.optim.Adam runs on the MI355X only (no CPU fallback): move the model first")
        sizes = np.array([p.numel() for p in ps], dtype=np.int64)
        total = int(sizes.sum())
        st = {"n": len(ps), "sizes": sizes,
              "params": np.array([p.data_ptr() for p in ps], dtype=np.uint64),
              "grads": np.zeros(len(ps), dtype=np.uint64),
              "m": torch.zeros(total, device=dev), "v": torch.zeros(total, device=dev),
              "steps": torch.zeros(len(ps), device=dev),
              "arrivals": torch.zeros(len(ps), device=dev, dtype=torch.int32),
              "lr": torch.full((1,), float(group["lr"]), device=dev), "lr_host": float(group["lr"])}
        self._dev_state[id(group)] = st
        return st

    def sync_lr(self):
        """Copy ``param_groups[i]["lr"]`` (a Python float, e.g. after a scheduler step) to the device scalar the
        launches read.  ``step()`` does this itself when it runs eagerly; a loop that only REPLAYS captured steps
        calls it after changing the rate (Trainer does)."""
        for group in self.param_groups:
            st = self._dev_state.get(id(group))
            if st is not None and st["lr_host"] != float(group["lr"]):
                st["lr"].fill_(float(group["lr"]))
                st["lr_host"] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        capturing = None
        for group in self.param_groups:
            st = self._group_state(group)          # (refuses anything that is not on the GPU)
            if st is None:
                continue
            if capturing is None:
                capturing = torch.cuda.is_current_stream_capturing()
            if not capturing and st["lr_host"] != float(group["lr"]):
                st["lr"].fill_(float(group["lr"]))
                st["lr_host"] = float(group["lr"])
            grads, params = st["grads"], st["params"]
            touched = []
            for i, p in enumerate(group["params"]):
                g = p.grad
                if g is None:
                    grads[i] = 0
                    continue
                touched.append(p)
                if g.dtype != torch.float32 or g.is_sparse or g.device != p.device:
                    raise RuntimeError("desco_amd.optim.Adam: gradients must be dense fp32 tensors on the parameter's device")
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                grads[i] = g.data_ptr()
                params[i] = p.data_ptr()
            b1, b2 = group["betas"]
            with ops._Timed("adam_step_kernel", 12.0 * float(st["sizes"].sum()), 28.0 * float(st["sizes"].sum()),
                            launches=(st["n"] + 127) // 128):
                _lib.check(L.desco_adam_step_f32(
                    st["n"], params.ctypes.data_as(ctypes.c_void_p), grads.ctypes.data_as(ctypes.c_void_p),
                    st["sizes"].ctypes.data_as(ctypes.c_void_p), ops._dev(st["m"], "m"), ops._dev(st["v"], "v"),
                    ops._dev(st["steps"], "steps"), ops._dev(st["arrivals"], "arrivals", torch.int32),
                    ops._dev(st["lr"], "lr"), b1, b2, group["eps"], group["weight_decay"], ops._stream()), "adam_step")
            # the launch writes the parameters through raw pointers: tell torch they changed (the folded-weight, head
            # and query-embedding caches of the inference path are keyed on tensor._version)
            increment_version(touched)
        return loss

    # ---- checkpointing: moments and ages per parameter, torch.optim.Adam's layout -----------------------------------
    def state_dict(self):
        packed, k = {}, 0
        groups = []
        for group in self.param_groups:
            st = self._dev_state.get(id(group))
            ids = list(range(k, k + len(group["params"])))
            groups.append({**{kk: vv for kk, vv in group.items() if kk != "params"}, "params": ids})
            if st is not None:
                off = 0
                for j, n in enumerate(st["sizes"]):
                    n = int(n)
                    shape = group["params"][j].shape
                    packed[k + j] = {"step": st["steps"][j].clone(), "exp_avg": st["m"][off:off + n].view(shape).clone(),
                                     "exp_avg_sq": st["v"][off:off + n].view(shape).clone()}
                    off += n
            k += len(group["params"])
        return {"state": packed, "param_groups": groups}

    def load_state_dict(self, sd):
        k = 0
        for group, saved in zip(self.param_groups, sd["param_groups"]):
            for kk, vv in saved.items():
                if kk != "params":
                    group[kk] = vv
            st = self._group_state(group)
            if st is not None:
                off = 0
                for j, n in enumerate(st["sizes"]):
                    n = int(n)
                    e = sd["state"].get(k + j)
                    if e is not None:
                        st["steps"][j] = float(e["step"])
                        st["m"][off:off + n] = e["exp_avg"].reshape(-1).to(st["m"].device)
                        st["v"][off:off + n] = e["exp_avg_sq"].reshape(-1).to(st["v"].device)
                    off += n
            k += len(group["params"])
        self.sync_lr()
