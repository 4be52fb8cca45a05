"""Torch-facing wrappers over the C ABI of libdesco_hip.so.

PyTorch is plumbing here: device memory (``torch.empty``), the current HIP stream and autograd
bookkeeping.  All arithmetic happens in the hand-written gfx950 kernels.  Every wrapper refuses
CPU tensors -- there is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


class LaunchProfiler:
    """Optional per-launch HIP-event timing on the launch stream (used by bench.py's roofline leg).

    When enabled every wrapper brackets its C-ABI call (one kernel launch; the few calls that are two to four
    launches say so) with two events recorded on the current stream and notes its algorithmic flops / bytes."""

    def __init__(self):
        self.enabled = False
        self.by_shape = False  # tools: key GEMM launches by shape as well ("kernel[m x k x n]")
        self.only = None       # a set of kernel names: bracket only those launches (bench.py's timed region)
        self.records = []      # (kernel, flops, bytes, start_event, end_event, launches)

    def reset(self):
        self.records = []

    def summary(self):
        """kernel -> dict(calls, launches, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for name, fl, by, e0, e1, nl in self.records:
            d = out.setdefault(name, {"calls": 0, "launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["launches"] += nl
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


PROFILER = LaunchProfiler()


class _Timed:
    __slots__ = ("name", "flops", "bytes", "e0", "launches")

    def __init__(self, name, flops=0.0, nbytes=0.0, shape=None, launches=1):
        if shape is not None and PROFILER.by_shape:
            name = f"{name}[{'x'.join(str(v) for v in shape)}]"
        self.name, self.flops, self.bytes, self.launches = name, flops, nbytes, launches

    def __enter__(self):
        self.e0 = None
        if PROFILER.enabled and (PROFILER.only is None or self.name in PROFILER.only):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILER.records.append((self.name, self.flops, self.bytes, self.e0, e1, self.launches))
        return False


def _stream() -> int:
    """The current stream of the CURRENT device.  The C ABI launches on the current HIP device, and
    ``_dev`` only admits operands that live there, so this is the operands' own stream."""
    return torch.cuda.current_stream().cuda_stream


_current_device = torch.cuda.current_device


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"desco_amd.ops: `{name}` must be a tensor on the MI355X (cuda) device; "
                           "there is no CPU fallback (use oracle/ for CPU checks)")
    if t.device.index != _current_device():
        # launching on device A's stream with device-B pointers would fault (or silently use peer
        # access with no stream ordering); every operand of a launch has to be on the current device
        raise RuntimeError(f"desco_amd.ops: `{name}` lives on {t.device} but the current device is "
                           f"cuda:{_current_device()}; call torch.cuda.set_device({t.device.index}) "
                           "(Trainer / InferencePipeline do) -- operands on mixed devices are rejected")
    if t.dtype != dtype:
        raise TypeError(f"desco_amd.ops: `{name}` must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def _rows(t: torch.Tensor, name: str):
    """(ptr, leading dim) of a 2-D fp32 row-major view with unit inner stride."""
    p = _dev(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"desco_amd.ops: `{name}` must be 2-D with unit inner stride")
    return p, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _opt(t: Optional[torch.Tensor], name: str, dtype=torch.float32):
    return None if t is None else _dev(t, name, dtype)


def linear_smallk(feat: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor],
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = feat @ wt + bias for tiny K (pre_mp = nn.Linear(input_dim, 64), gnn_model.py:131)."""
    m, k = feat.shape
    n = wt.shape[1]
    if out is None:
        out = torch.empty((m, n), device=feat.device, dtype=torch.float32)
    fp, ldf = _rows(feat, "feat")
    op, ldo = _rows(out, "out")
    wt = wt.contiguous()
    L = _lib.lib()
    with _Timed("linear_smallk_kernel", 2.0 * m * k * n, 4.0 * (m * k + m * n)):
        _lib.check(L.desco_linear_smallk_f32(fp, ldf, k, _dev(wt, "wt"), _opt(bias, "bias"), op,
                                             ldo, m, n, _stream()), "linear_smallk")
    return out


def csr_gather_sum(x: torch.Tensor, vrowptr: torch.Tensor, vcol: torch.Tensor, num_rows: int,
                   slots: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """agg[i, s*64:(s+1)*64] = sum over virtual row i*slots+s of x[vcol[e]]  -> [num_rows, slots*64]"""
    if out is None:
        out = torch.empty((num_rows, slots * 64), device=x.device, dtype=torch.float32)
    assert out.is_contiguous() and x.shape[1] == 64
    xp, ldx = _rows(x, "x")
    L = _lib.lib()
    # algorithmic bytes: every source row once + indices + the S aggregate rows
    nb = 256.0 * x.shape[0] + 4.0 * (vcol.numel() + num_rows * slots + 1) + 256.0 * num_rows * slots
    with _Timed("csr_gather_sum_kernel", float(vcol.numel()) * 64, nb):
        if vcol.numel() == 0:       # the kernel reads vcol[0] unconditionally (branch-free loads)
            vcol = torch.zeros(1, device=x.device, dtype=torch.int32)
        _lib.check(L.desco_csr_gather_sum_f32(xp, ldx, _dev(vrowptr, "vrowptr", torch.int32),
                                              _dev(vcol, "vcol", torch.int32), num_rows, slots,
                                              _dev(out, "out"), _stream()), "csr_gather_sum")
    return out


def csr_gather_sum_add(x: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, extra: torch.Tensor,
                       out: torch.Tensor) -> torch.Tensor:
    """out[j] = extra[j] + sum over row j of x[col[e]]  ([rows, 64]; extra may be out): the transposed gather
    of the training backward accumulating onto an existing gradient."""
    n = out.shape[0]
    xp, ldx = _rows(x, "x")
    ep, lde = _rows(extra, "extra")
    op, ldo = _rows(out, "out")
    assert x.shape[1] == 64 and out.shape[1] == 64 and extra.shape == out.shape
    with _Timed("csr_gather_sum_kernel", float(col.numel()) * 64, 256.0 * (x.shape[0] + 2 * n) + 4.0 * (col.numel() + n)):
        if col.numel() == 0:
            col = torch.zeros(1, device=x.device, dtype=torch.int32)
        _lib.check(_lib.lib().desco_csr_gather_sum_add_f32(xp, ldx, _dev(rowptr, "rowptr", torch.int32),
                                                          _dev(col, "col", torch.int32), n, ep, lde, op, ldo,
                                                          _stream()), "csr_gather_sum_add")
    return out


def shmp_bwd_dx(d: torch.Tensor, t_rowptr: torch.Tensor, t_col: torch.Tensor, num_count: int, off_count: int,
                off_canon: int, dpool: torch.Tensor, seg_id: torch.Tensor, dcanon: Optional[torch.Tensor],
                relu_src: Optional[torch.Tensor], mask_scale: float = 1.0) -> torch.Tensor:
    """Input-row gradient of one SHMP layer of the training trunk (desco_shmp_bwd_dx_f32): seed (pooling
    broadcast / anchor operand) + self block + transposed gather of the slot blocks, masked by relu' (times
    ``mask_scale``: the factor 1 / (1 - p) of a dropout behind the relu, whose kept elements are the positive ones)."""
    n = d.shape[0]
    out = torch.empty((n, 64), device=d.device, dtype=torch.float32)
    dp, ldd = _rows(d, "d")
    pp, ldp = _rows(dpool, "dpool")
    cp_, ldc = (None, 0) if dcanon is None else _rows(dcanon, "dcanon")
    rs = None if relu_src is None else _dev(relu_src, "relu_src")
    assert relu_src is None or (relu_src.is_contiguous() and tuple(relu_src.shape) == (n, 64))
    with _Timed("shmp_bwd_dx_kernel", float(t_col.numel()) * 64, 256.0 * (3 * n) + 4.0 * (t_col.numel() + 2 * n) +
                256.0 * t_col.numel() / 4):
        tc = t_col if t_col.numel() else torch.zeros(1, device=d.device, dtype=torch.int32)
        _lib.check(_lib.lib().desco_shmp_bwd_dx_f32(dp, ldd, _dev(t_rowptr, "t_rowptr", torch.int32),
                                                   _dev(tc, "t_col", torch.int32), n, int(num_count), int(off_count),
                                                   int(off_canon), pp, ldp, _dev(seg_id, "seg_id", torch.int32),
                                                   cp_, ldc, rs, float(mask_scale), _dev(out, "out"), _stream()),
                   "shmp_bwd_dx")
    return out


def add_rows(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst += src for 2-D fp32 views with unit inner stride (row strides free)."""
    assert dst.shape == src.shape and dst.shape[1] % 4 == 0
    dp, ldd = _rows(dst, "dst")
    sp, lds = _rows(src, "src")
    with _Timed("add_rows_kernel", float(dst.numel()), 12.0 * dst.numel()):
        _lib.check(_lib.lib().desco_add_rows_f32(dp, ldd, sp, lds, dst.shape[0], dst.shape[1], _stream()), "add_rows")
    return dst


def gemm(a1: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor] = None,
         a2: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.0,
         s: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act([a1 | a2] @ wt + bias[row % bias_rows] + s @ ws)  on the f32 MFMA path.

    ``wt`` is [(k1+k2), n] (torch weight transposed); ``bias`` is [n] or [bias_rows, n]."""
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = wt.shape[1]
    assert wt.shape[0] == k1 + k2 and wt.is_contiguous()
    if out is None:
        out = torch.empty((m, n), device=a1.device, dtype=torch.float32)
    a1p, lda1 = _rows(a1, "a1")
    a2p, lda2 = (None, 0) if a2 is None else _rows(a2, "a2")
    op, ldo = _rows(out, "out")
    bias_rows = 1
    if bias is not None:
        bias = bias.contiguous()
        bias_rows = 1 if bias.dim() == 1 else bias.shape[0]
    ns = 0
    if s is not None:
        assert s.is_contiguous() and s.shape[0] == m and ws is not None and ws.is_contiguous()
        ns = s.shape[1]
    L = _lib.lib()
    kk = k1 + k2
    with _Timed("gemm_f32_kernel", 2.0 * m * kk * n, 4.0 * (m * kk + kk * n + m * n + m * ns),
                (m, kk, n)):
        _lib.check(L.desco_gemm_f32(a1p, lda1, k1, a2p, lda2, k2, _dev(wt, "wt"), n,
                                    _opt(bias, "bias"), bias_rows, _opt(s, "s"), ns,
                                    _opt(ws, "ws"), act, slope, op, ldo, m, _stream()), "gemm")
    return out


def gemm_multi(problems) -> None:
    """Up to four independent ``gemm`` problems in one launch (desco_gemm_f32_multi).  ``problems``: dicts with the
    keyword arguments of ``gemm`` (a1, wt, bias, a2, act, slope, out -- ``out`` required) plus, for the backward pass,
    gate / gate_act / gate_slope (out = v * act'(gate), gate = the saved activation output) and accum (out += v);
    drop (a DropSite): the stored value times the dropout factor of its (row, col); empty ones (no rows) are skipped."""
    descs = (_lib.GemmDesc * len(problems))()
    flops = nbytes = 0.0
    for d, pr in zip(descs, problems):
        a1, wt, out, a2, bias = pr["a1"], pr["wt"], pr["out"], pr.get("a2"), pr.get("bias")
        m, k1 = a1.shape
        k2 = 0 if a2 is None else a2.shape[1]
        n = wt.shape[1]
        assert wt.shape[0] == k1 + k2 and wt.is_contiguous() and tuple(out.shape) == (m, n)
        d.m = m
        if m == 0:
            continue
        d.a1, d.lda1 = _rows(a1, "a1")
        d.k1, d.k2 = k1, k2
        if a2 is not None:
            d.a2, d.lda2 = _rows(a2, "a2")
        d.wt, d.n = _dev(wt, "wt"), n
        if bias is not None:
            assert bias.is_contiguous() and bias.dim() == 1
            d.bias, d.bias_rows = _dev(bias, "bias"), 1
        d.act, d.slope = pr.get("act", ACT_NONE), pr.get("slope", 0.0)
        d.c, d.ldc = _rows(out, "out")
        gate = pr.get("gate")
        if gate is not None:          # (backward: multiply by act'(saved output) in the epilogue)
            assert tuple(gate.shape) == (m, n)
            d.gate, d.ldg = _rows(gate, "gate")
            d.gate_act, d.gate_slope = pr["gate_act"], pr.get("gate_slope", 0.0)
        d.accum = int(bool(pr.get("accum", False)))
        drop = pr.get("drop")
        if drop is not None:          # (dropout factor of (row, col) on the stored value: DropSite.desc)
            d.drop = drop.desc()
        flops += 2.0 * m * (k1 + k2) * n
        nbytes += 4.0 * (m * (k1 + k2) + (k1 + k2) * n + m * n)
    L = _lib.lib()
    with _Timed("gemm_f32_multi_kernel", flops, nbytes):
        _lib.check(L.desco_gemm_f32_multi(len(problems), descs, _stream()), "gemm_multi")


def split_bf16_planes_t(wt: torch.Tensor) -> torch.Tensor:
    """[3, n, k] int16 planes of ``wt.t()`` for a [k, n] matrix with unit inner stride (rows may be strided): the n-major
    operand of ``gemm_split`` / ``gemm_split_desc`` from a weight kept as [in, out] (desco_split_bf16x3_t_f32)."""
    k, n = wt.shape
    planes = torch.empty((3, n, k), device=wt.device, dtype=torch.int16)
    wp, ldw = _rows(wt, "wt")
    _lib.check(_lib.lib().desco_split_bf16x3_t_f32(wp, k, n, ldw, _dev(planes, "planes", torch.int16), _stream()),
               "split_bf16x3_t")
    return planes


def gemm_split_desc(pr: dict, planes: torch.Tensor) -> None:
    """One ``gemm_multi`` problem (dict: a1, a2, bias, act, slope, out, gate / gate_act / gate_slope, drop -- no accum)
    on the bf16x6 pipe (desco_gemm_bf16x6_desc_f32); ``planes`` [3, n, k1 + k2] = the n-major split weight.
    ``s`` [m, ns] with ``ws`` [QV, ns, n]: out = act(product + bias + sum_j s[r, j] ws[r % QV, j, :]) -- ``affine_rows`` on
    the product, in its epilogue."""
    a1, out, a2, bias = pr["a1"], pr["out"], pr.get("a2"), pr.get("bias")
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = planes.shape[1]
    assert tuple(planes.shape) == (3, n, k1 + k2) and planes.is_contiguous() and tuple(out.shape) == (m, n)
    assert not pr.get("accum", False)
    if m == 0:
        return
    d = _lib.GemmDesc()
    d.m = m
    d.a1, d.lda1 = _rows(a1, "a1")
    d.k1, d.k2 = k1, k2
    if a2 is not None:
        d.a2, d.lda2 = _rows(a2, "a2")
    d.n = n
    if bias is not None:
        assert bias.is_contiguous() and bias.dim() == 1
        d.bias, d.bias_rows = _dev(bias, "bias"), 1
    d.act, d.slope = pr.get("act", ACT_NONE), pr.get("slope", 0.0)
    d.c, d.ldc = _rows(out, "out")
    gate = pr.get("gate")
    if gate is not None:
        assert tuple(gate.shape) == (m, n)
        d.gate, d.ldg = _rows(gate, "gate")
        d.gate_act, d.gate_slope = pr["gate_act"], pr.get("gate_slope", 0.0)
    drop = pr.get("drop")
    if drop is not None:
        d.drop = drop.desc()
    ws_rows = 1
    sc, ws = pr.get("s"), pr.get("ws")
    if sc is not None:
        ws_rows, ns = ws.shape[0], ws.shape[1]
        assert tuple(sc.shape) == (m, ns) and sc.is_contiguous() and tuple(ws.shape) == (ws_rows, ns, n) and ns <= 4
        ws = ws.contiguous()
        d.s, d.ns, d.ws = _dev(sc, "s"), ns, _dev(ws, "ws")
    with _Timed("gemm_split_kernel", 2.0 * m * (k1 + k2) * n, 4.0 * (m * (k1 + k2) + m * n * (2 if gate is not None else 1))):
        _lib.check(_lib.lib().desco_gemm_bf16x6_desc_f32(ctypes.byref(d), _dev(planes, "planes", torch.int16), ws_rows,
                                                         _stream()), "gemm_split_desc")


def split_bf16_planes_batch(w: torch.Tensor, transpose: bool, num_planes: int = 3) -> torch.Tensor:
    """[nb, P, rows, cols] (or, ``transpose``, [nb, P, cols, rows]) int16 planes of the nb contiguous matrices
    w[nb, rows, cols] in one launch (desco_split_bf16x3_batch_f32): a training trunk's stacked weights.  P = 3: the
    bf16x6 truncation split; P = 1: round-to-nearest bf16 (the bf16 training mode)."""
    nb, rows, cols = w.shape
    w = w.contiguous()
    planes = torch.empty((nb, num_planes, cols, rows) if transpose else (nb, num_planes, rows, cols), device=w.device,
                         dtype=torch.int16)
    _lib.check(_lib.lib().desco_split_bf16x3_batch_f32(_dev(w, "w"), nb, rows, cols, int(bool(transpose)), num_planes,
                                                       _dev(planes, "planes", torch.int16), _stream()), "split_bf16x3_batch")
    return planes


def gemm_split_multi(problems, planes) -> None:
    """Up to four independent ``gemm_split_desc`` problems (no scalar tail) in one launch (desco_gemm_bf16x6_multi_f32);
    ``planes[i]`` [3, n_i, k_i] = problem i's n-major split weight -- or [1, n_i, k_i] for all of them: plain bf16
    products (desco_gemm_bf16_multi_f32, the bf16 training mode)."""
    assert len(problems) == len(planes) <= 4
    npl = planes[0].shape[0]
    assert npl in (1, 3)
    descs = (_lib.GemmDesc * len(problems))()
    pl = (ctypes.c_void_p * len(problems))()
    flops = nbytes = 0.0
    for i, (d, pr, w) in enumerate(zip(descs, problems, planes)):
        a1, out, a2, bias = pr["a1"], pr["out"], pr.get("a2"), pr.get("bias")
        m, k1 = a1.shape
        k2 = 0 if a2 is None else a2.shape[1]
        n = w.shape[1]
        assert w.shape[0] == npl and tuple(w.shape[1:]) == (n, k1 + k2) and w.is_contiguous() and tuple(out.shape) == (m, n)
        assert not pr.get("accum", False) and pr.get("s") is None
        d.m = m
        if m == 0:
            continue
        pl[i] = _dev(w, "planes", torch.int16)
        d.a1, d.lda1 = _rows(a1, "a1")
        d.k1, d.k2 = k1, k2
        if a2 is not None:
            d.a2, d.lda2 = _rows(a2, "a2")
        d.n = n
        if bias is not None:
            assert bias.is_contiguous() and bias.dim() == 1
            d.bias, d.bias_rows = _dev(bias, "bias"), 1
        d.act, d.slope = pr.get("act", ACT_NONE), pr.get("slope", 0.0)
        d.c, d.ldc = _rows(out, "out")
        gate = pr.get("gate")
        if gate is not None:
            assert tuple(gate.shape) == (m, n)
            d.gate, d.ldg = _rows(gate, "gate")
            d.gate_act, d.gate_slope = pr["gate_act"], pr.get("gate_slope", 0.0)
        drop = pr.get("drop")
        if drop is not None:
            d.drop = drop.desc()
        flops += 2.0 * m * (k1 + k2) * n
        nbytes += 4.0 * (m * (k1 + k2) + m * n)
    with _Timed("gemm_split_multi_kernel", flops, nbytes):
        fn = _lib.lib().desco_gemm_bf16x6_multi_f32 if npl == 3 else _lib.lib().desco_gemm_bf16_multi_f32
        _lib.check(fn(len(problems), descs, pl, _stream()), "gemm_split_multi")


def linear_bwd_w_multi(problems) -> None:
    """Up to 16 independent ``linear_bwd_w`` problems in two launches (desco_linear_bwd_w_multi_f32).  ``problems``:
    dicts a1, a2 (or None), dz, dwt (contiguous [(k1+k2), n]), dbias ([n] or None)."""
    descs = (_lib.BwdWDesc * len(problems))()
    flops = nbytes = 0.0
    for d, pr in zip(descs, problems):
        a1, a2, dz, dwt, dbias = pr["a1"], pr.get("a2"), pr["dz"], pr["dwt"], pr.get("dbias")
        m, k1 = a1.shape
        k2 = 0 if a2 is None else a2.shape[1]
        n = dz.shape[1]
        assert dwt.is_contiguous() and tuple(dwt.shape) == (k1 + k2, n) and dz.shape[0] == m
        assert dbias is None or (dbias.is_contiguous() and dbias.numel() == n)
        d.a1, d.lda1 = _rows(a1, "a1")
        d.k1, d.k2 = k1, k2
        if a2 is not None:
            d.a2, d.lda2 = _rows(a2, "a2")
        d.dz, d.lddz = _rows(dz, "dz")
        d.m, d.n = m, n
        d.dwt, d.dbias = _dev(dwt, "dwt"), _opt(dbias, "dbias")
        flops += 2.0 * m * (k1 + k2) * n
        nbytes += 4.0 * (m * (k1 + k2) + 2 * m * n)
    L = _lib.lib()
    nb = L.desco_linear_bwd_w_multi_workspace(len(problems), descs)
    ws = torch.empty((max(nb // 4, 1),), device=problems[0]["dz"].device, dtype=torch.float32)
    with _Timed("linear_bwd_w_multi_kernel", flops, nbytes, launches=2):
        _lib.check(L.desco_linear_bwd_w_multi_f32(len(problems), descs, _dev(ws, "ws"), _stream()), "linear_bwd_w_multi")


def shmp_layer(x: torch.Tensor, vrowptr: torch.Tensor, vcol: torch.Tensor, row0: int,
               num_rows: int, slots_stored: int, slots_mfma: int, wt: torch.Tensor,
               bias: torch.Tensor, out: Optional[torch.Tensor], ytab: Optional[torch.Tensor] = None,
               ytab_row0: int = 0, out2: Optional[torch.Tensor] = None,
               pool: Optional[tuple] = None, row_absmax: Optional[torch.Tensor] = None,
               xself: Optional[torch.Tensor] = None, self_coef: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Fused gather + folded Linear + relu for rows [row0, row0+num_rows) (see desco_hip.h).
    ``xself`` [num_rows, >= 64] (f16x3 form): the launch's own rows (self block) read from this view -- row i of the launch
    at xself[i - row0] -- instead of from ``x``; ``out`` may then be None when ``out2`` is given.
    ``ytab`` [n_src, 64*st]: pre-transformed sources of the table slots sm .. sm+st-1.
    ``out2`` [num_rows, >=64] (optional): second copy of the produced rows (see desco_hip.h).
    ``wt``: fp32 [(sm+1)*64, 64] (f32 MFMA) or int16 planes [3, 64, (sm+1)*64] =
    ``split_bf16_planes(wt.t())`` (fp32-accurate bf16x6 arithmetic, sm <= 2).
    ``self_coef`` [slots_stored + 1, 64] (pooled f16x3 launch): the launch's own rows are recomputed from their slot
    degrees, relu(coef[S] + sum_s d_s coef[s]), instead of read (desco_shmp_layer_pool_table_f16x3_f32: ``x`` is then a
    table of the layer input's distinct rows and ``vcol`` addresses its rows).
    ``pool`` = (pool_bits, pool_slot, pool_part): also leave the per-(tile, segment) partial sums of
    the produced rows in ``pool_part`` (fused global_add_pool, finished by ``pool_reduce``); ``out``
    may then be None (rows not stored)."""
    f16 = isinstance(wt, F16Planes)
    x6 = f16 or wt.dtype == torch.int16
    if pool is not None:
        return _shmp_layer_pool(x, vrowptr, vcol, row0, num_rows, slots_stored, slots_mfma, wt, bias, out,
                                ytab, ytab_row0, pool, xself, self_coef)
    if self_coef is not None:
        raise ValueError("shmp_layer: self_coef is implemented by the pooled f16x3 launch only")
    if f16:
        assert wt.planes.is_contiguous() and wt.planes.shape == (2, 64, (slots_mfma + 1) * 64)
    elif x6:
        assert wt.is_contiguous() and wt.shape == (3, 64, (slots_mfma + 1) * 64)
    else:
        assert wt.is_contiguous() and wt.shape == ((slots_mfma + 1) * 64, 64)
    xp, ldx = _rows(x, "x")
    op, ldo = (None, 64) if out is None else _rows(out, "out")
    xsp, ldxs = None, 0
    if xself is not None:
        if not f16:
            raise ValueError("shmp_layer: xself is implemented by the f16x3 form only")
        assert xself.shape[0] == num_rows
        xsp, ldxs = _rows(xself, "xself")
        xsp -= 4 * ldxs * row0                       # the entry point indexes it by the global row id
    st, yp, ldy = 0, None, 0
    if ytab is not None:
        st = ytab.shape[1] // 64
        yp, ldy = _rows(ytab, "ytab")
    o2p, ldo2 = (None, 0) if out2 is None else _rows(out2, "out2")
    L = _lib.lib()
    # executed MFMA flops; compulsory bytes: x once + out once + this range's share of the indices
    fl = 2.0 * num_rows * (slots_mfma + 1) * 64 * 64
    nb = 512.0 * num_rows + 4.0 * (num_rows * slots_stored +
                                   vcol.numel() * num_rows / max((vrowptr.numel() - 1) // max(slots_stored, 1), 1))
    # profiler key = the device kernel's template instance (KB = sm + 1 weight blocks, ST table slots)
    with _Timed(shmp_kernel_name(slots_mfma + 1, st, x6, f16), fl, nb):
        head = (xp, ldx, _dev(vrowptr, "vrowptr", torch.int32), _dev(vcol, "vcol", torch.int32), row0, num_rows,
                slots_stored, slots_mfma, st)
        tail = (_dev(bias.contiguous(), "bias"), yp, ldy, ytab_row0, op, ldo, o2p, ldo2, _stream())
        if row_absmax is not None and not f16:
            raise ValueError("shmp_layer: row_absmax is an output of the f16x3 form only")
        if f16:
            rc = L.desco_shmp_layer_f16x3_f32(*head, _dev(wt.planes, "wt", torch.int16), _dev(wt.scale, "w_scale"),
                                              *tail[:-1], _opt(row_absmax, "row_absmax"), xsp, ldxs, tail[-1])
        else:
            fn = L.desco_shmp_layer_bf16x6_f32 if x6 else L.desco_shmp_layer_f32
            rc = fn(*head, _dev(wt, "wt", wt.dtype), *tail)
        _lib.check(rc, "shmp_layer")
    return out


def _shmp_layer_pool(x, vrowptr, vcol, row0, num_rows, slots_stored, slots_mfma, wt, bias, out, ytab,
                     ytab_row0, pool, xself=None, self_coef=None):
    bits, slot, part = pool
    f16 = isinstance(wt, F16Planes)
    if xself is not None or (self_coef is not None and not f16):
        raise ValueError("shmp_layer(pool=...): no xself; self_coef with the f16x3 form only")
    assert ytab is not None and (f16 or (wt.dtype == torch.int16 and wt.is_contiguous()))
    xp, ldx = _rows(x, "x")
    op, ldo = (None, 0) if out is None else _rows(out, "out")
    st = ytab.shape[1] // 64
    yp, ldy = _rows(ytab, "ytab")
    L = _lib.lib()
    fl = 2.0 * num_rows * (slots_mfma + 1) * 64 * 64
    # x once (+ out once when stored) + indices + the partial rows (about one per 32 rows + one per segment)
    # (table form: the input rows are not read from HBM either -- a few thousand table rows stand for all of them)
    nb = ((0.0 if self_coef is not None else 256.0) + (0.0 if out is None else 256.0)) * num_rows + 4.0 * (
        num_rows * slots_stored + vcol.numel() * num_rows / max((vrowptr.numel() - 1) // max(slots_stored, 1), 1))
    name = shmp_kernel_name(slots_mfma + 1, st, True, f16)
    if self_coef is not None:           # (its own instantiation of the device kernel: listed on its own)
        name = name[:-1] + ",selfdeg>"
    with _Timed(name, fl, nb):
        head = (xp, ldx, _dev(vrowptr, "vrowptr", torch.int32), _dev(vcol, "vcol", torch.int32), row0, num_rows,
                slots_stored, slots_mfma, st)
        tail = (_dev(bias.contiguous(), "bias"), yp, ldy, ytab_row0, op, ldo, _dev(bits, "pool_bits", torch.int32),
                _dev(slot, "pool_slot", torch.int32), _dev(part, "pool_part"), _stream())
        if self_coef is not None:
            assert tuple(self_coef.shape) == (slots_stored + 1, 64) and self_coef.is_contiguous()
            rc = L.desco_shmp_layer_pool_table_f16x3_f32(*head, _dev(wt.planes, "wt", torch.int16),
                                                         _dev(wt.scale, "w_scale"), *tail[:-1],
                                                         _dev(self_coef, "self_coef"), tail[-1])
        elif f16:
            rc = L.desco_shmp_layer_pool_f16x3_f32(*head, _dev(wt.planes, "wt", torch.int16),
                                                   _dev(wt.scale, "w_scale"), *tail)
        else:
            rc = L.desco_shmp_layer_pool_bf16x6_f32(*head, _dev(wt, "wt", torch.int16), *tail)
        _lib.check(rc, "shmp_layer_pool")
    return out


def pool_reduce(part: torch.Tensor, bits: torch.Tensor, slot: torch.Tensor, seg_ptr: torch.Tensor,
                num_seg: int, extra: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, tile_rows: Optional[int] = None) -> torch.Tensor:
    """out[b] = sum of segment b's partial rows (left by ``shmp_layer(pool=...)``) + extra[b];
    ``tile_rows`` = the tile size the index was built for (default: ``pool_tile_rows()``)."""
    if out is None:
        out = torch.empty((num_seg, 64), device=part.device, dtype=torch.float32)
    op, ldo = _rows(out, "out")
    ep, lde = (None, 0) if extra is None else _rows(extra, "extra")
    L = _lib.lib()
    with _Timed("pool_reduce_kernel", float(part.shape[0]) * 64,
                256.0 * part.shape[0] + 4.0 * (2 * bits.numel() + num_seg) + 512.0 * num_seg):
        _lib.check(L.desco_pool_reduce_f32(_dev(part, "pool_part"), _dev(bits, "pool_bits", torch.int32),
                                           _dev(slot, "pool_slot", torch.int32),
                                           _dev(seg_ptr, "seg_ptr", torch.int32), num_seg, ep, lde, op, ldo,
                                           pool_tile_rows() if tile_rows is None else tile_rows,
                                           _stream()), "pool_reduce")
    return out


def pool_reduce_multi(parts, bits: torch.Tensor, slot: torch.Tensor, seg_ptr: torch.Tensor, num_seg: int, extras, outs,
                      tile_rows: Optional[int] = None) -> None:
    """``pool_reduce`` for several layers in one launch (desco_pool_reduce_multi_f32, groups of 8): parts[i] -> outs[i]
    (+ extras[i], or None); the layers share the tile index and the segments.  Bit-identical to the single calls."""
    L = _lib.lib()
    tr = pool_tile_rows() if tile_rows is None else tile_rows
    for i0 in range(0, len(parts), 8):
        ps, es, os_ = parts[i0:i0 + 8], extras[i0:i0 + 8], outs[i0:i0 + 8]
        n = len(ps)
        pa, ea, oa = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        lde = ldo = None
        for i, (p_, e_, o_) in enumerate(zip(ps, es, os_)):
            pa[i] = _dev(p_, "pool_part")
            op, lo = _rows(o_, "out")
            oa[i] = op
            assert ldo in (None, lo)
            ldo = lo
            if e_ is not None:
                ep, le = _rows(e_, "extra")
                ea[i] = ep
                assert lde in (None, le)
                lde = le
        with _Timed("pool_reduce_kernel", float(sum(p_.shape[0] for p_ in ps)) * 64,
                    sum(256.0 * p_.shape[0] + 4.0 * (2 * bits.numel() + num_seg) + 512.0 * num_seg for p_ in ps)):
            _lib.check(L.desco_pool_reduce_multi_f32(n, pa, _dev(bits, "pool_bits", torch.int32),
                                                     _dev(slot, "pool_slot", torch.int32),
                                                     _dev(seg_ptr, "seg_ptr", torch.int32), num_seg, ea, lde or 0, oa, ldo,
                                                     tr, _stream()), "pool_reduce_multi")


def pool_post(anch: torch.Tensor, parts, bits: torch.Tensor, slot: torch.Tensor, seg_ptr: torch.Tensor, x0: torch.Tensor,
              w_planes: torch.Tensor, bias: torch.Tensor, act: int, slope: float) -> torch.Tensor:
    """act(pooled @ W^T + bias) [B, 64] with pooled = [anch block 0 + rows * x0 | anch block l + segment sums of parts[l-1]]
    formed inside the product (desco_pool_post_bf16x6_f32): ``parts`` = the pooled layers' partial arrays (layers 1..L),
    ``w_planes`` = split_bf16_planes(W) [3, 64, 64 (L + 1)].  Every segment must span at most three 16-row tiles."""
    B, L = anch.shape[0], len(parts)
    assert tuple(w_planes.shape) == (3, 64, 64 * (L + 1)) and w_planes.is_contiguous() and anch.shape[1] >= 64 * (L + 1)
    assert x0.is_contiguous() and x0.numel() == 64 and pool_tile_rows() == 16
    out = torch.empty((B, 64), device=anch.device, dtype=torch.float32)
    ap, lda = _rows(anch, "anch")
    pa = (ctypes.c_void_p * L)(*[_dev(p_, "pool_part") for p_ in parts])
    with _Timed("gemm_split_kernel", 2.0 * B * 64 * 64 * (L + 1),
                4.0 * B * 64 * (L + 2) + sum(256.0 * p_.shape[0] for p_ in parts)):
        _lib.check(_lib.lib().desco_pool_post_bf16x6_f32(
            ap, lda, L, _dev(w_planes, "w_planes", torch.int16), 64, _dev(bias.contiguous(), "bias"), act, slope,
            _dev(out, "out"), 64, B, _dev(seg_ptr, "seg_ptr", torch.int32), _dev(bits, "pool_bits", torch.int32),
            _dev(slot, "pool_slot", torch.int32), pa, _dev(x0, "x0"), 16, _stream()), "pool_post")
    return out


def post_mp_tail(x: torch.Tensor, w1, b1, w2, b2, w3, b3, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W3 relu(W2 relu(W1 x + b1) + b2) + b3 for the [64 -> 64 -> 256 -> 64] tail of post_mp in one launch
    (desco_post_mp_tail_f16x3_f32); the weights as ``split_f16_planes`` of the [out, in] matrices."""
    m = x.shape[0]
    assert x.shape[1] == 64 and all(isinstance(w, F16Planes) for w in (w1, w2, w3))
    assert tuple(w1.shape) == (2, 64, 64) and tuple(w2.shape) == (2, 256, 64) and tuple(w3.shape) == (2, 64, 256)
    if out is None:
        out = torch.empty((m, 64), device=x.device, dtype=torch.float32)
    if m == 0:
        return out
    xp, ldx = _rows(x, "x")
    op, ldo = _rows(out, "out")
    bp = [None if b is None else _dev(b.contiguous(), "bias") for b in (b1, b2, b3)]
    with _Timed("post_tail_kernel", 2.0 * m * (64 * 64 + 64 * 256 + 256 * 64), 4.0 * m * 128):
        _lib.check(_lib.lib().desco_post_mp_tail_f16x3_f32(
            xp, ldx, m, _dev(w1.planes, "w1", torch.int16), _dev(w1.scale, "w1_scale"), bp[0],
            _dev(w2.planes, "w2", torch.int16), _dev(w2.scale, "w2_scale"), bp[1],
            _dev(w3.planes, "w3", torch.int16), _dev(w3.scale, "w3_scale"), bp[2], op, ldo, _stream()), "post_mp_tail")
    return out


def shmp_kernel_name(kb: int, st: int, x6: bool, f16: bool = False) -> str:
    """Profiler key of a fused-layer launch: the kernel family that runs it (16-row wave tiles for
    the bf16x6 form unless DESCO_SHMP_ROWS=32; always for the fp16 three-product form) and its
    <weight blocks, table slots>."""
    if f16:
        return f"shmp_layer16_kernel<{kb},{st},f16x3>"
    if x6 and pool_tile_rows() == 16:
        return f"shmp_layer16_kernel<{kb},{st}>"
    return f"shmp_layer_f32_kernel<{kb},{st},{'x6' if x6 else 'f32'}>"


def pool_tile_rows() -> int:
    """Rows per wave tile of the fused layer kernel (16, or 32 with DESCO_SHMP_ROWS=32): the
    granularity of the fused-pooling index (``NeighborhoodBatch.pool_index``)."""
    return int(_lib.lib().desco_shmp_pool_tile_rows())


def linear64_planes(w: torch.Tensor) -> torch.Tensor:
    """[N/64, 3, 64, 64] int16: per 64-row block of a [N, 64] weight its bf16 planes (operand of
    ``linear64``)."""
    n, k = w.shape
    assert k == 64 and n % 64 == 0
    return torch.stack([split_bf16_planes(w[j:j + 64]) for j in range(0, n, 64)]).contiguous()


def linear64(x: torch.Tensor, planes: torch.Tensor, bias: Optional[torch.Tensor] = None,
             act: int = ACT_NONE, slope: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(x @ W.T + bias) for a [N, 64] weight given as ``linear64_planes(W)``: streaming
    launches, two 64-column output blocks per pass over x (memory-shaped projections, see desco_hip.h)."""
    m = x.shape[0]
    nb = planes.shape[0]
    assert x.shape[1] == 64 and planes.shape[1:] == (3, 64, 64) and planes.dtype == torch.int16
    if out is None:
        out = torch.empty((m, 64 * nb), device=x.device, dtype=torch.float32)
    xp, ldx = _rows(x, "x")
    op, ldo = _rows(out, "out")
    if bias is not None:
        bias = bias.contiguous()
    L = _lib.lib()
    with _Timed("linear64_kernel", 2.0 * m * 64 * 64 * nb, 256.0 * m * ((nb + 1) // 2) + 256.0 * m * nb,
                ("linear64", m, nb)):
        _lib.check(L.desco_linear64_bf16x6_f32(xp, ldx, _dev(planes, "planes", torch.int16), nb,
                                               _opt(bias, "bias"), act, slope, op, ldo, m, _stream()),
                   "linear64")
    return out


def degree_affine(vrowptr: torch.Tensor, row0: int, num_rows: int, slots: int, coef: torch.Tensor,
                  act: int, slope: float, out: torch.Tensor,
                  extra: Optional[torch.Tensor] = None, out_row0: Optional[int] = None,
                  row_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[row0+i] = act(coef[slots] + sum_s deg_s(i) * coef[s]) + extra[i]   (see desco_hip.h).
    ``out_row0``: row of ``out`` that receives row ``row0`` (default ``row0``: out is indexed like the CSR)."""
    assert coef.is_contiguous() and coef.shape == (slots + 1, 64)
    op, ldo = _rows(out, "out")
    if out_row0 is not None:
        assert out.shape[0] >= out_row0 + num_rows
        op += 4 * ldo * (out_row0 - row0)          # the entry point indexes out by the CSR row id
    ep, lde = (None, 0) if extra is None else _rows(extra, "extra")
    L = _lib.lib()
    with _Timed("degree_affine_kernel", 2.0 * num_rows * slots * 64,
                256.0 * num_rows + 4.0 * num_rows * (slots + 1)):
        _lib.check(L.desco_degree_affine_f32(_dev(vrowptr, "vrowptr", torch.int32), row0, num_rows,
                                             slots, _dev(coef, "coef"), act, slope, ep, lde, op, ldo,
                                             _opt(row_absmax, "row_absmax"), _stream()), "degree_affine")
    return out


def degree_affine_pool(vrowptr: torch.Tensor, num_rows: int, slots: int, coef: torch.Tensor, act: int, slope: float,
                       out: Optional[torch.Tensor], pool: tuple) -> None:
    """``degree_affine`` for rows [0, num_rows) with the rows' segment sums fused in (desco_degree_affine_pool_f32):
    ``pool`` = (pool_bits, pool_slot, pool_part) as for ``shmp_layer(pool=...)`` with 16-row tiles; ``out`` may be None."""
    assert coef.is_contiguous() and coef.shape == (slots + 1, 64) and pool_tile_rows() == 16
    bits, slot, part = pool
    op, ldo = (None, 64) if out is None else _rows(out, "out")
    with _Timed("degree_affine_kernel", 2.0 * num_rows * slots * 64, 256.0 * num_rows + 4.0 * num_rows * (slots + 1)):
        _lib.check(_lib.lib().desco_degree_affine_pool_f32(
            _dev(vrowptr, "vrowptr", torch.int32), num_rows, slots, _dev(coef, "coef"), act, slope, op, ldo,
            _dev(bits, "pool_bits", torch.int32), _dev(slot, "pool_slot", torch.int32), _dev(part, "pool_part"),
            _stream()), "degree_affine_pool")


def gemm_split(a1: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
               a2: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act([a1 | a2] @ w.T + bias) with fp32-level accuracy on the bf16 matrix pipe (bf16x6).
    ``w`` is the pre-split N-MAJOR weight: ``split_bf16_planes(weight)`` of torch's [n, k1+k2]
    (a float weight is accepted and split on the fly, for tests)."""
    if w.dtype != torch.int16:
        w = split_bf16_planes(w)
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = w.shape[1]
    assert w.dim() == 3 and w.shape[0] == 3 and w.shape[2] == k1 + k2 and w.is_contiguous()
    if out is None:
        out = torch.empty((m, n), device=a1.device, dtype=torch.float32)
    a1p, lda1 = _rows(a1, "a1")
    a2p, lda2 = (None, 0) if a2 is None else _rows(a2, "a2")
    op, ldo = _rows(out, "out")
    bias_rows = 1
    if bias is not None:
        bias = bias.contiguous()
        bias_rows = 1 if bias.dim() == 1 else bias.shape[0]
    L = _lib.lib()
    kk = k1 + k2
    with _Timed("gemm_split_kernel", 2.0 * m * kk * n, 4.0 * (m * kk + kk * n + m * n), (m, kk, n)):
        _lib.check(L.desco_gemm_bf16x6_f32(a1p, lda1, k1, a2p, lda2, k2, _dev(w, "w", torch.int16), n,
                                           _opt(bias, "bias"), bias_rows, None, 0, None, act, slope,
                                           op, ldo, m, _stream()), "gemm_split")
    return out


class F16Planes:
    """Weight operand of the f16x3 kernels: (hi, lo) fp16 planes [2, *w.shape] of scale * w and the device pair
    {scale, 1/scale} (one power of two per matrix), made once per weight version by ``split_f16_planes``."""
    __slots__ = ("planes", "scale")

    def __init__(self, planes, scale):
        self.planes, self.scale = planes, scale

    @property
    def shape(self):
        return self.planes.shape


def split_f16_planes(w: torch.Tensor) -> F16Planes:
    w = w.contiguous()
    planes = torch.empty((2,) + tuple(w.shape), device=w.device, dtype=torch.int16)
    scale = torch.empty((2,), device=w.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.desco_split_f16x2_f32(_dev(w, "w"), w.numel(), _dev(planes, "planes", torch.int16),
                                       _dev(scale, "scale"), _stream()), "split_f16x2")
    return F16Planes(planes, scale)


def row_scale_f16(a1: torch.Tensor, a2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[m] row bounds max_k |[a1 | a2][i, k]| (the per-row operand of ``gemm_f16x3``, which derives its power-of-two
    scales from them)."""
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    out = torch.empty((m,), device=a1.device, dtype=torch.float32)
    a1p, lda1 = _rows(a1, "a1")
    a2p, lda2 = (None, 0) if a2 is None else _rows(a2, "a2")
    with _Timed("row_scale_kernel", 0.0, 4.0 * m * (k1 + k2 + 1)):
        _lib.check(_lib.lib().desco_row_absmax_f32(a1p, lda1, k1, a2p, lda2, k2, m, _dev(out, "row_scale"),
                                                  _stream()), "row_scale_f16")
    return out


def gemm_f16x3(a1: torch.Tensor, w, bias: Optional[torch.Tensor] = None,
               a2: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.0,
               out: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act([a1 | a2] @ w.T + bias) with fp32-level accuracy on the fp16 matrix pipe in three products
    (csrc/gemm_f16x3.hip).  ``w``: ``split_f16_planes(weight)`` of torch's [n, k1+k2] weight (a float weight is
    split on the fly, for tests); ``row_scale``: ``row_scale_f16(a1, a2)`` when the caller already has it."""
    if not isinstance(w, F16Planes):
        w = split_f16_planes(w)
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = w.planes.shape[1]
    assert w.planes.dim() == 3 and w.planes.shape[0] == 2 and w.planes.shape[2] == k1 + k2
    if out is None:
        out = torch.empty((m, n), device=a1.device, dtype=torch.float32)
    if m == 0:
        return out
    if row_scale is None:
        row_scale = row_scale_f16(a1, a2)
    a1p, lda1 = _rows(a1, "a1")
    a2p, lda2 = (None, 0) if a2 is None else _rows(a2, "a2")
    op, ldo = _rows(out, "out")
    bias_rows = 1
    if bias is not None:
        bias = bias.contiguous()
        bias_rows = 1 if bias.dim() == 1 else bias.shape[0]
    L = _lib.lib()
    kk = k1 + k2
    with _Timed("gemm_f16x3_kernel", 2.0 * m * kk * n, 4.0 * (m * kk + kk * n + m * n), (m, kk, n)):
        _lib.check(L.desco_gemm_f16x3_f32(a1p, lda1, k1, a2p, lda2, k2, _dev(w.planes, "w", torch.int16),
                                          _dev(w.scale, "w_scale"), n, _opt(bias, "bias"), bias_rows, None, 0,
                                          None, act, slope, op, ldo, m, _dev(row_scale, "row_scale"), _stream()),
                   "gemm_f16x3")
    return out


def vcsr_transpose_sym(vrowptr: torch.Tensor, vcol: torch.Tensor, num_rows: int, slots: int,
                       num_count: Optional[int] = None):
    """(t_rowptr [num_rows+1], t_col [E]): the backward gather's index of a symmetric virtual-row
    CSR, built on the device (see desco_hip.h)."""
    t_rowptr = torch.empty(num_rows + 1, device=vrowptr.device, dtype=torch.int32)
    t_col = torch.empty(vcol.numel(), device=vrowptr.device, dtype=torch.int32)
    L = _lib.lib()
    _lib.check(L.desco_vcsr_transpose_sym(_dev(vrowptr, "vrowptr", torch.int32),
                                          _dev(vcol, "vcol", torch.int32) if vcol.numel() else None,
                                          num_rows, slots, num_rows if num_count is None else num_count,
                                          _dev(t_rowptr, "t_rowptr", torch.int32),
                                          _dev(t_col, "t_col", torch.int32) if vcol.numel() else None,
                                          _stream()), "vcsr_transpose_sym")
    return t_rowptr, t_col


def segment_ids(seg_ptr: torch.Tensor, num_rows: int) -> torch.Tensor:
    """seg_id [num_rows] int32 of contiguous segments (device)."""
    out = torch.empty(num_rows, device=seg_ptr.device, dtype=torch.int32)
    L = _lib.lib()
    _lib.check(L.desco_segment_ids(_dev(seg_ptr, "seg_ptr", torch.int32), seg_ptr.numel() - 1,
                                   _dev(out, "seg_id", torch.int32) if num_rows else None, _stream()),
               "segment_ids")
    return out


def round_bf16(w: torch.Tensor) -> torch.Tensor:
    """int16 tensor of w's shape: round-to-nearest-even bf16 bit patterns (weight operand of
    ``gemm_bf16``)."""
    w = w.contiguous()
    out = torch.empty(w.shape, device=w.device, dtype=torch.int16)
    L = _lib.lib()
    _lib.check(L.desco_round_bf16_f32(_dev(w, "w"), w.numel(), _dev(out, "out", torch.int16),
                                      _stream()), "round_bf16")
    return out


def gemm_bf16(a1: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
              a2: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act([a1 | a2] @ w.T + bias) with bf16 operands (A rounded in the kernel, ``w`` =
    ``round_bf16(weight)`` of torch's [n, k1+k2], or a float weight rounded on the fly), fp32
    accumulation and output: the matrix product of the bf16 training mode."""
    if w.dtype != torch.int16:
        w = round_bf16(w)
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = w.shape[0]
    assert w.dim() == 2 and w.shape[1] == k1 + k2 and w.is_contiguous()
    if out is None:
        out = torch.empty((m, n), device=a1.device, dtype=torch.float32)
    a1p, lda1 = _rows(a1, "a1")
    a2p, lda2 = (None, 0) if a2 is None else _rows(a2, "a2")
    op, ldo = _rows(out, "out")
    bias_rows = 1
    if bias is not None:
        bias = bias.contiguous()
        bias_rows = 1 if bias.dim() == 1 else bias.shape[0]
    L = _lib.lib()
    kk = k1 + k2
    with _Timed("gemm_bf16_kernel", 2.0 * m * kk * n, 4.0 * (m * kk + m * n) + 2.0 * kk * n, (m, kk, n)):
        _lib.check(L.desco_gemm_bf16_f32(a1p, lda1, k1, a2p, lda2, k2, _dev(w, "w", torch.int16), n,
                                         _opt(bias, "bias"), bias_rows, None, 0, None, act, slope,
                                         op, ldo, m, _stream()), "gemm_bf16")
    return out


def segment_sum(x: torch.Tensor, seg_ptr: torch.Tensor, num_seg: int,
                extra: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = sum(x[seg_ptr[b]:seg_ptr[b+1]]) + extra[b]   (global_add_pool, gnn_model.py:107)."""
    ncols = x.shape[1]
    if out is None:
        out = torch.empty((num_seg, ncols), device=x.device, dtype=torch.float32)
    xp, ldx = _rows(x, "x")
    op, ldo = _rows(out, "out")
    ep, lde = (None, 0) if extra is None else _rows(extra, "extra")
    L = _lib.lib()
    with _Timed("segment_sum_kernel", float(x.shape[0]) * ncols,
                4.0 * (x.shape[0] * ncols + 2 * num_seg * ncols + num_seg)):
        _lib.check(L.desco_segment_sum_f32(xp, ldx, ncols, _dev(seg_ptr, "seg_ptr", torch.int32),
                                           num_seg, ep, lde, op, ldo, _stream()), "segment_sum")
    return out


def segment_sum_layers(xall: torch.Tensor, num_rows: int, seg_ptr: torch.Tensor, num_seg: int,
                       extra: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """out[b, 64 l : 64 l + 64] = sum over segment b of xall[l, r, :] (r < num_rows) + extra[b, 64 l : ...] for every
    layer l of ``xall`` [Lx, N, 64] in one launch (``out`` / ``extra``: [num_seg, >= 64 Lx] views)."""
    Lx, N, H = xall.shape
    assert H == 64 and xall.is_contiguous() and num_rows <= N
    op, ldo = _rows(out, "out")
    ep, lde = (None, 0) if extra is None else _rows(extra, "extra")
    with _Timed("segment_sum_kernel", float(num_rows) * 64 * Lx, 4.0 * Lx * (num_rows * 64 + 2 * num_seg * 64)):
        _lib.check(_lib.lib().desco_segment_sum_layers_f32(_dev(xall, "xall"), 64, N * 64, Lx,
                                                          _dev(seg_ptr, "seg_ptr", torch.int32), num_seg, ep, lde,
                                                          op, ldo, _stream()), "segment_sum_layers")
    return out


def count_head(t: torch.Tensor, qh: torch.Tensor, w2: torch.Tensor, b2, slope: float,
               exp2_minus_1: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,Q] logits (or 2**logit - 1) of the separable count head (lightning_model.py:176-221).
    ``b2``: a float, or a 0-d / 1-element device tensor read by the kernel (no host sync).
    ``out``: optional contiguous [B, Q] destination (a slice of a persistent result buffer)."""
    B, hid = t.shape
    Q = qh.shape[0]
    if out is None:
        out = torch.empty((B, Q), device=t.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, Q) or not out.is_contiguous():
        raise ValueError("count_head: `out` must be a contiguous [B, Q] tensor")
    tp, ldt = _rows(t, "t")
    qp, ldq = _rows(qh, "qh")
    L = _lib.lib()
    with _Timed("count_head_kernel", 4.0 * B * Q * hid, 4.0 * (B * hid + Q * hid + B * Q)):
        b2_dev = None
        if isinstance(b2, torch.Tensor):
            b2_dev, b2 = _dev(b2.detach().reshape(1).contiguous(), "b2"), 0.0
        _lib.check(L.desco_count_head_f32(tp, ldt, qp, ldq, hid, _dev(w2.contiguous(), "w2"), b2,
                                          b2_dev, slope, int(exp2_minus_1), _dev(out, "out"), Q, B, Q,
                                          _stream()), "count_head")
    return out


def count_head_emb(emb: torch.Tensor, wt: "F16Planes", qh: torch.Tensor, w2: torch.Tensor, b2, slope: float,
                   exp2_minus_1: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``count_head(emb @ Wt.T, qh, ...)`` without the [B, 256] tensor in between (desco_count_head_emb_f16x3_f32):
    ``wt`` = split_f16_planes of count_model.0's target half [256, 64]; 29 queries, 256 hidden features."""
    B, Q, hid = emb.shape[0], qh.shape[0], qh.shape[1]
    assert emb.shape[1] == 64 and isinstance(wt, F16Planes) and tuple(wt.shape) == (2, hid, 64)
    if out is None:
        out = torch.empty((B, Q), device=emb.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, Q) or not out.is_contiguous():
        raise ValueError("count_head_emb: `out` must be a contiguous [B, Q] tensor")
    if B == 0:
        return out
    ep, lde = _rows(emb, "emb")
    qp, ldq = _rows(qh, "qh")
    with _Timed("count_head_emb_kernel", 2.0 * B * 64 * hid + 4.0 * B * Q * hid, 4.0 * (B * 64 + Q * hid + B * Q)):
        b2_dev = None
        if isinstance(b2, torch.Tensor):
            b2_dev, b2 = _dev(b2.detach().reshape(1).contiguous(), "b2"), 0.0
        _lib.check(_lib.lib().desco_count_head_emb_f16x3_f32(
            ep, lde, B, _dev(wt.planes, "wt", torch.int16), _dev(wt.scale, "wt_scale"), qp, ldq, hid,
            _dev(w2.contiguous(), "w2"), b2, b2_dev, slope, int(exp2_minus_1), _dev(out, "out"), Q, Q, _stream()),
            "count_head_emb")
    return out


def scatter_rows(src: torch.Tensor, rows: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[rows[b], :] = src[b, :]  (GossipDataset.apply_neighborhood_count, workload.py:107-112)."""
    sp, lds = _rows(src, "src")
    dp, ldd = _rows(dst, "dst")
    L = _lib.lib()
    with _Timed("scatter_rows_kernel", 0.0, 8.0 * src.numel() + 4.0 * src.shape[0]):
        _lib.check(L.desco_scatter_rows_f32(sp, lds, _dev(rows, "rows", torch.int32),
                                            src.shape[0], src.shape[1], dp, ldd, _stream()),
                   "scatter_rows")
    torch.autograd.graph.increment_version(dst)      # written through its raw pointer: caches keyed on the version see it
    return dst


def gossip_layer0(x: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, g0, g1, p, r, t, z):
    """Closed-form first gossip layer for all queries -> (h1 [N*Q,64], scal [N*Q,2])."""
    N, Q = x.shape
    h1 = torch.empty((N * Q, 64), device=x.device, dtype=torch.float32)
    scal = torch.empty((N * Q, 2), device=x.device, dtype=torch.float32)
    xp, ldx = _rows(x, "x")
    L = _lib.lib()
    with _Timed("gossip_layer0_kernel", 8.0 * N * Q * 64,
                4.0 * (N * Q + col.numel() + N) + N * Q * (256.0 + 8.0)):
      _lib.check(L.desco_gossip_layer0_f32(xp, ldx, _dev(rowptr, "rowptr", torch.int32),
                                         _dev(col, "col", torch.int32), N, Q,
                                         _dev(g0.contiguous(), "g0"), _dev(g1.contiguous(), "g1"),
                                         _dev(p.contiguous(), "p"), _dev(r.contiguous(), "r"),
                                         _dev(t.contiguous(), "t"), _dev(z.contiguous(), "z"),
                                         _dev(h1, "h1"), _dev(scal, "scal"), _stream()),
               "gossip_layer0")
    return h1, scal


def gossip_gather(h: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int,
                  num_q: int, g: Optional[torch.Tensor]) -> torch.Tensor:
    """out[i,q,:] = sum_j (j<i ? g[q] : 1-g[q]) * h[j,q,:]   (h: [N*Q, 64] contiguous);
    g=None: the signed form sum_{j<i} h[j] - sum_{j>i} h[j]."""
    assert h.is_contiguous()
    out = torch.empty_like(h)
    L = _lib.lib()
    with _Timed("gossip_gather_kernel", 2.0 * col.numel() * num_q * 64,
                512.0 * num_nodes * num_q + 4.0 * (col.numel() + num_nodes)):
      _lib.check(L.desco_gossip_gather_f32(_dev(h, "h"), _dev(rowptr, "rowptr", torch.int32),
                                         _dev(col, "col", torch.int32), num_nodes, num_q,
                                         None if g is None else _dev(g.contiguous(), "g"),
                                         _dev(out, "out"), _stream()),
               "gossip_gather")
    return out


def split_bf16_planes(w: torch.Tensor) -> torch.Tensor:
    """[3, *w.shape] int16: the truncation split w = hi + mid + lo into bf16 bit patterns -- the
    weight operand format of the bf16x6 kernels (one launch per weight version)."""
    w = w.contiguous()
    planes = torch.empty((3,) + tuple(w.shape), device=w.device, dtype=torch.int16)
    L = _lib.lib()
    _lib.check(L.desco_split_bf16x3_f32(_dev(w, "w"), w.numel(), _dev(planes, "planes", torch.int16),
                                        _stream()), "split_bf16x3")
    return planes


def gossip_scalars(x: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, g0, g1) -> torch.Tensor:
    """scal4[i*Q+q] = (a0, b0, a1, x[i,q]) -- the per-(node, query) scalars of the gossip stage."""
    N, Q = x.shape
    scal = torch.empty((N * Q, 4), device=x.device, dtype=torch.float32)
    xp, ldx = _rows(x, "x")
    L = _lib.lib()
    with _Timed("gossip_scalars_kernel", 6.0 * col.numel() * Q, 4.0 * (col.numel() * (Q + 1) + N) + 20.0 * N * Q):
        _lib.check(L.desco_gossip_scalars_f32(xp, ldx, _dev(rowptr, "rowptr", torch.int32),
                                              _dev(col, "col", torch.int32), N, Q,
                                              _dev(g0.contiguous(), "g0"), _dev(g1.contiguous(), "g1"),
                                              _dev(scal, "scal"), _stream()), "gossip_scalars")
    return scal


# executed MFMA flops per (node, query) row of the fused gossip kernel: K=128,128,64 (N=64), 64 (N=256)
GOSSIP_FUSED_FLOPS_PER_ROW = 2.0 * 64 * (128 + 128 + 64) + 2.0 * 64 * 256


def gossip_tile_order(rowptr: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """uint8 [ceil(N/128)*128]: per 128-node tile the degree-balanced row order of the fused kernel's neighbour-sum
    phase (desco_gossip_tile_order); depends on the CSR only, so a batch computes it once."""
    tiles = (num_nodes + 127) // 128
    perm = torch.empty((tiles * 128,), device=rowptr.device, dtype=torch.uint8)
    if num_nodes:
        _lib.check(_lib.lib().desco_gossip_tile_order(_dev(rowptr, "rowptr", torch.int32), num_nodes,
                                                      _dev(perm, "perm", torch.uint8), _stream()),
                   "gossip_tile_order")
    return perm


def gossip_fused(scal: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int,
                 num_q: int, v: dict, tile_perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One on-chip pass per (128-node tile, query): returns pred [N, Q] (see desco_hip.h)."""
    out = torch.empty((num_nodes, num_q), device=scal.device, dtype=torch.float32)
    if tile_perm is not None and tile_perm.numel() < ((num_nodes + 127) // 128) * 128:
        raise ValueError("gossip_fused: tile_perm is shorter than the tiles of this batch")
    L = _lib.lib()
    names = ("g1", "p", "z", "zp", "r", "t", "u", "tp", "d1", "w1s", "wps", "w3s", "b3", "w5s", "b5", "w7")
    ptrs = []
    for n in names:
        if not v[n].is_contiguous():
            raise ValueError(f"gossip_fused: operand {n} must be contiguous")
        ptrs.append(_dev(v[n], n, torch.int16 if n in ("w1s", "wps", "w3s", "w5s") else torch.float32))
    rows = float(num_nodes) * num_q
    with _Timed("gossip_fused_kernel", rows * GOSSIP_FUSED_FLOPS_PER_ROW,
                rows * 20.0 + 4.0 * (col.numel() * (1 + 4 * num_q) + num_nodes)):
        _lib.check(L.desco_gossip_fused_f32(_dev(scal, "scal"), _dev(rowptr, "rowptr", torch.int32),
                                            _dev(col, "col", torch.int32), num_nodes, num_q, *ptrs,
                                            float(v["b7"]), _dev(out, "out"),
                                            None if tile_perm is None else _dev(tile_perm, "tile_perm", torch.uint8),
                                            _stream()),
                   "gossip_fused")
    return out


def gossip_f16_stream(w1: F16Planes, wp: F16Planes, w3: F16Planes, w5: F16Planes):
    """(wstream int16 [9, 2, 4096], winv float32 [4]): the nine 64x64 weight blocks of the fp16 gossip kernel in its
    LDS image order (k slots of the register-fed blocks permuted, desco_gossip_f16_stream) and 1/scale per matrix."""
    for name, w, shape in (("w1", w1, (2, 64, 128)), ("wp", wp, (2, 64, 128)), ("w3", w3, (2, 64, 64)),
                           ("w5", w5, (2, 256, 64))):
        if tuple(w.planes.shape) != shape:
            raise ValueError(f"gossip_f16_stream: {name} planes must be {shape}, got {tuple(w.planes.shape)}")
    stream = torch.empty((9, 2, 4096), device=w1.planes.device, dtype=torch.int16)
    _lib.check(_lib.lib().desco_gossip_f16_stream(*[_dev(w.planes, "planes", torch.int16) for w in (w1, wp, w3, w5)],
                                                  _dev(stream, "wstream", torch.int16), _stream()), "gossip_f16_stream")
    winv = torch.stack([w.scale[1] for w in (w1, wp, w3, w5)]).contiguous()
    return stream, winv


def gossip_fused_f16(scal: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int,
                     num_q: int, v: dict, queue: torch.Tensor, tile_perm: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The whole gossip network on chip in the three-product fp16 form, one wave per (16 nodes, 8 queries) work unit
    (csrc/gossip_f16.hip): returns
    pred [N, Q].  ``queue``: two zeroed int64 words (see desco_hip.h); ``out``: optional contiguous [N, Q] destination."""
    if out is None:
        out = torch.empty((num_nodes, num_q), device=scal.device, dtype=torch.float32)
    elif tuple(out.shape) != (num_nodes, num_q) or not out.is_contiguous():
        raise ValueError("gossip_fused_f16: `out` must be a contiguous [N, Q] tensor")
    if tile_perm is not None and tile_perm.numel() < ((num_nodes + 127) // 128) * 128:
        raise ValueError("gossip_fused_f16: tile_perm is shorter than the tiles of this batch")
    if queue.dtype != torch.int64 or queue.numel() < 2:
        raise ValueError("gossip_fused_f16: queue must hold two int64 words")
    L = _lib.lib()
    names = ("g1", "p", "z", "zp", "r", "t", "u", "tp", "d1", "wstream", "winv", "b3", "b5", "w7")
    ptrs = []
    for n in names:
        if not v[n].is_contiguous():
            raise ValueError(f"gossip_fused_f16: operand {n} must be contiguous")
        ptrs.append(_dev(v[n], n, torch.int16 if n == "wstream" else torch.float32))
    rows = float(num_nodes) * num_q
    with _Timed("gossip_fused_f16_kernel", rows * GOSSIP_FUSED_FLOPS_PER_ROW,
                rows * 20.0 + 4.0 * (col.numel() * (1 + 4 * num_q) + num_nodes)):
        _lib.check(L.desco_gossip_fused_f16x3_f32(_dev(scal, "scal"), _dev(rowptr, "rowptr", torch.int32),
                                                  _dev(col, "col", torch.int32), num_nodes, num_q, *ptrs,
                                                  float(v["b7"]), _dev(out, "out"),
                                                  None if tile_perm is None else _dev(tile_perm, "tile_perm", torch.uint8),
                                                  _dev(queue, "queue", torch.int64), _stream()),
                   "gossip_fused_f16")
    return out


def rowdot_add(y: torch.Tensor, w: torch.Tensor, b: float, add: Optional[torch.Tensor]):
    """out[r] = add[r] + y[r,:] . w + b"""
    R, n = y.shape
    out = torch.empty((R,), device=y.device, dtype=torch.float32)
    yp, ldy = _rows(y, "y")
    L = _lib.lib()
    with _Timed("rowdot_add_kernel", 2.0 * R * n, 4.0 * (R * n + 2 * R)):
        _lib.check(L.desco_rowdot_add_f32(yp, ldy, n, _dev(w.contiguous(), "w"), b,
                                          _opt(add, "add"), _dev(out, "out"), R, _stream()),
                   "rowdot_add")
    return out


# ------------------------------------------------------------------------------------------------
# backward-pass entry points (csrc/train_ops.hip)
# ------------------------------------------------------------------------------------------------
def shmp_trunk_small_max_rows() -> int:
    return int(_lib.lib().desco_shmp_trunk_small_max_rows())


def shmp_trunk_small_fwd(x0, vrowptr, vcol, wt, bias, seg_ptr, num_seg):
    """One-workgroup SHMP trunk (desco_shmp_trunk_small_fwd_f32): returns (xall [L, n, 64], pooled [num_seg, 64 (L + 1)])."""
    n, L = x0.shape[0], wt.shape[0]
    assert x0.is_contiguous() and wt.is_contiguous() and bias.is_contiguous() and tuple(wt.shape) == (L, 192, 64)
    xall = torch.empty((L, n, 64), device=x0.device, dtype=torch.float32)
    pooled = torch.empty((num_seg, 64 * (L + 1)), device=x0.device, dtype=torch.float32)
    with _Timed("shmp_small_fwd_kernel", 2.0 * n * 192 * 64 * L, 4.0 * (L * 192 * 64 + (L + 1) * n * 64)):
        _lib.check(_lib.lib().desco_shmp_trunk_small_fwd_f32(
            _dev(x0, "x0"), _dev(vrowptr, "vrowptr", torch.int32), _dev(vcol, "vcol", torch.int32), n, L, _dev(wt, "wt"),
            _dev(bias, "bias"), _dev(seg_ptr, "seg_ptr", torch.int32), num_seg, _dev(xall, "xall"),
            _dev(pooled, "pooled"), pooled.shape[1], _stream()), "shmp_trunk_small_fwd")
    return xall, pooled


def shmp_trunk_small_bwd(x0, xall, vrowptr, vcol, t_rowptr, t_col_s1, seg_id, wt_t, dpooled):
    """Backward of shmp_trunk_small_fwd: (dwt [L, 192, 64], dbias [L, 64], dx0 [n, 64])."""
    L, n = xall.shape[0], xall.shape[1]
    assert wt_t.is_contiguous() and tuple(wt_t.shape) == (L, 64, 192)
    dev = x0.device
    dwt = torch.empty((L, 192, 64), device=dev, dtype=torch.float32)
    dbias = torch.empty((L, 64), device=dev, dtype=torch.float32)
    dx0 = torch.empty((n, 64), device=dev, dtype=torch.float32)
    dp, ldp = _rows(dpooled, "dpooled")
    with _Timed("shmp_small_bwd_kernel", 6.0 * n * 192 * 64 * L, 4.0 * (3 * L * 192 * 64 + (L + 1) * n * 64)):
        _lib.check(_lib.lib().desco_shmp_trunk_small_bwd_f32(
            _dev(x0, "x0"), _dev(xall, "xall"), _dev(vrowptr, "vrowptr", torch.int32), _dev(vcol, "vcol", torch.int32),
            _dev(t_rowptr, "t_rowptr", torch.int32), _dev(t_col_s1, "t_col", torch.int32),
            _dev(seg_id, "seg_id", torch.int32), n, L, _dev(wt_t, "wt_t"), dp, ldp, _dev(dwt, "dwt"),
            _dev(dbias, "dbias"), _dev(dx0, "dx0"), _stream()), "shmp_trunk_small_bwd")
    return dwt, dbias, dx0


def shmp_trunk_graphs_max_rows() -> int:
    return int(_lib.lib().desco_shmp_trunk_graphs_max_rows())


def shmp_trunk_graphs_fwd(x0, vrowptr, vcol, wt, bias, seg_ptr, num_seg, drop: "Optional[DropSite]" = None):
    """The SHMP trunk with one workgroup per graph (desco_shmp_trunk_graphs_fwd_f32; graphs of at most
    shmp_trunk_graphs_max_rows() rows): returns (xall [L, n, 64], pooled [num_seg, 64 (L + 1)]).  ``drop``: F.dropout
    behind every layer's relu, layer l at site drop.site + 2 l."""
    n, L = x0.shape[0], wt.shape[0]
    assert x0.is_contiguous() and wt.is_contiguous() and bias.is_contiguous() and tuple(wt.shape) == (L, 192, 64)
    xall = torch.empty((L, n, 64), device=x0.device, dtype=torch.float32)
    pooled = torch.empty((num_seg, 64 * (L + 1)), device=x0.device, dtype=torch.float32)
    with _Timed("shmp_graphs_fwd_kernel", 2.0 * n * 192 * 64 * L, 4.0 * (num_seg * L * 192 * 64 + (L + 1) * n * 64)):
        _lib.check(_lib.lib().desco_shmp_trunk_graphs_fwd_f32(
            _dev(x0, "x0"), _dev(vrowptr, "vrowptr", torch.int32), _dev(vcol, "vcol", torch.int32), n, L, _dev(wt, "wt"),
            _dev(bias, "bias"), _dev(seg_ptr, "seg_ptr", torch.int32), num_seg,
            None if drop is None else ctypes.byref(drop.desc()), _dev(xall, "xall"),
            _dev(pooled, "pooled"), pooled.shape[1], _stream()), "shmp_trunk_graphs_fwd")
    return xall, pooled


def shmp_trunk_graphs_bwd(x0, xall, vrowptr, vcol, t_rowptr, t_col_s1, seg_ptr, num_seg, wt, dpooled, mask_scale=1.0):
    """Backward of shmp_trunk_graphs_fwd: (dwt [L, 192, 64], dbias [L, 64], dx0 [n, 64]); two launches."""
    L, n = xall.shape[0], xall.shape[1]
    assert wt.is_contiguous() and tuple(wt.shape) == (L, 192, 64)
    dev = x0.device
    dwt = torch.empty((L, 192, 64), device=dev, dtype=torch.float32)
    dbias = torch.empty((L, 64), device=dev, dtype=torch.float32)
    dx0 = torch.empty((n, 64), device=dev, dtype=torch.float32)
    ws = torch.empty((L, n, 64), device=dev, dtype=torch.float32)
    dp, ldp = _rows(dpooled, "dpooled")
    with _Timed("shmp_graphs_bwd_kernel", 6.0 * n * 192 * 64 * L, 4.0 * (2 * num_seg * L * 192 * 64 + (L + 1) * n * 64),
                launches=2):
        _lib.check(_lib.lib().desco_shmp_trunk_graphs_bwd_f32(
            _dev(x0, "x0"), _dev(xall, "xall"), _dev(vrowptr, "vrowptr", torch.int32), _dev(vcol, "vcol", torch.int32),
            _dev(t_rowptr, "t_rowptr", torch.int32), _dev(t_col_s1, "t_col", torch.int32),
            _dev(seg_ptr, "seg_ptr", torch.int32), num_seg, n, L, _dev(wt, "wt"), dp, ldp, float(mask_scale), _dev(dwt, "dwt"),
            _dev(dbias, "dbias"), _dev(dx0, "dx0"), _dev(ws, "workspace"), _stream()), "shmp_trunk_graphs_bwd")
    return dwt, dbias, dx0


def linear_smallk_bwd(feat: torch.Tensor, dout: torch.Tensor):
    """(dwt [K, 64], dbias [64]) of out = feat @ wt + bias (tiny K, 64 output columns) given dout [M, 64]."""
    m, k = feat.shape
    assert dout.shape == (m, 64)
    dwb = torch.empty((k + 1, 64), device=dout.device, dtype=torch.float32)
    ws = torch.empty((512 * (k + 1) * 64,), device=dout.device, dtype=torch.float32)
    fp, ldf = _rows(feat, "feat")
    dp, ldd = _rows(dout, "dout")
    with _Timed("smallk_bwd_partial_kernel", 2.0 * m * (k + 1) * 64, 4.0 * m * (64 + k), launches=2):
        _lib.check(_lib.lib().desco_linear_smallk_bwd_f32(fp, ldf, k, dp, ldd, m, _dev(dwb, "dwb"), _dev(ws, "ws"),
                                                          _stream()), "linear_smallk_bwd")
    return dwb[:k], dwb[k]


def rowdot_bwd(y: torch.Tensor, w: torch.Tensor, dout: torch.Tensor):
    """Backward of ``rowdot_add`` on y = relu(z): (dz [R, n], dwb [n + 1] = (dw, db)) in one pass over y."""
    R, n = y.shape
    assert dout.is_contiguous() and dout.numel() == R and w.is_contiguous() and w.numel() == n
    dz = torch.empty((R, n), device=y.device, dtype=torch.float32)
    dwb = torch.empty((n + 1,), device=y.device, dtype=torch.float32)
    ws = torch.empty((1024 * (n + 1),), device=y.device, dtype=torch.float32)
    yp, ldy = _rows(y, "y")
    with _Timed("rowdot_bwd_kernel", 4.0 * R * n, 4.0 * (2 * R * n + R), launches=2):
        _lib.check(_lib.lib().desco_rowdot_bwd_f32(yp, ldy, n, _dev(w, "w"), _dev(dout, "dout"), R, _dev(dz, "dz"), n,
                                                   _dev(dwb, "dwb"), _dev(ws, "ws"), _stream()), "rowdot_bwd")
    return dz, dwb



def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None,
            accumulate: bool = False) -> torch.Tensor:
    """out[k, n] (+)= a[m, k]^T @ b[m, n]  (weight gradient; deterministic split-M reduction)."""
    import ctypes
    m, k = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty((k, n), device=a.device, dtype=torch.float32)
        accumulate = False
    ap, lda = _rows(a, "a")
    bp, ldb = _rows(b, "b")
    op, ldo = _rows(out, "out")
    L = _lib.lib()
    splits = ctypes.c_int(0)
    nbytes = L.desco_gemm_tn_workspace(m, k, n, ctypes.byref(splits))
    ws = torch.empty((max(nbytes // 4, 1),), device=a.device, dtype=torch.float32)
    with _Timed("gemm_tn_partial_kernel", 2.0 * m * k * n, 4.0 * (m * k + m * n), launches=2):
        _lib.check(L.desco_gemm_tn_f32(ap, lda, bp, ldb, m, k, n, op, ldo, int(accumulate),
                                       _dev(ws, "ws"), _stream()), "gemm_tn")
    return out


def linear_bwd_w(a1: torch.Tensor, a2: Optional[torch.Tensor], dz: torch.Tensor, want_bias: bool,
                 dwt: Optional[torch.Tensor] = None, dbias: Optional[torch.Tensor] = None):
    """(dwt [(k1+k2), n], dbias [n] or None) of c = act([a1 | a2] @ wt + bias) given dz: weight and
    bias gradient in two launches (desco_linear_bwd_w_f32); ``dwt`` / ``dbias``: contiguous outputs to fill."""
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = dz.shape[1]
    if dwt is None:
        dwt = torch.empty((k1 + k2, n), device=dz.device, dtype=torch.float32)
    assert dwt.is_contiguous() and tuple(dwt.shape) == (k1 + k2, n)
    if want_bias and dbias is None:
        dbias = torch.empty((n,), device=dz.device, dtype=torch.float32)
    if not want_bias:
        dbias = None
    assert dbias is None or (dbias.is_contiguous() and dbias.numel() == n)
    a1p, lda1 = _rows(a1, "a1")
    a2p, lda2 = (None, 0) if a2 is None else _rows(a2, "a2")
    zp, ldz = _rows(dz, "dz")
    L = _lib.lib()
    nbytes = L.desco_linear_bwd_w_workspace(m, k1 + k2, n)
    ws = torch.empty((max(nbytes // 4, 1),), device=dz.device, dtype=torch.float32)
    with _Timed("linear_bwd_w_kernel", 2.0 * m * (k1 + k2) * n, 4.0 * (m * (k1 + k2) + 2 * m * n),
                launches=1 if nbytes <= 4 * (k1 + k2 + 1) * n else 2):        # (one M slab: results written in place)
        _lib.check(L.desco_linear_bwd_w_f32(a1p, lda1, k1, a2p, lda2, k2, zp, ldz, m, n, _dev(dwt, "dwt"), n,
                                            _opt(dbias, "dbias"), _dev(ws, "ws"), _stream()), "linear_bwd_w")
    return dwt, dbias


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """out[n] (+)= sum_m x[m, n]  (bias gradient)."""
    m, n = x.shape
    if out is None:
        out = torch.empty((n,), device=x.device, dtype=torch.float32)
        accumulate = False
    xp, ldx = _rows(x, "x")
    ws = torch.empty((512 * n,), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    with _Timed("colsum_partial_kernel", float(m) * n, 4.0 * m * n, launches=2):
        _lib.check(L.desco_colsum_f32(xp, ldx, m, n, _dev(out, "out"), int(accumulate),
                                      _dev(ws, "ws"), _stream()), "colsum")
    return out


def act_grad(dc: torch.Tensor, c: torch.Tensor, act: int, slope: float, drop: "Optional[DropSite]" = None) -> torch.Tensor:
    """dz = dc * act'(c) with c the activation OUTPUT (contiguous tensors of equal shape); with ``drop`` the output was
    dropout(act(z)) and dz = dc * factor * act'(c), the factor regenerated from the step's key."""
    if act == ACT_NONE and drop is None:
        return dc
    dc, c = dc.contiguous(), c.contiguous()
    dz = torch.empty_like(dc)
    L = _lib.lib()
    if drop is not None:
        assert dc.dim() == 2
        d = drop.desc()
        with _Timed("act_grad_dropout_kernel", float(dc.numel()), 12.0 * dc.numel()):
            _lib.check(L.desco_act_grad_dropout_f32(_dev(dc, "dc"), _dev(c, "c"), act, slope, ctypes.byref(d), _dev(dz, "dz"),
                                                    dc.shape[0], dc.shape[1], _stream()), "act_grad_dropout")
        return dz
    with _Timed("act_grad_kernel", float(dc.numel()), 12.0 * dc.numel()):
        _lib.check(L.desco_act_grad_f32(_dev(dc, "dc"), _dev(c, "c"), act, slope, _dev(dz, "dz"),
                                        dc.numel(), _stream()), "act_grad")
    return dz


def count_head_bwd(t: torch.Tensor, qh: torch.Tensor, w2: torch.Tensor, slope: float,
                   dl: torch.Tensor):
    """Backward of count_head (logit mode): returns (dT [B,hid], dQh [Q,hid], dw2 [hid])."""
    B, hid = t.shape
    Q = qh.shape[0]
    dl = dl.contiguous()
    dt = torch.empty((B, hid), device=t.device, dtype=torch.float32)
    dqh = torch.empty((Q, hid), device=t.device, dtype=torch.float32)
    dw2 = torch.empty((hid,), device=t.device, dtype=torch.float32)
    L = _lib.lib()
    ws = torch.empty((L.desco_count_head_bwd_workspace(B, Q, hid) // 4,), device=t.device, dtype=torch.float32)
    tp, ldt = _rows(t, "t")
    qp, ldq = _rows(qh, "qh")
    with _Timed("count_head_bwd", 6.0 * B * Q * hid, 4.0 * (2 * B * hid + Q * hid + B * Q), launches=4):
        _lib.check(L.desco_count_head_bwd_f32(tp, ldt, qp, ldq, hid, _dev(w2.contiguous(), "w2"),
                                              slope, _dev(dl, "dl"), Q, B, Q, _dev(dt, "dt"), hid,
                                              _dev(dqh, "dqh"), _dev(dw2, "dw2"), _dev(ws, "ws"),
                                              _stream()), "count_head_bwd")
    return dt, dqh, dw2


def affine_rows(base: Optional[torch.Tensor], c: torch.Tensor, v: torch.Tensor, act: int,
                slope: float, drop: "Optional[DropSite]" = None) -> torch.Tensor:
    """out[r,:] = act(base[r,:] + sum_k c[r,k] * v[r % QV, k, :]);  v: [QV, KS, 64], c: [R, KS]; with ``drop`` the
    result times the dropout factor of its (row, col) (desco_affine_rows_dropout_f32)."""
    R, ks = c.shape
    qv = v.shape[0]
    assert v.shape == (qv, ks, 64) and c.is_contiguous()
    v = v.contiguous()
    out = torch.empty((R, 64), device=c.device, dtype=torch.float32)
    if base is not None:
        base = base.contiguous()
    L = _lib.lib()
    with _Timed("affine_rows_kernel", 2.0 * R * ks * 64, 4.0 * R * (ks + 128)):
        if drop is None:
            _lib.check(L.desco_affine_rows_f32(_opt(base, "base"), _dev(c, "c"), ks, _dev(v, "v"), qv,
                                               act, slope, _dev(out, "out"), R, _stream()),
                       "affine_rows")
        else:
            d = drop.desc()
            _lib.check(L.desco_affine_rows_dropout_f32(_opt(base, "base"), _dev(c, "c"), ks, _dev(v, "v"), qv, act, slope,
                                                       ctypes.byref(d), _dev(out, "out"), R, _stream()),
                       "affine_rows_dropout")
    return out


def affine_rows_bwd(c: torch.Tensor, dz: torch.Tensor, qv: int) -> torch.Tensor:
    """dv[q, k, :] = sum_{r = i*qv + q} c[r, k] * dz[r, :]."""
    R, ks = c.shape
    dz = dz.contiguous()
    dv = torch.empty((qv, ks, 64), device=c.device, dtype=torch.float32)
    ws = torch.empty((64 * qv * ks * 64,), device=c.device, dtype=torch.float32)
    L = _lib.lib()
    with _Timed("affine_rows_bwd_kernel", 2.0 * R * ks * 64, 4.0 * R * (ks + 64), launches=2):
        _lib.check(L.desco_affine_rows_bwd_f32(_dev(c, "c"), ks, _dev(dz, "dz"), qv, R,
                                               _dev(dv, "dv"), _dev(ws, "ws"), _stream()),
                   "affine_rows_bwd")
    return dv


def rowdot2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out[r] = a[r,:] . b[r,:]"""
    a, b = a.contiguous(), b.contiguous()
    R, n = a.shape
    out = torch.empty((R,), device=a.device, dtype=torch.float32)
    L = _lib.lib()
    with _Timed("rowdot2_kernel", 2.0 * R * n, 8.0 * R * n):
        _lib.check(L.desco_rowdot2_f32(_dev(a, "a"), _dev(b, "b"), n, _dev(out, "out"), R,
                                       _stream()), "rowdot2")
    return out


# ---- round 5: the glue of the training steps (csrc/train_native.hip) ------------------------------------------------
def copy2d_multi(problems) -> None:
    """Strided 2-D copies / transposes in one launch per 24 (desco_copy2d_multi_f32).  ``problems``: (src, dst) or
    (src, dst, transpose) or (src, dst, transpose, accumulate) with 2-D fp32 tensors whose rows are contiguous;
    transpose: dst [cols, rows] = src^T."""
    descs = (_lib.Copy2dDesc * len(problems))()
    nbytes = 0.0
    for d, pr in zip(descs, problems):
        src, dst = pr[0], pr[1]
        tr = bool(pr[2]) if len(pr) > 2 else False
        rows, cols = src.shape
        assert tuple(dst.shape) == ((cols, rows) if tr else (rows, cols)), (tuple(src.shape), tuple(dst.shape), tr)
        d.rows, d.cols, d.transpose = rows, cols, int(tr)
        d.accumulate = int(bool(pr[3])) if len(pr) > 3 else 0
        if rows and cols:
            d.src, d.lds = _rows(src, "src")
            d.dst, d.ldd = _rows(dst, "dst")
        nbytes += 8.0 * rows * cols
    with _Timed("copy2d_multi_kernel", 0.0, nbytes):
        _lib.check(_lib.lib().desco_copy2d_multi_f32(len(problems), descs, _stream()), "copy2d_multi")


def transposed(w: torch.Tensor) -> torch.Tensor:
    """w^T as a new contiguous tensor (one copy2d launch)."""
    out = torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=torch.float32)
    copy2d_multi([(w, out, True)])
    return out


def fold_shmp_fwd(table: torch.Tensor, L: int, S: int, NU: int):
    """(wt [L, (S+1) 64, 64], fb [L, 64]) from the parameter address table of one row type (desco_fold_shmp_fwd_f32)."""
    wt = torch.empty((L, (S + 1) * 64, 64), device=table.device, dtype=torch.float32)
    fb = torch.empty((L, 64), device=table.device, dtype=torch.float32)
    with _Timed("fold_shmp_fwd_kernel", 2.0 * L * S * 64 ** 3, 4.0 * L * ((S + 2) * 4096 + (S + 1) * 4096)):
        _lib.check(_lib.lib().desco_fold_shmp_fwd_f32(_dev(table, "table", torch.int64), L, S, NU, _dev(wt, "wt"),
                                                      _dev(fb, "fb"), _stream()), "fold_shmp_fwd")
    return wt, fb


def fold_shmp_bwd(table: torch.Tensor, goff: torch.Tensor, L: int, S: int, NU: int, dwt: torch.Tensor,
                  dfb: torch.Tensor, grads: torch.Tensor) -> None:
    """the gradients of every parameter of the table into the flat buffer ``grads`` (desco_fold_shmp_bwd_f32)"""
    assert dwt.is_contiguous() and dfb.is_contiguous() and tuple(dwt.shape) == (L, (S + 1) * 64, 64)
    with _Timed("fold_shmp_bwd_kernel", 4.0 * L * S * 64 ** 3, 4.0 * L * (2 * (S + 2) * 4096 + (S + 1) * 4096)):
        _lib.check(_lib.lib().desco_fold_shmp_bwd_f32(_dev(table, "table", torch.int64), _dev(goff, "goff", torch.int64),
                                                      L, S, NU, _dev(dwt, "dwt"), _dev(dfb, "dfb"), _dev(grads, "grads"),
                                                      _stream()), "fold_shmp_bwd")


def loss_fwd(pred: torch.Tensor, y: torch.Tensor, mode: int):
    """(loss [] , dpred) of desco_loss_f32: mode 0 = mean smooth_l1(pred - log2(y + 1)), mode 1 = sum log2(|pred - y| + 1)"""
    assert pred.is_contiguous() and y.is_contiguous() and pred.shape == y.shape
    dev = pred.device
    if pred.numel() == 0:               # an empty batch: the loss of nothing is 0 (what the torch criterion's sum gives)
        return zeros((), dev), torch.empty_like(pred)
    # the partial sums' workspace (4 KB) is allocated per call from torch's stream-aware caching allocator: one buffer
    # per device would be shared by every stream that trains on it (Trainer's side stream, two models side by side)
    ws = torch.empty(1024, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    with _Timed("loss_partial_kernel", 0.0, 12.0 * pred.numel()):
        _lib.check(_lib.lib().desco_loss_f32(_dev(pred, "pred"), _dev(y, "y"), pred.numel(), mode, _dev(loss, "loss"),
                                             _dev(dpred, "dpred"), _dev(ws, "ws"), _stream()), "loss")
    return loss, dpred


def affine_scalar(a: torch.Tensor, mul: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
                  addv: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a * mul[0] + add[0] + addv with one-element device tensors mul / add (no host read) and an optional addend."""
    assert a.is_contiguous() and (addv is None or (addv.is_contiguous() and addv.numel() == a.numel()))
    out = torch.empty_like(a)
    _lib.check(_lib.lib().desco_affine_scalar_f32(_dev(a, "a"), _opt(mul, "mul"), _opt(add, "add"), _opt(addv, "addv"),
                                                  _dev(out, "out"), a.numel(), _stream()), "affine_scalar")
    return out


def scale_by_scalar(a: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    return affine_scalar(a, mul=s)


def _gossip_fold_params(P: dict) -> "_lib.GossipFoldParams":
    p = _lib.GossipFoldParams()
    for n in ("E", "w_pre", "b_pre", "C0", "cb0", "D0", "db0", "C1", "cb1", "D1", "db1", "P0", "p0", "P3", "P5"):
        t = P[n]
        assert t.is_contiguous(), n
        setattr(p, n, _dev(t, n))
    for n in ("G0", "gb0", "g2", "gb2"):
        arr = getattr(p, n)
        for i in range(2):
            assert P[n][i].is_contiguous(), n
            arr[i] = _dev(P[n][i], n)
    p.num_q = P["E"].shape[0]
    return p


def _gossip_fold_out(O: dict) -> "_lib.GossipFoldOut":
    o = _lib.GossipFoldOut()
    for n in ("V0", "V1", "Vp", "wt1", "wtp", "w3t", "w5t", "g0", "g1", "g1c", "a", "h0", "h1"):
        setattr(o, n, _dev(O[n], n))
    return o


def gossip_fold_fwd(P: dict) -> dict:
    """desco_gossip_fold_fwd_f32: P = the parameter tensors by the names of desco_gossip_fold_params; returns the outputs
    by the names of desco_gossip_fold_out."""
    Q, dev = P["E"].shape[0], P["E"].device
    e = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)          # noqa: E731
    O = dict(V0=e(Q, 6, 64), V1=e(Q, 3, 64), Vp=e(Q, 2, 64), wt1=e(128, 64), wtp=e(128, 64), w3t=e(64, 64), w5t=e(64, 256),
             g0=e(Q), g1=e(Q), g1c=e(Q), a=e(Q, 64), h0=e(Q, 64), h1=e(Q, 64))
    p, o = _gossip_fold_params(P), _gossip_fold_out(O)
    with _Timed("gossip_fold_fwd_kernel", 0.0, 0.0):
        _lib.check(_lib.lib().desco_gossip_fold_fwd_f32(ctypes.byref(p), ctypes.byref(o), _stream()), "gossip_fold_fwd")
    return O


def gossip_fold_bwd(P: dict, O: dict, dO: dict) -> dict:
    """desco_gossip_fold_bwd_f32: gradients of the parameters (dict by desco_gossip_fold_grads' output names)"""
    Q, dev = P["E"].shape[0], P["E"].device
    G = {("d" + n): torch.empty_like(P[n]) for n in ("C0", "cb0", "D0", "db0", "C1", "cb1", "D1", "db1", "P0", "p0", "P3", "P5")}
    for n in ("G0", "gb0", "g2", "gb2"):
        G["d" + n] = [torch.empty_like(P[n][i]) for i in range(2)]
    d = _lib.GossipFoldGrads()
    for n in ("dV0", "dV1", "dVp", "dwt1", "dwtp", "dw3t", "dw5t"):
        assert dO[n].is_contiguous(), n
        setattr(d, n, _dev(dO[n], n))
    d.dg1 = _opt(dO.get("dg1"), "dg1")
    for n in ("dC0", "dcb0", "dD0", "ddb0", "dC1", "dcb1", "dD1", "ddb1", "dP0", "dp0", "dP3", "dP5"):
        setattr(d, n, _dev(G[n], n))
    for n in ("dG0", "dgb0", "dg2", "dgb2"):
        arr = getattr(d, n)
        for i in range(2):
            arr[i] = _dev(G[n][i], n)
    scratch = torch.empty((4 * Q * 64,), device=dev, dtype=torch.float32)
    d.scratch = _dev(scratch, "scratch")
    p, o = _gossip_fold_params(P), _gossip_fold_out(O)
    with _Timed("gossip_fold_bwd_kernel", 0.0, 0.0):
        _lib.check(_lib.lib().desco_gossip_fold_bwd_f32(ctypes.byref(p), ctypes.byref(o), ctypes.byref(d), _stream()),
                   "gossip_fold_bwd")
    return G


def fill(t: torch.Tensor, value: float) -> torch.Tensor:
    assert t.is_contiguous() and t.dtype == torch.float32
    _lib.check(_lib.lib().desco_fill_f32(_dev(t, "t"), float(value), t.numel(), _stream()), "fill")
    return t


def zeros(shape, device) -> torch.Tensor:
    """torch.zeros without torch's fill kernel"""
    return fill(torch.empty(shape, device=device, dtype=torch.float32), 0.0)


# ---- round 6: counter-based dropout (include/desco_hip.h: desco_dropout) ----------------------------------------------
_RNG_STATE = {}


def rng_state(device) -> torch.Tensor:
    """(seed, step) of this device's dropout stream: two int64 words in device memory.  Seeded from
    ``torch.initial_seed()`` on first use (so ``torch.manual_seed`` before the first training step selects the stream);
    ``manual_seed`` resets it."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _RNG_STATE.get(device)
    if st is None:
        st = _RNG_STATE[device] = torch.tensor([torch.initial_seed() & 0x7fffffffffffffff, 0], dtype=torch.int64).to(device)
    return st


def manual_seed(seed: int, device=None, step: int = 0) -> None:
    """Restart the dropout stream of ``device`` (default: the current one) at (seed, step)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    rng_state(device).copy_(torch.tensor([seed & 0x7fffffffffffffff, step], dtype=torch.int64))


def rng_next(device) -> torch.Tensor:
    """The key of one training forward pass: a fresh [2] int64 device tensor holding (seed, step); the device's step
    counter moves on (desco_rng_next: one launch, capturable -- a replayed step draws the next mask)."""
    st = rng_state(device)
    key = torch.empty((2,), device=st.device, dtype=torch.int64)
    L = _lib.lib()
    with _Timed("rng_next_kernel", 0.0, 32.0):
        _lib.check(L.desco_rng_next(_dev(st, "state", torch.int64), _dev(key, "key", torch.int64), _stream()), "rng_next")
    return key


class DropSite:
    """One dropout call site of a step: (key tensor, site id, p)."""
    __slots__ = ("key", "site", "p")

    def __init__(self, key: torch.Tensor, site: int, p: float):
        assert key.dtype == torch.int64 and key.numel() == 2 and key.is_contiguous() and 0 <= site < 256 and 0.0 <= p <= 1.0
        self.key, self.site, self.p = key, site, float(p)

    @property
    def scale(self) -> float:
        """the factor of the kept elements, as the kernels apply it (fp32)"""
        return 0.0 if self.p >= 1.0 else float(torch.tensor(1.0 / (1.0 - self.p), dtype=torch.float32))

    def desc(self) -> "_lib.Dropout":
        d = _lib.Dropout()
        d.key, d.site = _dev(self.key, "dropout key", torch.int64), self.site
        if self.p >= 1.0:
            d.threshold, d.scale = 0xffffffff, 0.0
        else:
            d.threshold = min(int(round(self.p * 4294967296.0)), 0xffffffff)
            d.scale = 1.0 / (1.0 - self.p)
        return d


def dropout_mask(drop: DropSite, num_rows: int, num_cols: int) -> torch.Tensor:
    """The factor tensor [num_rows, num_cols] (0 or 1 / (1 - p)) the fused epilogues multiply by."""
    out = torch.empty((num_rows, num_cols), device=drop.key.device, dtype=torch.float32)
    d = drop.desc()
    L = _lib.lib()
    with _Timed("dropout_mask_kernel", 0.0, 4.0 * num_rows * num_cols):
        _lib.check(L.desco_dropout_mask_f32(ctypes.byref(d), num_rows, num_cols, _dev(out, "out"), num_cols, _stream()),
                   "dropout_mask")
    return out
