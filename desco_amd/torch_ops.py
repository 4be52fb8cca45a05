"""``torch.ops.desco.*``: the inner boundary SURVEY 8b names ("PyTorch-ROCm custom ops"), registered
with torch.library on top of the C ABI (include/desco_hip.h).

Tensors in / tensors out, outputs allocated by the op, no retained pointers, errors as
``RuntimeError``, launches on the current HIP stream of the operands' device, deterministic (no float
atomics).  The device ops are registered for the CUDA (= HIP on ROCm) dispatch key ONLY: a CPU tensor
finds no kernel and raises -- there is no CPU fallback; ``build_canonical_partition`` is the host
builder (CPU key), as in the survey's minimum set.  The classes of this package call the same C-ABI
entry points directly through ``desco_amd.ops`` (one ctypes call, ~2 us; the dispatcher adds ~5 us
per launch, which the 70-launch small-dataset pass would feel); the registered ops are for callers
that want them in torch-native form (torch.compile graphs, TorchScript-free export, other code
bases).

    import desco_amd.torch_ops                      # registers the namespace
    agg = torch.ops.desco.shmp_aggregate(x, vrowptr, vcol, num_rows, 4)
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import ops

_DEF = torch.library.Library("desco", "DEF")
_CUDA = torch.library.Library("desco", "IMPL", "CUDA")
_CPU = torch.library.Library("desco", "IMPL", "CPU")

SCHEMAS = {
    "build_canonical_partition": "(Tensor graph_ptr, Tensor rowptr, Tensor col, int depth, int quirk_batch=0, "
                                 "int num_threads=0) -> Tensor[]",
    "shmp_aggregate": "(Tensor x, Tensor vrowptr, Tensor vcol, int num_rows, int slots) -> Tensor",
    "shmp_aggregate_backward": "(Tensor grad_agg, Tensor t_rowptr, Tensor t_col, int num_src) -> Tensor",
    "shmp_transpose_index": "(Tensor vrowptr, Tensor vcol, int num_rows, int slots, int num_count) -> Tensor[]",
    "shmp_layer_fused": "(Tensor x, Tensor vrowptr, Tensor vcol, int row0, int num_rows, int slots_stored, "
                        "int slots_mfma, Tensor weight_planes, Tensor bias, Tensor? ytab, int ytab_row0, "
                        "Tensor(a!) out) -> Tensor(a!)",
    "segment_sum": "(Tensor x, Tensor seg_ptr, int num_seg, Tensor? extra) -> Tensor",
    "count_head": "(Tensor t, Tensor qh, Tensor w2, Tensor b2, float slope, bool exp2_minus_1) -> Tensor",
    "count_head_backward": "(Tensor t, Tensor qh, Tensor w2, float slope, Tensor grad_logits) -> Tensor[]",
    "gossip_aggregate": "(Tensor h, Tensor rowptr, Tensor col, int num_nodes, int num_q, Tensor? gate) -> Tensor",
    "gossip_aggregate_backward": "(Tensor grad_out, Tensor rowptr, Tensor col, int num_nodes, int num_q, "
                                 "Tensor gate) -> Tensor",
    "gossip_fused": "(Tensor scal4, Tensor rowptr, Tensor col, int num_nodes, int num_q, Tensor[] operands, "
                    "float b7) -> Tensor",
    "split_bf16_planes": "(Tensor w) -> Tensor",
    # the three-product fp16 forms the product path runs since round 4 (csrc/gemm_f16x3.hip, gossip_f16.hip,
    # shmp_layer16.hip): planes int16 [2, n, k] + scale float32 [2] of desco_split_f16x2_f32
    "split_f16_planes": "(Tensor w) -> Tensor[]",
    "gemm_f16x3": "(Tensor a1, Tensor planes, Tensor scale, Tensor? bias, Tensor? a2, int act, float slope) -> Tensor",
    "shmp_layer_fused_f16x3": "(Tensor x, Tensor vrowptr, Tensor vcol, int row0, int num_rows, int slots_stored, "
                              "int slots_mfma, Tensor planes, Tensor scale, Tensor bias, Tensor? ytab, int ytab_row0, "
                              "Tensor(a!) out) -> Tensor(a!)",
    "gossip_f16_stream": "(Tensor[] planes, Tensor[] scales) -> Tensor[]",
    "gossip_fused_f16x3": "(Tensor scal4, Tensor rowptr, Tensor col, int num_nodes, int num_q, Tensor[] operands, "
                          "float b7, Tensor(a!) queue, Tensor? tile_perm) -> Tensor",
}
for _name, _schema in SCHEMAS.items():
    _DEF.define(_name + _schema)

GOSSIP_FUSED_OPERANDS = ("g1", "p", "z", "zp", "r", "t", "u", "tp", "d1", "w1s", "wps", "w3s", "b3", "w5s",
                         "b5", "w7")


GOSSIP_F16_OPERANDS = ("g1", "p", "z", "zp", "r", "t", "u", "tp", "d1", "wstream", "winv", "b3", "b5", "w7")


def _gossip_fused_f16x3(scal4, rowptr, col, num_nodes, num_q, operands, b7, queue, tile_perm):
    if len(operands) != len(GOSSIP_F16_OPERANDS):
        raise RuntimeError(f"desco::gossip_fused_f16x3 expects {len(GOSSIP_F16_OPERANDS)} operands {GOSSIP_F16_OPERANDS}")
    v = dict(zip(GOSSIP_F16_OPERANDS, operands))
    v["b7"] = b7
    return ops.gossip_fused_f16(scal4, rowptr, col, num_nodes, num_q, v, queue, tile_perm=tile_perm)


def _gossip_f16_stream(planes, scales):
    if len(planes) != 4 or len(scales) != 4:
        raise RuntimeError("desco::gossip_f16_stream expects the planes / scales of W1, Wp, W3, W5")
    return list(ops.gossip_f16_stream(*[ops.F16Planes(p, s) for p, s in zip(planes, scales)]))


def _shmp_layer_fused_f16x3(x, vrowptr, vcol, row0, num_rows, slots_stored, slots_mfma, planes, scale, bias, ytab,
                            ytab_row0, out):
    return ops.shmp_layer(x, vrowptr, vcol, row0, num_rows, slots_stored, slots_mfma, ops.F16Planes(planes, scale), bias,
                          out, ytab=ytab, ytab_row0=ytab_row0)


def _build_canonical_partition(graph_ptr, rowptr, col, depth, quirk_batch=0, num_threads=0) -> List[torch.Tensor]:
    """Host builder (csrc/partition.cpp): [neigh_index int64 [B,2], indicator bool [N], count_ptr int32,
    count_orig int32, vrowptr int32, vcol int32] (workload.py:243-294 + transforms.py:180-255)."""
    from .graphs import GraphSet
    from .partition import build_partition
    gs = GraphSet(graph_ptr.cpu().numpy(), rowptr.cpu().numpy(), col.cpu().numpy())
    p = build_partition(gs, int(depth), int(quirk_batch), int(num_threads))
    return [torch.from_numpy(np.ascontiguousarray(a)) for a in
            (p.neigh_index, p.indicator, p.count_ptr, p.count_orig, p.vrowptr, p.vcol)]


def _shmp_layer_fused(x, vrowptr, vcol, row0, num_rows, slots_stored, slots_mfma, weight_planes, bias, ytab,
                      ytab_row0, out):
    return ops.shmp_layer(x, vrowptr, vcol, row0, num_rows, slots_stored, slots_mfma, weight_planes, bias, out,
                          ytab=ytab, ytab_row0=ytab_row0)


def _gossip_fused(scal4, rowptr, col, num_nodes, num_q, operands, b7):
    if len(operands) != len(GOSSIP_FUSED_OPERANDS):
        raise RuntimeError(f"desco::gossip_fused expects {len(GOSSIP_FUSED_OPERANDS)} operands "
                           f"{GOSSIP_FUSED_OPERANDS}")
    v = dict(zip(GOSSIP_FUSED_OPERANDS, operands))
    v["b7"] = b7
    return ops.gossip_fused(scal4, rowptr, col, num_nodes, num_q, v)


_CPU.impl("build_canonical_partition", _build_canonical_partition)
_CUDA.impl("shmp_aggregate", lambda x, vrowptr, vcol, num_rows, slots:
           ops.csr_gather_sum(x.contiguous(), vrowptr, vcol, num_rows, slots))
_CUDA.impl("shmp_aggregate_backward", lambda g, t_rowptr, t_col, num_src:
           ops.csr_gather_sum(g.contiguous().view(-1, 64), t_rowptr, t_col, num_src, 1))
_CUDA.impl("shmp_transpose_index", lambda vrowptr, vcol, num_rows, slots, num_count:
           list(ops.vcsr_transpose_sym(vrowptr, vcol, num_rows, slots, num_count)))
_CUDA.impl("shmp_layer_fused", _shmp_layer_fused)
_CUDA.impl("segment_sum", lambda x, seg_ptr, num_seg, extra: ops.segment_sum(x, seg_ptr, num_seg, extra=extra))
_CUDA.impl("count_head", lambda t, qh, w2, b2, slope, e: ops.count_head(t, qh, w2, b2, slope, e))
_CUDA.impl("count_head_backward", lambda t, qh, w2, slope, dl: list(ops.count_head_bwd(t, qh, w2, slope, dl)))
_CUDA.impl("gossip_aggregate", lambda h, rowptr, col, n, q, gate: ops.gossip_gather(h, rowptr, col, n, q, gate))
_CUDA.impl("gossip_aggregate_backward", lambda g, rowptr, col, n, q, gate:
           ops.gossip_gather(g.contiguous(), rowptr, col, n, q, (1.0 - gate).contiguous()))
_CUDA.impl("gossip_fused", _gossip_fused)
_CUDA.impl("split_bf16_planes", ops.split_bf16_planes)
def _split_f16_planes(w):
    fp = ops.split_f16_planes(w)
    return [fp.planes, fp.scale]


_CUDA.impl("split_f16_planes", _split_f16_planes)
_CUDA.impl("gemm_f16x3", lambda a1, planes, scale, bias, a2, act, slope:
           ops.gemm_f16x3(a1, ops.F16Planes(planes, scale), bias, a2=a2, act=act, slope=slope))
_CUDA.impl("shmp_layer_fused_f16x3", _shmp_layer_fused_f16x3)
_CUDA.impl("gossip_f16_stream", _gossip_f16_stream)
_CUDA.impl("gossip_fused_f16x3", _gossip_fused_f16x3)
