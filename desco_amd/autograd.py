"""torch.autograd bookkeeping around the HIP kernels (training path).

Every Function's forward AND backward are C-ABI kernel launches (desco_amd.ops); torch only wires
the graph, so ``loss.backward()`` + ``torch.optim.Adam`` train the reference-named parameters.
The weight folding of gnn_model.pack_* is done with (tiny) differentiable torch ops, so gradients
reach the original ``lin`` / ``updates`` / ``anchor_mlp`` / ``post_mp`` / ``count_model`` tensors.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

# Matrix-product precision of the training path: "fp32" (v_mfma_f32_32x32x2_f32, the default) or
# "bf16" (BASELINE config 3: operands rounded to nearest-even bf16, one v_mfma_f32_32x32x16_bf16
# product per multiply-add, fp32 accumulation; activations, gathers, pooling, the loss, the weight
# gradients' accumulation and the master weights stay fp32).
PRECISION = "fp32"


def set_precision(p: str) -> None:
    global PRECISION
    p = {"32": "fp32", "32-true": "fp32", "bf16-mixed": "bf16"}.get(str(p), str(p))
    if p not in ("fp32", "bf16"):
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {p!r}")
    PRECISION = p



class GatherSum(torch.autograd.Function):
    """agg = csr_gather_sum(x); backward = gather over the transposed index (csr_t)."""

    @staticmethod
    def forward(ctx, x, vrowptr, vcol, t_rowptr, t_col, num_rows, slots):
        ctx.save_for_backward(t_rowptr, t_col)
        ctx.n_src = x.shape[0]
        return ops.csr_gather_sum(x.contiguous(), vrowptr, vcol, num_rows, slots)

    @staticmethod
    def backward(ctx, dagg):
        t_rowptr, t_col = ctx.saved_tensors
        d = dagg.contiguous().view(-1, 64)                     # [num_rows*slots, 64] virtual rows
        dx = ops.csr_gather_sum(d, t_rowptr, t_col, ctx.n_src, 1)
        return dx, None, None, None, None, None, None


class Linear(torch.autograd.Function):
    """c = act([a1 | a2] @ wt + bias);  wt is [(k1+k2), n] (a differentiable function of params)."""

    @staticmethod
    def forward(ctx, a1, a2, wt, bias, act, slope):
        wt = wt.contiguous()
        ctx.bf16 = PRECISION == "bf16" and wt.shape[1] % 64 == 0 and wt.shape[0] % 64 == 0
        if ctx.bf16:
            c = ops.gemm_bf16(a1, ops.round_bf16(wt.t()), bias, a2=a2, act=act, slope=slope)
        else:
            c = ops.gemm(a1, wt, bias, a2=a2, act=act, slope=slope)
        ctx.save_for_backward(a1, a2 if a2 is not None else a1.new_empty(0), wt, c)
        ctx.has_a2, ctx.act, ctx.slope, ctx.has_bias = a2 is not None, act, slope, bias is not None
        return c

    @staticmethod
    def backward(ctx, dc):
        a1, a2, wt, c = ctx.saved_tensors
        dz = ops.act_grad(dc.contiguous(), c, ctx.act, ctx.slope)
        k1 = a1.shape[1]
        da1 = da2 = dwt = dbias = None
        need_a1, need_a2 = ctx.needs_input_grad[0], ctx.has_a2 and ctx.needs_input_grad[1]
        if need_a1 or need_a2:
            if ctx.bf16:     # dA[m,k] = sum_n dZ[m,n] wt[k,n]: wt is already the n-major operand
                da = ops.gemm_bf16(dz, ops.round_bf16(wt))
            else:
                da = ops.gemm(dz, wt.t().contiguous())         # dA = dZ @ Wt^T  (k % 64 == 0)
            if need_a1:
                da1 = da[:, :k1]
            if need_a2:
                da2 = da[:, k1:]
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        fused_ok = (k1 % 64 == 0 and dz.shape[1] % 64 == 0 and (not ctx.has_a2 or a2.shape[1] % 64 == 0)
                    and a1.shape[0] > 0)
        if ctx.needs_input_grad[2] and fused_ok:
            # weight and bias gradient in one pass over the rows (two launches instead of six)
            dwt, dbias = ops.linear_bwd_w(a1, a2 if ctx.has_a2 else None, dz, want_b)
        else:
            if ctx.needs_input_grad[2]:
                dwt = torch.empty_like(wt)
                ops.gemm_tn(a1, dz, out=dwt[:k1])
                if ctx.has_a2:
                    ops.gemm_tn(a2, dz, out=dwt[k1:])
            if want_b:
                dbias = ops.colsum(dz)
        return da1, da2, dwt, dbias, None, None


class SmallKLinear(torch.autograd.Function):
    """pre_mp: out = feat @ wt + bias with tiny K (gnn_model.py:131); feat carries no gradient."""

    @staticmethod
    def forward(ctx, feat, wt, bias):
        ctx.save_for_backward(feat)
        return ops.linear_smallk(feat, wt.contiguous(), bias)

    @staticmethod
    def backward(ctx, dout):
        (feat,) = ctx.saved_tensors
        dout = dout.contiguous()
        dwt = None
        if ctx.needs_input_grad[1]:
            # K is 1 in the reference pipeline; a [K, n] product over M rows: K column-sums
            dwt = torch.stack([ops.colsum(dout * feat[:, k:k + 1]) for k in range(feat.shape[1])])
        return None, dwt, ops.colsum(dout)


class SegmentSum(torch.autograd.Function):
    """out[b] = sum_{rows of b} x + extra[b]; backward = broadcast (gather by segment id)."""

    @staticmethod
    def forward(ctx, x, seg_ptr, seg_id, ident_ptr, extra):
        ctx.save_for_backward(seg_id, ident_ptr)
        ctx.has_extra = extra is not None
        return ops.segment_sum(x.contiguous(), seg_ptr, seg_ptr.numel() - 1, extra=extra)

    @staticmethod
    def backward(ctx, dout):
        seg_id, ident_ptr = ctx.saved_tensors
        dout = dout.contiguous()
        dx = ops.csr_gather_sum(dout, ident_ptr, seg_id, seg_id.numel(), 1)   # dx[r] = dout[seg(r)]
        return dx, None, None, None, (dout if ctx.has_extra else None)


class CountHead(torch.autograd.Function):
    """logit[b,q] = sum_c w2[c]*leaky(T[b,c]+Qh[q,c]) + b2 (lightning_model.py:176-193)."""

    @staticmethod
    def forward(ctx, t, qh, w2, b2, slope):
        ctx.save_for_backward(t, qh, w2)
        ctx.slope = slope
        return ops.count_head(t, qh, w2, b2, slope, False)       # b2 read on the device: capturable

    @staticmethod
    def backward(ctx, dl):
        t, qh, w2 = ctx.saved_tensors
        dt, dqh, dw2 = ops.count_head_bwd(t, qh, w2, ctx.slope, dl)
        db2 = ops.colsum(dl.contiguous().view(-1, 1)).view(())
        return dt, dqh, dw2, db2, None


class AffineRows(torch.autograd.Function):
    """out = act(base + sum_k c[:,k] * v[row % QV, k, :]); c is a constant, v (and base) learn."""

    @staticmethod
    def forward(ctx, base, c, v, act, slope):
        out = ops.affine_rows(base, c, v, act, slope)
        ctx.save_for_backward(c, out)
        ctx.qv, ctx.act, ctx.slope, ctx.has_base = v.shape[0], act, slope, base is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        c, out = ctx.saved_tensors
        dz = ops.act_grad(dout.contiguous(), out, ctx.act, ctx.slope)
        dv = ops.affine_rows_bwd(c, dz, ctx.qv) if ctx.needs_input_grad[2] else None
        return (dz if ctx.has_base else None), None, dv, None, None


class GossipGather(torch.autograd.Function):
    """out[i,q] = sum_j (j<i ? g[q] : 1-g[q]) * h[j,q]  (GossipConv message+aggregate,
    gnn_model.py:335-344, aggregate-then-transform); differentiable in h and in the gate g."""

    @staticmethod
    def forward(ctx, h, rowptr, col, num_nodes, num_q, g):
        ctx.save_for_backward(h, rowptr, col, g)
        ctx.n, ctx.q = num_nodes, num_q
        return ops.gossip_gather(h.contiguous(), rowptr, col, num_nodes, num_q, g)

    @staticmethod
    def backward(ctx, dout):
        h, rowptr, col, g = ctx.saved_tensors
        dout = dout.contiguous()
        dh = dg = None
        if ctx.needs_input_grad[0]:     # transpose of the gated sum: the same kernel with 1 - g
            dh = ops.gossip_gather(dout, rowptr, col, ctx.n, ctx.q, (1.0 - g).contiguous())
        if ctx.needs_input_grad[5]:     # d out / d g = sum_{j<i} h_j - sum_{j>i} h_j
            d = ops.gossip_gather(h.contiguous(), rowptr, col, ctx.n, ctx.q, None)
            dg = ops.colsum(ops.rowdot2(dout, d).view(ctx.n, ctx.q))
        return dh, None, None, None, None, dg
