"""torch.autograd bookkeeping around the HIP kernels (training path).

Every Function's forward AND backward are C-ABI kernel launches (desco_amd.ops); torch only wires
the graph, so ``loss.backward()`` + the optimizer train the reference-named parameters.  Since round 5
that includes the glue: the weight folding (FoldShmp / FoldGossip) reads the parameters where torch
keeps them and writes their gradients into one buffer, the moves between nn.Linear's [out, in] layout
and the kernels' K-major operands are copy2d launches (TransposedMany, SplitT), the losses are one
kernel pair each (Loss) -- a step launches no kernel of torch's or hipBLASLt's
(tools/check_pass_is_native.sh --train; ``backward(loss)`` below seeds the graph without torch's ones_like).
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
# ShmpTrunkSmall: one workgroup per graph (csrc/shmp_small.hip, round 6) when the graphs are small enough; False: the
# one-workgroup kernels (their cross-check)
GRAPH_TRUNK_KERNEL = True
# the gossip step's forward and input-gradient products (a million (node, query) rows through 64- to 256-wide weights) on
# the bf16x6 pipe -- fp32-accurate at the streaming rate -- instead of the exact-fp32 matrix pipe, which runs them at
# 2-4 x their HBM time (desco_gemm_bf16x6_desc_f32: gate and dropout factor in the epilogue); weight gradients stay fp32
import os as _os
TRAIN_GEMM_BF16X6 = _os.environ.get("DESCO_TRAIN_GEMM_BF16X6", "1") != "0"


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
        if not wt.is_contiguous():
            wt = ops.transposed(wt.t()) if wt.t().is_contiguous() else wt.contiguous()
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
                da = ops.gemm(dz, ops.transposed(wt))          # dA = dZ @ Wt^T  (k % 64 == 0)
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


# (the remaining fp32 products -- anchor, post_mp, head: 512-row operands -- run on v_mfma_f32_32x32x2_f32; the layers'
#  products went to the bf16x6 pipe in round 6: TRAIN_GEMM_BF16X6 above)
def _mm_fwd(a1, a2, wt, bias, act, slope, out=None):
    """act([a1 | a2] @ wt + bias) in the training precision (see Linear)."""
    if PRECISION == "bf16" and wt.shape[1] % 64 == 0 and wt.shape[0] % 64 == 0:
        return ops.gemm_bf16(a1, ops.round_bf16(wt.t()), bias, a2=a2, act=act, slope=slope, out=out)
    return ops.gemm(a1, wt, bias, a2=a2, act=act, slope=slope, out=out)


def _mm_bwd_da(dz, wt, out=None):
    """dA = dZ @ wt^T  ([m, n] x [k, n]^T -> [m, k])."""
    if PRECISION == "bf16" and wt.shape[1] % 64 == 0 and wt.shape[0] % 64 == 0:
        return ops.gemm_bf16(dz, ops.round_bf16(wt), out=out)
    return ops.gemm(dz, ops.transposed(wt), out=out)


class ShmpTrunk(torch.autograd.Function):
    """BaseGNNCore.forward's SAGE loop (gnn_model.py:253-277) + anchor_mlp on the canonical rows (:69-73) +
    global_add_pool per layer block (:88-89, 107) as ONE autograd node: pooled [B, 64 (L+1)] from x0 [N, 64]
    and the folded weights STACKED over the layers.  Forward and backward are C-ABI launches on buffers this
    node owns -- no torch.cat / slice / accumulate kernels between them (the per-op autograd wiring of round 2
    spent as much GPU time in ~900 torch glue launches per step as in the 216 kernels that do the work).

    ``drop`` = None, or (key, p): F.dropout behind every layer's relu (gnn_model.py:274; --neigh_dropout, default 0.0) as
    the counter-based factor of (key, site 2 l + row type, row inside the row type, column) in the epilogue of the layer's
    product (desco_gemm_desc.drop); its backward is the scale 1 / (1 - p) on the relu mask (the kept elements of
    dropout(relu(z)) are its positive ones).  Layer products of a dropout step run in fp32 in either precision mode.

    args: x0, batch (NeighborhoodBatch / QueryBatch), groups [(type, r0, r1, su)], has_anchor, drop, then tensors:
          [anchor wt (K-major [P, P]), anchor bias] if has_anchor, then per group Wt [L, (su+1) 64, 64] and
          bias [L, 64] (gnn_model.pack_shmp_stacked).
    Gradient of a row of X_l = pooling broadcast + (canonical rows) its column block of d(anchor operand) +
    self block + transposed gather of the aggregate blocks of layer l: assembled in one buffer per layer."""

    @staticmethod
    def forward(ctx, x0, batch, groups, has_anchor, drop, *w):
        N, S = batch.num_rows, batch.slots
        dev = x0.device
        H = 64
        k = 2 if has_anchor else 0
        aw, ab = (w[0], w[1]) if has_anchor else (None, None)
        Wt = [w[k + 2 * g].contiguous() for g in range(len(groups))]
        Bs = [w[k + 2 * g + 1].contiguous() for g in range(len(groups))]
        num_layers = Wt[0].shape[0]
        Nc = groups[0][2] if has_anchor else N
        seg_ptr = batch.count_ptr if has_anchor else batch.graph_ptr
        B = batch.num_graphs
        P = H * (num_layers + 1)
        X, AGG = [x0.contiguous()], []
        xall = torch.empty((num_layers, N, H), device=dev)        # X_1 .. X_L in one buffer (pooled in one launch)
        # fp32 mode: the layers' products on the bf16x6 pipe (fp32-accurate; 38 -> 26 us per product on a real-size
        # batch): the stacked weights' n-major planes for all layers in one launch per row type
        # (bf16 mode: the same launches with ONE rounded plane per weight and A rounded in the kernel; a dropout step's
        #  products are fp32-accurate in either mode)
        x6 = TRAIN_GEMM_BF16X6 and len(groups) > 1
        npl = 3 if (PRECISION == "fp32" or drop is not None) else 1
        planes_f = [ops.split_bf16_planes_batch(w_, True, npl) for w_ in Wt] if x6 else None      # [L, P, 64, K_g]
        for l in range(num_layers):
            agg = ops.csr_gather_sum(X[-1], batch.vrowptr, batch.vcol, N, S)          # [N, S*64]
            xn = xall[l]
            if x6:
                ops.gemm_split_multi([dict(a1=agg[r0:r1, :su * H], a2=X[-1][r0:r1], bias=Bs[g][l], act=ops.ACT_RELU,
                                           out=xn[r0:r1],
                                           drop=None if drop is None else ops.DropSite(drop[0], 2 * l + g, drop[1]))
                                      for g, (t, r0, r1, su) in enumerate(groups)],
                                     [planes_f[g][l] for g in range(len(groups))])
            elif drop is not None or (PRECISION == "fp32" and len(groups) > 1):
                # the row types' products of one layer are independent: one launch (desco_gemm_f32_multi)
                ops.gemm_multi([dict(a1=agg[r0:r1, :su * H], a2=X[-1][r0:r1], wt=Wt[g][l], bias=Bs[g][l],
                                     act=ops.ACT_RELU, out=xn[r0:r1],
                                     drop=None if drop is None else ops.DropSite(drop[0], 2 * l + g, drop[1]))
                                for g, (t, r0, r1, su) in enumerate(groups)])
            else:
                for g, (t, r0, r1, su) in enumerate(groups):
                    if r1 > r0:
                        _mm_fwd(agg[r0:r1, :su * H], X[-1][r0:r1], Wt[g][l], Bs[g][l], ops.ACT_RELU, 0.0, out=xn[r0:r1])
            AGG.append(agg)
            X.append(xn)
        pooled = torch.empty((B, P), device=dev)
        canon = anch = None
        if has_anchor:
            canon = torch.empty((B, P), device=dev)
            ops.copy2d_multi([(xl[Nc:], canon[:, l * H:(l + 1) * H]) for l, xl in enumerate(X)])
            anch = _mm_fwd(canon, None, aw.contiguous(), ab, ops.ACT_LEAKY, 0.1)
        ops.segment_sum(X[0][:Nc], seg_ptr, B, extra=None if anch is None else anch[:, :H], out=pooled[:, :H])
        ops.segment_sum_layers(xall, Nc, seg_ptr, B, None if anch is None else anch[:, H:], pooled[:, H:])
        ctx.batch, ctx.groups, ctx.has_anchor = batch, groups, has_anchor
        ctx.mask_scale = 1.0 if drop is None else ops.DropSite(drop[0], 0, drop[1]).scale
        ctx.X, ctx.AGG, ctx.canon, ctx.anch = X, AGG, canon, anch
        ctx.x6, ctx.npl = x6, npl
        ctx.save_for_backward(*w)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        w = ctx.saved_tensors
        batch, groups, has_anchor = ctx.batch, ctx.groups, ctx.has_anchor
        X, AGG = ctx.X, ctx.AGG
        if AGG is None or any(a is None for a in AGG):
            # the per-layer aggregates are released as the backward consumes them (they are the largest saved
            # tensors of the step); a second backward through the same node would otherwise die on a None
            raise RuntimeError("ShmpTrunk.backward ran twice on the same forward (retain_graph=True is not supported: "
                               "the saved aggregates are freed layer by layer); run the forward again")
        ti = batch.train_index()
        N, S = batch.num_rows, batch.slots
        H = 64
        dev = dpooled.device
        k = 2 if has_anchor else 0
        Wt = [w[k + 2 * g].contiguous() for g in range(len(groups))]
        L = Wt[0].shape[0]
        Nc = groups[0][2] if has_anchor else N
        dpooled = dpooled.contiguous()
        grads = [None] * len(w)
        for g in range(len(groups)):        # stacked gradients, filled layer by layer
            grads[k + 2 * g] = torch.empty_like(Wt[g])
            grads[k + 2 * g + 1] = torch.empty((L, H), device=dev)
        dcanon = None
        if has_anchor:
            aw, canon, anch = w[0], ctx.canon, ctx.anch
            dza = ops.act_grad(dpooled, anch, ops.ACT_LEAKY, 0.1)                 # extra rows pass the gradient on
            grads[0], grads[1] = ops.linear_bwd_w(canon, None, dza, True)
            dcanon = _mm_bwd_da(dza, aw.contiguous())                             # [B, P]

        su_of = [su for _, _, _, su in groups]
        off_count = su_of[0] * H                       # column of the self block in a row of D, per row type
        off_canon = su_of[1] * H if has_anchor else 0

        def dx(l, d_rows, mask):
            """gradient w.r.t. the rows of X_l: seed (pooling broadcast; canonical rows: anchor operand) + self
            block + transposed gather of the slot blocks of d_rows, times relu'(X_l) (one launch)"""
            return ops.shmp_bwd_dx(d_rows, ti["t_rowptr"], ti["t_col_s1"], Nc, off_count, off_canon,
                                   dpooled[:, l * H:(l + 1) * H], ti["seg_id"],
                                   None if dcanon is None else dcanon[:, l * H:(l + 1) * H], mask, ctx.mask_scale)

        # dZ of the last layer: its rows feed only the pooling / the anchor operand (one zero row stands in for D:
        # row stride 0, no transposed edges)
        if "zero_d" not in ti:         # constants of the batch, made once
            ti["zero_d"] = torch.zeros((1, (S + 1) * H), device=dev)
            ti["empty_ptr"] = torch.zeros(N + 1, device=dev, dtype=torch.int32)
        zero_d, empty_ptr = ti["zero_d"].expand(N, -1), ti["empty_ptr"]
        dz = ops.shmp_bwd_dx(zero_d, empty_ptr, ti["t_col_s1"], Nc, off_count, off_canon,
                             dpooled[:, L * H:(L + 1) * H], ti["seg_id"],
                             None if dcanon is None else dcanon[:, L * H:(L + 1) * H], X[L], ctx.mask_scale)
        D = torch.empty((N, (S + 1) * H), device=dev)
        fp32 = PRECISION == "fp32"
        # transposed weights of all layers in one copy per row type (dA = dZ Wt^T wants the n-major operand)
        WtT = planes_b = None
        if ctx.x6:
            # dA = dZ Wt^T on the bf16x6 pipe: its n-major operand [n = in][k = out] is Wt as stored -- no transposes
            planes_b = [ops.split_bf16_planes_batch(w_, False, ctx.npl) for w_ in Wt]        # [L, P, K_g, 64]
        elif fp32:
            WtT = [torch.empty((L, w_.shape[2], w_.shape[1]), device=dev) for w_ in Wt]
            ops.copy2d_multi([(w_[l], t_[l], True) for w_, t_ in zip(Wt, WtT) for l in range(L)])
        wgrad = []                     # the weight / bias gradients feed nothing but the optimizer: formed together,
        for l in range(L - 1, -1, -1):  # 16 per launch pair, after the chain of input gradients
            live = [(g, r0, r1, su) for g, (t, r0, r1, su) in enumerate(groups)]
            for g, r0, r1, su in live:
                wgrad.append(dict(a1=AGG[l][r0:r1, :su * H], a2=X[l][r0:r1], dz=dz[r0:r1], dwt=grads[k + 2 * g][l],
                                  dbias=grads[k + 2 * g + 1][l]))
            if ctx.x6:
                ops.gemm_split_multi([dict(a1=dz[r0:r1], out=D[r0:r1, :(su + 1) * H]) for g, r0, r1, su in live],
                                     [planes_b[g][l] for g, r0, r1, su in live])
            elif fp32 and len(groups) > 1:
                ops.gemm_multi([dict(a1=dz[r0:r1], wt=WtT[g][l], out=D[r0:r1, :(su + 1) * H]) for g, r0, r1, su in live])
            else:
                for g, r0, r1, su in live:
                    if r1 > r0:
                        _mm_bwd_da(dz[r0:r1], Wt[g][l], out=D[r0:r1, :(su + 1) * H])
            dz = dx(l, D, X[l] if l > 0 else None)        # (l == 0: the gradient of x0 itself)
        for i in range(0, len(wgrad), 16):
            ops.linear_bwd_w_multi(wgrad[i:i + 16])
        del wgrad
        for l in range(L):
            AGG[l] = None
        dxn = dz
        return (dxn, None, None, None, None) + tuple(grads)


class ShmpTrunkSmall(torch.autograd.Function):
    """ShmpTrunk for a small single-type batch (the query graphs, at most ops.shmp_trunk_small_max_rows() rows, two
    relation slots): the whole trunk is one launch forward and one backward (csrc/shmp_small.hip) instead of 37.
    args: x0 [n, 64], batch (QueryBatch), Wt [L, 192, 64], bias [L, 64]; returns pooled [B, 64 (L + 1)].
    Products are fp32 in either training precision (1.7 MFLOP per layer: nothing to gain from bf16 operands)."""

    @staticmethod
    def per_graph(batch) -> bool:
        """One workgroup per graph (round 6: 29 workgroups instead of one) when every graph of the batch has at most
        ops.shmp_trunk_graphs_max_rows() rows -- the standard queries have 3..5."""
        pg = batch.__dict__.get("_small_per_graph")
        if pg is None:
            import numpy as np
            gp = getattr(batch, "graph_ptr_host", None)
            sizes = np.diff(np.asarray(batch.graph_ptr.cpu()) if gp is None else np.asarray(gp))
            pg = batch.__dict__["_small_per_graph"] = bool(GRAPH_TRUNK_KERNEL and len(sizes) and
                                                           int(sizes.max()) <= ops.shmp_trunk_graphs_max_rows())
        return pg

    @staticmethod
    def forward(ctx, x0, batch, drop, wt, bias):
        x0, wt, bias = x0.contiguous(), wt.contiguous(), bias.contiguous()
        ctx.per_graph = ShmpTrunkSmall.per_graph(batch)
        ctx.mask_scale = 1.0
        if drop is not None:          # (key, p): dropout behind every layer's relu -- the per-graph kernels only
            assert ctx.per_graph
            site = ops.DropSite(drop[0], 0, drop[1])
            ctx.mask_scale = site.scale
            xall, pooled = ops.shmp_trunk_graphs_fwd(x0, batch.vrowptr, batch.vcol, wt, bias, batch.graph_ptr,
                                                     batch.num_graphs, drop=site)
        else:
            fwd = ops.shmp_trunk_graphs_fwd if ctx.per_graph else ops.shmp_trunk_small_fwd
            xall, pooled = fwd(x0, batch.vrowptr, batch.vcol, wt, bias, batch.graph_ptr, batch.num_graphs)
        ctx.batch = batch
        ctx.save_for_backward(x0, xall, wt)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x0, xall, wt = ctx.saved_tensors
        batch = ctx.batch
        ti = batch.train_index()
        if ctx.per_graph:
            dwt, dbias, dx0 = ops.shmp_trunk_graphs_bwd(x0, xall, batch.vrowptr, batch.vcol, ti["t_rowptr"], ti["t_col_s1"],
                                                        batch.graph_ptr, batch.num_graphs, wt, dpooled.contiguous(),
                                                        ctx.mask_scale)
            return dx0, None, None, dwt, dbias
        wt_t = torch.empty((wt.shape[0], wt.shape[2], wt.shape[1]), device=wt.device)
        ops.copy2d_multi([(wt[l], wt_t[l], True) for l in range(wt.shape[0])])
        dwt, dbias, dx0 = ops.shmp_trunk_small_bwd(x0, xall, batch.vrowptr, batch.vcol, ti["t_rowptr"], ti["t_col_s1"],
                                                   ti["seg_id"], wt_t, dpooled.contiguous())
        return dx0, None, None, dwt, dbias


class SmallKLinear(torch.autograd.Function):
    """pre_mp: out = feat @ wt + bias with tiny K (gnn_model.py:131); feat carries no gradient."""

    @staticmethod
    def forward(ctx, feat, wt, bias):
        ctx.save_for_backward(feat)
        return ops.linear_smallk(feat, wt.contiguous(), bias)

    @staticmethod
    def backward(ctx, dout):
        (feat,) = ctx.saved_tensors
        if dout.shape[0] == 0:
            return None, ops.zeros((feat.shape[1], dout.shape[1]), dout.device), ops.zeros((dout.shape[1],), dout.device)
        if dout.shape[1] == 64 and feat.shape[1] <= 16:
            dwt, db = ops.linear_smallk_bwd(feat, dout)         # weight rows and bias row in one pass (two launches)
            return None, dwt, db
        dout = dout.contiguous()
        dwt = None
        if ctx.needs_input_grad[1]:
            # a [K, n] product over M rows: K column-sums
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


class GossipTrunk(torch.autograd.Function):
    """The gossip model's per-(node, query) pipeline -- GossipConv x 2 with the closed-form first layer, post_mp
    (gnn_model.py:58-103, 303-350 of the reference; algebra in DESIGN.md 4.2) -- as ONE autograd node, like ShmpTrunk:
    forward and backward are C-ABI launches on buffers this node owns; torch sees the folded operands going in and the
    correction coming out.  (As separate autograd Functions the step spent a tenth of its GPU time in torch's gradient
    accumulations, strided copies and zero fills between our kernels, and the activation derivatives were five extra
    passes over [N Q, 64..256] tensors: here they ride in the epilogue of the GEMM that produces the gradient.)

    args: batch (rowptr / col), num_nodes, num_q, C6 [R,6], C3 [R,3], C2 [R,2] (constants), x [R] (the input counts, no
    gradient), g1c = 1 - g1 and the raw post_mp.3 / .5 weights ([out, in]: the operands of dA = dZ W, no gradient asked
    of them here), then tensors with gradients: V0 [Q,6,64], g1 [Q], wt1 [128,64], V1 [Q,3,64], wtp [128,64],
    Vp [Q,2,64], w3t [64,64], b3 [64], w5t [64,256], b5 [256], w7 [256], b7 [1].
    Returns pred [R] = x + post_mp.7(...).

    ``drop`` = None, or (p_layer, p_post) in training mode with --gossip_dropout > 0 (default 0.01, config.py:316):
    F.dropout behind each layer's relu (gnn_model.py:274; sites 0, 1) and post_mp.1 = nn.Dropout between post_mp.0 and
    its LeakyReLU (:46; site 2) ride in the epilogues that produce h1, h2 and y -- relu / leaky commute with the
    non-negative factor -- as a counter-based factor of (step key, site, row, col) that the backward kernels regenerate
    (csrc/common_device.hpp); no mask tensor exists.  The step's key is drawn here (ops.rng_next: capturable)."""

    SITE_H1, SITE_H2, SITE_POST = 0, 1, 2

    @staticmethod
    def forward(ctx, rowptr, col, n, q, C6, C3, C2, x, g1c, w3, w5, drop, V0, g1, wt1, V1, wtp, Vp, w3t, b3, w5t, b5, w7, b7):
        V0, V1, Vp = V0.contiguous(), V1.contiguous(), Vp.contiguous()
        wt1, wtp, w3t, w5t = wt1.contiguous(), wtp.contiguous(), w3t.contiguous(), w5t.contiguous()
        g1, b3, b5, w7 = g1.contiguous(), b3.contiguous(), b5.contiguous(), w7.contiguous()
        d1 = d2 = dp = None
        key = C6.new_empty(0, dtype=torch.int64)
        if drop is not None:
            key = ops.rng_next(C6.device)
            d1, d2 = ops.DropSite(key, GossipTrunk.SITE_H1, drop[0]), ops.DropSite(key, GossipTrunk.SITE_H2, drop[0])
            dp = ops.DropSite(key, GossipTrunk.SITE_POST, drop[1])
        h1 = ops.affine_rows(None, C6, V0, ops.ACT_RELU, 0.0, d1)                   # layer 0 (closed form)
        hh = ops.gossip_gather(h1, rowptr, col, n, q, g1)                           # layer 1 aggregate
        x6 = TRAIN_GEMM_BF16X6 and PRECISION == "fp32"
        if x6:
            w3, w5 = w3.contiguous(), w5.contiguous()
            # (the per-query affine terms of the layer and of post_mp.0 -- affine_rows -- ride in the products' epilogues)
            h2 = torch.empty_like(h1)
            ops.gemm_split_desc(dict(a1=hh, a2=h1, out=h2, act=ops.ACT_RELU, s=C3, ws=V1, drop=d2),
                                ops.split_bf16_planes_t(wt1))
            y = torch.empty_like(h1)
            ops.gemm_split_desc(dict(a1=h1, a2=h2, out=y, act=ops.ACT_LEAKY, slope=0.1, s=C2, ws=Vp, drop=dp),
                                ops.split_bf16_planes_t(wtp))
            y3 = ops.gemm_split(y, ops.split_bf16_planes(w3), b3, act=ops.ACT_RELU)
            y5 = ops.gemm_split(y3, ops.split_bf16_planes(w5), b5, act=ops.ACT_RELU)
        else:
            h2 = ops.affine_rows(ops.gemm(hh, wt1, a2=h1), C3, V1, ops.ACT_RELU, 0.0, d2)
            y = ops.affine_rows(ops.gemm(h1, wtp, a2=h2), C2, Vp, ops.ACT_LEAKY, 0.1, dp)   # post_mp.0 + .1 + .2
            y3 = ops.gemm(y, w3t, b3, act=ops.ACT_RELU)
            y5 = ops.gemm(y3, w5t, b5, act=ops.ACT_RELU)
        pred = ops.affine_scalar(ops.rowdot_add(y5, w7, 0.0, None), add=b7, addv=x)
        ctx.save_for_backward(rowptr, col, C6, C3, C2, g1c, w3, w5, wt1, wtp, w7, h1, hh, h2, y, y3, y5, key)
        ctx.n, ctx.q, ctx.drop, ctx.x6 = n, q, drop, x6
        return pred

    @staticmethod
    def backward(ctx, dcorr):
        rowptr, col, C6, C3, C2, g1c, w3, w5, wt1, wtp, w7, h1, hh, h2, y, y3, y5, key = ctx.saved_tensors
        n, q = ctx.n, ctx.q
        d1 = d2 = dp = None
        if ctx.drop is not None:      # the same factors as the forward pass, from the same key
            d1, d2 = ops.DropSite(key, GossipTrunk.SITE_H1, ctx.drop[0]), ops.DropSite(key, GossipTrunk.SITE_H2, ctx.drop[0])
            dp = ops.DropSite(key, GossipTrunk.SITE_POST, ctx.drop[1])
        R = h1.shape[0]
        dev = h1.device
        dz5, dwb7 = ops.rowdot_bwd(y5, w7, dcorr.contiguous())                      # [R,256], (dw7 | db7)
        if ctx.x6:
            return GossipTrunk._backward_x6(ctx, dz5, dwb7, d1, d2, dp)
        # dA = dZ W^T wants W itself as the [K = out, N = in] operand: post_mp.3 / .5 as torch keeps them, and the
        # transposed blocks of the two folded weights (one copy2d launch)
        wp = torch.empty((2, 64, 64), device=dev)                                    # [(h1 | h2) block][out][in]
        w1 = torch.empty((2, 64, 64), device=dev)                                    # [(hh | h1) block][out][in]
        ops.copy2d_multi([(wtp[:64], wp[0], True), (wtp[64:], wp[1], True), (wt1[:64], w1[0], True),
                          (wt1[64:], w1[1], True)])
        dz3 = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dz5, wt=w5, out=dz3, gate=y3, gate_act=ops.ACT_RELU)])
        dzp = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dz3, wt=w3, out=dzp, gate=y, gate_act=ops.ACT_LEAKY, gate_slope=0.1, drop=dp)])
        dVp = ops.affine_rows_bwd(C2, dzp, q)
        dh1 = torch.empty((R, 64), device=dev)
        dz1 = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dzp, wt=wp[0], out=dh1),
                        dict(a1=dzp, wt=wp[1], out=dz1, gate=h2, gate_act=ops.ACT_RELU, drop=d2)])
        dV1 = ops.affine_rows_bwd(C3, dz1, q)
        dhh = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dz1, wt=w1[0], out=dhh),
                        dict(a1=dz1, wt=w1[1], out=dh1, accum=True)])
        # transpose of the gated sum: the same kernel with 1 - g (GossipGather.backward)
        ops.add_rows(dh1, ops.gossip_gather(dhh, rowptr, col, n, q, g1c))
        dsig = ops.gossip_gather(h1, rowptr, col, n, q, None)                       # d out / d g1
        dg1 = ops.colsum(ops.rowdot2(dhh, dsig).view(n, q))
        dz0 = ops.act_grad(dh1, h1, ops.ACT_RELU, 0.0, d1)
        dV0 = ops.affine_rows_bwd(C6, dz0, q)
        # the four weight / bias gradients wait for nothing and nothing but the optimizer waits for them: one launch pair
        dw5t, db5 = torch.empty((64, 256), device=dev), torch.empty((256,), device=dev)
        dw3t, db3 = torch.empty((64, 64), device=dev), torch.empty((64,), device=dev)
        dwtp, dwt1 = torch.empty_like(wtp), torch.empty_like(wt1)
        ops.linear_bwd_w_multi([dict(a1=y3, dz=dz5, dwt=dw5t, dbias=db5), dict(a1=y, dz=dz3, dwt=dw3t, dbias=db3),
                                dict(a1=h1, a2=h2, dz=dzp, dwt=dwtp), dict(a1=hh, a2=h1, dz=dz1, dwt=dwt1)])
        return (None, None, None, None, None, None, None, None, None, None, None, None, dV0, dg1, dwt1, dV1, dwtp, dVp,
                dw3t, db3, dw5t, db5, dwb7[:256], dwb7[256:257])


def _gossip_trunk_backward_x6(ctx, dz5, dwb7, d1, d2, dp):
    """GossipTrunk.backward with the input-gradient products on the bf16x6 pipe.  dA = dZ W wants the n-major planes of
    [n = in][k = out]: the transposing split of torch's [out, in] weights (post_mp.3 / .5) and the folded weights' blocks
    as they are stored ([in, out]); dh1 = dzp Wp_h1 + dz1 W1_h1 is ONE K = 128 product over [dzp | dz1]."""
    rowptr, col, C6, C3, C2, g1c, w3, w5, wt1, wtp, w7, h1, hh, h2, y, y3, y5, key = ctx.saved_tensors
    n, q = ctx.n, ctx.q
    R, dev = h1.shape[0], h1.device
    dz3 = torch.empty((R, 64), device=dev)
    ops.gemm_split_desc(dict(a1=dz5, out=dz3, gate=y3, gate_act=ops.ACT_RELU), ops.split_bf16_planes_t(w5))
    dzp = torch.empty((R, 64), device=dev)
    ops.gemm_split_desc(dict(a1=dz3, out=dzp, gate=y, gate_act=ops.ACT_LEAKY, gate_slope=0.1, drop=dp),
                        ops.split_bf16_planes_t(w3))
    dVp = ops.affine_rows_bwd(C2, dzp, q)
    dz1 = torch.empty((R, 64), device=dev)
    ops.gemm_split_desc(dict(a1=dzp, out=dz1, gate=h2, gate_act=ops.ACT_RELU, drop=d2), ops.split_bf16_planes(wtp[64:]))
    dV1 = ops.affine_rows_bwd(C3, dz1, q)
    dhh = torch.empty((R, 64), device=dev)
    ops.gemm_split_desc(dict(a1=dz1, out=dhh), ops.split_bf16_planes(wt1[:64]))
    wc = torch.empty((64, 128), device=dev)                     # [in_h1][out of post_mp.0 | out of layer 1]
    ops.copy2d_multi([(wtp[:64], wc[:, :64]), (wt1[64:], wc[:, 64:])])
    dh1 = torch.empty((R, 64), device=dev)
    ops.gemm_split_desc(dict(a1=dzp, a2=dz1, out=dh1), ops.split_bf16_planes(wc))
    ops.add_rows(dh1, ops.gossip_gather(dhh, rowptr, col, n, q, g1c))
    dsig = ops.gossip_gather(h1, rowptr, col, n, q, None)                       # d out / d g1
    dg1 = ops.colsum(ops.rowdot2(dhh, dsig).view(n, q))
    dz0 = ops.act_grad(dh1, h1, ops.ACT_RELU, 0.0, d1)
    dV0 = ops.affine_rows_bwd(C6, dz0, q)
    dw5t, db5 = torch.empty((64, 256), device=dev), torch.empty((256,), device=dev)
    dw3t, db3 = torch.empty((64, 64), device=dev), torch.empty((64,), device=dev)
    dwtp, dwt1 = torch.empty_like(wtp), torch.empty_like(wt1)
    ops.linear_bwd_w_multi([dict(a1=y3, dz=dz5, dwt=dw5t, dbias=db5), dict(a1=y, dz=dz3, dwt=dw3t, dbias=db3),
                            dict(a1=h1, a2=h2, dz=dzp, dwt=dwtp), dict(a1=hh, a2=h1, dz=dz1, dwt=dwt1)])
    return (None, None, None, None, None, None, None, None, None, None, None, None, dV0, dg1, dwt1, dV1, dwtp, dVp,
            dw3t, db3, dw5t, db5, dwb7[:256], dwb7[256:257])


GossipTrunk._backward_x6 = staticmethod(_gossip_trunk_backward_x6)


class Mlp(torch.autograd.Function):
    """post_mp (gnn_model.py:40-53 of the reference) as ONE autograd node: h_{i+1} = act_i(h_i @ wt_i + b_i).

    args: x [M, k0], acts ((act, slope), ...), w_nk (the nn.Linear weights [out, in] as they are stored: the operand
    of dA = dZ W, no gradient asked of them here), then wt_0, b_0, wt_1, b_1, ... (wt_i = weight_i^T [in, out],
    differentiable).  Backward: one GEMM per layer with the activation derivative in its epilogue (gemm_multi's gate),
    all weight / bias gradients in one launch pair."""

    @staticmethod
    def forward(ctx, x, acts, w_nk, drop, *wb):
        """``drop`` = None, or a DropSite: post_mp.1 = nn.Dropout between post_mp.0 and its LeakyReLU (gnn_model.py:44-53)
        as the factor of the first layer's epilogue (leaky commutes with a non-negative factor); regenerated in the
        backward product's epilogue."""
        hs = [x.contiguous()]
        wts = []
        for i, (act, slope) in enumerate(acts):
            wt, b = wb[2 * i].contiguous(), wb[2 * i + 1]
            wts.append(wt)
            if i == 0 and drop is not None:
                out = torch.empty((hs[-1].shape[0], wt.shape[1]), device=x.device)
                ops.gemm_multi([dict(a1=hs[-1], wt=wt, bias=b, act=act, slope=slope, out=out, drop=drop)])
                hs.append(out)
            else:
                hs.append(_mm_fwd(hs[-1], None, wt, b, act, slope))
        ctx.acts, ctx.drop = acts, drop
        key = x.new_empty(0, dtype=torch.int64) if drop is None else drop.key
        ctx.save_for_backward(*hs, *[w.detach() for w in w_nk], *wts, key)
        return hs[-1]

    @staticmethod
    def backward(ctx, dout):
        n = len(ctx.acts)
        sv = ctx.saved_tensors
        hs, w_nk, wts = sv[:n + 1], sv[n + 1:2 * n + 1], sv[2 * n + 1:3 * n + 1]
        drop = None if ctx.drop is None else ops.DropSite(sv[3 * n + 1], ctx.drop.site, ctx.drop.p)
        dev = dout.device
        dz = dout.contiguous()
        act, slope = ctx.acts[n - 1]
        if act != ops.ACT_NONE:
            dz = ops.act_grad(dz, hs[n], act, slope)
        grads = [None] * (2 * n)
        wgrad = []
        for i in range(n - 1, -1, -1):
            grads[2 * i] = torch.empty_like(wts[i])
            grads[2 * i + 1] = torch.empty((wts[i].shape[1],), device=dev)
            wgrad.append(dict(a1=hs[i], dz=dz, dwt=grads[2 * i], dbias=grads[2 * i + 1]))
            if i == 0 and not ctx.needs_input_grad[0]:
                break
            da = torch.empty((dz.shape[0], wts[i].shape[0]), device=dev)
            w = w_nk[i]
            dsite = drop if i == 1 else None          # hs[1] = dropout(act_0(z)): its gradient carries the factor again
            if PRECISION == "bf16" and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
                _mm_bwd_da(dz, wts[i], out=da)
                if i > 0 and (ctx.acts[i - 1][0] != ops.ACT_NONE or dsite is not None):
                    da = ops.act_grad(da, hs[i], *ctx.acts[i - 1], drop=dsite)
            else:
                pr = dict(a1=dz, wt=w if w.is_contiguous() else w.contiguous(), out=da)
                if i > 0 and (ctx.acts[i - 1][0] != ops.ACT_NONE or dsite is not None):
                    pr.update(gate=hs[i], gate_act=ctx.acts[i - 1][0], gate_slope=ctx.acts[i - 1][1], drop=dsite)
                ops.gemm_multi([pr])
            dz = da
        ops.linear_bwd_w_multi(wgrad)
        return (dz if ctx.needs_input_grad[0] else None, None, None, None) + tuple(grads)



# ---- round 5: the glue of a training step as autograd nodes on this library's kernels -------------------------------
_ONES = {}


def backward(loss: torch.Tensor, weight: float = 1.0) -> None:
    """``(loss * weight).backward()`` seeded with a cached device scalar (torch's implicit seed is a ones_like fill
    launch per step, the product another launch); ``weight``: a rank's share of a data-parallel step's mean loss
    (distributed.mean_loss_weight)."""
    if not loss.is_cuda:
        (loss if weight == 1.0 else loss * weight).backward()
        return
    key = (loss.device, float(weight))
    seed = _ONES.get(key)
    if seed is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("autograd.backward: the seed of this weight has to be made outside a capture "
                               "(run the step eagerly once)")
        seed = _ONES[key] = ops.fill(torch.empty((), device=loss.device, dtype=torch.float32), float(weight))
    loss.backward(gradient=seed)


class Loss(torch.autograd.Function):
    """mode 0: mean smooth_l1(pred - log2(y + 1)) (lightning_model.py:246-254, 285-289); mode 1: sum log2(|pred - y| + 1)
    (:630-635).  The gradient is formed in the forward pass (desco_loss_f32) and scaled by the upstream scalar."""

    @staticmethod
    def forward(ctx, pred, y, mode):
        loss, dpred = ops.loss_fwd(pred.contiguous(), y.contiguous(), mode)
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dpred,) = ctx.saved_tensors
        return ops.scale_by_scalar(dpred, dloss), None, None


class TransposedMany(torch.autograd.Function):
    """(w_0^T, w_1^T, ...) of 2-D weights as contiguous tensors: one copy2d launch forward, one backward
    (nn.Linear keeps [out, in]; the GEMM entry points take the K-major [in, out])."""

    @staticmethod
    def forward(ctx, *ws):
        outs = [torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=torch.float32) for w in ws]
        ops.copy2d_multi([(w, o, True) for w, o in zip(ws, outs)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dws):
        outs = [torch.empty((d.shape[1], d.shape[0]), device=d.device, dtype=torch.float32) for d in dws]
        ops.copy2d_multi([(d.contiguous(), o, True) for d, o in zip(dws, outs)])
        return tuple(outs)


class SplitT(torch.autograd.Function):
    """(w[:, :h]^T, w[:, h:]^T) of one [out, in] weight (count_model.0 on cat(target, query), lightning_model.py:176-193);
    the backward writes both halves of ONE gradient tensor, so the parameter receives a single gradient."""

    @staticmethod
    def forward(ctx, w, h):
        a = torch.empty((h, w.shape[0]), device=w.device, dtype=torch.float32)
        b = torch.empty((w.shape[1] - h, w.shape[0]), device=w.device, dtype=torch.float32)
        ops.copy2d_multi([(w[:, :h], a, True), (w[:, h:], b, True)])
        ctx.h, ctx.shape = h, tuple(w.shape)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        dw = torch.empty(ctx.shape, device=da.device, dtype=torch.float32)
        ops.copy2d_multi([(da.contiguous(), dw[:, :ctx.h], True), (db.contiguous(), dw[:, ctx.h:], True)])
        return dw, None


class PreLinear(torch.autograd.Function):
    """pre_mp of every row type into ONE buffer x [N, 64] (gnn_model.py:131; no torch.cat): args feat [N, K], groups
    [(type, r0, r1, su)], then per group wt [K, 64], bias [64]."""

    @staticmethod
    def forward(ctx, feat, groups, *wb):
        x = torch.empty((feat.shape[0], 64), device=feat.device, dtype=torch.float32)
        for g, (_, r0, r1, _) in enumerate(groups):
            if r1 > r0:
                ops.linear_smallk(feat[r0:r1], wb[2 * g].contiguous(), wb[2 * g + 1], out=x[r0:r1])
        ctx.save_for_backward(feat)
        ctx.groups = groups
        return x

    @staticmethod
    def backward(ctx, dx):
        (feat,) = ctx.saved_tensors
        dx = dx.contiguous()
        K = feat.shape[1]
        out = []
        for _, r0, r1, _ in ctx.groups:
            if r1 > r0 and K <= 16:
                dwt, db = ops.linear_smallk_bwd(feat[r0:r1], dx[r0:r1])
            elif r1 > r0:
                dwt = torch.stack([ops.colsum(dx[r0:r1] * feat[r0:r1, k:k + 1]) for k in range(K)])
                db = ops.colsum(dx[r0:r1])
            else:
                dwt, db = ops.zeros((K, 64), dx.device), ops.zeros((64,), dx.device)
            out += [dwt, db]
        return (None, None) + tuple(out)


class FoldSpec:
    """The address table of one row type's parameters for desco_fold_shmp_* (built once per model and device; the
    parameters are updated in place by the optimizer, so their addresses hold)."""

    def __init__(self, core, t):
        L = core.layer_num
        keys = core.slot_keys(t)
        uniq = list(dict.fromkeys(keys))
        self.L, self.S, self.NU = L, len(keys), len(uniq)
        params, index = [], {}

        def add(p):
            if id(p) not in index:
                index[id(p)] = len(params)
                params.append(p)
            return index[id(p)]

        rows = []
        for l in range(L):
            row = [add(core.updates[l][t].weight), add(core.updates[l][t].bias)]
            row += [add(core.convs[l][k].lin.weight) for k in keys]
            row += [add(core.convs[l][k].lin.bias) for k in uniq]
            rows.append(row)
        self.params = params
        self._core, self._t, self._keys, self._uniq = core, t, keys, uniq
        offs, total = [], 0
        for p in params:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError("FoldShmp: parameters must be contiguous fp32")
            offs.append(total)
            total += p.numel()
        self.offsets, self.total = offs, total
        dev = params[0].device
        self.key = tuple(p.data_ptr() for p in params)
        self.table = torch.tensor([[params[i].data_ptr() for i in row] for row in rows], dtype=torch.int64).to(dev)
        self.goff = torch.tensor([[offs[i] for i in row] for row in rows], dtype=torch.int64).to(dev)

    def valid(self):
        """Still the table of the module's LIVE parameters?  The module is walked again: a Parameter object replaced
        since the table was built (load_state_dict(assign=True), a re-initialisation) leaves the captured object --
        and its address -- untouched, so comparing the captured objects' addresses alone would keep folding the stale
        storage and hand the gradients to orphaned tensors."""
        core, t = self._core, self._t
        live = []
        for l in range(self.L):
            live += [core.updates[l][t].weight, core.updates[l][t].bias]
            live += [core.convs[l][k].lin.weight for k in self._keys]
            live += [core.convs[l][k].lin.bias for k in self._uniq]
        seen, uniq_live = set(), []
        for p in live:
            if id(p) not in seen:
                seen.add(id(p))
                uniq_live.append(p)
        return (len(uniq_live) == len(self.params) and all(a is b for a, b in zip(uniq_live, self.params))
                and self.key == tuple(p.data_ptr() for p in self.params))


class FoldShmp(torch.autograd.Function):
    """(Wt [L, (S+1) 64, 64], fb [L, 64]) of one row type from its raw parameters (spec.params, passed so that autograd
    routes their gradients): desco_fold_shmp_fwd / _bwd.  Replaces gnn_model.pack_shmp_stacked's torch ops."""

    @staticmethod
    def forward(ctx, spec, *params):
        ctx.spec = spec
        return ops.fold_shmp_fwd(spec.table, spec.L, spec.S, spec.NU)

    @staticmethod
    def backward(ctx, dwt, dfb):
        sp = ctx.spec
        flat = torch.empty((sp.total,), device=dwt.device, dtype=torch.float32)
        ops.fold_shmp_bwd(sp.table, sp.goff, sp.L, sp.S, sp.NU, dwt.contiguous(), dfb.contiguous(), flat)
        return (None,) + tuple(flat[o:o + p.numel()].view(p.shape) for o, p in zip(sp.offsets, sp.params))


_GF_ORDER = ("C0", "cb0", "D0", "db0", "C1", "cb1", "D1", "db1", "G0_0", "gb0_0", "g2_0", "gb2_0", "G0_1", "gb0_1", "g2_1",
             "gb2_1", "P0", "p0", "P3", "P5")


class FoldGossip(torch.autograd.Function):
    """The operands of GossipTrunk from the gossip model's raw parameters (desco_gossip_fold_fwd / _bwd; formulas in
    csrc/train_native.hip, algebra DESIGN.md 4.2).  args: E [Q, 64], w_pre [64], b_pre [64] (no gradient), then the
    parameters in _GF_ORDER (g2_i = lin_gate.2.weight [1, 64]).  Returns (V0, g1, g1c, wt1, V1, wtp, Vp, w3t, w5t)."""

    @staticmethod
    def _pack(E, w_pre, b_pre, ps):
        d = dict(zip(_GF_ORDER, ps))
        return dict(E=E, w_pre=w_pre, b_pre=b_pre, C0=d["C0"], cb0=d["cb0"], D0=d["D0"], db0=d["db0"], C1=d["C1"],
                    cb1=d["cb1"], D1=d["D1"], db1=d["db1"], G0=[d["G0_0"], d["G0_1"]], gb0=[d["gb0_0"], d["gb0_1"]],
                    g2=[d["g2_0"].view(-1), d["g2_1"].view(-1)], gb2=[d["gb2_0"], d["gb2_1"]], P0=d["P0"], p0=d["p0"],
                    P3=d["P3"], P5=d["P5"])

    @staticmethod
    def forward(ctx, E, w_pre, b_pre, *ps):
        P = FoldGossip._pack(E, w_pre, b_pre, ps)
        O = ops.gossip_fold_fwd(P)
        # inputs and outputs through save_for_backward (an output kept as a plain ctx attribute is a reference cycle
        # output -> grad_fn -> ctx -> output: the step's autograd graph and its AccumulateGrad nodes then live until the
        # cyclic collector runs -- on whatever stream that is -- which breaks a later capture on another stream)
        ctx.okeys = tuple(O.keys())
        ctx.save_for_backward(E, w_pre, b_pre, *ps, *[O[k] for k in ctx.okeys])
        ctx.mark_non_differentiable(O["g1c"])
        ctx.set_materialize_grads(False)        # (an absent gradient arrives as None, not as a zero fill of torch's)
        return O["V0"], O["g1"], O["g1c"], O["wt1"], O["V1"], O["wtp"], O["Vp"], O["w3t"], O["w5t"]

    @staticmethod
    def backward(ctx, dV0, dg1, _dg1c, dwt1, dV1, dwtp, dVp, dw3t, dw5t):
        sv = ctx.saved_tensors
        n_in = 3 + len(_GF_ORDER)
        P = FoldGossip._pack(sv[0], sv[1], sv[2], sv[3:n_in])
        O = dict(zip(ctx.okeys, sv[n_in:]))
        z = lambda t, like: ops.zeros(tuple(like.shape), like.device) if t is None else t.contiguous()   # noqa: E731
        dO = dict(dV0=z(dV0, O["V0"]), dV1=z(dV1, O["V1"]), dVp=z(dVp, O["Vp"]), dwt1=z(dwt1, O["wt1"]),
                  dwtp=z(dwtp, O["wtp"]), dw3t=z(dw3t, O["w3t"]), dw5t=z(dw5t, O["w5t"]),
                  dg1=None if dg1 is None else dg1.contiguous())
        G = ops.gossip_fold_bwd(P, O, dO)
        g = {"C0": G["dC0"], "cb0": G["dcb0"], "D0": G["dD0"], "db0": G["ddb0"], "C1": G["dC1"], "cb1": G["dcb1"],
             "D1": G["dD1"], "db1": G["ddb1"], "P0": G["dP0"], "p0": G["dp0"], "P3": G["dP3"], "P5": G["dP5"]}
        for i in range(2):
            g[f"G0_{i}"], g[f"gb0_{i}"] = G["dG0"][i], G["dgb0"][i]
            g[f"g2_{i}"], g[f"gb2_{i}"] = G["dg2"][i].view(1, -1), G["dgb2"][i]
        return (None, None, None) + tuple(g[n] for n in _GF_ORDER)
