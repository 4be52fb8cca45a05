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


# (fp32 precision runs on v_mfma_f32_32x32x2_f32: the six-product bf16 form of gemm_split.hip was measured on the
#  real-size training batches -- 54 k rows, N = 64 / 320 -- at 35 us per launch against 26 us, 7.02 vs 6.58 ms per
#  replayed step, and pushes one gradient tensor to 2e-3 of the oracle's: not used here)
def _mm_fwd(a1, a2, wt, bias, act, slope, out=None):
    """act([a1 | a2] @ wt + bias) in the training precision (see Linear)."""
    if PRECISION == "bf16" and wt.shape[1] % 64 == 0 and wt.shape[0] % 64 == 0:
        return ops.gemm_bf16(a1, ops.round_bf16(wt.t()), bias, a2=a2, act=act, slope=slope, out=out)
    return ops.gemm(a1, wt, bias, a2=a2, act=act, slope=slope, out=out)


def _mm_bwd_da(dz, wt, out=None):
    """dA = dZ @ wt^T  ([m, n] x [k, n]^T -> [m, k])."""
    if PRECISION == "bf16" and wt.shape[1] % 64 == 0 and wt.shape[0] % 64 == 0:
        return ops.gemm_bf16(dz, ops.round_bf16(wt), out=out)
    return ops.gemm(dz, wt.t().contiguous(), out=out)


class ShmpTrunk(torch.autograd.Function):
    """BaseGNNCore.forward's SAGE loop (gnn_model.py:253-277) + anchor_mlp on the canonical rows (:69-73) +
    global_add_pool per layer block (:88-89, 107) as ONE autograd node: pooled [B, 64 (L+1)] from x0 [N, 64]
    and the folded weights STACKED over the layers.  Forward and backward are C-ABI launches on buffers this
    node owns -- no torch.cat / slice / accumulate kernels between them (the per-op autograd wiring of round 2
    spent as much GPU time in ~900 torch glue launches per step as in the 216 kernels that do the work).

    args: x0, batch (NeighborhoodBatch / QueryBatch), groups [(type, r0, r1, su)], has_anchor, then tensors:
          [anchor wt (K-major [P, P]), anchor bias] if has_anchor, then per group Wt [L, (su+1) 64, 64] and
          bias [L, 64] (gnn_model.pack_shmp_stacked).
    Gradient of a row of X_l = pooling broadcast + (canonical rows) its column block of d(anchor operand) +
    self block + transposed gather of the aggregate blocks of layer l: assembled in one buffer per layer."""

    @staticmethod
    def forward(ctx, x0, batch, groups, has_anchor, *w):
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
        for l in range(num_layers):
            agg = ops.csr_gather_sum(X[-1], batch.vrowptr, batch.vcol, N, S)          # [N, S*64]
            xn = xall[l]
            if PRECISION == "fp32" and len(groups) > 1:
                # the row types' products of one layer are independent: one launch (desco_gemm_f32_multi)
                ops.gemm_multi([dict(a1=agg[r0:r1, :su * H], a2=X[-1][r0:r1], wt=Wt[g][l], bias=Bs[g][l],
                                     act=ops.ACT_RELU, out=xn[r0:r1]) for g, (t, r0, r1, su) in enumerate(groups)])
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
            for l, xl in enumerate(X):
                canon[:, l * H:(l + 1) * H] = xl[Nc:]
            anch = _mm_fwd(canon, None, aw.contiguous(), ab, ops.ACT_LEAKY, 0.1)
        ops.segment_sum(X[0][:Nc], seg_ptr, B, extra=None if anch is None else anch[:, :H], out=pooled[:, :H])
        ops.segment_sum_layers(xall, Nc, seg_ptr, B, None if anch is None else anch[:, H:], pooled[:, H:])
        ctx.batch, ctx.groups, ctx.has_anchor = batch, groups, has_anchor
        ctx.X, ctx.AGG, ctx.canon, ctx.anch = X, AGG, canon, anch
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
                                   None if dcanon is None else dcanon[:, l * H:(l + 1) * H], mask)

        # dZ of the last layer: its rows feed only the pooling / the anchor operand (one zero row stands in for D:
        # row stride 0, no transposed edges)
        zero_d = torch.zeros((1, (S + 1) * H), device=dev).expand(N, -1)
        empty_ptr = torch.zeros(N + 1, device=dev, dtype=torch.int32)
        dz = ops.shmp_bwd_dx(zero_d, empty_ptr, ti["t_col_s1"], Nc, off_count, off_canon,
                             dpooled[:, L * H:(L + 1) * H], ti["seg_id"],
                             None if dcanon is None else dcanon[:, L * H:(L + 1) * H], X[L])
        D = torch.empty((N, (S + 1) * H), device=dev)
        fp32 = PRECISION == "fp32"
        # transposed weights of all layers in one copy per row type (dA = dZ Wt^T wants the n-major operand)
        WtT = [w_.transpose(1, 2).contiguous() for w_ in Wt] if fp32 else None
        wgrad = []                     # the weight / bias gradients feed nothing but the optimizer: formed together,
        for l in range(L - 1, -1, -1):  # 16 per launch pair, after the chain of input gradients
            live = [(g, r0, r1, su) for g, (t, r0, r1, su) in enumerate(groups)]
            for g, r0, r1, su in live:
                wgrad.append(dict(a1=AGG[l][r0:r1, :su * H], a2=X[l][r0:r1], dz=dz[r0:r1], dwt=grads[k + 2 * g][l],
                                  dbias=grads[k + 2 * g + 1][l]))
            if fp32 and len(groups) > 1:
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
        return (dxn, None, None, None) + tuple(grads)


class ShmpTrunkSmall(torch.autograd.Function):
    """ShmpTrunk for a small single-type batch (the query graphs, at most ops.shmp_trunk_small_max_rows() rows, two
    relation slots): the whole trunk is one launch forward and one backward (csrc/shmp_small.hip) instead of 37.
    args: x0 [n, 64], batch (QueryBatch), Wt [L, 192, 64], bias [L, 64]; returns pooled [B, 64 (L + 1)].
    Products are fp32 in either training precision (1.7 MFLOP per layer: nothing to gain from bf16 operands)."""

    @staticmethod
    def forward(ctx, x0, batch, wt, bias):
        x0, wt, bias = x0.contiguous(), wt.contiguous(), bias.contiguous()
        xall, pooled = ops.shmp_trunk_small_fwd(x0, batch.vrowptr, batch.vcol, wt, bias, batch.graph_ptr,
                                                batch.num_graphs)
        ctx.batch = batch
        ctx.save_for_backward(x0, xall, wt)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x0, xall, wt = ctx.saved_tensors
        batch = ctx.batch
        ti = batch.train_index()
        dwt, dbias, dx0 = ops.shmp_trunk_small_bwd(x0, xall, batch.vrowptr, batch.vcol, ti["t_rowptr"], ti["t_col_s1"],
                                                   ti["seg_id"], wt.transpose(1, 2).contiguous(), dpooled.contiguous())
        return dx0, None, dwt, dbias


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
            return None, torch.zeros((feat.shape[1], dout.shape[1]), device=dout.device), \
                torch.zeros((dout.shape[1],), device=dout.device)
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

    args: batch (rowptr / col), num_nodes, num_q, C6 [R,6], C3 [R,3], C2 [R,2] (constants), then tensors with
    gradients: V0 [Q,6,64], g1 [Q], wt1 [128,64], V1 [Q,3,64], wtp [128,64], Vp [Q,2,64], w3t [64,64], b3 [64],
    w5t [64,256], b5 [256], w7 [256].  Returns corr0 [R] = post_mp.7's product WITHOUT its bias."""

    @staticmethod
    def forward(ctx, rowptr, col, n, q, C6, C3, C2, V0, g1, wt1, V1, wtp, Vp, w3t, b3, w5t, b5, w7):
        V0, V1, Vp = V0.contiguous(), V1.contiguous(), Vp.contiguous()
        wt1, wtp, w3t, w5t = wt1.contiguous(), wtp.contiguous(), w3t.contiguous(), w5t.contiguous()
        g1, b3, b5, w7 = g1.contiguous(), b3.contiguous(), b5.contiguous(), w7.contiguous()
        h1 = ops.affine_rows(None, C6, V0, ops.ACT_RELU, 0.0)                       # layer 0 (closed form)
        hh = ops.gossip_gather(h1, rowptr, col, n, q, g1)                           # layer 1 aggregate
        h2 = ops.affine_rows(ops.gemm(hh, wt1, a2=h1), C3, V1, ops.ACT_RELU, 0.0)
        y = ops.affine_rows(ops.gemm(h1, wtp, a2=h2), C2, Vp, ops.ACT_LEAKY, 0.1)   # post_mp.0 + .2
        y3 = ops.gemm(y, w3t, b3, act=ops.ACT_RELU)
        y5 = ops.gemm(y3, w5t, b5, act=ops.ACT_RELU)
        corr = ops.rowdot_add(y5, w7, 0.0, None)
        ctx.save_for_backward(rowptr, col, C6, C3, C2, g1, wt1, wtp, w3t, w5t, w7, h1, hh, h2, y, y3, y5)
        ctx.n, ctx.q = n, q
        return corr

    @staticmethod
    def backward(ctx, dcorr):
        rowptr, col, C6, C3, C2, g1, wt1, wtp, w3t, w5t, w7, h1, hh, h2, y, y3, y5 = ctx.saved_tensors
        n, q = ctx.n, ctx.q
        R = h1.shape[0]
        dev = h1.device
        dz5, dwb7 = ops.rowdot_bwd(y5, w7, dcorr.contiguous())                      # [R,256], (dw7 | .)
        # dA = dZ W^T wants W itself as the [K = out, N = in] operand: the transposes of the (tiny) folded weights
        w5 = w5t.t().contiguous()                                                    # [256, 64]
        w3 = w3t.t().contiguous()
        wp = wtp.view(2, 64, 64).transpose(1, 2).contiguous()                        # [(h1 | h2) block][out][in]
        w1 = wt1.view(2, 64, 64).transpose(1, 2).contiguous()                        # [(hh | h1) block][out][in]
        dz3 = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dz5, wt=w5, out=dz3, gate=y3, gate_act=ops.ACT_RELU)])
        dzp = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dz3, wt=w3, out=dzp, gate=y, gate_act=ops.ACT_LEAKY, gate_slope=0.1)])
        dVp = ops.affine_rows_bwd(C2, dzp, q)
        dh1 = torch.empty((R, 64), device=dev)
        dz1 = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dzp, wt=wp[0], out=dh1),
                        dict(a1=dzp, wt=wp[1], out=dz1, gate=h2, gate_act=ops.ACT_RELU)])
        dV1 = ops.affine_rows_bwd(C3, dz1, q)
        dhh = torch.empty((R, 64), device=dev)
        ops.gemm_multi([dict(a1=dz1, wt=w1[0], out=dhh),
                        dict(a1=dz1, wt=w1[1], out=dh1, accum=True)])
        # transpose of the gated sum: the same kernel with 1 - g (GossipGather.backward)
        ops.add_rows(dh1, ops.gossip_gather(dhh, rowptr, col, n, q, (1.0 - g1).contiguous()))
        dsig = ops.gossip_gather(h1, rowptr, col, n, q, None)                       # d out / d g1
        dg1 = ops.colsum(ops.rowdot2(dhh, dsig).view(n, q))
        dz0 = ops.act_grad(dh1, h1, ops.ACT_RELU, 0.0)
        dV0 = ops.affine_rows_bwd(C6, dz0, q)
        # the four weight / bias gradients wait for nothing and nothing but the optimizer waits for them: one launch pair
        dw5t, db5 = torch.empty_like(w5t), torch.empty((256,), device=dev)
        dw3t, db3 = torch.empty_like(w3t), torch.empty((64,), device=dev)
        dwtp, dwt1 = torch.empty_like(wtp), torch.empty_like(wt1)
        ops.linear_bwd_w_multi([dict(a1=y3, dz=dz5, dwt=dw5t, dbias=db5), dict(a1=y, dz=dz3, dwt=dw3t, dbias=db3),
                                dict(a1=h1, a2=h2, dz=dzp, dwt=dwtp), dict(a1=hh, a2=h1, dz=dz1, dwt=dwt1)])
        return (None, None, None, None, None, None, None, dV0, dg1, dwt1, dV1, dwtp, dVp, dw3t, db3, dw5t, db5,
                dwb7[:256])


class Mlp(torch.autograd.Function):
    """post_mp (gnn_model.py:40-53 of the reference) as ONE autograd node: h_{i+1} = act_i(h_i @ wt_i + b_i).

    args: x [M, k0], acts ((act, slope), ...), w_nk (the nn.Linear weights [out, in] as they are stored: the operand
    of dA = dZ W, no gradient asked of them here), then wt_0, b_0, wt_1, b_1, ... (wt_i = weight_i^T [in, out],
    differentiable).  Backward: one GEMM per layer with the activation derivative in its epilogue (gemm_multi's gate),
    all weight / bias gradients in one launch pair."""

    @staticmethod
    def forward(ctx, x, acts, w_nk, *wb):
        hs = [x.contiguous()]
        wts = []
        for i, (act, slope) in enumerate(acts):
            wt, b = wb[2 * i].contiguous(), wb[2 * i + 1]
            wts.append(wt)
            hs.append(_mm_fwd(hs[-1], None, wt, b, act, slope))
        ctx.acts = acts
        ctx.save_for_backward(*hs, *[w.detach() for w in w_nk], *wts)
        return hs[-1]

    @staticmethod
    def backward(ctx, dout):
        n = len(ctx.acts)
        sv = ctx.saved_tensors
        hs, w_nk, wts = sv[:n + 1], sv[n + 1:2 * n + 1], sv[2 * n + 1:]
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
            if PRECISION == "bf16" and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
                _mm_bwd_da(dz, wts[i], out=da)
                if i > 0 and ctx.acts[i - 1][0] != ops.ACT_NONE:
                    da = ops.act_grad(da, hs[i], *ctx.acts[i - 1])
            else:
                pr = dict(a1=dz, wt=w if w.is_contiguous() else w.contiguous(), out=da)
                if i > 0 and ctx.acts[i - 1][0] != ops.ACT_NONE:
                    pr.update(gate=hs[i], gate_act=ctx.acts[i - 1][0], gate_slope=ctx.acts[i - 1][1])
                ops.gemm_multi([pr])
            dz = da
        ops.linear_bwd_w_multi(wgrad)
        return (dz if ctx.needs_input_grad[0] else None, None, None) + tuple(grads)

