"""Kernel-level parity: every C-ABI device entry point vs. a plain torch fp32/fp64 formula."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from desco_amd import ops  # noqa: E402
from helpers import assert_counts_close, assert_logits_close  # noqa: E402

DEV = "cuda"


def _close(got, ref, rtol=2e-5, atol=2e-5):
    """One kernel against the fp64 evaluation of the same operation (NOT a model-level gate: those all go through
    helpers.assert_logits_close / assert_counts_close).  Defaults 2e-5; a call that passes its own atol states the
    accumulation it covers (split-K sums over m rows, gradients over a batch)."""
    torch.testing.assert_close(got.detach().cpu().double(), ref.double(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("m", [1, 29, 127, 128, 129, 1000, 4097])
@pytest.mark.parametrize("k1,k2,n", [(64, 0, 64), (256, 64, 64), (128, 64, 64), (576, 0, 576),
                                     (64, 0, 256), (64, 64, 64), (576, 0, 64), (256, 0, 64)])
def test_gemm_shapes(m, k1, k2, n):
    g = torch.Generator().manual_seed(m * 7 + k1 + n)
    a1 = torch.randn(m, k1, generator=g)
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    wt = torch.randn(k1 + k2, n, generator=g) / np.sqrt(k1 + k2)
    bias = torch.randn(n, generator=g)
    A = a1 if a2 is None else torch.cat([a1, a2], 1)
    ref = torch.relu(A.double() @ wt.double() + bias.double())
    got = ops.gemm(a1.to(DEV), wt.to(DEV), bias.to(DEV), a2=None if a2 is None else a2.to(DEV),
                   act=ops.ACT_RELU)
    _close(got, ref)


def test_gemm_epilogue_and_strides():
    g = torch.Generator().manual_seed(3)
    Q, N = 29, 37
    m = Q * N
    slab = torch.randn(m + 5, 576, generator=g).to(DEV)
    a1 = slab[:m, 64:128]                    # strided view, ld = 576
    agg = torch.randn(m, 256, generator=g).to(DEV)
    a2 = agg[:, :128]                        # ld = 256, k = 128
    wt = (torch.randn(64 + 128, 64, generator=g) / 14).to(DEV)
    bias = torch.randn(Q, 64, generator=g).to(DEV)
    s = torch.randn(m, 2, generator=g).to(DEV)
    ws = torch.randn(2, 64, generator=g).to(DEV)
    out = torch.zeros(m, 576, device=DEV)
    ops.gemm(a1, wt, bias, a2=a2, act=ops.ACT_LEAKY, slope=0.1, s=s, ws=ws, out=out[:, 128:192])
    A = torch.cat([a1, a2], 1).double().cpu()
    ref = A @ wt.double().cpu() + bias.double().cpu().repeat(N, 1) + s.double().cpu() @ ws.double().cpu()
    ref = torch.nn.functional.leaky_relu(ref, 0.1)
    _close(out[:, 128:192], ref)
    assert out[:, :128].abs().sum().item() == 0 and out[:, 192:].abs().sum().item() == 0


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric B catches a transposed C/D map (guide section 3)."""
    wt = torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64)
    a = torch.eye(64)
    got = ops.gemm(a.to(DEV), wt.to(DEV))
    assert torch.equal(got.cpu(), wt)


def test_gemm_rejects_cpu_and_bad_shapes():
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 64), torch.zeros(64, 64))
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 48, device=DEV), torch.zeros(48, 64, device=DEV))


def _random_vcsr(num_rows, slots, max_deg, n_src, g):
    cnt = torch.randint(0, max_deg + 1, (num_rows * slots,), generator=g)
    cnt[::7] = 0
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(cnt, 0)])
    col = torch.randint(0, n_src, (int(ptr[-1]),), generator=g)
    return ptr.to(torch.int32), col.to(torch.int32), cnt


@pytest.mark.parametrize("slots", [1, 2, 4])
@pytest.mark.parametrize("num_rows,max_deg", [(1, 3), (257, 5), (1000, 40)])
def test_csr_gather_sum(slots, num_rows, max_deg):
    g = torch.Generator().manual_seed(slots * 100 + num_rows)
    n_src = num_rows + 11
    slab = torch.randn(n_src, 576, generator=g)
    x = slab[:, 64:128]
    ptr, col, cnt = _random_vcsr(num_rows, slots, max_deg, n_src, g)
    ref = torch.zeros(num_rows * slots, 64, dtype=torch.double)
    vrow = torch.repeat_interleave(torch.arange(num_rows * slots), cnt)
    ref.index_add_(0, vrow, x.double()[col.long()])
    got = ops.csr_gather_sum(slab.to(DEV)[:, 64:128], ptr.to(DEV), col.to(DEV), num_rows, slots)
    _close(got.reshape(num_rows * slots, 64), ref, atol=1e-5)


def test_segment_sum_and_extra():
    g = torch.Generator().manual_seed(5)
    sizes = torch.randint(0, 30, (200,), generator=g)
    sizes[3] = 0
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(sizes, 0)])
    x = torch.randn(int(ptr[-1]), 64, generator=g)
    extra = torch.randn(200, 576, generator=g)
    ref = torch.zeros(200, 64, dtype=torch.double)
    ref.index_add_(0, torch.repeat_interleave(torch.arange(200), sizes), x.double())
    ref += extra[:, 128:192].double()
    out = torch.zeros(200, 576, device=DEV)
    ops.segment_sum(x.to(DEV), ptr.to(torch.int32).to(DEV), 200, extra=extra.to(DEV)[:, 128:192],
                    out=out[:, 64:128])
    _close(out[:, 64:128], ref, atol=1e-5)


@pytest.mark.parametrize("B,Q,hid,slope", [(333, 29, 256, 0.01),        # small batch: query groups of 8
                                            (140_000, 29, 256, 0.01),   # the 29-accumulator kernel
                                            (135_000, 7, 128, 0.01),    # padded accumulators, hid < 256
                                            (500, 32, 64, 1.5),         # slope outside [0, 1]
                                            (131_073, 29, 256, 0.0)])   # ragged last block, plain relu
def test_count_head(B, Q, hid, slope):
    g = torch.Generator().manual_seed(6)
    t, qh = torch.randn(B, hid, generator=g), torch.randn(Q, hid, generator=g)
    w2, b2 = torch.randn(hid, generator=g) / 16, 0.3
    idx = torch.randperm(B, generator=g)[:300]         # reference on a row sample (fp64, CPU)
    pre = torch.nn.functional.leaky_relu(t[idx].double()[:, None, :] + qh.double()[None], slope)
    ref = pre @ w2.double() + b2
    got = ops.count_head(t.to(DEV), qh.to(DEV), w2.to(DEV), b2, slope, False)
    assert got.shape == (B, Q)
    _close(got[idx.to(DEV)], ref)
    got2 = ops.count_head(t.to(DEV), qh.to(DEV), w2.to(DEV), torch.tensor(b2, device=DEV), slope, True)
    assert_counts_close("count head, exp2 - 1 form", got2[idx.to(DEV)], 2 ** ref - 1)


@pytest.mark.parametrize("B", [1, 31, 33, 1000, 70001])
@pytest.mark.parametrize("slope", [0.01, 0.2])
def test_count_head_from_the_embeddings(B, slope):
    """desco_count_head_emb_f16x3_f32: the count head with count_model.0's target half formed inside the launch
    (lightning_model.py:127-131, 176-193) against fp64 and against the two launches it replaces (linear64 + count_head);
    ragged 32-row tiles, rows of very different magnitude, both output forms."""
    g = torch.Generator().manual_seed(B)
    emb = torch.randn(B, 64, generator=g) * (torch.rand(B, 1, generator=g) * 30 + 0.01)
    wt = torch.randn(256, 64, generator=g) / 8
    qh = torch.randn(29, 256, generator=g)
    w2, b2 = torch.randn(256, generator=g) / 16, 0.3
    idx = torch.randperm(B, generator=g)[:300]
    t64 = emb[idx].double() @ wt.double().t()
    pre = torch.nn.functional.leaky_relu(t64[:, None, :] + qh.double()[None], slope)
    ref = pre @ w2.double() + b2
    mag = (emb[idx].double().abs() @ wt.double().abs().t())[:, None, :] + qh.double().abs()[None]
    mag = mag @ w2.double().abs() + abs(b2)
    planes = ops.split_f16_planes(wt.to(DEV))
    got = ops.count_head_emb(emb.to(DEV), planes, qh.to(DEV), w2.to(DEV), b2, slope, False)
    two = ops.count_head(ops.linear64(emb.to(DEV), ops.linear64_planes(wt.to(DEV))), qh.to(DEV), w2.to(DEV), b2, slope,
                         False)
    assert got.shape == (B, 29)
    e1 = ((got[idx.to(DEV)].cpu().double() - ref).abs() / mag).max().item()
    e2 = ((two[idx.to(DEV)].cpu().double() - ref).abs() / mag).max().item()
    print(f"[head from emb] B={B} slope={slope}: one launch {e1:.2e}  two launches {e2:.2e} (max err / magnitude sum)")
    assert e1 <= max(2.0 * e2, 1.2e-7)          # (one fp32 rounding unit of the magnitude sum: the max(t, -q) + q form)
    _close(got[idx.to(DEV)], ref)
    got2 = ops.count_head_emb(emb.to(DEV), planes, qh.to(DEV), w2.to(DEV), torch.tensor(b2, device=DEV), slope, True)
    small = ref.abs().max(1).values < 20                  # (2^logit - 1 within fp32 range)
    assert_counts_close("count head from emb, exp2 - 1 form", got2[idx.to(DEV)][small.to(DEV)], 2 ** ref[small] - 1)
    with pytest.raises(Exception):
        ops.count_head_emb(emb.to(DEV), planes, qh[:28].to(DEV), w2.to(DEV), b2, slope, False)      # 29 queries only


def test_scatter_rows_and_linear_smallk_and_rowdot():
    g = torch.Generator().manual_seed(8)
    src = torch.randn(50, 29, generator=g)
    rows = torch.randperm(80, generator=g)[:50].to(torch.int32)
    dst = torch.zeros(80, 29, device=DEV)
    ops.scatter_rows(src.to(DEV), rows.to(DEV), dst)
    ref = torch.zeros(80, 29)
    ref[rows.long()] = src
    assert torch.equal(dst.cpu(), ref)

    feat = torch.randn(123, 3, generator=g)
    wt, b = torch.randn(3, 64, generator=g), torch.randn(64, generator=g)
    got = ops.linear_smallk(feat.to(DEV), wt.to(DEV), b.to(DEV))
    _close(got, feat.double() @ wt.double() + b.double())

    y, w, add = torch.randn(777, 256, generator=g), torch.randn(256, generator=g), torch.randn(777, generator=g)
    got = ops.rowdot_add(y.to(DEV), w.to(DEV), 0.25, add.to(DEV))
    _close(got, y.double() @ w.double() + 0.25 + add.double(), atol=1e-4)


def _sym_csr(n, m, g):
    a = torch.randint(0, n, (m,), generator=g)
    b = torch.randint(0, n, (m,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * n + b, b * n + a]))
    r, c = key // n, key % n
    ptr = torch.zeros(n + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return ptr.to(torch.int32), c.to(torch.int32), r, c


def test_gossip_layer0_and_gather():
    g = torch.Generator().manual_seed(9)
    n, Q = 150, 29
    ptr, col, r, c = _sym_csr(n, 400, g)
    x = torch.rand(n, Q, generator=g) * 20
    g0, g1 = torch.rand(Q, generator=g), torch.rand(Q, generator=g)
    p, z = torch.randn(Q, 64, generator=g), torch.randn(Q, 64, generator=g)
    rv, tv = torch.randn(64, generator=g), torch.randn(64, generator=g)
    h1, scal = ops.gossip_layer0(x.to(DEV), ptr.to(DEV), col.to(DEV), g0.to(DEV), g1.to(DEV),
                                 p.to(DEV), rv.to(DEV), tv.to(DEV), z.to(DEV))
    # dense reference: edge (src=c -> dst=r); lo iff src < dst
    lo = (c < r).double()
    xd = x.double()
    deg_lo = torch.zeros(n, dtype=torch.double).index_add_(0, r, lo)
    deg_hi = torch.zeros(n, dtype=torch.double).index_add_(0, r, 1 - lo)
    s_lo = torch.zeros(n, Q, dtype=torch.double).index_add_(0, r, xd[c] * lo[:, None])
    s_hi = torch.zeros(n, Q, dtype=torch.double).index_add_(0, r, xd[c] * (1 - lo)[:, None])
    a0 = g0.double() * deg_lo[:, None] + (1 - g0.double()) * deg_hi[:, None]
    b0 = g0.double() * s_lo + (1 - g0.double()) * s_hi
    a1 = g1.double() * deg_lo[:, None] + (1 - g1.double()) * deg_hi[:, None]
    ref_h1 = torch.relu(a0[..., None] * p.double() + b0[..., None] * rv.double() +
                        xd[..., None] * tv.double() + z.double())
    _close(h1.view(n, Q, 64), ref_h1, rtol=1e-4, atol=1e-3)
    _close(scal.view(n, Q, 2)[..., 0], a1, atol=1e-5)
    _close(scal.view(n, Q, 2)[..., 1], xd)

    h = torch.randn(n * Q, 64, generator=g)
    got = ops.gossip_gather(h.to(DEV), ptr.to(DEV), col.to(DEV), n, Q, g1.to(DEV))
    hd = h.double().view(n, Q, 64)
    w = lo[:, None] * g1.double() + (1 - lo)[:, None] * (1 - g1.double())      # [E,Q]
    ref = torch.zeros(n, Q, 64, dtype=torch.double).index_add_(0, r, hd[c] * w[..., None])
    _close(got.view(n, Q, 64), ref, atol=1e-4)


@pytest.mark.parametrize("S,sm,st", [(4, 3, 0), (4, 2, 0), (2, 2, 0), (4, 2, 2), (1, 1, 0), (4, 0, 1), (4, 3, 1)])
@pytest.mark.parametrize("num_rows,row0,max_deg", [(1, 0, 3), (63, 5, 4), (256, 0, 2), (1000, 17, 9),
                                                   (333, 0, 70), (70000, 3, 3), (70000, 0, 6)])
@pytest.mark.parametrize("x6", [False, True, "f16x3"])
def test_fused_shmp_layer(S, sm, st, num_rows, row0, max_deg, x6):
    if x6 and sm > 2:
        pytest.skip("the split forms hold at most three resident weight blocks")
    g = torch.Generator().manual_seed(S * 1000 + sm * 100 + st * 10 + num_rows)
    n_all = row0 + num_rows + 9
    x = torch.randn(n_all, 64, generator=g)
    ptr, col, cnt = _random_vcsr(n_all, S, max_deg, n_all, g)
    wt = torch.randn((sm + 1) * 64, 64, generator=g) / 12
    bias = torch.randn(64, generator=g)
    agg = torch.zeros(n_all * S, 64, dtype=torch.double)
    agg.index_add_(0, torch.repeat_interleave(torch.arange(n_all * S), cnt), x.double()[col.long()])
    aggv = agg.view(n_all, S * 64)
    A = torch.cat([aggv[:, :sm * 64], x.double()], 1)
    ref = A @ wt.double() + bias.double()
    ytab = None
    if st:
        wtab = torch.randn(64, 64 * st, generator=g) / 8
        ytab_cpu = x @ wtab                                   # fp32 table, as the product path builds it
        for s in range(st):
            ref = ref + aggv[:, (sm + s) * 64:(sm + s + 1) * 64] @ wtab.double()[:, s * 64:(s + 1) * 64]
        ytab = ytab_cpu.to(DEV)
    ref = torch.relu(ref)[row0:row0 + num_rows]
    out = torch.full((n_all, 64), -7.0, device=DEV)
    w_dev = (ops.split_f16_planes(wt.t().contiguous().to(DEV)) if x6 == "f16x3" else
             ops.split_bf16_planes(wt.t().contiguous().to(DEV)) if x6 else wt.to(DEV))
    ops.shmp_layer(x.to(DEV), ptr.to(DEV), col.to(DEV), row0, num_rows, S, sm, w_dev,
                   bias.to(DEV), out, ytab=ytab, ytab_row0=0)
    assert_logits_close("fused SHMP layer vs fp64", out[row0:row0 + num_rows], ref)
    rest = torch.cat([out[:row0], out[row0 + num_rows:]])
    assert (rest == -7.0).all()          # rows outside the range are untouched


@pytest.mark.parametrize("S,sm,st", [(4, 2, 2), (4, 2, 0), (2, 1, 0)])
@pytest.mark.parametrize("f16", [False, True])
def test_fused_shmp_layer_strided_operands(S, sm, st, f16):
    """The general-stride instantiations of the bf16x6 layer kernel (LD64 = false): x, the table and the
    output are column blocks of wider tensors (ldx = 96, ldy = 64 st + 32, ldo = 128)."""
    num_rows, row0, max_deg = 3000, 16, 7
    g = torch.Generator().manual_seed(77 + S + sm + st)
    n_all = row0 + num_rows + 5
    xw = torch.randn(n_all, 96, generator=g)
    x = xw[:, 16:80]
    ptr, col, cnt = _random_vcsr(n_all, S, max_deg, n_all, g)
    wt = torch.randn((sm + 1) * 64, 64, generator=g) / 12
    bias = torch.randn(64, generator=g)
    agg = torch.zeros(n_all * S, 64, dtype=torch.double)
    agg.index_add_(0, torch.repeat_interleave(torch.arange(n_all * S), cnt), x.double()[col.long()])
    aggv = agg.view(n_all, S * 64)
    ref = torch.cat([aggv[:, :sm * 64], x.double()], 1) @ wt.double() + bias.double()
    ytab = None
    if st:
        wtab = torch.randn(64, 64 * st, generator=g) / 8
        yw = torch.zeros(n_all, 64 * st + 32)
        yw[:, :64 * st] = x @ wtab
        for s in range(st):
            ref = ref + aggv[:, (sm + s) * 64:(sm + s + 1) * 64] @ wtab.double()[:, s * 64:(s + 1) * 64]
        ytab = yw.to(DEV)[:, :64 * st]
    ref = torch.relu(ref)[row0:row0 + num_rows]
    outw = torch.full((n_all, 128), -7.0, device=DEV)
    out = outw[:, 32:96]
    split = ops.split_f16_planes if f16 else ops.split_bf16_planes
    ops.shmp_layer(xw.to(DEV)[:, 16:80], ptr.to(DEV), col.to(DEV), row0, num_rows, S, sm,
                   split(wt.t().contiguous().to(DEV)), bias.to(DEV), out, ytab=ytab, ytab_row0=0)
    assert_logits_close("fused SHMP layer vs fp64", out[row0:row0 + num_rows], ref)
    assert (outw[:, :32] == -7.0).all() and (outw[:, 96:] == -7.0).all() and (outw[:row0] == -7.0).all()


@pytest.mark.parametrize("m,k1,k2,n", [(1, 64, 0, 64), (300, 256, 64, 64), (5000, 128, 64, 64), (70001, 576, 0, 576),
                                       (4097, 64, 0, 256)])
def test_linear_bwd_w_fused_weight_and_bias_gradient(m, k1, k2, n):
    """desco_linear_bwd_w_f32: dWt = [A1 | A2]^T dZ and db = colsum(dZ) in one pass (strided operands as
    the training path passes them: A1 is a column block of the wider aggregate tensor)."""
    g = torch.Generator().manual_seed(m + k1 + n)
    wide = torch.randn(m, k1 + 64, generator=g)
    a1 = wide[:, :k1]
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    dz = torch.randn(m, n, generator=g)
    A = a1 if a2 is None else torch.cat([a1, a2], 1)
    ref_w = A.double().T @ dz.double()
    ref_b = dz.double().sum(0)
    dwt, db = ops.linear_bwd_w(wide.to(DEV)[:, :k1], None if a2 is None else a2.to(DEV), dz.to(DEV), True)
    _close(dwt, ref_w, rtol=1e-4, atol=1e-3 * max(1.0, m ** 0.5 / 10))
    _close(db, ref_b, rtol=1e-4, atol=1e-3 * max(1.0, m ** 0.5 / 10))
    dwt2, none = ops.linear_bwd_w(wide.to(DEV)[:, :k1], None if a2 is None else a2.to(DEV), dz.to(DEV), False)
    assert none is None and torch.equal(dwt2, dwt)          # deterministic, bias row optional
    # same numbers as the separate entry points
    sep = ops.gemm_tn(wide.to(DEV)[:, :k1], dz.to(DEV))
    _close(dwt[:k1], sep.double().cpu(), rtol=1e-5, atol=1e-4 * max(1.0, m ** 0.5 / 10))


@pytest.mark.parametrize("num_rows,seg_kind", [(1, "one"), (31, "tiny"), (32, "tiny"), (1000, "mixed"),
                                               (70001, "mixed"), (70000, "huge"), (4096, "single-rows")])
def test_fused_pooling_equals_segment_sum(num_rows, seg_kind):
    """desco_shmp_layer_pool_bf16x6_f32 + desco_pool_reduce_f32 == the layer followed by
    desco_segment_sum_f32 (global_add_pool, gnn_model.py:107): produced rows bit-identical to the plain
    launch, segment sums equal to fp32 rounding (a segment's partials are added tile by tile);
    segments of one row, segments spanning 60+ tiles, ragged last tile, rows not stored."""
    g = torch.Generator().manual_seed(num_rows + len(seg_kind))
    S, sm, st = 4, 2, 2
    n_tab = 50
    x = torch.randn(num_rows + n_tab, 64, generator=g)
    ptr, col, cnt = _random_vcsr(num_rows + n_tab, S, 3, num_rows, g)
    # table slots (2, 3) read the table rows [num_rows, num_rows + n_tab)
    vrow = torch.repeat_interleave(torch.arange((num_rows + n_tab) * S), cnt)
    is_tab = (vrow % S) >= sm
    col = torch.where(is_tab, num_rows + col % n_tab, col % num_rows).to(torch.int32)
    if seg_kind == "one":
        lens = [num_rows]
    elif seg_kind == "single-rows":
        lens = [1] * num_rows
    else:
        hi = {"tiny": 3, "mixed": 40, "huge": 3000}[seg_kind]
        lens, left = [], num_rows
        while left > 0:
            k = int(torch.randint(1, hi + 1, (1,), generator=g))
            k = min(k, left)
            lens.append(k)
            left -= k
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    B = len(lens)

    class _P:           # the slice of NeighborhoodPartition that NeighborhoodBatch.pool_index reads
        count_ptr = seg
    from desco_amd.batch import NeighborhoodBatch
    nb = NeighborhoodBatch.__new__(NeighborhoodBatch)
    nb.part, nb.device = _P, torch.device(DEV, torch.cuda.current_device())
    bits, slot, nslots = nb.pool_index()
    TR = ops.pool_tile_rows()
    assert TR in (16, 32) and nslots <= B + (num_rows + TR - 1) // TR
    wt = torch.randn(3 * 64, 64, generator=g) / 12
    bias = torch.randn(64, generator=g)
    planes = ops.split_bf16_planes(wt.t().contiguous().to(DEV))
    ytab = torch.randn(n_tab, 128, generator=g).to(DEV)
    xd, ptrd, cold, segd = x.to(DEV), ptr.to(DEV), col.to(DEV), torch.from_numpy(seg).to(DEV)
    plain = torch.empty(num_rows, 64, device=DEV)
    ops.shmp_layer(xd, ptrd, cold, 0, num_rows, S, sm, planes, bias.to(DEV), plain, ytab=ytab, ytab_row0=num_rows)
    ref = ops.segment_sum(plain, segd, B)
    extra = torch.randn(B, 64, generator=g).to(DEV)
    for store in (True, False):
        part = torch.full((nslots, 64), float("nan"), device=DEV)
        out = torch.full((num_rows, 64), -7.0, device=DEV) if store else None
        ops.shmp_layer(xd, ptrd, cold, 0, num_rows, S, sm, planes, bias.to(DEV), out, ytab=ytab,
                       ytab_row0=num_rows, pool=(bits, slot, part))
        assert torch.isfinite(part).all()                       # every slot written exactly once
        if store:
            assert torch.equal(out, plain)
        got = ops.pool_reduce(part, bits, slot, segd, B)
        _close(got, ref.double().cpu(), rtol=1e-5, atol=1e-4)
        got_e = ops.pool_reduce(part, bits, slot, segd, B, extra=extra)
        _close(got_e, (ref + extra).double().cpu(), rtol=1e-5, atol=1e-4)
    exact = torch.zeros(B, 64, dtype=torch.double).index_add_(
        0, torch.from_numpy(np.repeat(np.arange(B), lens)), plain.double().cpu())
    _close(got, exact, rtol=1e-5, atol=1e-4 * max(1, max(lens)) ** 0.5)


@pytest.mark.parametrize("m,k1,k2,n", [(1, 64, 0, 64), (300, 128, 64, 128), (1000, 576, 0, 576), (4097, 64, 0, 192)])
def test_gemm_bf16_rounds_operands_to_nearest_even(m, k1, k2, n):
    """bf16 training GEMM == exact product of the RNE-bf16-rounded operands (fp32 accumulation)."""
    g = torch.Generator().manual_seed(m + n + 1)
    a1 = torch.randn(m, k1, generator=g) * 5
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    bias = torch.randn(n, generator=g)
    A = a1 if a2 is None else torch.cat([a1, a2], 1)
    Ar, wr = A.bfloat16().double(), w.bfloat16().double()
    ref = torch.relu(Ar @ wr.T + bias.double())
    got = ops.gemm_bf16(a1.to(DEV), w.to(DEV), bias.to(DEV), a2=None if a2 is None else a2.to(DEV),
                        act=ops.ACT_RELU)
    _close(got, ref, rtol=2e-5, atol=2e-4)
    planes = ops.round_bf16(w.to(DEV))
    assert torch.equal(planes.cpu(), w.bfloat16().view(torch.int16))


@pytest.mark.parametrize("m,n,act", [(1, 64, 0), (257, 128, 1), (5000, 256, 2), (70001, 128, 0)])
def test_linear64_streaming(m, n, act):
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, 80, generator=g)[:, 8:72] * 3          # strided input view (ldx = 80)
    w = torch.randn(n, 64, generator=g) / 8
    bias = torch.randn(n, generator=g)
    ref = x.double() @ w.double().T + bias.double()
    ref = [ref, torch.relu(ref), torch.nn.functional.leaky_relu(ref, 0.1)][act]
    xd = x.to(DEV)
    got = ops.linear64(xd, ops.linear64_planes(w.to(DEV)), bias.to(DEV), act=act, slope=0.1)
    _close(got, ref, rtol=1e-5, atol=1e-5)
    nob = ops.linear64(xd, ops.linear64_planes(w.to(DEV)))
    _close(nob, x.double() @ w.double().T, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("quirk", [0, 7])
def test_device_transposed_index_matches_stable_sort(quirk):
    """desco_vcsr_transpose_sym == stable argsort by source (batch._transpose_index) on canonical
    partitions (4 slots, with and without the PyG-quirk edge drops) and on query blocks (2 slots)."""
    from helpers import golden_graphs, standard_queries
    from desco_amd.batch import NeighborhoodBatch, QueryBatch, _transpose_index
    from desco_amd.graphs import GraphSet
    from desco_amd.partition import build_partition
    part = build_partition(GraphSet.from_edge_lists(golden_graphs(max_n=60)), 4, quirk_batch=quirk)
    nb = NeighborhoodBatch(part, DEV)
    qb = QueryBatch(standard_queries()[1], DEV)
    for b in (nb, qb):
        ti = b.train_index()
        rp, tc = _transpose_index(b.vrowptr.cpu().numpy(), b.vcol.cpu().numpy(), b.num_rows)
        assert np.array_equal(ti["t_rowptr"].cpu().numpy(), rp)
        assert np.array_equal(ti["t_col"].cpu().numpy(), tc)
        seg_ptr = b._seg_ptr_host()
        seg_id = np.repeat(np.arange(len(seg_ptr) - 1), np.diff(seg_ptr))
        assert np.array_equal(ti["seg_id"].cpu().numpy(), seg_id)


def test_degree_affine():
    g = torch.Generator().manual_seed(11)
    n, S, row0 = 500, 4, 37
    ptr, col, cnt = _random_vcsr(n + row0, S, 6, n, g)
    coef = torch.randn(S + 1, 64, generator=g)
    extra = torch.randn(n, 576, generator=g)
    deg = cnt.view(n + row0, S).double()[row0:]
    ref = torch.nn.functional.leaky_relu(deg @ coef[:S].double() + coef[S].double(), 0.1) + extra[:, 64:128].double()
    out = torch.full((n + row0, 64), -3.0, device=DEV)
    ops.degree_affine(ptr.to(DEV), row0, n, S, coef.to(DEV), ops.ACT_LEAKY, 0.1, out,
                      extra=extra.to(DEV)[:, 64:128])
    _close(out[row0:], ref, atol=1e-4)
    assert (out[:row0] == -3.0).all()
    # the optional row bound (start of the anchor operand's per-row bound): exactly the row maxima of what was stored
    bound = torch.full((n + 3,), -1.0, device=DEV)
    out2 = torch.empty((n, 64), device=DEV)
    ops.degree_affine(ptr.to(DEV), row0, n, S, coef.to(DEV), ops.ACT_LEAKY, 0.1, out2, out_row0=0, row_absmax=bound)
    assert torch.equal(bound[:n], out2.abs().amax(1)) and (bound[n:] == -1.0).all()


@pytest.mark.parametrize("num_rows,row0", [(1, 0), (4099, 5), (70001, 0)])
def test_fused_shmp_layer_f16x3_accumulates_the_row_bound(num_rows, row0):
    """Canonical launches (second output = a column block of the anchor operand) raise row_absmax[i - row0] to the
    row's largest stored |value| and never lower it (desco_hip.h: accumulated over the launches of one operand)."""
    S, sm = 2, 2
    g = torch.Generator().manual_seed(num_rows)
    n_all = row0 + num_rows + 3
    x = torch.randn(n_all, 64, generator=g) * (torch.rand(n_all, 1, generator=g) * 50)
    ptr, col, cnt = _random_vcsr(n_all, S, 5, n_all, g)
    wt = torch.randn((sm + 1) * 64, 64, generator=g) / 12
    bias = torch.randn(64, generator=g)
    out = torch.empty((n_all, 64), device=DEV)
    canon = torch.zeros((num_rows, 192), device=DEV)
    prior = torch.rand(num_rows + 2, generator=g).to(DEV) * 40
    bound = prior.clone()
    ops.shmp_layer(x.to(DEV), ptr.to(DEV), col.to(DEV), row0, num_rows, S, sm,
                   ops.split_f16_planes(wt.t().contiguous().to(DEV)), bias.to(DEV), out, out2=canon[:, 64:128],
                   row_absmax=bound)
    assert torch.equal(canon[:, 64:128], out[row0:row0 + num_rows])
    assert torch.equal(bound[:num_rows], torch.maximum(prior[:num_rows], canon.abs().amax(1)))
    assert torch.equal(bound[num_rows:], prior[num_rows:])
    with pytest.raises(ValueError):
        ops.shmp_layer(x.to(DEV), ptr.to(DEV), col.to(DEV), row0, num_rows, S, sm, wt.to(DEV), bias.to(DEV), out,
                       row_absmax=bound)


@pytest.mark.parametrize("m,k1,k2,n", [(1, 64, 0, 64), (129, 128, 64, 64), (1000, 576, 0, 576), (4097, 64, 0, 256),
                                       (1153, 64, 32, 192), (2049, 96, 0, 384), (900, 64, 0, 128), (131, 32, 0, 320)])
def test_gemm_bf16x6_is_fp32_accurate(m, k1, k2, n):
    g = torch.Generator().manual_seed(m + n)
    a1 = torch.randn(m, k1, generator=g) * torch.rand(m, 1, generator=g) * 30
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    bias = torch.randn(n, generator=g)
    A = a1 if a2 is None else torch.cat([a1, a2], 1)
    ref = torch.nn.functional.leaky_relu(A.double() @ w.double().T + bias.double(), 0.1)
    got = ops.gemm_split(a1.to(DEV), w.to(DEV), bias.to(DEV), a2=None if a2 is None else a2.to(DEV),
                         act=ops.ACT_LEAKY, slope=0.1)
    f32 = ops.gemm(a1.to(DEV), w.t().contiguous().to(DEV), bias.to(DEV),
                   a2=None if a2 is None else a2.to(DEV), act=ops.ACT_LEAKY, slope=0.1)
    e_split = (got.cpu().double() - ref).abs().max().item()
    e_f32 = (f32.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"[accuracy] m={m} k={k1}+{k2} n={n}: bf16x6 err {e_split:.2e}  f32-MFMA err {e_f32:.2e}  scale {scale:.1f}")
    assert e_split <= max(4 * e_f32, 2e-6 * scale)


# ---- fp16 hi/lo three-product GEMM (csrc/gemm_f16x3.hip) ------------------------------------------------------------
def _f16x3_case(m, k1, k2, n, row_mag, seed):
    g = torch.Generator().manual_seed(seed)
    a1 = torch.randn(m, k1, generator=g) * row_mag(m, g)
    a2 = torch.randn(m, k2, generator=g) * row_mag(m, g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    bias = torch.randn(n, generator=g)
    return a1, a2, w, bias


def _f16x3_errors(a1, a2, w, bias):
    A = a1 if a2 is None else torch.cat([a1, a2], 1)
    lin = A.double() @ w.double().T
    ref = torch.nn.functional.leaky_relu(lin + bias.double(), 0.1)
    mag = A.double().abs() @ w.double().abs().T + bias.double().abs()      # the scale rounding errors live on
    dev = lambda t: None if t is None else t.to(DEV)                     # noqa: E731
    got = ops.gemm_f16x3(dev(a1), dev(w), dev(bias), a2=dev(a2), act=ops.ACT_LEAKY, slope=0.1)
    f32 = ops.gemm(dev(a1), dev(w.t().contiguous()), dev(bias), a2=dev(a2), act=ops.ACT_LEAKY, slope=0.1)
    x6 = ops.gemm_split(dev(a1), dev(w), dev(bias), a2=dev(a2), act=ops.ACT_LEAKY, slope=0.1)
    rel = lambda t: ((t.cpu().double() - ref).abs() / mag.clamp_min(1e-300)).max().item()   # noqa: E731
    return rel(got), rel(f32), rel(x6)


@pytest.mark.parametrize("m,k1,k2,n", [(300, 576, 0, 576), (1000, 512, 0, 576), (257, 64, 64, 64), (129, 64, 0, 256),
                                       (64, 32, 32, 128), (5, 96, 0, 192)])
def test_gemm_f16x3_is_fp32_accurate(m, k1, k2, n):
    """the three-product fp16 form must be as accurate as the f32 MFMA (error measured relative to
    sum |a||w|, the scale fp32 rounding errors live on), on rows whose magnitudes span 30x."""
    a1, a2, w, bias = _f16x3_case(m, k1, k2, n, lambda m_, g: torch.rand(m_, 1, generator=g) * 30, m + n)
    e16, e32, e6 = _f16x3_errors(a1, a2, w, bias)
    print(f"[accuracy] m={m} k={k1}+{k2} n={n}: f16x3 {e16:.2e}  f32-MFMA {e32:.2e}  bf16x6 {e6:.2e} (max err / sum|a||w|)")
    assert e16 <= 2.0 * e32


@pytest.mark.parametrize("kind", ["1e5", "1e-4", "rows_2^+-20", "inrow_2^-20", "zero_rows", "fp16_overflow_edge"])
def test_gemm_f16x3_range(kind):
    """fp16 has a 5-bit exponent: the per-row power-of-two scale must carry activations of any magnitude (the
    range guard VERDICT r3 asks for is a scale here, not a fallback), tiny rows must not fall into the subnormal
    hole, and an all-zero row must stay zero."""
    m, k, n = 384, 128, 64
    g = torch.Generator().manual_seed(7)
    mags = {
        "1e5": lambda m_, g_: torch.full((m_, 1), 1e5),
        "1e-4": lambda m_, g_: torch.full((m_, 1), 1e-4),
        "rows_2^+-20": lambda m_, g_: 2.0 ** torch.randint(-20, 21, (m_, 1), generator=g_).float(),
        "inrow_2^-20": lambda m_, g_: torch.ones(m_, 1),
        "zero_rows": lambda m_, g_: (torch.rand(m_, 1, generator=g_) > 0.5).float(),
        "fp16_overflow_edge": lambda m_, g_: torch.full((m_, 1), 65504.0 * 3),
    }
    a1, a2, w, bias = _f16x3_case(m, k, 0, n, mags[kind], 11)
    if kind == "inrow_2^-20":          # half of every row is 2^-20 of the other half
        a1[:, ::2] *= 2.0 ** -20
    e16, e32, e6 = _f16x3_errors(a1, a2, w, bias * 0)
    print(f"[range] {kind}: f16x3 {e16:.2e}  f32-MFMA {e32:.2e}  bf16x6 {e6:.2e} (max err / sum|a||w|)")
    assert e16 <= 2.0 * e32 + 1e-30
    if kind == "zero_rows":
        got = ops.gemm_f16x3(a1.to(DEV), w.to(DEV))
        zero = (a1.abs().sum(1) == 0)
        assert zero.any() and torch.all(got.cpu()[zero] == 0)


def test_gemm_f16x3_weight_scale_and_planes():
    """The planes are (hi, lo) of scale * w with ONE power of two per matrix; hi + lo reproduces the scaled weight
    to 2^-22 and tiny weights next to large ones degrade gradually (subnormal lo), never to garbage."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 96, generator=g) * 0.05
    w[0, 0] = 3.0            # one large element sets the scale
    w[1, :8] = 1e-7          # far below it
    P = ops.split_f16_planes(w.to(DEV))
    scale = P.scale.cpu()
    assert scale[0] * scale[1] == 1.0 and 2 ** 14 <= scale[0] * 3.0 < 2 ** 15
    assert float(np.log2(scale[0].item())).is_integer()
    planes = P.planes.cpu().view(torch.float16).double()
    rec = (planes[0] + planes[1]) / scale[0].double()
    err = (rec - w.double()).abs()
    assert bool((err <= torch.clamp(2.0 ** -22 * w.double().abs(), min=2.0 ** -25 / scale[0].item())).all())


@pytest.mark.parametrize("kind", ["O(1)", "rows_2^+-16", "1e5", "1e-4", "hub"])
def test_fused_shmp_layer_f16x3_is_fp32_accurate_over_the_fp32_range(kind):
    """The fp16 three-product layer against fp64, next to the bf16x6 and f32-MFMA forms of the same launch: error
    relative to sum |a||w| per output (the scale fp32 rounding errors live on).  Rows of very different magnitude
    share 16-row tiles (per-row power-of-two scales), all-zero slots and rows keep their scale, hub rows sum hundreds
    of sources."""
    S, sm, st, num_rows, row0 = 4, 2, 2, 5000, 0
    g = torch.Generator().manual_seed(len(kind))
    n_all = num_rows + 16
    x = torch.randn(n_all, 64, generator=g).abs()
    if kind == "rows_2^+-16":
        x = x * 2.0 ** torch.randint(-16, 17, (n_all, 1), generator=g).float()
    elif kind == "1e5":
        x = x * 1e5
    elif kind == "1e-4":
        x = x * 1e-4
    ptr, col, cnt = _random_vcsr(n_all, S, 300 if kind == "hub" else 5, n_all, g)
    wt = torch.randn((sm + 1) * 64, 64, generator=g) / 12
    bias = torch.zeros(64)
    wtab = torch.randn(64, 64 * st, generator=g) / 8
    agg = torch.zeros(n_all * S, 64, dtype=torch.double)
    agg.index_add_(0, torch.repeat_interleave(torch.arange(n_all * S), cnt), x.double()[col.long()])
    aggv = agg.view(n_all, S * 64)
    A = torch.cat([aggv[:, :sm * 64], x.double()], 1)
    ytab_cpu = x @ wtab
    ref = A @ wt.double()
    mag = A.abs() @ wt.double().abs()
    for s_ in range(st):
        t_agg = torch.zeros(n_all, 64, dtype=torch.double)
        rows = torch.repeat_interleave(torch.arange(n_all * S), cnt)
        sel = (rows % S) == (sm + s_)
        t_agg.index_add_(0, rows[sel] // S, ytab_cpu.double()[col.long()[sel], s_ * 64:(s_ + 1) * 64])
        ref = ref + t_agg
        mag = mag + t_agg.abs()
    ref = torch.relu(ref)[:num_rows]
    mag = mag[:num_rows].clamp_min(1e-300)
    errs = {}
    for name, w_dev in (("f16x3", ops.split_f16_planes(wt.t().contiguous().to(DEV))),
                        ("bf16x6", ops.split_bf16_planes(wt.t().contiguous().to(DEV))), ("f32", wt.to(DEV))):
        out = torch.empty((n_all, 64), device=DEV)
        ops.shmp_layer(x.to(DEV), ptr.to(DEV), col.to(DEV), row0, num_rows, S, sm, w_dev, bias.to(DEV), out,
                       ytab=ytab_cpu.to(DEV), ytab_row0=0)
        errs[name] = ((out[:num_rows].cpu().double() - ref).abs() / mag).max().item()
    print(f"[accuracy] shmp layer {kind}: max err / sum|a||w|  f16x3 {errs['f16x3']:.2e}  bf16x6 {errs['bf16x6']:.2e}  "
          f"f32-MFMA {errs['f32']:.2e}")
    assert errs["f16x3"] <= 2.0 * max(errs["f32"], errs["bf16x6"])


# ---- the optimizer launch (csrc/adam.hip, desco_amd/optim.py) -----------------------------------------------------------
def _tail_case(m, seed, row_mag=None, w_gain=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, 64, generator=g)
    if row_mag is not None:
        x = x * row_mag(m, g)
    ws = [torch.randn(o, i, generator=g) * (w_gain / i ** 0.5) for o, i in ((64, 64), (256, 64), (64, 256))]
    bs = [torch.randn(o, generator=g) * 0.1 for o in (64, 256, 64)]
    return x, ws, bs


def _tail_fp64(x, ws, bs):
    """(result, per-output error scale): the chain in fp64, and the magnitude sum an fp32 chain's rounding errors live
    on (|W3| (|W2| (|W1||x| + |b1|) + |b2|) + |b3|)"""
    h = torch.relu(x.double() @ ws[0].double().t() + bs[0].double())
    h = torch.relu(h @ ws[1].double().t() + bs[1].double())
    ref = h @ ws[2].double().t() + bs[2].double()
    mag = x.double().abs() @ ws[0].double().abs().t() + bs[0].double().abs()
    mag = mag @ ws[1].double().abs().t() + bs[1].double().abs()
    mag = mag @ ws[2].double().abs().t() + bs[2].double().abs()
    return ref, mag


def _tail_both(x, ws, bs):
    """the one-launch tail and the three launches it replaces (linear64, linear64, gemm_split: bf16x6 arithmetic)"""
    xd = x.to(DEV)
    planes = [ops.split_f16_planes(w.to(DEV)) for w in ws]
    bd = [b.to(DEV) for b in bs]
    fused = ops.post_mp_tail(xd, planes[0], bd[0], planes[1], bd[1], planes[2], bd[2])
    h = ops.linear64(xd, ops.linear64_planes(ws[0].to(DEV)), bd[0], act=ops.ACT_RELU)
    h = ops.linear64(h, ops.linear64_planes(ws[1].to(DEV)), bd[1], act=ops.ACT_RELU)
    chain = ops.gemm_split(h, ops.split_bf16_planes(ws[2].to(DEV)), bd[2])
    torch.cuda.synchronize()
    return fused.cpu().double(), chain.cpu().double()


@pytest.mark.parametrize("m", [1, 31, 32, 33, 257, 5000, 70001])
def test_post_mp_tail_in_one_launch_is_fp32_accurate(m):
    """post_mp.3 -> .5 -> .7 in one launch (reference gnn_model.py:44-53) against fp64, next to the three launches it
    replaces: the f16x3 chain must be as accurate as the bf16x6 one (error relative to the magnitude sum of the chain),
    for every ragged tail of the 32-row wave tiles and rows whose magnitudes span 30x."""
    x, ws, bs = _tail_case(m, 100 + m, lambda m_, g: torch.rand(m_, 1, generator=g) * 30 + 0.01)
    ref, mag = _tail_fp64(x, ws, bs)
    fused, chain = _tail_both(x, ws, bs)
    ef = ((fused - ref).abs() / mag).max().item()
    ec = ((chain - ref).abs() / mag).max().item()
    print(f"[tail] m={m}: one launch {ef:.2e}  three launches {ec:.2e} (max err / magnitude sum)")
    assert ef <= max(2.0 * ec, 2e-7)
    assert (fused - chain).abs().max().item() <= 5e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kind", ["1e5", "1e-5", "rows_2^+-20", "zero_rows", "late_block_dominates", "big_weights"])
def test_post_mp_tail_range(kind):
    """The activations' per-row power-of-two scales (a running one over the 256 hidden features) must carry any
    magnitude: huge and tiny rows, all-zero rows, a hidden row whose largest feature sits in the LAST 32-feature block
    (every earlier block's accumulators are rescaled), weights far from O(1)."""
    m = 1000
    mags = {
        "1e5": lambda m_, g_: torch.full((m_, 1), 1e5),
        "1e-5": lambda m_, g_: torch.full((m_, 1), 1e-5),
        "rows_2^+-20": lambda m_, g_: 2.0 ** torch.randint(-20, 21, (m_, 1), generator=g_).float(),
        "zero_rows": lambda m_, g_: (torch.rand(m_, 1, generator=g_) > 0.5).float(),
    }
    x, ws, bs = _tail_case(m, 7 + len(kind), mags.get(kind), w_gain=300.0 if kind == "big_weights" else 1.0)
    if kind == "late_block_dominates":
        ws[1][:224] *= 2.0 ** -12          # hidden features 224..255 are 4096x the others
        bs[1][:224] *= 2.0 ** -12
    if kind in ("1e5", "1e-5", "rows_2^+-20", "zero_rows"):
        bs = [b * 0 for b in bs]           # (a bias would swamp the tiny rows and hide their error)
    ref, mag = _tail_fp64(x, ws, bs)
    fused, chain = _tail_both(x, ws, bs)
    live = mag > 0
    ef = ((fused - ref).abs()[live] / mag[live]).max().item()
    ec = ((chain - ref).abs()[live] / mag[live]).max().item()
    print(f"[tail range] {kind}: one launch {ef:.2e}  three launches {ec:.2e}")
    assert ef <= max(2.0 * ec, 2e-7)
    if kind == "zero_rows":
        zero = x.abs().sum(1) == 0
        assert zero.any() and torch.all(fused[zero] == 0)


def test_post_mp_tail_strided_in_place_and_bad_arguments():
    x, ws, bs = _tail_case(300, 5)
    planes = [ops.split_f16_planes(w.to(DEV)) for w in ws]
    bd = [b.to(DEV) for b in bs]
    want = ops.post_mp_tail(x.to(DEV), planes[0], bd[0], planes[1], bd[1], planes[2], bd[2])
    wide = torch.zeros(300, 192, device=DEV)
    wide[:, 64:128] = x.to(DEV)
    got = ops.post_mp_tail(wide[:, 64:128], planes[0], bd[0], planes[1], bd[1], planes[2], bd[2], out=wide[:, 128:192])
    assert torch.equal(got, want) and torch.all(wide[:, :64] == 0)
    inpl = x.to(DEV).clone()
    ops.post_mp_tail(inpl, planes[0], bd[0], planes[1], bd[1], planes[2], bd[2], out=inpl)
    assert torch.equal(inpl, want)
    nob = ops.post_mp_tail(x.to(DEV), planes[0], None, planes[1], None, planes[2], None)
    ref = torch.relu(torch.relu(x.double() @ ws[0].double().t()) @ ws[1].double().t()) @ ws[2].double().t()
    assert (nob.cpu().double() - ref).abs().max().item() < 1e-5
    assert ops.post_mp_tail(x[:0].to(DEV), planes[0], bd[0], planes[1], bd[1], planes[2], bd[2]).shape == (0, 64)
    with pytest.raises(Exception):
        ops.post_mp_tail(wide[:, 1:65], planes[0], bd[0], planes[1], bd[1], planes[2], bd[2])     # misaligned rows


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_matches_torch_adam(wd):
    """desco_amd.optim.Adam against torch.optim.Adam (the reference's optimizer, lightning_model.py:160-173) on a
    ragged parameter list: unaligned sizes, one tensor > one workgroup's 1024 elements, one parameter that never gets a
    gradient (skipped, not aged), one that gets one only on some steps, > 128 tensors (two launches)."""
    from desco_amd.optim import Adam
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 64), (7,), (1,), (300, 17), (64,), (5, 3)] + [(3 + i % 11, 1 + i % 5) for i in range(140)]
    ref = [torch.randn(*s, generator=g).to(DEV).requires_grad_() for s in shapes]
    own = [p.detach().clone().requires_grad_() for p in ref]
    o_ref = torch.optim.Adam(ref, lr=3e-3, weight_decay=wd)
    o_own = Adam(own, lr=3e-3, weight_decay=wd)
    for step in range(7):
        if step == 4:          # a scheduler halves the rate (ReduceLROnPlateau writes param_groups[i]["lr"])
            for o in (o_ref, o_own):
                o.param_groups[0]["lr"] *= 0.5
        for i, (a, b) in enumerate(zip(ref, own)):
            if i == 1 or (i == 4 and step % 2):
                a.grad = b.grad = None
                continue
            gr = (torch.randn(a.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        o_ref.step()
        o_own.step()
    worst = 0.0
    for a, b in zip(ref, own):
        worst = max(worst, float(((a - b).abs() / (1e-3 + a.abs())).max()))
    print(f"[parity] Adam, 7 steps: max |p_own - p_torch| / (1e-3 + |p|) = {worst:.2e}")
    assert worst < 2e-6
    assert torch.equal(ref[1], own[1]) and own[1]._version == ref[1]._version      # never updated
    assert own[0]._version >= 7          # updated through raw pointers, but torch is told (caches key on _version)
    sd = o_own.state_dict()
    assert float(sd["state"][4]["step"]) == 4 and float(sd["state"][0]["step"]) == 7 and float(sd["state"][1]["step"]) == 0
    tsd = o_ref.state_dict()["state"]
    torch.testing.assert_close(sd["state"][3]["exp_avg_sq"], tsd[3]["exp_avg_sq"], rtol=1e-5, atol=1e-30)
    # state round trip
    o2 = Adam([p.detach().clone().requires_grad_() for p in own], lr=1.0)
    o2.load_state_dict(sd)
    sd2 = o2.state_dict()
    assert all(torch.equal(sd2["state"][k]["exp_avg"], sd["state"][k]["exp_avg"]) for k in sd["state"])
    assert o2.param_groups[0]["lr"] == o_own.param_groups[0]["lr"]


# ---- the single-workgroup SHMP trunk of small batches (csrc/shmp_small.hip) -----------------------------------------------
@pytest.mark.parametrize("layers", [1, 8])
def test_small_trunk_matches_the_layerwise_trunk(layers):
    """autograd.ShmpTrunkSmall (one launch per direction) against autograd.ShmpTrunk (the general per-layer launches)
    on the 29 standard query graphs: pooled output and the gradients of x0, the stacked weights and biases."""
    from helpers import standard_queries
    from desco_amd import autograd as AG
    from desco_amd.batch import QueryBatch
    qb = QueryBatch(standard_queries()[1], DEV)
    n, B = qb.num_rows, qb.num_graphs
    assert 0 < n <= ops.shmp_trunk_small_max_rows()
    g = torch.Generator().manual_seed(layers)
    x0 = torch.randn(n, 64, generator=g)
    wt = torch.randn(layers, 192, 64, generator=g) / 10
    bias = torch.randn(layers, 64, generator=g) / 4
    seed = torch.randn(B, 64 * (layers + 1), generator=g)
    res = []
    for small in (True, False):
        a, w, b = (t.clone().to(DEV).requires_grad_() for t in (x0, wt, bias))
        if small:
            qb.__dict__["_small_per_graph"] = False          # (the per-graph form has its own test below)
            pooled = AG.ShmpTrunkSmall.apply(a, qb, None, w, b)
        else:
            pooled = AG.ShmpTrunk.apply(a, qb, [("union_node", 0, n, 2)], False, None, w, b)
        (pooled * seed.to(DEV)).sum().backward()
        res.append((pooled.detach(), a.grad, w.grad, b.grad))
    worst = 0.0
    for name, u, v in zip(("pooled", "dx0", "dWt", "dbias"), res[0], res[1]):
        err = float((u - v).abs().max() / (1e-6 + v.abs().max()))
        worst = max(worst, err)
        assert err < 2e-6, (name, err)
    print(f"[parity] single-workgroup trunk vs layer-wise trunk, L={layers}: worst max|d| / max|ref| = {worst:.2e}")


@pytest.mark.parametrize("layers", [1, 8])
def test_per_graph_trunk_matches_the_one_workgroup_trunk(layers):
    """autograd.ShmpTrunkSmall's two forms: one workgroup per graph (round 6) against the one-workgroup kernels, on the
    29 standard query graphs and on a batch with 1-row, 2-row and 8-row graphs (the per-graph limit)."""
    from helpers import standard_queries
    from desco_amd import autograd as AG
    from desco_amd.batch import QueryBatch
    ring8 = (8, [(i, (i + 1) % 8) for i in range(8)] + [(0, 4), (1, 5), (0, 2)])
    extra = [(1, []), (2, [(0, 1)]), ring8, (3, [(0, 1), (1, 2), (0, 2)]), ring8]
    for name, graphs in (("standard queries", standard_queries()[1]), ("1..8-row graphs", extra)):
        qb = QueryBatch(graphs, DEV)
        n, B = qb.num_rows, qb.num_graphs
        assert AG.ShmpTrunkSmall.per_graph(qb)
        g = torch.Generator().manual_seed(layers)
        x0 = torch.randn(n, 64, generator=g)
        wt = torch.randn(layers, 192, 64, generator=g) / 10
        bias = torch.randn(layers, 64, generator=g) / 4
        seed = torch.randn(B, 64 * (layers + 1), generator=g)
        res = []
        for per_graph in (True, False):
            qb.__dict__["_small_per_graph"] = per_graph
            a, w, b = (t.clone().to(DEV).requires_grad_() for t in (x0, wt, bias))
            pooled = AG.ShmpTrunkSmall.apply(a, qb, None, w, b)
            (pooled * seed.to(DEV)).sum().backward()
            res.append((pooled.detach(), a.grad, w.grad, b.grad))
        worst = 0.0
        for what, u, v in zip(("pooled", "dx0", "dWt", "dbias"), res[0], res[1]):
            err = float((u - v).abs().max() / (1e-6 + v.abs().max()))
            worst = max(worst, err)
            assert err < 2e-6, (name, what, err)
        print(f"[parity] per-graph trunk vs one-workgroup trunk, {name}, L={layers}: worst max|d| / max|ref| = {worst:.2e}")
    # a graph above the limit keeps the one-workgroup form
    big = QueryBatch([(9, [(i, i + 1) for i in range(8)])], DEV)
    assert not AG.ShmpTrunkSmall.per_graph(big)


def test_small_trunk_refuses_large_batches():
    L = ops._lib.lib()
    z = torch.zeros(8, device=DEV)
    zi = torch.zeros(8, device=DEV, dtype=torch.int32)
    rc = L.desco_shmp_trunk_small_fwd_f32(z.data_ptr(), zi.data_ptr(), zi.data_ptr(), 145, 1, z.data_ptr(), z.data_ptr(),
                                          zi.data_ptr(), 1, z.data_ptr(), z.data_ptr(), 128, None)
    assert rc == -1 and b"desco_shmp_trunk_small_fwd_f32" in L.desco_last_error()


# ---- round-4 training entry points, each against plain torch fp64 ---------------------------------------------------------
def test_gemm_multi_with_gate_and_accumulate():
    """desco_gemm_f32_multi: independent products in one launch; gate = multiply by act'(saved output) (the act_grad pass
    fused into the epilogue), accum = add into the existing output; an empty problem is skipped."""
    g = torch.Generator().manual_seed(7)
    a0, w0, b0 = torch.randn(1000, 192, generator=g), torch.randn(192, 64, generator=g) / 8, torch.randn(64, generator=g)
    a1, w1 = torch.randn(77, 64, generator=g), torch.randn(64, 256, generator=g) / 8
    a2, w2 = torch.randn(300, 256, generator=g), torch.randn(256, 64, generator=g) / 8
    gate_r = torch.randn(77, 256, generator=g)                   # "saved relu output": its sign is what matters
    gate_l = torch.randn(300, 64, generator=g)
    prev = torch.randn(300, 64, generator=g)
    o0 = torch.empty(1000, 64, device=DEV)
    o1 = torch.empty(77, 256, device=DEV)
    o2 = prev.clone().to(DEV)
    oe = torch.empty(0, 64, device=DEV)
    ops.gemm_multi([dict(a1=a0.to(DEV)[:, :128], a2=a0.to(DEV)[:, 128:], wt=w0.to(DEV), bias=b0.to(DEV), act=ops.ACT_RELU, out=o0),
                    dict(a1=a1.to(DEV), wt=w1.to(DEV), out=o1, gate=gate_r.to(DEV), gate_act=ops.ACT_RELU),
                    dict(a1=a2.to(DEV), wt=w2.to(DEV), out=o2, gate=gate_l.to(DEV), gate_act=ops.ACT_LEAKY, gate_slope=0.1,
                         accum=True),
                    dict(a1=torch.empty(0, 64, device=DEV), wt=w1.to(DEV)[:, :64].contiguous(), out=oe)])
    _close(o0, torch.relu(a0.double() @ w0.double() + b0.double()), atol=1e-4)
    _close(o1, (a1.double() @ w1.double()) * (gate_r.double() > 0), atol=1e-4)
    _close(o2, prev.double() + (a2.double() @ w2.double()) * torch.where(gate_l.double() > 0, 1.0, 0.1), atol=1e-4)


def test_gemm_split_desc_gate_dropout_and_row_class_tail():
    """desco_gemm_bf16x6_desc_f32 (the gossip training step's products): the transposing plane split, the backward
    epilogue (dropout factor regenerated from the key, activation-derivative gate on the saved output) against the
    exact-fp32 gemm_multi with the same descriptor, and the per-row-class scalar tail against affine_rows on the product."""
    g = torch.Generator().manual_seed(21)
    m, k1, k2, n, Q = 3001, 64, 64, 64, 29
    a1, a2 = torch.randn(m, k1, generator=g).to(DEV), torch.randn(m, k2, generator=g).to(DEV)
    wt = (torch.randn(k1 + k2, n, generator=g) / 11).to(DEV)                  # [in, out]
    assert torch.equal(ops.split_bf16_planes_t(wt), ops.split_bf16_planes(wt.t().contiguous()))
    assert torch.equal(ops.split_bf16_planes_t(wt[:64]), ops.split_bf16_planes(wt[:64].t().contiguous()))
    planes = ops.split_bf16_planes_t(wt)
    gate = torch.randn(m, n, generator=g).to(DEV)
    key = ops.rng_next(DEV)
    drop = ops.DropSite(key, 5, 0.3)
    for kw in (dict(), dict(gate=gate, gate_act=ops.ACT_RELU), dict(gate=gate, gate_act=ops.ACT_LEAKY, gate_slope=0.1, drop=drop),
               dict(act=ops.ACT_RELU, drop=drop)):
        want, got = torch.empty(m, n, device=DEV), torch.empty(m, n, device=DEV)
        ops.gemm_multi([dict(a1=a1, a2=a2, wt=wt, out=want, **kw)])
        ops.gemm_split_desc(dict(a1=a1, a2=a2, out=got, **kw), planes)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-6 * scale, kw.keys()
        assert torch.equal(got == 0, want == 0) or "drop" not in kw           # the same elements are dropped / gated out
    # scalar tail per row class == affine_rows(product)
    c = torch.randn(m, 3, generator=g).to(DEV)
    v = torch.randn(Q, 3, n, generator=g).to(DEV)
    base = ops.gemm_split(a1, planes, a2=a2)
    for act, slope, d in ((ops.ACT_RELU, 0.0, None), (ops.ACT_LEAKY, 0.1, drop)):
        want = ops.affine_rows(base, c, v, act, slope, d)
        got = torch.empty(m, n, device=DEV)
        ops.gemm_split_desc(dict(a1=a1, a2=a2, out=got, act=act, slope=slope, s=c, ws=v, drop=d), planes)
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    with pytest.raises(Exception):
        ops.gemm_split_desc(dict(a1=a1, a2=a2, out=got, accum=True), planes)


def test_linear_bwd_w_multi_matches_the_single_problem_form():
    g = torch.Generator().manual_seed(8)
    probs, refs = [], []
    for m, k1, k2 in ((5000, 128, 64), (0, 64, 64), (40, 64, 0), (700, 256, 64)):
        a1 = torch.randn(m, k1, generator=g)
        a2 = torch.randn(m, k2, generator=g) if k2 else None
        dz = torch.randn(m, 64, generator=g)
        A = a1 if a2 is None else torch.cat([a1, a2], 1)
        refs.append((A.double().T @ dz.double(), dz.double().sum(0)))
        probs.append(dict(a1=a1.to(DEV) if m else torch.empty(0, k1, device=DEV), a2=None if a2 is None else a2.to(DEV),
                          dz=dz.to(DEV), dwt=torch.empty(k1 + k2, 64, device=DEV), dbias=torch.empty(64, device=DEV)))
    ops.linear_bwd_w_multi(probs)
    for pr, (rw, rb) in zip(probs, refs):
        _close(pr["dwt"], rw, rtol=1e-4, atol=2e-3)
        _close(pr["dbias"], rb, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("rows", [1, 1000, 300001])
def test_rowdot_bwd_and_smallk_bwd(rows):
    g = torch.Generator().manual_seed(rows)
    y = torch.relu(torch.randn(rows, 256, generator=g))
    w = torch.randn(256, generator=g)
    d = torch.randn(rows, generator=g)
    dz, dwb = ops.rowdot_bwd(y.to(DEV), w.to(DEV), d.to(DEV))
    _close(dz, d.double()[:, None] * w.double()[None, :] * (y.double() > 0), atol=1e-5)
    _close(dwb[:256], (y.double() * d.double()[:, None]).sum(0), rtol=1e-4, atol=1e-2)
    _close(dwb[256:], d.double().sum().reshape(1), rtol=1e-4, atol=1e-2)
    feat = torch.randn(rows, 2, generator=g)
    dout = torch.randn(rows + 3, 64, generator=g)[3:]                     # a row-offset view
    dwt, db = ops.linear_smallk_bwd(feat.to(DEV), dout.to(DEV))
    _close(dwt, feat.double().T @ dout.double(), rtol=1e-4, atol=1e-2)
    _close(db, dout.double().sum(0), rtol=1e-4, atol=1e-2)

