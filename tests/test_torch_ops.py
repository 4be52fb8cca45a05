"""torch.ops.desco.* (desco_amd/torch_ops.py): registration and schemas (CPU), the host partition op
(CPU), and on the GPU that every registered device op IS the C-ABI path."""
import numpy as np
import pytest
import torch

import desco_amd.torch_ops as TO
from desco_amd.graphs import GraphSet
from helpers import golden_graphs


def test_namespace_is_registered_with_the_survey_minimum_set():
    for name in ("build_canonical_partition", "shmp_aggregate", "shmp_aggregate_backward", "shmp_layer_fused",
                 "segment_sum", "count_head", "gossip_aggregate", "gossip_aggregate_backward"):
        assert hasattr(torch.ops.desco, name), name
        assert name in TO.SCHEMAS


def test_host_partition_op_equals_python_api():
    from desco_amd.partition import build_partition
    gs = GraphSet.from_edge_lists(golden_graphs(max_n=41))
    outs = torch.ops.desco.build_canonical_partition(torch.from_numpy(gs.graph_ptr), torch.from_numpy(gs.rowptr),
                                                     torch.from_numpy(gs.col), 4)
    p = build_partition(gs, 4)
    for got, ref in zip(outs, (p.neigh_index, p.indicator, p.count_ptr, p.count_orig, p.vrowptr, p.vcol)):
        assert np.array_equal(got.numpy(), ref)


def test_device_ops_have_no_cpu_kernel():
    x = torch.zeros(4, 64)
    ptr = torch.zeros(5, dtype=torch.int32)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.desco.segment_sum(x, ptr, 4, None)


@pytest.mark.gpu
def test_registered_device_ops_are_the_c_abi_path():
    from desco_amd import ops
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.partition import build_partition
    dev = "cuda"
    part = build_partition(GraphSet.from_edge_lists(golden_graphs(max_n=41)), 4)
    b = NeighborhoodBatch(part, dev)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b.num_rows, 64, generator=g).to(dev)
    agg = torch.ops.desco.shmp_aggregate(x, b.vrowptr, b.vcol, b.num_rows, 4)
    assert torch.equal(agg, ops.csr_gather_sum(x, b.vrowptr, b.vcol, b.num_rows, 4))
    t_rowptr, t_col = torch.ops.desco.shmp_transpose_index(b.vrowptr, b.vcol, b.num_rows, 4, b.num_count)
    dagg = torch.randn(b.num_rows, 256, generator=g).to(dev)
    dx = torch.ops.desco.shmp_aggregate_backward(dagg, t_rowptr, t_col, b.num_rows)
    # <A x, g> == <x, A^T g>
    lhs = float((agg.double() * dagg.double()).sum())
    rhs = float((x.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * float((agg.double() * dagg.double()).abs().sum()) / agg.numel() ** 0.5 + 1e-5 * abs(lhs)
    pooled = torch.ops.desco.segment_sum(x[:b.num_count], b.count_ptr, b.num_graphs, None)
    ref = torch.zeros(b.num_graphs, 64, dtype=torch.double)
    ref.index_add_(0, torch.from_numpy(np.repeat(np.arange(b.num_graphs), np.diff(part.count_ptr))),
                   x[:b.num_count].double().cpu())
    torch.testing.assert_close(pooled.double().cpu(), ref, rtol=1e-5, atol=1e-5)
    # fused layer op == aggregate + GEMM
    wt = torch.randn(3 * 64, 64, generator=g).to(dev) / 12
    bias = torch.randn(64, generator=g).to(dev)
    planes = torch.ops.desco.split_bf16_planes(wt.t().contiguous())
    out = torch.empty(b.num_count, 64, device=dev)
    torch.ops.desco.shmp_layer_fused(x, b.vrowptr, b.vcol, 0, b.num_count, 4, 2, planes, bias, None, 0, out)
    ref = torch.relu(torch.cat([agg[:b.num_count, :128], x[:b.num_count]], 1).double() @ wt.double() + bias.double())
    torch.testing.assert_close(out.double(), ref, rtol=2e-5, atol=2e-5)
    with pytest.raises(RuntimeError):
        torch.ops.desco.shmp_aggregate(x, b.vrowptr.cpu(), b.vcol, b.num_rows, 4)     # mixed devices
    # ---- the three-product fp16 forms (what the product path runs): the registered ops ARE the ops.* calls ----------
    fp = ops.split_f16_planes(wt.t().contiguous())
    pl, sc = torch.ops.desco.split_f16_planes(wt.t().contiguous())
    assert torch.equal(pl, fp.planes) and torch.equal(sc, fp.scale)
    out16 = torch.empty(b.num_count, 64, device=dev)
    torch.ops.desco.shmp_layer_fused_f16x3(x, b.vrowptr, b.vcol, 0, b.num_count, 4, 2, pl, sc, bias, None, 0, out16)
    torch.testing.assert_close(out16.double(), ref, rtol=2e-5, atol=2e-5)
    a = torch.randn(300, 192, generator=g).to(dev)
    w = (torch.randn(64, 192, generator=g) / 14).to(dev)
    wp, ws = torch.ops.desco.split_f16_planes(w)
    got = torch.ops.desco.gemm_f16x3(a, wp, ws, bias, None, ops.ACT_RELU, 0.0)
    assert torch.equal(got, ops.gemm_f16x3(a, ops.F16Planes(wp, ws), bias, act=ops.ACT_RELU))
    torch.testing.assert_close(got.double(), torch.relu(a.double() @ w.double().t() + bias.double()), rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_registered_gossip_f16x3_is_the_product_kernel():
    """torch.ops.desco.gossip_fused_f16x3 / gossip_f16_stream launch desco_gossip_fused_f16x3_f32 (the kernel the pipeline
    runs): same bits as the ops.* call, on random operands."""
    from desco_amd import ops
    from desco_amd.batch import GossipBatch
    dev = "cuda"
    gs = GraphSet.from_edge_lists(golden_graphs(max_n=41))
    Q = 7
    torch.manual_seed(2)
    x = torch.rand(gs.num_nodes, Q) * 20
    batch = GossipBatch(gs, dev, x=x)
    g0, g1 = torch.rand(Q, device=dev) * 0.8 + 0.1, torch.rand(Q, device=dev) * 0.8 + 0.1
    scal = ops.gossip_scalars(batch.x, batch.rowptr, batch.col, g0, g1)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.2).contiguous()     # noqa: E731
    W = [r(64, 128), r(64, 128), r(64, 64), r(256, 64)]
    split = [torch.ops.desco.split_f16_planes(w) for w in W]
    wstream, winv = torch.ops.desco.gossip_f16_stream([p for p, _ in split], [s for _, s in split])
    v = dict(g1=g1, p=r(Q, 64), z=r(Q, 64), zp=r(Q, 64), r=r(64), t=r(64), u=r(64), tp=r(64), d1=r(64), wstream=wstream,
             winv=winv, b3=r(64), b5=r(256), w7=r(256), b7=0.25)
    queue = torch.zeros(2, dtype=torch.int64, device=dev)
    got = torch.ops.desco.gossip_fused_f16x3(scal, batch.rowptr, batch.col, gs.num_nodes, Q,
                                             [v[n] for n in TO.GOSSIP_F16_OPERANDS], 0.25, queue, None)
    ref = ops.gossip_fused_f16(scal, batch.rowptr, batch.col, gs.num_nodes, Q, v, queue)
    assert torch.equal(got, ref) and torch.isfinite(got).all() and int(queue.abs().sum()) == 0
