"""The neighborhood-resident multi-layer SHMP kernel (csrc/shmp_resident.hip) against the CPU oracle in the
reference's form (gnn_model.py:58-109, 230-277, 372-404) and against the layer-by-layer kernels, incl.
neighborhoods above the pack limits (sub-batch through the layer-by-layer path), hub rows (cooperative
gather) and bit-exact independence of a neighborhood's result from its placement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import desco_amd.gnn_model as GM  # noqa: E402
from desco_amd import ops, synthetic  # noqa: E402
from desco_amd.batch import NeighborhoodBatch  # noqa: E402
from desco_amd.graphs import GraphSet  # noqa: E402
from desco_amd.partition import build_partition  # noqa: E402
from oracle import model as OM  # noqa: E402
from oracle import partition as OP  # noqa: E402

from helpers import cpu_sd, golden_graphs, make_models, report, standard_queries  # noqa: E402

DEV = "cuda"


def _model(gains=(1.3, 1.4)):
    nm, _ = make_models(seed=0, gains=gains)
    qids, queries = standard_queries()
    nm = nm.to(DEV)
    nm.set_queries(qids)
    return nm, queries


def _logits(nm, batch, mode, min_rows=1):
    old = GM.RESIDENT_SHMP, GM.RESIDENT_MIN_ROWS
    GM.RESIDENT_SHMP, GM.RESIDENT_MIN_ROWS = mode, min_rows
    try:
        with torch.no_grad():
            return nm._logits(batch, exp2=False)
    finally:
        GM.RESIDENT_SHMP, GM.RESIDENT_MIN_ROWS = old


def test_fragment_layout_is_the_mfma_b_operand():
    """resident_fragments: lane (n = lane & 15, q = lane >> 4) of column tile t holds
    W[k = 64 b + 32 h + 8 q .. + 7][16 t + n] of every bf16 plane."""
    torch.manual_seed(0)
    wt_tab, wt_can, wt_cnt = torch.randn(64, 128, device=DEV), torch.randn(192, 64, device=DEV), torch.randn(192, 64, device=DEV)
    fr = ops.resident_fragments(wt_tab, wt_can, wt_cnt).cpu().numpy()          # [16][3][4][64][8]
    assert fr.shape == (16, 3, 4, 64, 8)
    pl = ops.split_bf16_planes(wt_cnt.t().contiguous()).cpu().numpy()          # [3][64 n][192 k]
    for b in range(3):
        for h in range(2):
            step = fr[10 + 2 * b + h]
            for lane in (0, 5, 17, 40, 63):
                for t in range(4):
                    ref = pl[:, 16 * t + (lane & 15), 64 * b + 32 * h + 8 * (lane >> 4):][:, :8]
                    assert (step[:, t, lane] == ref).all()
    pt = ops.split_bf16_planes(wt_tab.t().contiguous()).cpu().numpy()          # [3][128 n][64 k]
    for s, (h, j) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        for lane in (3, 31, 62):
            for t in range(4):
                ref = pt[:, 64 * j + 16 * t + (lane & 15), 32 * h + 8 * (lane >> 4):][:, :8]
                assert (fr[s][:, t, lane] == ref).all()


def test_resident_vs_oracle_and_layerwise_on_golden_graphs():
    nm, queries = _model()
    graphs = golden_graphs(max_n=60)
    part = build_partition(GraphSet.from_edge_lists(graphs), 4)
    batch = NeighborhoodBatch(part, DEV)
    plan = batch.resident_plan(1)
    assert plan["num_packs"] > 1 and plan["rest_batch"] is None
    res = _logits(nm, batch, True)
    lay = _logits(nm, batch, False)
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    ref = OM.neighborhood_logits(cpu_sd(nm), OP.neighborhood_batch(neighs), OP.query_batch(queries),
                                 emulate_quirk=False)[0]
    report("resident vs oracle", res, ref)
    report("resident vs layer-by-layer", res, lay)
    torch.testing.assert_close(res.cpu(), ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res, lay, rtol=1e-4, atol=1e-4)


def _dense_graphs():
    """Syn_1827-shaped graphs incl. some of the largest (neighborhoods above the pack limits) plus a hub
    graph (rows with hundreds of sources: cooperative gather) and molecule-sized ones."""
    syn = synthetic.syn_1827_shaped(1827).edge_lists()
    pick = [syn[i] for i in (5, 300, 900, 1400, 1500, 1650, 1750, 1800, 1826)]
    hub_n = 420
    hub = (hub_n, [(400, v) for v in range(0, 399)] + [(v, v + 1) for v in range(0, 380, 3)] +
           [(410, v) for v in range(0, 300, 2)])
    return golden_graphs(max_n=41)[:5] + pick + [hub]


def test_resident_dense_shapes_vs_layerwise_incl_oversize_and_hubs():
    nm, queries = _model(gains=(0.8, 1.2))
    part = build_partition(GraphSet.from_edge_lists(_dense_graphs()), 4)
    batch = NeighborhoodBatch(part, DEV)
    plan = batch.resident_plan(1)
    n = np.diff(part.count_ptr)
    print(f"[shape] {part.num_neigh} neighborhoods, {part.num_rows} rows, max {n.max()} count rows; "
          f"{plan['num_packs']} packs, oversize {0 if plan['rest_index'] is None else len(plan['rest_index'])}")
    assert plan["rest_batch"] is not None and n.max() > 500
    rmax = ops.resident_limits()[0]
    assert int((n <= rmax).sum()) + len(plan["rest_index"]) == part.num_neigh
    res = _logits(nm, batch, True)
    lay = _logits(nm, batch, False)
    # the product setting: small neighborhoods take the layer-by-layer path too
    mix = _logits(nm, batch, True, min_rows=GM.RESIDENT_MIN_ROWS)
    pm = batch.resident_plan(GM.RESIDENT_MIN_ROWS)
    assert 0 < pm["neighborhoods"] < plan["neighborhoods"]
    torch.testing.assert_close(mix, lay, rtol=1e-4, atol=1e-4)
    assert torch.isfinite(lay).all() and float(lay.std()) > 0
    report("resident vs layer-by-layer (dense)", res, lay)
    torch.testing.assert_close(res, lay, rtol=1e-4, atol=1e-4)


def test_resident_vs_oracle_on_a_dense_graph():
    nm, queries = _model(gains=(0.8, 1.2))
    syn = synthetic.syn_1827_shaped(1827).edge_lists()
    graphs = [syn[1100], syn[1450]]
    part = build_partition(GraphSet.from_edge_lists(graphs), 4)
    batch = NeighborhoodBatch(part, DEV)
    res = _logits(nm, batch, True)
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    ref = OM.neighborhood_logits(cpu_sd(nm), OP.neighborhood_batch(neighs), OP.query_batch(queries),
                                 emulate_quirk=False)[0]
    report("resident vs oracle (dense)", res, ref)
    torch.testing.assert_close(res.cpu(), ref, rtol=1e-4, atol=1e-4)


def test_resident_results_do_not_depend_on_placement():
    """Bit for bit: a neighborhood's pooled sums are added up in an order fixed by the neighborhood alone
    (tiles aligned to its first row, cooperative gather by a row's own degree), so any shard / pack /
    launch composition gives the same logits (N ranks == 1 rank)."""
    nm, queries = _model(gains=(0.8, 1.2))
    graphs = _dense_graphs()
    part = build_partition(GraphSet.from_edge_lists(graphs), 4)
    full = _logits(nm, NeighborhoodBatch(part, DEV), True)
    B = part.num_neigh
    for b0, b1 in ((0, B // 3), (B // 3, B // 3 + 77), (B // 3 + 77, B)):
        piece = _logits(nm, NeighborhoodBatch(part.slice(b0, b1), DEV), True)
        assert torch.equal(piece, full[b0:b1]), (b0, b1)
    idx = np.arange(B)[::3]
    sub = _logits(nm, NeighborhoodBatch(part.select(idx), DEV), True)
    want = full[torch.from_numpy(idx).to(DEV)]
    # neighborhoods above the pack limits run through the layer-by-layer kernels, whose pooling partials
    # depend on where 16-row tiles cut a neighborhood: those agree to fp32 rounding, the packed ones bit for bit
    rmax = ops.resident_limits()[0]
    packed = torch.from_numpy(np.diff(part.count_ptr)[idx] <= rmax).to(DEV)
    assert int(packed.sum()) > 100 and int((~packed).sum()) > 10
    assert torch.equal(sub[packed], want[packed])
    torch.testing.assert_close(sub[~packed], want[~packed], rtol=1e-4, atol=1e-4)
    # the product setting (neighborhoods under RESIDENT_MIN_ROWS count rows stay on the layer-by-layer kernels):
    # the packed ones are still bit-identical in any composition
    mr = GM.RESIDENT_MIN_ROWS
    full2 = _logits(nm, NeighborhoodBatch(part, DEV), True, min_rows=mr)
    sub2 = _logits(nm, NeighborhoodBatch(part.select(idx), DEV), True, min_rows=mr)
    nsel = np.diff(part.count_ptr)[idx]
    packed2 = torch.from_numpy((nsel <= rmax) & (nsel >= mr)).to(DEV)
    assert int(packed2.sum()) > 50
    assert torch.equal(sub2[packed2], full2[torch.from_numpy(idx).to(DEV)][packed2])
