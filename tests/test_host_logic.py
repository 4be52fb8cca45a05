"""Host-side mirror of the reference API: config defaults, metrics, queries, state-dict names,
workload bookkeeping -- all CPU."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import gossip_args, make_models, neigh_args, standard_queries


def test_config_defaults_match_reference():
    from desco_amd import config
    p = argparse.ArgumentParser()
    config.parse_neighborhood(p)
    config.parse_gossip(p)
    config.parse_optimizer(p)
    ns = vars(p.parse_args([]))
    assert ns == load_golden("config_defaults.json")
    n, g, o = config.split_namespaces(p.parse_args(["--neigh_layer_num", "3", "--gpu", "0", "1"]))
    assert n.layer_num == 3 and n.conv_type == "SAGE" and n.depth == 4 and g.lr == 1e-3
    assert g.conv_type == "GOSSIP" and o.gpu == [0, 1] and n.batch_size == 512 and g.batch_size == 256


def test_metrics_match_reference():
    from desco_amd import analysis
    m = load_golden("metrics.json")
    pred, truth = np.array(m["pred"], dtype=np.float32), np.array(m["truth"], dtype=np.float32)
    np.testing.assert_allclose(analysis.norm_mse(pred, truth, m["groups"]), m["norm_mse"], rtol=1e-12)
    np.testing.assert_allclose(analysis.mse(pred, truth, m["groups"]), m["mse"], rtol=1e-12)
    np.testing.assert_allclose(analysis.mae(pred, truth, m["groups"]), m["mae"], rtol=1e-6)
    np.testing.assert_allclose(analysis.norm_mse(pred, truth), m["norm_mse_all"], rtol=1e-12)


def test_query_ids_and_atlas(queries_golden):
    from desco_amd import data
    assert data.gen_query_ids([3, 4, 5]) == queries_golden["query_ids"] == data.STANDARD_QUERY_IDS
    for q in queries_golden["queries"]:
        g = data.graph_atlas_plus(q["atlas_id"])
        assert g.number_of_nodes() == q["n"]
        assert sorted((min(a, b), max(a, b)) for a, b in g.edges()) == [tuple(e) for e in q["edges"]]
    with pytest.raises(NotImplementedError):
        data.graph_atlas_plus(8000)


def test_state_dict_names_and_sizes():
    nm, gm = make_models()
    ks = set(nm.state_dict())
    for k in ["emb_model.gnn_core.pre_mp.0.count.weight",
              "emb_model.gnn_core.convs.7.canonical__union_tride__count.lin.bias",
              "emb_model.gnn_core.convs.0.count__union_triangle__canonical.lin.weight",
              "emb_model.gnn_core.updates.3.canonical.weight", "emb_model.anchor_mlp.0.weight",
              "emb_model.post_mp.7.bias", "emb_model_query.gnn_core.pre_mp.0.union_node.bias",
              "emb_model_query.gnn_core.convs.2.union_node__union_tride__union_node.lin.weight",
              "count_model.0.weight", "count_model.2.bias"]:
        assert k in ks, k
    assert sum(p.numel() for p in nm.parameters()) == 1311105
    assert sum(p.numel() for p in nm.emb_model.parameters()) == 738560
    assert sum(p.numel() for p in nm.emb_model_query.parameters()) == 539264
    assert sum(p.numel() for p in gm.parameters()) == 144899
    gk = set(gm.state_dict())
    assert {"emb_model.gnn_core.convs.1.lin_gate.2.weight", "emb_model.gnn_core.convs.0.lin_com.weight",
            "emb_model.gnn_core.convs.0.lin_update.bias", "emb_model.post_mp.7.weight"} <= gk
    assert gm.emb_model.gnn_core.convs[0].lin_update.weight.shape == (64, 192)
    assert gm.emb_model.post_mp[0].weight.shape == (64, 256)


def test_checkpoint_roundtrip(tmp_path):
    from desco_amd.lightning_model import GossipCountingModel, NeighborhoodCountingModel
    nm, gm = make_models(seed=3)
    nm.save_checkpoint(tmp_path / "n.ckpt")
    gm.save_checkpoint(tmp_path / "g.ckpt")
    nm2 = NeighborhoodCountingModel.load_from_checkpoint(tmp_path / "n.ckpt")
    gm2 = GossipCountingModel.load_from_checkpoint(tmp_path / "g.ckpt")
    for a, b in ((nm, nm2), (gm, gm2)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    assert nm2.depth == 4 and nm2.emb_model.gnn_core.node_types == ["count", "canonical"]


def test_model_refuses_cpu_batches():
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.graphs import GraphSet
    from desco_amd.partition import build_partition
    nm, _ = make_models()
    nm.set_queries(standard_queries()[0])
    part = build_partition(GraphSet.from_edge_lists([(3, [(0, 1), (1, 2)])]), 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nm.graph_to_count(NeighborhoodBatch(part, "cpu"))


def test_set_queries_warns_on_small_depth():
    from desco_amd.lightning_model import NeighborhoodCountingModel
    nm = NeighborhoodCountingModel(1, 64, neigh_args(depth=1)).to_hetero_old(True, True)
    with pytest.warns(UserWarning, match="too small"):
        nm.set_queries([6, 14])


def test_workload_bookkeeping(tmp_path):
    from desco_amd.graphs import GraphSet
    from desco_amd.workload import Workload
    graphs = [(8, [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (3, 4), (4, 5), (5, 6), (6, 7), (7, 2)]),
              (3, [(0, 1)])]
    w = Workload(GraphSet.from_edge_lists(graphs), str(tmp_path))
    w.generate_pipeline_datasets(depth_neigh=4)
    nd = w.neighborhood_dataset
    assert nd.nx_neighs_indicator.tolist() == [False] + [True] * 7 + [False, True, False]
    assert nd.nx_neighs_index.tolist() == [[0, v] for v in range(1, 8)] + [[1, 1]]
    # reference file names / dtypes of the interchange pair (workload.py:197-213, 293-294)
    pdir = tmp_path / "NeighborhoodDataset" / "processed"
    idx = np.load(pdir / "neighs_index_depth_4.npy")
    ind = np.load(pdir / "neighs_indicator_depth_4.npy")
    assert idx.dtype == np.int64 and ind.dtype == bool and idx.shape == (8, 2)
    count = torch.arange(8 * 3, dtype=torch.float).reshape(8, 3)
    w.apply_neighborhood_count(count)
    x = w.gossip_dataset.x
    assert x.shape == (11, 3) and x[0].abs().sum() == 0 and torch.equal(x[1], count[0]) and torch.equal(x[9], count[7])
    assert torch.equal(nd.aggregate_neighborhood_count(count),
                       torch.stack([count[:7].sum(0), count[7]]))
    assert torch.equal(w.gossip_dataset.aggregate_neighborhood_count(x),
                       torch.stack([x[:8].sum(0), x[8:].sum(0)]))
    # cache is re-used
    w2 = Workload(GraphSet.from_edge_lists(graphs), str(tmp_path))
    w2.generate_pipeline_datasets(depth_neigh=4)
    assert (w2.neighborhood_dataset.partition.vcol == nd.partition.vcol).all()
    sizes = [b.num_graphs for b in nd.batches(3)]
    assert sizes == [3, 3, 2]
    truth = torch.arange(11 * 2, dtype=torch.double).reshape(11, 2)
    nd.apply_truth_from_dataset(truth)
    assert torch.equal(nd.y, truth[torch.from_numpy(nd.nx_neighs_indicator)])


def test_unsupported_transforms_are_refused_not_ignored(tmp_path):
    """A per-item PyG transform the native pipeline cannot run (workload.py:443-449 runs it in DataLoader workers) must
    raise; the two transforms main.py passes are accepted."""
    from desco_amd.graphs import GraphSet
    from desco_amd.transforms import ToTconvHetero, ZeroNodeFeat
    from desco_amd.workload import GossipDataset, NeighborhoodDataset, Workload
    gs = GraphSet.from_edge_lists([(4, [(0, 1), (1, 2), (2, 3), (3, 0)])])
    w = Workload(gs, str(tmp_path))
    w.generate_pipeline_datasets(depth_neigh=4, neighborhood_transform=ToTconvHetero(), gossip_transform=ZeroNodeFeat())
    with pytest.raises(NotImplementedError, match="transform=function"):
        w.generate_pipeline_datasets(depth_neigh=4, neighborhood_transform=lambda d: d)
    with pytest.raises(NotImplementedError, match="pre_transform"):
        w.generate_pipeline_datasets(depth_neigh=4, pre_transform=lambda d: d)
    with pytest.raises(NotImplementedError, match="pre_transform / pre_filter"):
        NeighborhoodDataset(4, None, dataset=gs, pre_filter=lambda d: True)
    with pytest.raises(NotImplementedError, match="GossipDataset"):
        GossipDataset(gs, transform=object())


def test_syn_edgelist_text_format_round_trip(tmp_path):
    """The reference's synthetic-dataset text files (data.py:644-750): global ids, per-graph edge
    counts; local ids = order of first appearance (from_networkx of add_edges_from)."""
    import os
    from desco_amd.data import load_data, read_syn_edgelist, write_syn_edgelist
    from desco_amd.graphs import GraphSet
    graphs = [(5, [(3, 4), (0, 3), (1, 0), (2, 1)]), (3, [(0, 2), (1, 2)]), (4, [(0, 1), (1, 2), (2, 3), (0, 3)])]
    raw = tmp_path / "Syn_3" / "raw"
    os.makedirs(raw)
    name = "Synthetic_size_min_10_max_500_graph_num_3"
    # hand-written file in the reference's layout (arbitrary line order inside a graph)
    with open(raw / f"{name}_edgelist.txt", "w") as f:
        f.write("# 12 10\n")
        base = 0
        for n, edges in graphs:
            for u, v in edges:
                f.write(f"{base + u} {base + v}\n")
            base += n
    with open(raw / f"{name}_graph_indicator.txt", "w") as f:
        f.write("# 3\n" + "".join(f"{len(e)}\n" for _, e in graphs))
    got = read_syn_edgelist(raw / f"{name}_edgelist.txt", raw / f"{name}_graph_indicator.txt")
    # expected: relabel every graph by first appearance in its edge lines
    exp = []
    for n, edges in graphs:
        order = {}
        for u, v in edges:
            order.setdefault(u, len(order))
            order.setdefault(v, len(order))
        exp.append((n, [(order[u], order[v]) for u, v in edges]))
    ref = GraphSet.from_edge_lists(exp)
    assert np.array_equal(got.graph_ptr, ref.graph_ptr)
    assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.col, ref.col)
    # load_data finds the files under <root>/Syn_3/raw like the reference's DeSCoSyntheticDataset
    via = load_data("Syn_3", root_folder=str(tmp_path))
    assert np.array_equal(via.col, ref.col)
    # writer -> reader is the identity up to the first-appearance relabelling of sorted edge lines
    write_syn_edgelist(ref, raw / "w_edgelist.txt", raw / "w_graph_indicator.txt")
    again = read_syn_edgelist(raw / "w_edgelist.txt", raw / "w_graph_indicator.txt")
    assert again.num_graphs == 3 and again.num_nodes == 12
    assert sorted(np.diff(again.rowptr).tolist()) == sorted(np.diff(ref.rowptr).tolist())


def test_committed_bench_line_has_the_contract_fields():
    """The newest committed bench line (profiles/r1_*_bench.json) carries every field of the
    driver's contract plus the roofline / cpu_baseline objects, with consistent values."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r1_*_bench.json")) if "unfused" not in f and "fused" not in f)
    assert files, "no committed bench line"
    d = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "graphs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["graphs_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_pool_index_slots_are_consistent_between_kernel_and_reduce():
    """The fused pooling's index (NeighborhoodBatch.pool_index): emulate the layer kernel's slot
    walk (one slot per segment end in a tile + one for a segment that runs on) and the reduce
    kernel's slot lookup (slot_base[t] + popcount of the ends before the segment's first row in the
    tile) -- every (tile, segment) pair must get one private slot and the partials must add up."""
    import torch
    from desco_amd.batch import NeighborhoodBatch
    rng = np.random.default_rng(3)
    for lens in ([1], [32], [33], [5, 1, 1, 90, 2, 31, 64, 1], list(rng.integers(1, 70, size=200)),
                 [1] * 100, [700, 3, 640]):
        cp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)

        class _P:
            count_ptr = cp
        nb = NeighborhoodBatch.__new__(NeighborhoodBatch)
        nb.part, nb.device = _P, torch.device("cpu")
        for TR in (16, 32):
            bits, slot, nslots = nb.pool_index(TR)
            bits = bits.numpy().view(np.uint32)
            slot = slot.numpy()
            nc = int(cp[-1])
            vals = rng.standard_normal(nc)
            part = np.full(nslots, np.nan)
            for t in range((nc + TR - 1) // TR):                # the layer kernel's walk over tile t
                s, run, nr = int(slot[t]), 0.0, min(TR, nc - TR * t)
                for r in range(nr):
                    run += vals[TR * t + r]
                    if (bits[t] >> r) & 1:
                        assert np.isnan(part[s])
                        part[s] = run
                        s, run = s + 1, 0.0
                if not (bits[t] >> (nr - 1)) & 1:
                    assert np.isnan(part[s])
                    part[s] = run
            assert not np.isnan(part).any()
            for b in range(len(lens)):                          # the reduce kernel's lookup
                r0, r1 = int(cp[b]), int(cp[b + 1])
                tot = 0.0
                for t in range(r0 // TR, (r1 - 1) // TR + 1):
                    first = max(r0 - TR * t, 0)
                    k = bin(int(bits[t]) & ((1 << first) - 1)).count("1")
                    tot += part[slot[t] + k]
                assert abs(tot - vals[r0:r1].sum()) < 1e-9


# ---- round 3: partition select, stacked weight folding ------------------------------------------------------
def _golden_partition():
    from desco_amd.graphs import GraphSet
    from desco_amd.partition import build_partition
    from helpers import golden_graphs
    return build_partition(GraphSet.from_edge_lists(golden_graphs()), 4)


def test_partition_select_equals_slice_and_keeps_neighborhoods_intact():
    part = _golden_partition()
    a, b = part.slice(10, 40), part.select(np.arange(10, 40))
    for f in ("count_ptr", "vrowptr", "vcol", "count_orig"):
        assert (getattr(a, f) == getattr(b, f)).all(), f
    idx = np.array([0, 3, 7, 8, 100, 200, part.num_neigh - 1])
    c = part.select(idx)
    assert c.num_neigh == len(idx) and (c.neigh_index == part.neigh_index[idx]).all()
    for i, bi in enumerate(idx):                       # every selected neighborhood is the same self-contained block
        one, two = part.slice(int(bi), int(bi) + 1), c.slice(i, i + 1)
        assert (one.vrowptr == two.vrowptr).all() and (one.vcol == two.vcol).all()


def _degree_sorted_numpy(part):
    """numpy restatement of desco_partition_degree_sort (the checker of the C++ entry point)."""
    from desco_amd.partition import NeighborhoodPartition
    Nc, B = part.num_count, part.num_neigh
    v = part.vrowptr.astype(np.int64)
    deg4 = np.diff(v)[:4 * Nc].reshape(Nc, 4).astype(np.int64)
    seg = np.repeat(np.arange(B, dtype=np.int64), np.diff(part.count_ptr.astype(np.int64)))
    # primary slot and direction are per NEIGHBORHOOD: its own slot totals, the parity of its node id inside the graph
    tot = np.zeros((B, 2), np.int64)
    np.add.at(tot, seg, deg4[:, :2])
    ps = (tot[:, 1] >= tot[:, 0]).astype(np.int64)[seg]
    nkey = part.neigh_index[:, 1].astype(np.int64)
    sign = np.where(nkey[seg] % 2 == 1, 1, -1)
    rows = np.arange(Nc)
    key = sign * ((deg4[rows, ps] << 32) + deg4[rows, 1 - ps])
    order = np.lexsort((np.arange(Nc), key, seg))            # stable: ties keep the old order
    new_of_old = np.empty(Nc, np.int64)
    new_of_old[order] = np.arange(Nc)
    four = np.arange(4, dtype=np.int64)
    vr_old = np.concatenate([(order[:, None] * 4 + four).ravel(), np.arange(4 * Nc, 4 * (Nc + B), dtype=np.int64)])
    deg = v[vr_old + 1] - v[vr_old]
    vr2 = np.concatenate([[0], np.cumsum(deg)])
    e_old = np.repeat(v[vr_old] - vr2[:-1], deg) + np.arange(int(vr2[-1]), dtype=np.int64)
    col_old = part.vcol[e_old].astype(np.int64)
    col_new = np.where(col_old < Nc, new_of_old[np.minimum(col_old, Nc - 1)], col_old)
    vrow_of_e = np.repeat(np.arange(len(deg), dtype=np.int64), deg)
    col_new = np.sort(vrow_of_e * (Nc + B) + col_new) % (Nc + B)
    return NeighborhoodPartition(part.neigh_index, part.indicator, part.count_ptr, part.count_orig[order],
                                 vr2.astype(np.int32), col_new.astype(np.int32), part.depth, part.quirk_batch)


def _typed_edges_in_original_ids(p):
    Nc, B = p.num_count, p.num_neigh
    v = p.vrowptr.astype(np.int64)
    vrow = np.repeat(np.arange(4 * (Nc + B)), np.diff(v))
    dst, slot, src = vrow // 4, vrow % 4, p.vcol.astype(np.int64)
    seg = np.repeat(np.arange(B), np.diff(p.count_ptr))
    node = lambda r: np.where(r < Nc, p.count_orig[np.minimum(r, Nc - 1)], -1)        # -1 = the canonical row
    nb = lambda r: np.where(r < Nc, seg[np.minimum(r, Nc - 1)], r - Nc)
    assert (nb(dst) == nb(src)).all()
    return set(zip(nb(dst).tolist(), node(dst).tolist(), node(src).tolist(), slot.tolist()))


def test_degree_sorted_rows_keep_every_neighborhood_intact():
    """desco_partition_degree_sort: same typed edge sets in original node ids, sources ascending inside a slot, rows
    monotone in the sort key with the direction alternating between neighborhoods, and equal to the numpy restatement."""
    from desco_amd import synthetic
    from desco_amd.partition import build_partition
    for part in (_golden_partition(), build_partition(synthetic.syn_1827_shaped(40), 4),
                 build_partition(synthetic.msrc_imdb_mixed(6, 10), 4)):
        got, want = part.degree_sorted(), _degree_sorted_numpy(part)
        for f in ("count_ptr", "vrowptr", "vcol", "count_orig"):
            assert (getattr(got, f) == getattr(want, f)).all(), f
        assert _typed_edges_in_original_ids(got) == _typed_edges_in_original_ids(part)
        v = got.vrowptr.astype(np.int64)
        assert all((np.diff(got.vcol[v[i]:v[i + 1]]) > 0).all() for i in range(len(v) - 1))
        assert (got.degree_sorted().vcol == got.vcol).all()               # idempotent (stable sort)
    empty = part.slice(0, 0)
    assert empty.degree_sorted().num_count == 0
    # placement independence: a neighborhood's row order does not depend on where the block cuts fall
    part = _golden_partition()
    whole = part.degree_sorted()
    cut = 37 if part.num_neigh > 80 else part.num_neigh // 2
    a, b = part.slice(0, cut).degree_sorted(), part.slice(cut, part.num_neigh).degree_sorted()
    assert (np.concatenate([a.count_orig, b.count_orig]) == whole.count_orig).all()
    # ... nor on which rank's shard holds the graph: the partition of a sub-dataset that starts at an ODD graph (graph ids
    # are relative to the shard) orders every neighborhood as the partition of the whole dataset does
    gs = synthetic.syn_1827_shaped(9)
    whole = build_partition(gs, 4).degree_sorted()
    for g0 in (1, 4):
        sub = build_partition(gs.subset(g0, gs.num_graphs), 4).degree_sorted()
        first = int(np.searchsorted(whole.neigh_index[:, 0], g0))
        tail = whole.slice(first, whole.num_neigh)
        off = int(gs.graph_ptr[g0])                                        # count_orig holds dataset-wide node ids
        assert (sub.count_orig + off == tail.count_orig).all() and (sub.vcol == tail.vcol).all(), g0


@pytest.mark.parametrize("use_tconv", [True, False])
def test_stacked_weight_folding_equals_the_per_layer_folding(use_tconv):
    """gnn_model.pack_shmp_stacked (training trunk) == gnn_model.pack_shmp per (layer, type), on CPU tensors."""
    import torch
    import desco_amd.gnn_model as GM
    from desco_amd.lightning_model import NeighborhoodCountingModel
    from helpers import neigh_args
    torch.manual_seed(3)
    nm = NeighborhoodCountingModel(1, 64, neigh_args(use_tconv=use_tconv)).to_hetero_old(use_tconv, use_tconv)
    for gnn in (nm.emb_model, nm.emb_model_query):
        with torch.no_grad():
            pk = GM.pack_shmp(gnn, bf16_planes=False)
            st = GM.pack_shmp_stacked(gnn)
        for t in gnn.gnn_core.node_types:
            Wt, fb = st[t]
            for l in range(gnn.gnn_core.layer_num):
                torch.testing.assert_close(Wt[l], pk["layers"][l][t]["wt"], rtol=1e-6, atol=1e-6)
                torch.testing.assert_close(fb[l], pk["layers"][l][t]["b"], rtol=1e-6, atol=1e-6)


def test_degree_table_index_is_a_faithful_remap_of_the_count_rows():
    """NeighborhoodBatch.degree_table_index (round 6, gnn_model.FIRST_LAYER_TABLE): the count rows' slot-degree tuples as
    an index into their distinct tuples.  On a real partition: the table's tuples are distinct, row_id maps every count
    row to ITS tuple, the remapped column ids of the two count-row slots address the table row of the ORIGINAL source
    (so that gathering T[vcol_t] equals gathering X_1[vcol] for any X_1 = T[row_id]) and the canonical slots keep their
    ids; too many distinct tuples -> None."""
    import torch
    from desco_amd.batch import NeighborhoodBatch
    part = _golden_partition()
    nb = NeighborhoodBatch.__new__(NeighborhoodBatch)
    nb.part, nb.device = part, torch.device("cpu")
    nb.slots, nb.num_count, nb.num_rows = 4, part.num_count, part.num_rows
    nb.vrowptr = torch.from_numpy(np.asarray(part.vrowptr, dtype=np.int32))
    nb.vcol = torch.from_numpy(np.asarray(part.vcol, dtype=np.int32))
    idx = nb.degree_table_index()
    assert idx is not None
    uptr, row_id, vcol_t = (t.numpy().astype(np.int64) for t in idx)
    S, nc, n = 4, part.num_count, part.num_rows
    vr = np.asarray(part.vrowptr, dtype=np.int64)
    deg = np.diff(vr).reshape(n, S)
    ut = np.diff(uptr).reshape(-1, S)
    assert len(np.unique(ut, axis=0)) == len(ut) and len(ut) < nc
    assert (ut[row_id] == deg[:nc]).all()
    # any X_1 that is a function of the tuple: gathers through the table equal gathers through the rows
    T = np.random.default_rng(0).standard_normal((len(ut), 3))
    X1 = T[row_id]
    vcol = np.asarray(part.vcol, dtype=np.int64)
    slot_of = np.repeat(np.arange(n * S) % S, deg.reshape(-1))
    low = slot_of < 2
    assert (vcol[low] < nc).all() and (vcol_t[~low] == vcol[~low]).all()
    assert np.array_equal(T[vcol_t[low]], X1[vcol[low]])
    nb2 = NeighborhoodBatch.__new__(NeighborhoodBatch)
    nb2.__dict__.update({k: v for k, v in nb.__dict__.items() if k != "_degree_table"})
    assert nb2.degree_table_index(max_rows=2) is None
