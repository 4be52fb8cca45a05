"""Native canonical-count enumerator vs the reference's VF2 counts (golden) and brute force."""
import numpy as np
import torch

from desco_amd.graphs import GraphSet
from desco_amd.groundtruth import canonical_counts
from oracle import partition as OP


def test_counts_match_reference_vf2_golden(partition_golden, queries_golden, counts_golden):
    qs = [(q["n"], [tuple(e) for e in q["edges"]]) for q in queries_golden["queries"]]
    by_name = {g["name"]: g for g in partition_golden["graphs"]}
    graphs = [(by_name[c["name"]]["n"], [tuple(e) for e in by_name[c["name"]]["edges"]])
              for c in counts_golden]
    gs = GraphSet.from_edge_lists(graphs)
    got = canonical_counts(gs, qs, num_threads=4)
    want = torch.tensor(np.concatenate([np.array(c["count"]) for c in counts_golden]), dtype=torch.double)
    assert got.shape == want.shape and want.sum() > 1000
    assert torch.equal(got, want)                      # bit-exact integers
    # Appendix B spot checks: P3 -> {3:2,4:2,5:1,6:1,7:5}, triangle -> {2:1,3:1}
    toy = got[:8]
    assert toy[:, 0].tolist() == [0, 0, 0, 2, 2, 1, 1, 5] and toy[:, 1].tolist() == [0, 0, 1, 1, 0, 0, 0, 0]


def test_counts_size2_and_6_vs_bruteforce():
    rng = np.random.default_rng(0)
    edges = sorted({(int(min(a, b)), int(max(a, b))) for a, b in rng.integers(0, 11, size=(24, 2)) if a != b})
    qs = [(2, [(0, 1)]), (6, [(i, i + 1) for i in range(5)]), (6, [(i, (i + 1) % 6) for i in range(6)]),
          (4, [(0, 1), (0, 2), (0, 3)])]
    got = canonical_counts(GraphSet.from_edge_lists([(11, edges)]), qs)
    want = OP.canonical_counts_bruteforce(11, edges, qs)
    assert got.long().tolist() == want.tolist()


def test_random_graph_families_vs_bruteforce():
    """The 29 standard queries on small graphs of eleven families (stars, wheels, grids, barbells, cliques with tails,
    trees, G(n,p), ... with shuffled ids): the host enumerator equals the brute-force definition, node by node."""
    from helpers import random_family_graphs, standard_queries
    _, queries = standard_queries()
    graphs = [g for g in random_family_graphs(41, 120) if g[0] <= 13][:22]
    assert len(graphs) >= 15
    got = canonical_counts(GraphSet.from_edge_lists(graphs), queries, backend="host").long()
    want = np.concatenate([OP.canonical_counts_bruteforce(n, e, queries) for n, e in graphs])
    assert want.sum() > 500 and got.tolist() == want.tolist()


def test_workload_compute_groundtruth(tmp_path):
    from desco_amd.data import STANDARD_QUERY_IDS
    from desco_amd.workload import Workload
    g = [(8, [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (3, 4), (4, 5), (5, 6), (6, 7), (7, 2)])]
    w = Workload(GraphSet.from_edge_lists(g), str(tmp_path))
    assert not w.exist_groundtruth(STANDARD_QUERY_IDS)
    t = w.compute_groundtruth(query_ids=STANDARD_QUERY_IDS)
    assert t.shape == (8, 29) and w.exist_groundtruth(STANDARD_QUERY_IDS)
    assert (tmp_path / "CanonicalCountTruth" / "query_num_29_query_len_sum_135.pt").exists()
    w2 = Workload(GraphSet.from_edge_lists(g), str(tmp_path))
    assert torch.equal(w2.load_groundtruth(STANDARD_QUERY_IDS), t)
    w2.generate_pipeline_datasets(4)
    assert torch.equal(w2.neighborhood_dataset.y, t[1:]) and torch.equal(w2.gossip_dataset.y, t)
