"""Seeded, shape-matched synthetic target graphs (no dataset is reachable offline).

Shapes follow SURVEY.md section 8(d): public TU statistics for MUTAG / COX2 / MSRC-21 /
IMDB-BINARY and the reference's own Syn_1827 recipe (subgraph_counting/syn_data.py:658-746,
restated from its parameters, not copied).  Seeds: ``20240817 + config_id``, numpy PCG64.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from .graphs import GraphSet

BASE_SEED = 20240817
EdgeList = Tuple[int, List[Tuple[int, int]]]


def _random_tree_bfs(n: int, rng: np.random.Generator, max_deg: int = 4) -> List[Tuple[int, int]]:
    """Random tree with bounded degree whose labels are in BFS order (parent id < child id)."""
    deg = np.zeros(n, dtype=np.int64)
    edges = []
    open_nodes = [0]
    for v in range(1, n):
        k = int(rng.integers(max(0, len(open_nodes) - 6), len(open_nodes)))  # attach near the frontier
        p = open_nodes[k]
        edges.append((p, v))
        deg[p] += 1
        deg[v] += 1
        if deg[p] >= max_deg:
            open_nodes.pop(k)
        open_nodes.append(v)
    return edges


def _bfs_dist(adj: List[List[int]], s: int, limit: int) -> np.ndarray:
    n = len(adj)
    d = np.full(n, -1, dtype=np.int64)
    d[s] = 0
    q = [s]
    for u in q:
        if d[u] >= limit:
            continue
        for w in adj[u]:
            if d[w] < 0:
                d[w] = d[u] + 1
                q.append(w)
    return d


def molecule_like(n: int, extra: int, rng: np.random.Generator) -> EdgeList:
    """Tree + ``extra`` ring-closing edges between nodes at tree distance 4-5 (5/6-rings)."""
    edges = _random_tree_bfs(n, rng)
    adj = [[] for _ in range(n)]
    for a, b in edges:
        adj[a].append(b)
        adj[b].append(a)
    have = set(edges)
    tries = 0
    while extra > 0 and tries < 50 * (extra + 1):
        tries += 1
        a = int(rng.integers(n))
        d = _bfs_dist(adj, a, 5)
        cand = np.nonzero((d == 4) | (d == 5))[0]
        cand = [int(c) for c in cand if len(adj[a]) < 4 and len(adj[c]) < 4]
        if not cand:
            continue
        b = cand[int(rng.integers(len(cand)))]
        e = (min(a, b), max(a, b))
        if e in have:
            continue
        have.add(e)
        edges.append(e)
        adj[a].append(b)
        adj[b].append(a)
        extra -= 1
    return n, edges


def _tu_molecules(num_graphs, mean_n, std_n, mean_extra, seed) -> GraphSet:
    rng = np.random.default_rng(np.random.PCG64(seed))
    graphs = []
    for _ in range(num_graphs):
        n = max(4, int(round(rng.normal(mean_n, std_n))))
        extra = int(rng.poisson(mean_extra))
        graphs.append(molecule_like(n, extra, rng))
    return GraphSet.from_edge_lists(graphs)


def mutag_shaped(num_graphs: int = 188) -> GraphSet:
    """C1: MUTAG-shaped (188 graphs, 17.9 nodes, 19.8 edges on average)."""
    return _tu_molecules(num_graphs, 17.9, 2.7, 2.9, BASE_SEED + 1)


def cox2_shaped(num_graphs: int = 467) -> GraphSet:
    """C2: COX2-shaped (467 graphs, 41.2 nodes, 43.5 edges on average)."""
    return _tu_molecules(num_graphs, 41.2, 6.2, 3.2, BASE_SEED + 2)


def _gnm(n: int, m: int, rng) -> EdgeList:
    m = int(min(m, n * (n - 1) // 2))
    have = set()
    while len(have) < m:
        a, b = (int(v) for v in rng.integers(n, size=2))
        if a != b:
            have.add((min(a, b), max(a, b)))
    return n, sorted(have)


def _force_connected(n: int, edges, rng) -> EdgeList:
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for a, b in edges:
        parent[find(a)] = find(b)
    edges = list(edges)
    roots = sorted({find(v) for v in range(n)})
    for r0, r1 in zip(roots[:-1], roots[1:]):
        edges.append((min(r0, r1), max(r0, r1)))
        parent[find(r0)] = find(r1)
    return n, sorted(set(edges))


def _ba(n: int, m: int, rng) -> EdgeList:
    m = max(1, min(m, n - 1))
    targets = list(range(m))
    rep = []
    edges = []
    for v in range(m, n):
        for t in set(targets):
            edges.append((t, v))
        rep.extend(set(targets))
        rep.extend([v] * len(set(targets)))
        targets = [rep[int(rng.integers(len(rep)))] for _ in range(m)]
    return n, sorted(set((min(a, b), max(a, b)) for a, b in edges if a != b))


def _ws(n: int, k: int, p: float, rng) -> EdgeList:
    k = max(2, min(k - k % 2, n - 1 - (n - 1) % 2))
    have = set()
    for v in range(n):
        for j in range(1, k // 2 + 1):
            a, b = v, (v + j) % n
            if rng.random() < p:
                b = int(rng.integers(n))
            if a != b:
                have.add((min(a, b), max(a, b)))
    return n, sorted(have)


def syn_1827_shaped(num_graphs: int = 1827) -> GraphSet:
    """C3/C4: Syn_1827-shaped.  Size / density schedule of syn_data.py:658-746: sid < 1380 ->
    n = sid//23 + 10, degree parameter 0.5*(sid%23) + 1 + Tri(-.5,0,.5); else
    n = 5*((sid-1380)//3) + 60 + Tri(-5,0,5), degree parameter in {1,2,3} + jitter;
    m = clip(int(N(1,.1) * int(n*d)), n-1, n(n-1)/2); family drawn uniformly from
    {ER/G(n,m), WS, BA, ...}; force-connected; randomly relabelled."""
    rng = np.random.default_rng(np.random.PCG64(BASE_SEED + 3))
    graphs = []
    for sid in range(num_graphs):
        s = sid * 1827 // max(num_graphs, 1) if num_graphs != 1827 else sid
        if s < 1380:
            n = s // 23 + 10
            d = 0.5 * (s % 23) + 1 + rng.triangular(-0.5, 0, 0.5)
        else:
            n = int(5 * ((s - 1380) // 3) + 60 + rng.triangular(-5, 0, 5))
            d = (s - 1380) % 3 + 1 + rng.triangular(-0.5, 0, 0.5)
        n = max(int(n), 4)
        m = int(rng.normal(1, 0.1) * int(n * d))
        m = int(np.clip(m, n - 1, n * (n - 1) // 2))
        fam = int(rng.integers(4))
        if fam <= 1:
            g = _gnm(n, m, rng)
        elif fam == 2:
            g = _ws(n, max(2, int(round(2 * m / n))), 0.3, rng)
        else:
            g = _ba(n, max(1, int(round(m / n))), rng)
        n_, edges = _force_connected(g[0], g[1], rng)
        perm = rng.permutation(n_)
        edges = sorted((int(min(perm[a], perm[b])), int(max(perm[a], perm[b]))) for a, b in edges)
        graphs.append((n_, edges))
    return GraphSet.from_edge_lists(graphs)


def msrc_imdb_mixed(num_msrc: int = 563, num_imdb: int = 1000) -> GraphSet:
    """C5: MSRC-21-shaped G(n,m) graphs (n~N(77.5,11.6), m=2.56n) interleaved with
    IMDB-BINARY-shaped clique unions (n~N(19.8,3), m~4.9n)."""
    rng = np.random.default_rng(np.random.PCG64(BASE_SEED + 5))
    msrc, imdb = [], []
    for _ in range(num_msrc):
        n = max(10, int(round(rng.normal(77.5, 11.6))))
        msrc.append(_force_connected(*_gnm(n, int(2.56 * n), rng), rng))
    for _ in range(num_imdb):
        n = max(8, int(round(rng.normal(19.8, 3.0))))
        have = set()
        ego = 0
        rest = list(range(1, n))
        rng.shuffle(rest)
        k = int(rng.integers(1, 4))
        for part in np.array_split(np.array(rest), k):
            members = [ego] + [int(v) for v in part]
            for i, a in enumerate(members):
                for b in members[i + 1:]:
                    have.add((min(a, b), max(a, b)))
        imdb.append((n, sorted(have)))
    out = []
    i = j = 0
    while i < len(msrc) or j < len(imdb):
        if i < len(msrc):
            out.append(msrc[i])
            i += 1
        for _ in range(2):
            if j < len(imdb):
                out.append(imdb[j])
                j += 1
    return GraphSet.from_edge_lists(out)


WORKLOADS = {
    "mutag": mutag_shaped,
    "cox2": cox2_shaped,
    "syn_1827": syn_1827_shaped,
    "msrc_imdb": msrc_imdb_mixed,
}
