// Exact canonical ground-truth counts (C ABI: desco_canonical_counts), SURVEY.md 8f row N1.
//
// count[v][q] = number of node subsets S with max(S) = v whose INDUCED subgraph is isomorphic to
// query q -- what the reference obtains by running networkx VF2 (MatchSubgraphWorker,
// workload.py:327-348: one match per automorphism, keyed by max(vmap.keys())) and dividing by the
// query's symmetry factor (data.py:61-67).  Written from that definition with the ESU enumeration
// (every connected node subset of size <= kmax is visited exactly once, rooted at its maximum id)
// and an isomorphism-class lookup over induced adjacency bitmasks (<= 6 nodes -> <= 15 bits).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/desco_hip.h"
#include "common_host.hpp"

namespace {

constexpr int KMAX = 6;

inline int pair_bit(int a, int b) { return b * (b - 1) / 2 + a; }   // a < b

// canonical form of every adjacency mask on k nodes: minimum over all relabelings
std::vector<uint16_t> canon_table(int k) {
  const int nb = k * (k - 1) / 2;
  std::vector<uint16_t> tab((size_t)1 << nb, 0xffff);
  std::vector<int> perm(k);
  for (int i = 0; i < k; ++i) perm[i] = i;
  std::vector<std::vector<int>> perms;
  do perms.push_back(perm);
  while (std::next_permutation(perm.begin(), perm.end()));
  for (uint32_t m = 0; m < ((uint32_t)1 << nb); ++m) {
    uint32_t best = 0xffffffffu;
    for (const auto& p : perms) {
      uint32_t r = 0;
      for (int b = 1; b < k; ++b)
        for (int a = 0; a < b; ++a)
          if (m >> pair_bit(a, b) & 1) {
            const int x = std::min(p[a], p[b]), y = std::max(p[a], p[b]);
            r |= 1u << pair_bit(x, y);
          }
      best = std::min(best, r);
    }
    tab[m] = (uint16_t)best;
  }
  return tab;
}

struct Ctx {
  int64_t base, n;
  const int64_t* rowptr;
  const int32_t* col;
  int kmax;
  const std::vector<uint16_t>* canon;               // [k] -> table
  const std::vector<std::vector<int>>* cls2q;       // [k][canonical mask] -> query ids (flattened below)
  const std::vector<int32_t>* qlist;                 // per k: offsets into qids
  std::vector<uint64_t> bits;                        // adjacency bitset rows (n x words)
  int words;
  std::vector<uint8_t> seen;
  int64_t* out;
  int num_q;
  bool adj(int a, int b) const { return bits[(size_t)a * words + (b >> 6)] >> (b & 63) & 1; }
};

struct Classifier {
  // for size k: map canonical mask -> list of query indices
  std::vector<std::vector<int>> by_mask[KMAX + 1];
  std::vector<uint16_t> canon[KMAX + 1];
  bool used[KMAX + 1] = {false};
};

void classify(const Ctx& c, const Classifier& cl, const int* sub, int k, int v) {
  if (!cl.used[k]) return;
  uint32_t m = 0;
  for (int b = 1; b < k; ++b)
    for (int a = 0; a < b; ++a)
      if (c.adj(sub[a], sub[b])) m |= 1u << pair_bit(a, b);
  const auto& lst = cl.by_mask[k][cl.canon[k][m]];
  for (int q : lst) c.out[(c.base + v) * c.num_q + q] += 1;
}

void extend(Ctx& c, const Classifier& cl, int* sub, int nsub, std::vector<int>& ext, int v) {
  classify(c, cl, sub, nsub, v);
  if (nsub == c.kmax) return;
  std::vector<int> newly, ext2;
  while (!ext.empty()) {
    const int w = ext.back();
    ext.pop_back();
    newly.clear();
    const int64_t gw = c.base + w;
    for (int64_t e = c.rowptr[gw]; e < c.rowptr[gw + 1]; ++e) {
      const int u = (int)(c.col[e] - c.base);
      if (u >= v) break;                       // rows sorted ascending; only ids below the root
      if (!c.seen[u]) {
        c.seen[u] = 1;
        newly.push_back(u);
      }
    }
    ext2 = ext;
    ext2.insert(ext2.end(), newly.begin(), newly.end());
    sub[nsub] = w;
    extend(c, cl, sub, nsub + 1, ext2, v);
    for (int u : newly) c.seen[u] = 0;
  }
}

}  // namespace

extern "C" int desco_canonical_counts(const int64_t* graph_ptr, int64_t num_graphs,
                                      const int64_t* rowptr, const int32_t* col,
                                      const int32_t* q_nodes, const int32_t* q_edge_ptr,
                                      const int32_t* q_edges, int num_queries, int num_threads,
                                      int64_t* out) {
  if (!graph_ptr || !rowptr || !q_nodes || !q_edge_ptr || !out || num_graphs < 0 || num_queries < 0)
    return desco::fail(DESCO_EINVAL, "desco_canonical_counts: bad argument");
  try {
    Classifier cl;
    int kmax = 0;
    for (int q = 0; q < num_queries; ++q) {
      const int k = q_nodes[q];
      if (k < 2 || k > KMAX)
        return desco::fail(DESCO_EINVAL, "desco_canonical_counts: queries must have 2..6 nodes");
      kmax = std::max(kmax, k);
      if (!cl.used[k]) {
        cl.used[k] = true;
        cl.canon[k] = canon_table(k);
        cl.by_mask[k].assign((size_t)1 << (k * (k - 1) / 2), {});
      }
      uint32_t m = 0;
      for (int e = q_edge_ptr[q]; e < q_edge_ptr[q + 1]; ++e) {
        int a = q_edges[2 * e], b = q_edges[2 * e + 1];
        if (a == b || a < 0 || b < 0 || a >= k || b >= k)
          return desco::fail(DESCO_EINVAL, "desco_canonical_counts: bad query edge");
        if (a > b) std::swap(a, b);
        m |= 1u << pair_bit(a, b);
      }
      cl.by_mask[k][cl.canon[k][m]].push_back(q);
    }
    const int64_t total = graph_ptr[num_graphs];
    std::memset(out, 0, sizeof(int64_t) * (size_t)total * (size_t)num_queries);
    if (num_queries == 0) return 0;
#ifdef _OPENMP
    const int nt = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
    const int nt = 1;
    (void)num_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (int64_t g = 0; g < num_graphs; ++g) {
      Ctx c;
      c.base = graph_ptr[g];
      c.n = graph_ptr[g + 1] - c.base;
      c.rowptr = rowptr;
      c.col = col;
      c.kmax = kmax;
      c.out = out;
      c.num_q = num_queries;
      c.words = (int)((c.n + 63) / 64);
      c.bits.assign((size_t)c.n * c.words, 0);
      for (int64_t u = 0; u < c.n; ++u)
        for (int64_t e = rowptr[c.base + u]; e < rowptr[c.base + u + 1]; ++e) {
          const int w = (int)(col[e] - c.base);
          c.bits[(size_t)u * c.words + (w >> 6)] |= (uint64_t)1 << (w & 63);
        }
      c.seen.assign((size_t)c.n, 0);
      int sub[KMAX];
      std::vector<int> ext;
      for (int v = 0; v < (int)c.n; ++v) {
        ext.clear();
        c.seen[v] = 1;
        const int64_t gv = c.base + v;
        for (int64_t e = rowptr[gv]; e < rowptr[gv + 1]; ++e) {
          const int u = (int)(col[e] - c.base);
          if (u >= v) break;
          c.seen[u] = 1;
          ext.push_back(u);
        }
        std::vector<int> marked = ext;
        sub[0] = v;
        extend(c, cl, sub, 1, ext, v);
        for (int u : marked) c.seen[u] = 0;
        c.seen[v] = 0;
      }
    }
    return 0;
  } catch (const std::bad_alloc&) {
    return desco::fail(DESCO_ENOMEM, "desco_canonical_counts: out of memory");
  }
}
