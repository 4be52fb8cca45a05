// Exact canonical ground-truth counts on the GPU (C ABI: desco_canonical_counts_dev), SURVEY.md 8f
// row N1.  Same definition as the host enumerator (groundtruth.cpp):
//
//   count[v][q] = #{ node subsets S : max(S) = v, G[S] connected and isomorphic to query q }
//
// i.e. what the reference gets from networkx VF2 (MatchSubgraphWorker, workload.py:327-348, keyed by
// max(vmap.keys())) divided by the query's symmetry factor (data.py:61-67).
//
// Enumeration: ESU without extension lists.  In ESU a node u enters the extension set when the
// FIRST subset node adjacent to it is added, and choices are made in list order; with
// key(u) = (insertion index of that first neighbour, position of u in its adjacency row) this is
// exactly "chosen nodes have strictly increasing keys".  So a level is two nested loops over the
// adjacency rows of the subset's nodes, starting behind the previous key, and a candidate is taken
// iff it is below the root, not in the subset and adjacent to no earlier subset node -- the same
// adjacency bits then extend the induced-subgraph mask.  No per-thread lists: the subset (<= 5
// nodes), the mask and the loop cursors live in registers (levels are template-unrolled).
//
// Work item = one CSR entry (v, u0) with u0 < v: the subtree of subsets whose first chosen node is
// u0, handled by ONE WAVE: the candidates for the next node (the rest of v's row, then u0's row) are
// dealt round-robin to the 64 lanes and every lane walks the deeper levels of its candidates on its
// own.  (One thread per item left the launch waiting for the few items rooted at hubs, whose
// subtrees grow with the cube of the degree.)  Matches are tallied in an LDS column per thread
// (ds_add_u32, conflict-free), reduced over the wave and flushed with one 64-bit atomic per (item,
// query).  Adjacency tests read per-graph bitset rows (built by a
// first kernel from the CSR).  Integer work, bit-exact against the host enumerator.
#include "common_device.hpp"

namespace desco {

constexpr int GT_MAXQ = 32, GT_THREADS = 256, GT_KMAX = 5;
// class table: for k nodes the 2^(k(k-1)/2) adjacency masks start at GT_OFF[k]
__device__ __constant__ int GT_OFF_DEV[GT_KMAX + 2] = {0, 0, 0, 2, 10, 74, 1098};
static const int GT_OFF[GT_KMAX + 2] = {0, 0, 0, 2, 10, 74, 1098};

struct GtArgs {
  const int64_t* graph_ptr;
  const int64_t* rowptr;
  const int32_t* col;
  const int32_t* node_graph;
  const int64_t* bit_off;            // [G] first bitset word of graph g
  unsigned long long* bits;
  const int16_t* cls;                // [1098] query index of the mask's isomorphism class, or -1
  int kmax, Q;
  int64_t num_nodes, num_entries;
  unsigned long long* out;           // [N][Q]
};

__device__ __forceinline__ int64_t gt_row_of_entry(const int64_t* __restrict__ rowptr, int64_t n,
                                                   int64_t e) {
  int64_t lo = 0, hi = n;            // last row with rowptr[row] <= e
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(GT_THREADS) void gt_bits_kernel(GtArgs a) {
  const int64_t e = (int64_t)blockIdx.x * GT_THREADS + threadIdx.x;
  if (e >= a.num_entries) return;
  const int64_t v = gt_row_of_entry(a.rowptr, a.num_nodes, e);
  const int g = a.node_graph[v];
  const int64_t base = a.graph_ptr[g];
  const int words = (int)((a.graph_ptr[g + 1] - base + 63) >> 6);
  const int u = (int)(a.col[e] - base);
  atomicOr(a.bits + a.bit_off[g] + (v - base) * words + (u >> 6), 1ull << (u & 63));
}

struct GtCtx {
  const int64_t* rowptr;
  const int32_t* col;
  const unsigned long long* bits;    // this graph's rows
  const int16_t* cls;
  int64_t base;
  int words, lv, kmax;
  unsigned* cnt;                     // this thread's LDS column (stride GT_THREADS)
};

__device__ __forceinline__ unsigned gt_adj(const GtCtx& c, int a, int b) {
  return (unsigned)(c.bits[(int64_t)a * c.words + (b >> 6)] >> (b & 63)) & 1u;
}

template <int K>
__device__ __forceinline__ void gt_classify(const GtCtx& c, unsigned mask) {
  const int q = c.cls[GT_OFF_DEV[K] + mask];
  if (q >= 0) atomicAdd(c.cnt + q * GT_THREADS, 1u);
}

template <int NS>
__device__ void gt_extend(const GtCtx& c, const int (&S)[GT_KMAX], unsigned mask, int ci0, int cp0);

// candidate at position e of the adjacency row (starting at r0) of S[ci] for node number NS of the
// subset: taken iff it is below the root, new, and adjacent to no subset node before S[ci].
// Returns false when the row has passed the root's id (rows ascend: nothing further qualifies).
template <int NS>
__device__ __forceinline__ bool gt_try(const GtCtx& c, const int (&S)[GT_KMAX], unsigned mask, int ci,
                                       int64_t r0, int64_t e) {
  const int u = (int)(c.col[e] - c.base);
  if (u >= c.lv) return false;
  unsigned ab = 0;
  bool in_s = false;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    in_s |= u == S[j];
    ab |= gt_adj(c, S[j], u) << j;
  }
  if (in_s || (ab & ((1u << ci) - 1u))) return true;     // u entered the extension set earlier
  const unsigned m2 = mask | (ab << (NS * (NS - 1) / 2));
  gt_classify<NS + 1>(c, m2);
  if constexpr (NS + 1 < GT_KMAX) {
    if (NS + 1 < c.kmax) {
      int S2[GT_KMAX];
#pragma unroll
      for (int j = 0; j < GT_KMAX; ++j) S2[j] = j < NS ? S[j] : 0;
      S2[NS] = u;
      gt_extend<NS + 1>(c, S2, m2, ci, (int)(e - r0) + 1);
    }
  }
  return true;
}

// choose node number NS of the subset (S[0..NS-1] chosen, induced mask `mask`); candidates start at
// key (ci0, cp0)
template <int NS>
__device__ void gt_extend(const GtCtx& c, const int (&S)[GT_KMAX], unsigned mask, int ci0, int cp0) {
#pragma unroll
  for (int ci = 0; ci < NS; ++ci) {
    if (ci < ci0) continue;
    const int64_t r0 = c.rowptr[c.base + S[ci]], r1 = c.rowptr[c.base + S[ci] + 1];
    for (int64_t e = r0 + (ci == ci0 ? cp0 : 0); e < r1; ++e)
      if (!gt_try<NS>(c, S, mask, ci, r0, e)) break;
  }
}

__global__ __launch_bounds__(GT_THREADS) void gt_count_kernel(GtArgs a) {
  __shared__ unsigned cnt[GT_MAXQ * GT_THREADS];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int q = 0; q < a.Q; ++q) cnt[q * GT_THREADS + tid] = 0;
  const int64_t e = (int64_t)blockIdx.x * (GT_THREADS / 64) + (tid >> 6);     // one wave per entry
  if (e >= a.num_entries) return;                  // (no barrier below: columns are private)
  const int64_t v = gt_row_of_entry(a.rowptr, a.num_nodes, e);
  const int g = a.node_graph[v];
  GtCtx c;
  c.rowptr = a.rowptr;
  c.col = a.col;
  c.cls = a.cls;
  c.base = a.graph_ptr[g];
  c.words = (int)((a.graph_ptr[g + 1] - c.base + 63) >> 6);
  c.bits = a.bits + a.bit_off[g];
  c.lv = (int)(v - c.base);
  c.kmax = a.kmax;
  c.cnt = cnt + tid;
  const int u0 = (int)(a.col[e] - c.base);
  if (u0 >= c.lv) return;                          // the root is the maximum of its subsets
  const int S[GT_KMAX] = {c.lv, u0, 0, 0, 0};
  if (lane == 0) gt_classify<2>(c, 1u);
  if (c.kmax > 2) {
    // candidates for the third node: v's row behind u0, then u0's row; position p -> lane p % 64
    const int64_t v0 = a.rowptr[v], v1 = a.rowptr[v + 1];
    const int64_t w0 = a.rowptr[c.base + u0], w1 = a.rowptr[c.base + u0 + 1];
    const int64_t rest = v1 - (e + 1), total = rest + (w1 - w0);
    for (int64_t p = lane; p < total; p += 64) {
      if (p < rest)
        gt_try<2>(c, S, 1u, 0, v0, e + 1 + p);
      else
        gt_try<2>(c, S, 1u, 1, w0, w0 + (p - rest));
    }
  }
  for (int q = 0; q < a.Q; ++q) {
    unsigned long long n = cnt[q * GT_THREADS + tid];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0 && n) atomicAdd(a.out + v * a.Q + q, n);
  }
}

}  // namespace desco

using namespace desco;

namespace {
inline int gt_pair_bit(int a, int b) { return b * (b - 1) / 2 + a; }   // a < b
}

// HOST helper: the class table the device kernel reads.
extern "C" int desco_canonical_class_table(const int32_t* q_nodes, const int32_t* q_edge_ptr,
                                           const int32_t* q_edges, int num_queries, int16_t* table,
                                           int* kmax_out) {
  if (!q_nodes || !q_edge_ptr || !table || !kmax_out || num_queries < 0 || num_queries > GT_MAXQ)
    return fail(DESCO_EINVAL, "desco_canonical_class_table: bad argument (at most 32 queries)");
  for (int i = 0; i < GT_OFF[GT_KMAX + 1]; ++i) table[i] = -1;
  int kmax = 0;
  for (int q = 0; q < num_queries; ++q) {
    const int k = q_nodes[q];
    if (k < 2 || k > GT_KMAX)
      return fail(DESCO_EINVAL, "desco_canonical_class_table: the device path takes queries of 2..5 nodes");
    kmax = k > kmax ? k : kmax;
    uint32_t m = 0;
    for (int e = q_edge_ptr[q]; e < q_edge_ptr[q + 1]; ++e) {
      int a = q_edges[2 * e], b = q_edges[2 * e + 1];
      if (a == b || a < 0 || b < 0 || a >= k || b >= k)
        return fail(DESCO_EINVAL, "desco_canonical_class_table: bad query edge");
      if (a > b) { const int t = a; a = b; b = t; }
      m |= 1u << gt_pair_bit(a, b);
    }
    // every relabeling of the query is a mask of its class
    int perm[GT_KMAX];
    for (int i = 0; i < k; ++i) perm[i] = i;
    for (;;) {
      uint32_t r = 0;
      for (int b = 1; b < k; ++b)
        for (int a = 0; a < b; ++a)
          if (m >> gt_pair_bit(a, b) & 1) {
            const int x = perm[a] < perm[b] ? perm[a] : perm[b];
            const int y = perm[a] < perm[b] ? perm[b] : perm[a];
            r |= 1u << gt_pair_bit(x, y);
          }
      int16_t& slot = table[GT_OFF[k] + r];
      if (slot >= 0 && slot != q)
        return fail(DESCO_EINVAL, "desco_canonical_class_table: two queries are isomorphic");
      slot = (int16_t)q;
      // next permutation
      int i = k - 2;
      while (i >= 0 && perm[i] > perm[i + 1]) --i;
      if (i < 0) break;
      int j = k - 1;
      while (perm[j] < perm[i]) --j;
      { const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
      for (int lo = i + 1, hi = k - 1; lo < hi; ++lo, --hi) {
        const int t = perm[lo]; perm[lo] = perm[hi]; perm[hi] = t;
      }
    }
  }
  *kmax_out = kmax;
  return 0;
}

extern "C" int desco_canonical_counts_dev(const int64_t* graph_ptr, int64_t num_graphs,
                                          int64_t num_nodes, const int64_t* rowptr,
                                          int64_t num_entries, const int32_t* col,
                                          const int32_t* node_graph, const int64_t* bit_off,
                                          uint64_t* bits, int64_t num_words, const int16_t* cls,
                                          int kmax, int num_queries, int64_t* out,
                                          desco_stream_t stream) {
  if (num_nodes == 0 || num_queries == 0) return 0;
  if (!graph_ptr || !rowptr || !node_graph || !bit_off || !bits || !cls || !out || num_graphs < 0 ||
      num_nodes < 0 || num_entries < 0 || num_words < 0 || kmax < 2 || kmax > GT_KMAX ||
      num_queries < 0 || num_queries > GT_MAXQ || (num_entries > 0 && !col))
    return fail(DESCO_EINVAL, "desco_canonical_counts_dev: bad argument (queries of 2..5 nodes, at most 32)");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(bits, 0, (size_t)num_words * 8, s) != hipSuccess ||
      hipMemsetAsync(out, 0, (size_t)num_nodes * num_queries * 8, s) != hipSuccess)
    return launch_status("desco_canonical_counts_dev: memset");
  if (num_entries == 0) return 0;
  const int64_t blocks = (num_entries + GT_THREADS - 1) / GT_THREADS;
  const int64_t wblocks = (num_entries + GT_THREADS / 64 - 1) / (GT_THREADS / 64);
  if (wblocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_canonical_counts_dev: too many edges");
  GtArgs a{graph_ptr, rowptr, col, node_graph, bit_off, reinterpret_cast<unsigned long long*>(bits),
           cls, kmax, num_queries, num_nodes, num_entries, reinterpret_cast<unsigned long long*>(out)};
  hipLaunchKernelGGL(gt_bits_kernel, dim3((unsigned)blocks), dim3(GT_THREADS), 0, s, a);
  hipLaunchKernelGGL(gt_count_kernel, dim3((unsigned)wblocks), dim3(GT_THREADS), 0, s, a);
  return launch_status("desco_canonical_counts_dev");
}
