// Device-side canonical-partition builder (C ABI: desco_partition_dev_*), see include/desco_hip.h.
//
// The same integer algorithm as the host builder (partition.cpp) -- BFS ball of radius `depth` in
// the full graph, keep ids <= v, connected component of v, typed directed edges with the
// triangle / tride split by sorted-adjacency intersection (workload.py:243-294, data.py:329-396,
// transforms.py:180-255, 319-412) -- with one WAVEFRONT per target node v and the result streamed
// straight into the flat 4-slot CSR the kernels consume:
//
//   pass 1  desco_partition_dev_count   per node: neighborhood size, edges into its count rows, edges
//                                       into its canonical row (no triangle tests needed for sizes)
//   scan    desco_partition_dev_scan    exclusive prefix sums over the nodes -> neighborhood index,
//                                       row offset, edge offsets; totals (B, N_c, E_count, E_canon)
//   pass 2  desco_partition_dev_fill    recomputes each neighborhood and writes neigh_index,
//                                       count_ptr, count_orig, vrowptr, vcol at their final places
//
// Wave-level data structures live in LDS (two bitmaps over the graph's local ids, a rank-prefix
// array, two node queues, a slot-code buffer; ~9 bytes per node of the largest graph and wave, so
// graphs up to ~4400 nodes fit -- larger ones take the host builder): the component is kept as a
// BITMAP, so "ascending id, canonical last" needs no sort -- a node's local row is the rank of its
// bit, and the canonical node (the maximum id) is the last bit.  Output is bit-identical to the host
// builder's (tests/test_partition_dev_gpu.py).  The PyG remove_self_loops quirk emulation
// (quirk_batch) is only offered by the host builder.
#include "common_device.hpp"

namespace desco {

struct PartDevArgs {
  const int64_t* graph_ptr;     // [G+1]
  const int32_t* node_graph;    // [V] graph id of every node
  const int32_t* rowptr;        // [V+1] CSR over global node ids (symmetric, sorted rows)
  const int32_t* col;
  int64_t num_nodes;
  int depth;
  int64_t ws_words;             // LDS words per wave
  int n_max;                    // largest graph (nodes): sizes the bitmaps / queues
  // pass 1 outputs
  int32_t* nsize;               // nodes of the neighborhood (0 = skipped: no edge)
  int32_t* ecnt_count;          // directed edges into its count rows
  int32_t* ecnt_canon;          // directed edges into its canonical row
  // pass 2 inputs (exclusive scans over the nodes) and outputs
  const int64_t* b_index;
  const int64_t* row_off;
  const int64_t* eoff_count;
  const int64_t* eoff_canon;
  int64_t B, Nc, Ecount, Etotal;
  int64_t* neigh_index;         // [B,2]
  uint8_t* indicator;           // [V]
  int32_t* count_ptr;           // [B+1]
  int32_t* count_orig;          // [Nc]
  int32_t* vrowptr;             // [4*(Nc+B)+1]
  int32_t* vcol;                // [E]
};

__device__ __forceinline__ bool bm_test(const uint32_t* bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

// One BFS expansion sweep of the whole wave: for every node of `cur`, lanes stride over its
// adjacency; newly marked nodes are appended to `nxt` (wave-aggregated).  COMP = false: ball of the
// full graph; COMP = true: only neighbours <= v that are in the ball bitmap.
template <bool COMP>
__device__ __forceinline__ int expand(const PartDevArgs& g, int64_t base, int v, const int* cur, int ncur,
                                      int* nxt, uint32_t* bm_mark, const uint32_t* bm_ball, int lane) {
  int nn = 0;
  for (int i = 0; i < ncur; ++i) {
    const int u = cur[i];
    const int e0 = g.rowptr[base + u], e1 = g.rowptr[base + u + 1];
    for (int eb = e0; eb < e1; eb += 64) {
      const int e = eb + lane;
      bool fresh = false;
      int w = 0;
      if (e < e1) {
        w = g.col[e] - (int)base;
        const bool ok = COMP ? (w <= v && bm_test(bm_ball, w)) : true;
        if (ok) {
          const uint32_t bit = 1u << (w & 31);
          const uint32_t old = atomicOr(&bm_mark[w >> 5], bit);
          fresh = !(old & bit);
        }
      }
      const unsigned long long m = __ballot(fresh);
      if (fresh) nxt[nn + __popcll(m & ((1ull << lane) - 1ull))] = w;
      nn += __popcll(m);
    }
  }
  return nn;
}

// true iff a and b have a common neighbour inside the component (rows sorted ascending)
__device__ __forceinline__ bool share_neighbor(const int32_t* ra, int da, const int32_t* rb, int db,
                                               const uint32_t* bm_comp, int base) {
  int i = 0, j = 0;
  while (i < da && j < db) {
    const int x = ra[i], y = rb[j];
    if (x == y) {
      if (bm_test(bm_comp, x - base)) return true;
      ++i;
      ++j;
    } else if (x < y) {
      ++i;
    } else {
      ++j;
    }
  }
  return false;
}

template <bool FILL>
__global__ __launch_bounds__(256) void partition_dev_kernel(PartDevArgs g) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  extern __shared__ uint32_t part_lds[];
  const int nw = (g.n_max + 31) >> 5;                       // bitmap words
  uint32_t* wsw = part_lds + (threadIdx.x >> 6) * g.ws_words;
  if (FILL && wave == 0 && lane == 0) {                     // closing entries
    g.vrowptr[4 * (g.Nc + g.B)] = (int32_t)g.Etotal;
    if (g.B == 0) g.count_ptr[0] = 0;
  }
  uint32_t* bm_ball = wsw;
  uint32_t* bm_comp = wsw + nw;
  int* prefix = reinterpret_cast<int*>(wsw + 2 * nw);       // rank of the first bit of every word
  int* qa = prefix + nw;                                     // node queues [n_max]
  int* qb = qa + g.n_max;
  uint8_t* slotbuf = reinterpret_cast<uint8_t*>(qb + g.n_max);   // [n_max] slot code per neighbour

  for (int64_t gv = wave; gv < g.num_nodes; gv += nwaves) {
    const int gid = g.node_graph[gv];
    const int64_t base = g.graph_ptr[gid];
    const int n = (int)(g.graph_ptr[gid + 1] - base);
    const int v = (int)(gv - base);
    const int nwg = (n + 31) >> 5;
    for (int i = lane; i < nwg; i += 64) {
      bm_ball[i] = 0u;
      bm_comp[i] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (1) BFS ball of radius depth in the FULL graph (data.py:329-338)
    if (lane == 0) {
      bm_ball[v >> 5] = 1u << (v & 31);
      qa[0] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int* cur = qa;
    int* nxt = qb;
    int ncur = 1;
    for (int l = 0; l < g.depth && ncur > 0; ++l) {
      ncur = expand<false>(g, base, v, cur, ncur, nxt, bm_ball, bm_ball, lane);
      int* t = cur;
      cur = nxt;
      nxt = t;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // (2)+(3) ids <= v (applied after the BFS, data.py:385) and the component of v (data.py:387-390)
    if (lane == 0) {
      bm_comp[v >> 5] = 1u << (v & 31);
      qa[0] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    cur = qa;
    nxt = qb;
    ncur = 1;
    int nn = 1;
    while (ncur > 0) {
      ncur = expand<true>(g, base, v, cur, ncur, nxt, bm_comp, bm_ball, lane);
      nn += ncur;
      int* t = cur;
      cur = nxt;
      nxt = t;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (!FILL) {
      // sizes only: every neighbour <= v inside the component is one directed edge into this row
      int ec = 0, ek = 0;
      if (nn > 1) {
        for (int wi = 0; wi < nwg; ++wi) {
          uint32_t bits = bm_comp[wi];
          while (bits) {
            const int a = (wi << 5) + __builtin_ctz(bits);
            bits &= bits - 1;
            const int e0 = g.rowptr[base + a], e1 = g.rowptr[base + a + 1];
            int c = 0;
            for (int e = e0 + lane; e < e1; e += 64) {
              const int b = g.col[e] - (int)base;
              c += (b <= v && bm_test(bm_comp, b)) ? 1 : 0;
            }
            c = (int)wave_sum((float)c);        // degrees are far below 2^24: exact in fp32
            if (a == v)
              ek += c;
            else
              ec += c;
          }
        }
      }
      if (lane == 0) {
        g.nsize[gv] = nn > 1 ? nn : 0;        // 0 edges -> skipped (workload.py:252-256)
        g.ecnt_count[gv] = ec;
        g.ecnt_canon[gv] = ek;
      }
      continue;
    }
    // ---- pass 2: write the neighborhood ---------------------------------------------------------
    if (lane == 0) g.indicator[gv] = nn > 1 ? 1 : 0;
    if (nn <= 1) continue;
    // rank prefix per bitmap word (local row of a node = rank of its bit)
    {
      int run = 0;
      for (int w0 = 0; w0 < nwg; w0 += 64) {
        const int wi = w0 + lane;
        const int pc = wi < nwg ? __popc(bm_comp[wi]) : 0;
        int inc = pc;                                      // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        if (wi < nwg) prefix[wi] = run + inc - pc;
        run += __shfl(inc, 63, 64);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int64_t b = g.b_index[gv];
    const int64_t r0 = g.row_off[gv];                      // first count row
    const int64_t canon_row = g.Nc + b;
    if (lane == 0) {
      g.neigh_index[2 * b] = gid;
      g.neigh_index[2 * b + 1] = v;
      g.count_ptr[b] = (int32_t)r0;
      if (b == g.B - 1) g.count_ptr[g.B] = (int32_t)(r0 + nn - 1);
    }
    int64_t epos = g.eoff_count[gv];                       // running edge offset over the count rows
    for (int wi = 0; wi < nwg; ++wi) {
      uint32_t bits = bm_comp[wi];
      while (bits) {
        const int a = (wi << 5) + __builtin_ctz(bits);
        bits &= bits - 1;
        const bool is_canon = a == v;
        const int64_t row = is_canon ? canon_row : r0 + prefix[a >> 5] + __popc(bm_comp[a >> 5] & ((1u << (a & 31)) - 1u));
        if (!is_canon && lane == 0) g.count_orig[row] = (int32_t)(base + a);
        const int e0 = g.rowptr[base + a], e1 = g.rowptr[base + a + 1];
        const int32_t* ra = g.col + e0;
        const int da = e1 - e0;
        // sub-pass A: slot code of every neighbour (4 = not an edge of the neighborhood)
        int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
        for (int kb = 0; kb < da; kb += 64) {
          const int k = kb + lane;
          int code = 4;
          if (k < da) {
            const int bn = ra[k] - (int)base;
            if (bn <= v && bm_test(bm_comp, bn)) {
              const int eb0 = g.rowptr[base + bn], eb1 = g.rowptr[base + bn + 1];
              const bool tri = share_neighbor(ra, da, g.col + eb0, eb1 - eb0, bm_comp, (int)base);
              code = 2 * (bn == v ? 1 : 0) + (tri ? 0 : 1);
            }
            slotbuf[k] = (uint8_t)code;
          }
          cnt0 += __popcll(__ballot(code == 0));
          cnt1 += __popcll(__ballot(code == 1));
          cnt2 += __popcll(__ballot(code == 2));
          cnt3 += __popcll(__ballot(code == 3));
        }
        const int64_t o0 = is_canon ? g.Ecount + g.eoff_canon[gv] : epos;
        const int64_t o1 = o0 + cnt0, o2 = o1 + cnt1, o3 = o2 + cnt2;
        if (lane == 0) {
          g.vrowptr[4 * row] = (int32_t)o0;
          g.vrowptr[4 * row + 1] = (int32_t)o1;
          g.vrowptr[4 * row + 2] = (int32_t)o2;
          g.vrowptr[4 * row + 3] = (int32_t)o3;
        }
        if (!is_canon) epos = o3 + cnt3;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // sub-pass B: scatter the source rows (ascending inside a slot: adjacency is sorted)
        int run0 = 0, run1 = 0, run2 = 0, run3 = 0;
        for (int kb = 0; kb < da; kb += 64) {
          const int k = kb + lane;
          const int code = k < da ? (int)slotbuf[k] : 4;
          const unsigned long long lt = (1ull << lane) - 1ull;
          const unsigned long long m0 = __ballot(code == 0), m1 = __ballot(code == 1);
          const unsigned long long m2 = __ballot(code == 2), m3 = __ballot(code == 3);
          if (code < 4) {
            const int bn = ra[k] - (int)base;
            const int64_t src = bn == v ? canon_row
                                        : r0 + prefix[bn >> 5] + __popc(bm_comp[bn >> 5] & ((1u << (bn & 31)) - 1u));
            int64_t pos;
            if (code == 0)
              pos = o0 + run0 + __popcll(m0 & lt);
            else if (code == 1)
              pos = o1 + run1 + __popcll(m1 & lt);
            else if (code == 2)
              pos = o2 + run2 + __popcll(m2 & lt);
            else
              pos = o3 + run3 + __popcll(m3 & lt);
            g.vcol[pos] = (int32_t)src;
          }
          run0 += __popcll(m0);
          run1 += __popcll(m1);
          run2 += __popcll(m2);
          run3 += __popcll(m3);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }
}

// Exclusive prefix sums over the nodes (one block per output array, chunks of 1024 scanned in
// sequence: V is a few million at most and this runs once per dataset).
//   which = 0: kept (nsize > 0) -> b_index      1: nsize - 1 if kept -> row_off
//           2: ecnt_count -> eoff_count         3: ecnt_canon -> eoff_canon
// totals[which] = the grand total (B, Nc, Ecount, Ecanon).
__global__ __launch_bounds__(1024) void partition_scan_kernel(const int32_t* __restrict__ nsize,
                                                              const int32_t* __restrict__ ecnt_count,
                                                              const int32_t* __restrict__ ecnt_canon,
                                                              int64_t num_nodes, int64_t* b_index,
                                                              int64_t* row_off, int64_t* eoff_count,
                                                              int64_t* eoff_canon, int64_t* totals) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry;
  const int which = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t* out = which == 0 ? b_index : which == 1 ? row_off : which == 2 ? eoff_count : eoff_canon;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < num_nodes; c0 += 1024) {
    const int64_t i = c0 + tid;
    int64_t x = 0;
    if (i < num_nodes) {
      const int ns = nsize[i];
      x = which == 0 ? (ns > 0 ? 1 : 0)
                     : which == 1 ? (ns > 0 ? ns - 1 : 0)
                                  : which == 2 ? (ns > 0 ? ecnt_count[i] : 0) : (ns > 0 ? ecnt_canon[i] : 0);
    }
    int64_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int64_t base = carry;
    if (i < num_nodes) out[i] = base + woff + inc - x;
    __syncthreads();
    if (tid == 1023) carry = base + woff + inc;
    __syncthreads();
  }
  if (tid == 0) totals[which] = carry;
}

}  // namespace desco

using namespace desco;

static int64_t part_ws_words(int n_max) {
  if (n_max < 1) return 0;
  const int64_t nw = (n_max + 31) / 32;
  // 2 bitmaps + rank prefix + 2 queues + slot codes (bytes, rounded up to words)
  return 3 * nw + 2 * (int64_t)n_max + ((int64_t)n_max + 3) / 4 + 4;
}

static int part_dev_launch(bool fill, const PartDevArgs& a, int num_waves, hipStream_t st) {
  const unsigned blocks = (unsigned)((num_waves + 3) / 4);
  const size_t lds = (size_t)4 * a.ws_words * sizeof(uint32_t);
  if (lds > 160 * 1024)
    return fail(DESCO_EINVAL, "desco_partition_dev: largest graph does not fit the LDS workspace "
                              "(use desco_partition_build)");
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(partition_dev_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(partition_dev_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  if (fill)
    hipLaunchKernelGGL(partition_dev_kernel<true>, dim3(blocks), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL(partition_dev_kernel<false>, dim3(blocks), dim3(256), lds, st, a);
  return launch_status(fill ? "desco_partition_dev_fill" : "desco_partition_dev_count");
}

extern "C" int desco_partition_dev_count(const int64_t* graph_ptr, const int32_t* node_graph,
                                         const int32_t* rowptr, const int32_t* col, int64_t num_nodes,
                                         int depth, int n_max, int num_waves, int32_t* nsize, int32_t* ecnt_count, int32_t* ecnt_canon,
                                         desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  if (!graph_ptr || !node_graph || !rowptr || !nsize || !ecnt_count || !ecnt_canon ||
      num_nodes < 0 || depth < 0 || n_max < 1 || num_waves < 4 || num_waves % 4)
    return fail(DESCO_EINVAL, "desco_partition_dev_count: bad argument (num_waves % 4 == 0)");
  PartDevArgs a{};
  a.graph_ptr = graph_ptr;
  a.node_graph = node_graph;
  a.rowptr = rowptr;
  a.col = col;
  a.num_nodes = num_nodes;
  a.depth = depth;
  a.ws_words = part_ws_words(n_max);
  a.n_max = n_max;
  a.nsize = nsize;
  a.ecnt_count = ecnt_count;
  a.ecnt_canon = ecnt_canon;
  return part_dev_launch(false, a, num_waves, (hipStream_t)stream);
}

extern "C" int desco_partition_dev_scan(const int32_t* nsize, const int32_t* ecnt_count,
                                        const int32_t* ecnt_canon, int64_t num_nodes,
                                        int64_t* b_index, int64_t* row_off, int64_t* eoff_count,
                                        int64_t* eoff_canon, int64_t* totals4,
                                        desco_stream_t stream) {
  if (!nsize || !ecnt_count || !ecnt_canon || !b_index || !row_off || !eoff_count || !eoff_canon ||
      !totals4 || num_nodes < 0)
    return fail(DESCO_EINVAL, "desco_partition_dev_scan: bad argument");
  hipLaunchKernelGGL(partition_scan_kernel, dim3(4), dim3(1024), 0, (hipStream_t)stream, nsize,
                     ecnt_count, ecnt_canon, num_nodes, b_index, row_off, eoff_count, eoff_canon,
                     totals4);
  return launch_status("desco_partition_dev_scan");
}

extern "C" int desco_partition_dev_fill(const int64_t* graph_ptr, const int32_t* node_graph,
                                        const int32_t* rowptr, const int32_t* col, int64_t num_nodes,
                                        int depth, int n_max, int num_waves, const int64_t* b_index, const int64_t* row_off,
                                        const int64_t* eoff_count, const int64_t* eoff_canon,
                                        int64_t num_neigh, int64_t num_count, int64_t edges_count,
                                        int64_t edges_canon, int64_t* neigh_index, uint8_t* indicator,
                                        int32_t* count_ptr, int32_t* count_orig, int32_t* vrowptr,
                                        int32_t* vcol, desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  if (!graph_ptr || !node_graph || !rowptr || !b_index || !row_off || !eoff_count ||
      !eoff_canon || !indicator || !count_ptr || !vrowptr || num_nodes < 0 || depth < 0 || n_max < 1 ||
      num_waves < 4 || num_waves % 4 || num_neigh < 0 || num_count < 0 ||
      (num_neigh > 0 && (!neigh_index || (num_count > 0 && !count_orig))) ||
      4 * (num_count + num_neigh) + 1 > INT32_MAX || edges_count + edges_canon > INT32_MAX)
    return fail(DESCO_EINVAL, "desco_partition_dev_fill: bad argument or more than 2^31 rows / edges");
  PartDevArgs a{};
  a.graph_ptr = graph_ptr;
  a.node_graph = node_graph;
  a.rowptr = rowptr;
  a.col = col;
  a.num_nodes = num_nodes;
  a.depth = depth;
  a.ws_words = part_ws_words(n_max);
  a.n_max = n_max;
  a.b_index = b_index;
  a.row_off = row_off;
  a.eoff_count = eoff_count;
  a.eoff_canon = eoff_canon;
  a.B = num_neigh;
  a.Nc = num_count;
  a.Ecount = edges_count;
  a.neigh_index = neigh_index;
  a.indicator = indicator;
  a.count_ptr = count_ptr;
  a.count_orig = count_orig;
  a.vrowptr = vrowptr;
  a.vcol = vcol;
  a.Etotal = edges_count + edges_canon;
  return part_dev_launch(true, a, num_waves, (hipStream_t)stream);
}
