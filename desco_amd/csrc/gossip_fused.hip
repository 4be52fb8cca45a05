// Fused gossip stage for gfx950: for a tile of 128 nodes and ONE query the whole per-query network
// of the reference (BaseGNN gossip path, gnn_model.py:58-103, 230-260, 303-350; looped over queries
// in lightning_model.py:613-628) runs on chip:
//
//   scalars (pre-pass kernel)  a0,b0,a1,x per (node,query)          -> scal4 [N*Q] float4
//   h1   = relu(a0*p_q + b0*r + x*t + z_q)                           (layer 0, closed form)
//   hh   = sum_j (j<i ? g1 : 1-g1) * h1_j                            (h1_j recomputed from j's scalars)
//   h2   = relu([hh|h1] W1 + a1*u + d1)                              (layer 1, K=128)      MFMA
//   y1   = leaky([h1|h2] Wp + x*tp + zp_q, 0.1)                      (post_mp.0, K=128)    MFMA
//   y2   = relu(y1 W3 + b3)                                          (post_mp.3, K=64)     MFMA
//   out  = x + b7 + sum_c relu(y2 W5 + b5)[c] * w7[c]                (post_mp.5/.7, N=256) MFMA
//
// Nothing but the 16-byte scalar records and the [N,Q] result crosses HBM (the unfused path moves
// ~2.3 KB per (node,query)).  All four GEMMs run on the bf16 matrix pipe at fp32 accuracy (the
// 6-product split of gemm_split.hip): activations live in LDS only as three bf16 planes, the
// weights arrive pre-split from the host.
//
// Block = 8 waves = one CU, persistent over (tile, query) items.  LDS (151 KB):
//   two activation images [3 planes][128 rows][64 k] bf16 (h1/y1 and hh/h2/y2),
//   two weight-block buffers [3][64 n][64 k] bf16 (nine 64x64 blocks per item, double buffered:
//   block i+1 is fetched into registers under the MFMAs of block i and stored after them, so a
//   block costs ONE barrier), the tile's scalar records / row pointers, the bias vectors.
//   Rows are 128 B without padding; the 16-byte chunk index is XOR-swizzled with (row>>1)&7, which
//   makes every ds_read_b128 fragment read conflict-free.
// The GEMMs are computed TRANSPOSED (MFMA A operand = weight rows, B operand = activation rows):
// in the C/D layout a lane then owns a node and 4 consecutive output features per accumulator tile,
// so an epilogue packs bf16 pairs in registers and writes 8 bytes per plane, reads its
// per-node scalars once, and the final 256-wide dot product is a per-lane running sum.
// Wave (wm = wave&3, wn = wave>>2) owns nodes 32wm..+31 x features 32wn..+31 of every 64-wide block,
// as 2 x 2 tiles of v_mfma_f32_16x16x32_bf16 (a lane then owns two nodes and 2 x 4 features).
// The next item's CSR slice and scalar records are prefetched through registers in three stages
// under the current item's GEMMs.  Algebra: DESIGN.md 4.2.
#include <atomic>

#include "common_device.hpp"

namespace desco {

constexpr int GT = 128;            // rows (nodes) per tile
constexpr int GNT = 512;           // threads per block
constexpr int PLN = GT * 64;       // shorts per activation plane
constexpr int WPLN = 64 * 64;      // shorts per weight-block plane
constexpr int PCAP = 1216;         // neighbour records prefetched (up to three per thread) for the next tile: a 128-node tile
                                   // of MSRC-21 / IMDB-shaped graphs has ~900 (512 sent nearly every such tile through
                                   // the staged passes below: two barriers and two dependent global round trips each)
constexpr int ECAP = 1216;         // neighbour records staged per pass (aliases weight buffer 1: 20 B each)
constexpr int CST = 832;           // u, d1, tp, b3 (64 each), b5, w7 (256 each), zp_q (64)
constexpr size_t GOSSIP_LDS_BYTES = (size_t)2 * 3 * PLN * 2 + (size_t)2 * 3 * WPLN * 2 + GT * 16 +
                                    132 * 4 + CST * 4 + 2 * GT * 4 + GT + 16;
static_assert(ECAP * 20 <= 3 * WPLN * 2, "neighbour staging must fit in one weight buffer");
static_assert(GOSSIP_LDS_BYTES <= 160 * 1024, "gossip_fused: LDS budget exceeded");

struct GossipFusedArgs {
  const float4* scal;       // [N*Q] (a0, b0, a1, x)
  const int32_t* rowptr;
  const int32_t* col;
  int64_t num_nodes;
  int Q;
  const float* g1;          // [Q]
  const float* p;           // [Q,64]
  const float* z;           // [Q,64]
  const float* zp;          // [Q,64]
  const float* r;           // [64]
  const float* t;           // [64]
  const float* u;           // [64]  D1a c1
  const float* tp;          // [64]  P0[:,64:128] w_pre
  const float* d1;          // [64]
  const short* w1s;         // [3][64][128]  bf16 planes of the n-major (row n holds the 128 k) W1
  const short* wps;         // [3][64][128]
  const short* w3s;         // [3][64][64]
  const float* b3;          // [64]
  const short* w5s;         // [3][256][64]
  const float* b5;          // [256]
  const float* w7;          // [256]
  float b7;
  float* out;               // [N,Q]
  const uint8_t* tperm;     // [tiles*128] phase-1 slot -> row of the tile (desco_gossip_tile_order), or null
  unsigned long long* queue;  // {next ticket, finished blocks} of this launch (see gf_queue)
};

// scalars pre-pass: one HALF wave per node, lane & 31 = query (num_q <= 32; one wave per node above that).  A node is
// three dependent memory round trips (row pointer, column ids, neighbours' rows) and nothing else, so the launch's time
// is nodes in flight: two per wave, 0.46 -> 0.2x ms per 1.23 M nodes.
__global__ __launch_bounds__(256) void gossip_scalars_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             int64_t num_nodes, int Q,
                                                             const float* __restrict__ g0,
                                                             const float* __restrict__ g1,
                                                             float4* __restrict__ scal) {
  const int wave = threadIdx.x >> 6;
  const bool two = Q <= 32;
  const int lane = two ? (threadIdx.x & 31) : (threadIdx.x & 63);
  const int64_t i = two ? ((int64_t)blockIdx.x * 4 + wave) * 2 + ((threadIdx.x >> 5) & 1) : (int64_t)blockIdx.x * 4 + wave;
  if (i >= num_nodes) return;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  const int q = lane < Q ? lane : Q - 1;
  float slo = 0.f, shi = 0.f;
  int dlo = 0;
  // four neighbours per trip: their (dependent) index and value loads are in flight together;
  // a short tail re-reads the last neighbour with weight 0
  for (int e = e0; e < e1; e += 4) {
    const int last = e1 - 1;
    const int64_t j0 = col[e], j1 = col[e + 1 < e1 ? e + 1 : last];
    const int64_t j2 = col[e + 2 < e1 ? e + 2 : last], j3 = col[e + 3 < e1 ? e + 3 : last];
    const float x0 = x[j0 * ldx + q], x1 = x[j1 * ldx + q], x2 = x[j2 * ldx + q], x3 = x[j3 * ldx + q];
    const bool v1 = e + 1 < e1, v2 = e + 2 < e1, v3 = e + 3 < e1;
    slo += j0 < i ? x0 : 0.f;
    shi += j0 < i ? 0.f : x0;
    dlo += j0 < i ? 1 : 0;
    slo += (v1 && j1 < i) ? x1 : 0.f;
    shi += (v1 && !(j1 < i)) ? x1 : 0.f;
    dlo += (v1 && j1 < i) ? 1 : 0;
    slo += (v2 && j2 < i) ? x2 : 0.f;
    shi += (v2 && !(j2 < i)) ? x2 : 0.f;
    dlo += (v2 && j2 < i) ? 1 : 0;
    slo += (v3 && j3 < i) ? x3 : 0.f;
    shi += (v3 && !(j3 < i)) ? x3 : 0.f;
    dlo += (v3 && j3 < i) ? 1 : 0;
  }
  const float flo = (float)dlo, fhi = (float)(e1 - e0 - dlo);
  const float a = g0[q], b = g1[q];
  if (lane < Q)
    scal[i * Q + lane] = make_float4(a * flo + (1.f - a) * fhi, a * slo + (1.f - a) * shi,
                                     b * flo + (1.f - b) * fhi, x[i * ldx + q]);
}

// Phase-1 work order of a 128-node tile (desco_gossip_tile_order): in phase 1 of the fused kernel a half wave walks the
// neighbour list of one row, the two halves of a wave run in lock step (iterations = the longer list of the pair), a
// wave takes 8 pairs one after the other and the phase ends at a block barrier (time = the slowest wave).  With rows in
// node order that is 1.3-1.8x the balanced time (tools note in DESIGN.md section 8).  Here the tile's rows are sorted by
// degree, consecutive rows form a pair, and the pairs are dealt to the 8 waves in snake order; perm[16 w + 2 i + h] =
// the row that wave w handles as pair i, half h.  One block of 128 threads per tile, bitonic sort in LDS.
__global__ __launch_bounds__(128) void gossip_tile_order_kernel(const int32_t* __restrict__ rowptr, int64_t num_nodes,
                                                                uint8_t* __restrict__ perm) {
  __shared__ uint32_t key[128];
  const int t = threadIdx.x;
  const int64_t node = (int64_t)blockIdx.x * 128 + t;
  const uint32_t deg = node < num_nodes ? (uint32_t)(rowptr[node + 1] - rowptr[node]) : 0u;
  key[t] = ((0xffffffu - (deg < 0xffffffu ? deg : 0xffffffu)) << 8) | (uint32_t)t;   // ascending = high degree first
  __syncthreads();
  for (int k = 2; k <= 128; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int o = t ^ j;
      if (o > t) {
        const uint32_t a = key[t], b = key[o];
        const bool up = (t & k) == 0;
        if ((a > b) == up) {
          key[t] = b;
          key[o] = a;
        }
      }
      __syncthreads();
    }
  }
  const int pair = t >> 1, grp = pair >> 3, slot = pair & 7;
  const int w = (grp & 1) ? 7 - slot : slot;                  // snake: the 8 heaviest pairs go to waves 0..7, the next 8 to 7..0
  perm[(int64_t)blockIdx.x * 128 + 16 * w + 2 * grp + (t & 1)] = (uint8_t)(key[t] & 0xffu);
}

using bf16x8 = __attribute__((ext_vector_type(8))) short;

// element offset of (row, k) inside a [rows][64] bf16 plane with swizzled 16-byte chunks
__device__ __forceinline__ int gf_pidx(const int row, const int k) {
  return row * 64 + ((((k >> 3) ^ (row >> 1)) & 7) << 3) + (k & 7);
}

// weight block (64 n x 64 k, three planes): thread t moves 8 bf16 of row n = t>>3, chunk t&7, per plane
#define GF_WLOAD(base_, rows_total_, ldk_, nrow0_, koff_)                                          \
  {                                                                                                \
    const short* s_ = (base_) + (int64_t)((nrow0_) + (tid >> 3)) * (ldk_) + (koff_) + 8 * (tid & 7); \
    q0 = *reinterpret_cast<const uint4*>(s_);                                                      \
    q1 = *reinterpret_cast<const uint4*>(s_ + (int64_t)(rows_total_) * (ldk_));                    \
    q2 = *reinterpret_cast<const uint4*>(s_ + (int64_t)2 * (rows_total_) * (ldk_));                \
  }
#define GF_WSTORE(wb_)                                                                 \
  {                                                                                    \
    short* d_ = (wb_) + (tid >> 3) * 64 + ((((tid & 7) ^ (tid >> 4)) & 7) << 3);       \
    *reinterpret_cast<uint4*>(d_) = q0;                                                \
    *reinterpret_cast<uint4*>(d_ + WPLN) = q1;                                         \
    *reinterpret_cast<uint4*>(d_ + 2 * WPLN) = q2;                                     \
  }
// MFMA shape: v_mfma_f32_16x16x32_bf16.  At equal cycles per flop the chip holds a 12-14 % higher clock
// under it than under v_mfma_f32_32x32x16_bf16 (tools/micro/mfma_peak.hip: 2.2 vs 1.94 PFLOP/s on random
// data), and the swizzled plane layout is conflict-free for its fragment reads as it stands.
// A wave's 32 features x 32 nodes of a 64-wide block are 2 x 2 tiles of 16 x 16: acc_ij, i = feature
// tile, j = node tile.  Lane (r16 = lane & 15, q4 = lane >> 4) holds, per 32-deep k step t and plane,
// W[32 wn + 16 i + r16][32 t + 8 q4 + 0..7] and X[32 wm + 16 j + r16][32 t + 8 q4 + 0..7] (16-byte
// fragments), and after the MFMAs D^T[feature 32 wn + 16 i + 4 q4 + e][node 32 wm + 16 j + r16].
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define GF_M16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0);
// the six products (smallest terms first) of one weight tile w_ with both node tiles of x_
#define GF_MM(c0_, c1_, w_, x_)                                       \
  GF_M16(w_##l, x_##0h, c0_) GF_M16(w_##l, x_##1h, c1_)               \
  GF_M16(w_##h, x_##0l, c0_) GF_M16(w_##h, x_##1l, c1_)               \
  GF_M16(w_##m, x_##0m, c0_) GF_M16(w_##m, x_##1m, c1_)               \
  GF_M16(w_##m, x_##0h, c0_) GF_M16(w_##m, x_##1h, c1_)               \
  GF_M16(w_##h, x_##0m, c0_) GF_M16(w_##h, x_##1m, c1_)               \
  GF_M16(w_##h, x_##0h, c0_) GF_M16(w_##h, x_##1h, c1_)
// X fragments of k step t_ (both node tiles, three planes) into register set s_
#define GF_LDX(s_, xa_, t_)                                                                   \
  {                                                                                           \
    const int c_ = (((4 * (t_) + q4) ^ swz) & 7) << 3;                                        \
    s_##0h = *reinterpret_cast<const bf16x8*>((xa_) + c_);                                    \
    s_##0m = *reinterpret_cast<const bf16x8*>((xa_) + PLN + c_);                              \
    s_##0l = *reinterpret_cast<const bf16x8*>((xa_) + 2 * PLN + c_);                          \
    s_##1h = *reinterpret_cast<const bf16x8*>((xa_) + 16 * 64 + c_);                          \
    s_##1m = *reinterpret_cast<const bf16x8*>((xa_) + 16 * 64 + PLN + c_);                    \
    s_##1l = *reinterpret_cast<const bf16x8*>((xa_) + 16 * 64 + 2 * PLN + c_);                \
  }
// W fragments of feature tile i_, k step t_ into register set s_
#define GF_LDW(s_, wa_, i_, t_)                                                               \
  {                                                                                           \
    const int c_ = (((4 * (t_) + q4) ^ swz) & 7) << 3;                                        \
    s_##h = *reinterpret_cast<const bf16x8*>((wa_) + (i_) * 16 * 64 + c_);                    \
    s_##m = *reinterpret_cast<const bf16x8*>((wa_) + (i_) * 16 * 64 + WPLN + c_);             \
    s_##l = *reinterpret_cast<const bf16x8*>((wa_) + (i_) * 16 * 64 + 2 * WPLN + c_);         \
  }
// 48 MFMAs of one 64x64 weight block: D^T[n][node] += sum_k W[n][k] X[node][k].  The fragments of
// the next (feature tile, k step) are read while the MFMAs of the current one run.
#define GF_MFMA_BLOCK(wb_, img_)                                                                   \
  {                                                                                                \
    const short* wa_ = (wb_) + wrow * 64;                                                          \
    const short* xa_ = (img_) + xrow * 64;                                                         \
    bf16x8 fxa0h = {}, fxa0m = {}, fxa0l = {}, fxa1h = {}, fxa1m = {}, fxa1l = {}, fxb0h = {}, fxb0m = {}, fxb0l = {}, fxb1h = {}, fxb1m = {}, fxb1l = {};     \
    bf16x8 fwah = {}, fwam = {}, fwal = {}, fwbh = {}, fwbm = {}, fwbl = {};                                                     \
    GF_LDX(fxa, xa_, 0)                                                                            \
    GF_LDW(fwa, wa_, 0, 0)                                                                         \
    GF_LDW(fwb, wa_, 1, 0)                                                                         \
    GF_MM(acc00, acc01, fwa, fxa)                                                                  \
    GF_LDW(fwa, wa_, 0, 1)                                                                         \
    GF_LDX(fxb, xa_, 1)                                                                            \
    GF_MM(acc10, acc11, fwb, fxa)                                                                  \
    GF_LDW(fwb, wa_, 1, 1)                                                                         \
    GF_MM(acc00, acc01, fwa, fxb)                                                                  \
    GF_MM(acc10, acc11, fwb, fxb)                                                                  \
  }
// the four accumulators seeded with a per-feature vector v_ (+ a per-node scalar times a vector)
#define GF_SEED1(v_)                                                                       \
  {                                                                                        \
    const float4 a_ = *reinterpret_cast<const float4*>((v_) + fq_e);                       \
    const float4 b_ = *reinterpret_cast<const float4*>((v_) + fq_e + 16);                  \
    acc00 = f32x4{a_.x, a_.y, a_.z, a_.w};                                                 \
    acc01 = acc00;                                                                         \
    acc10 = f32x4{b_.x, b_.y, b_.z, b_.w};                                                 \
    acc11 = acc10;                                                                         \
  }
#define GF_SEED2(s0_, s1_, u_, v_)                                                         \
  {                                                                                        \
    const float4 ua_ = *reinterpret_cast<const float4*>((u_) + fq_e);                      \
    const float4 ub_ = *reinterpret_cast<const float4*>((u_) + fq_e + 16);                 \
    const float4 va_ = *reinterpret_cast<const float4*>((v_) + fq_e);                      \
    const float4 vb_ = *reinterpret_cast<const float4*>((v_) + fq_e + 16);                 \
    acc00 = f32x4{(s0_) * ua_.x + va_.x, (s0_) * ua_.y + va_.y, (s0_) * ua_.z + va_.z, (s0_) * ua_.w + va_.w}; \
    acc01 = f32x4{(s1_) * ua_.x + va_.x, (s1_) * ua_.y + va_.y, (s1_) * ua_.z + va_.z, (s1_) * ua_.w + va_.w}; \
    acc10 = f32x4{(s0_) * ub_.x + vb_.x, (s0_) * ub_.y + vb_.y, (s0_) * ub_.z + vb_.z, (s0_) * ub_.w + vb_.w}; \
    acc11 = f32x4{(s1_) * ub_.x + vb_.x, (s1_) * ub_.y + vb_.y, (s1_) * ub_.z + vb_.z, (s1_) * ub_.w + vb_.w}; \
  }
#define GF_SPLIT(a_, b_, h_, m_, l_) split2_bf16x3(a_, b_, h_, m_, l_)
// write 4 consecutive features (fb_ .. fb_+3) of node node_ as bf16 planes into image img_
#define GF_PUT4(img_, node_, fb_, v0_, v1_, v2_, v3_)                                    \
  {                                                                                      \
    uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                               \
    GF_SPLIT(v0_, v1_, h0_, m0_, l0_);                                                       \
    GF_SPLIT(v2_, v3_, h1_, m1_, l1_);                                                       \
    short* d_ = (img_) + gf_pidx((node_), (fb_));                                          \
    *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                                \
    *reinterpret_cast<uint2*>(d_ + PLN) = make_uint2(m0_, m1_);                          \
    *reinterpret_cast<uint2*>(d_ + 2 * PLN) = make_uint2(l0_, l1_);                      \
  }
// epilogue of a 64-wide block: activation + bf16 planes of the four accumulator tiles into img_
#define GF_EPI(img_, ACT_)                                                               \
  {                                                                                      \
    GF_PUT4(img_, xrow_e, fq_e, ACT_(acc00[0]), ACT_(acc00[1]), ACT_(acc00[2]), ACT_(acc00[3]))            \
    GF_PUT4(img_, xrow_e + 16, fq_e, ACT_(acc01[0]), ACT_(acc01[1]), ACT_(acc01[2]), ACT_(acc01[3]))       \
    GF_PUT4(img_, xrow_e, fq_e + 16, ACT_(acc10[0]), ACT_(acc10[1]), ACT_(acc10[2]), ACT_(acc10[3]))       \
    GF_PUT4(img_, xrow_e + 16, fq_e + 16, ACT_(acc11[0]), ACT_(acc11[1]), ACT_(acc11[2]), ACT_(acc11[3]))  \
  }
#define GF_RELU(v_) ((v_) > 0.f ? (v_) : 0.f)
#define GF_LEAKY01(v_) ((v_) > 0.f ? (v_) : 0.1f * (v_))

// Work queue: a block's first item is its block index, every further one a ticket (grid + atomicAdd(queue[0], 1)): the
// static map "item += grid" let the slowest block finish 2-5 % after the mean (profiles/r3_i_tail_probe_*.log).  The
// counters are self-cleaning -- the block that finishes last zeroes them -- so a launch needs no memset and a captured
// launch can be replayed; launches take the 64 slots in turn (at most 64 launches in flight per device).
constexpr int GF_QSLOTS = 64;
__device__ unsigned long long gf_queue[GF_QSLOTS][2];
__global__ __launch_bounds__(GNT) void gossip_fused_kernel(GossipFusedArgs g, int64_t num_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint4 gf_lds[];
  short* I0 = reinterpret_cast<short*>(gf_lds);          // h1, later y1
  short* I1 = I0 + 3 * PLN;                              // hh, later h2, later y2
  short* WB0 = I1 + 3 * PLN;                             // weight blocks 0, 2, 4, 6, 8
  short* WB1 = WB0 + 3 * WPLN;                           // weight blocks 1, 3, 5, 7; neighbour staging
  float4* srow = reinterpret_cast<float4*>(WB1 + 3 * WPLN);   // scalars of the tile rows
  int* rp = reinterpret_cast<int*>(srow + GT);                // rowptr[n0 .. n0+128]
  float* cst = reinterpret_cast<float*>(rp + 132);            // bias vectors (see CST)
  float* red = cst + CST;                                     // [2][128] head partials
  uint8_t* tperm = reinterpret_cast<uint8_t*>(red + 2 * GT);  // [128] phase-1 slot -> row
  unsigned long long* tick = reinterpret_cast<unsigned long long*>(tperm + GT);   // the item after the current one
  int* ecol = reinterpret_cast<int*>(WB1);                    // [ECAP]
  float4* escal = reinterpret_cast<float4*>(ecol + ECAP);     // [ECAP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int q4 = lane >> 4;                 // k chunk within a 32-deep step / feature quad of a 16x16 tile
  const int xrow = 32 * wm + (lane & 15);   // this lane's node of node tile 0 (+16: tile 1) (B operand, epilogues)
  const int wrow = 32 * wn + (lane & 15);   // this lane's weight row of feature tile 0 (+16: tile 1) (A operand)
  const int swz = (lane >> 1) & 7;          // chunk swizzle of all four rows: (row >> 1) & 7 with row = 16 j + (lane & 15)
  const int fq = 32 * wn + 4 * q4;          // first output feature of feature tile 0 (+16: tile 1)
  const int Q = g.Q;
  const int64_t nitems = num_tiles * Q;           // item = tile * Q + q: neighbours share a tile
  int64_t item = blockIdx.x;
  if (item >= nitems) return;

  for (int i = tid; i < 64; i += GNT) {
    cst[i] = g.u[i];
    cst[64 + i] = g.d1[i];
    cst[128 + i] = g.tp[i];
    cst[192 + i] = g.b3[i];
  }
  for (int i = tid; i < 256; i += GNT) {
    cst[256 + i] = g.b5[i];
    cst[512 + i] = g.w7[i];
  }

  // The (rowptr -> col -> scalar record) chain of the NEXT item is fetched into these registers
  // in three stages spread over the current item's GEMMs and published to LDS when the current
  // item is done: a tile does not start with three dependent global-memory latencies.
  float4 n_srow = make_float4(0.f, 0.f, 0.f, 0.f), n_scal = n_srow, n_scal2 = n_srow, n_scal3 = n_srow;
  int n_rp = 0, n_col = 0, n_col2 = 0, n_col3 = 0, n_ebeg = 0, n_cnt = 0;
  uint32_t n_perm = 0x03020100u + 0x04040404u * (uint32_t)(tid & 31);   // identity slots 4 tid .. 4 tid + 3
  float n_zp = 0.f, n_gq = 0.f;            // the next item's per-query vectors (this lane's slice)
  float2 n_pc = make_float2(0.f, 0.f), n_zc = n_pc;
  const int f0 = 2 * (lane & 31);          // phase-1 lane map: features (f0, f0+1)
#define GF_STAGE1(it_)                                                                     \
  {                                                                                        \
    const int q_ = (int)((it_) % Q);                                                       \
    const int64_t t0_ = ((it_) / Q) * GT;                                                  \
    const int nr_ = (int)((g.num_nodes - t0_) < GT ? (g.num_nodes - t0_) : GT);            \
    if (tid < GT) n_srow = g.scal[(t0_ + (tid < nr_ ? tid : nr_ - 1)) * Q + q_];           \
    if (tid <= GT) n_rp = g.rowptr[t0_ + (tid < nr_ ? tid : nr_)];                         \
    if (g.tperm && tid < GT / 4) n_perm = reinterpret_cast<const uint32_t*>(g.tperm + ((it_) / Q) * GT)[tid]; \
    n_ebeg = g.rowptr[t0_];                                                                \
    n_cnt = g.rowptr[t0_ + nr_] - n_ebeg;                                                  \
    n_cnt = n_cnt < PCAP ? n_cnt : PCAP;                                                   \
    if (tid < 64) n_zp = g.zp[q_ * 64 + tid];                                              \
    n_gq = g.g1[q_];                                                                       \
    n_pc = *reinterpret_cast<const float2*>(g.p + q_ * 64 + f0);                           \
    n_zc = *reinterpret_cast<const float2*>(g.z + q_ * 64 + f0);                           \
  }
#define GF_STAGE2() \
  {                                                          \
    if (tid < n_cnt) n_col = g.col[n_ebeg + tid];            \
    if (tid + GNT < n_cnt) n_col2 = g.col[n_ebeg + tid + GNT]; \
    if (tid + 2 * GNT < n_cnt) n_col3 = g.col[n_ebeg + tid + 2 * GNT]; \
  }
#define GF_STAGE3(it_) \
  {                                                                                          \
    if (tid < n_cnt) n_scal = g.scal[(int64_t)n_col * Q + (int)((it_) % Q)];                 \
    if (tid + GNT < n_cnt) n_scal2 = g.scal[(int64_t)n_col2 * Q + (int)((it_) % Q)];         \
    if (tid + 2 * GNT < n_cnt) n_scal3 = g.scal[(int64_t)n_col3 * Q + (int)((it_) % Q)];     \
  }

  if (tid == 0) *tick = gridDim.x + atomicAdd(g.queue, 1ull);
  GF_STAGE1(item)
  GF_STAGE2()
  GF_STAGE3(item)
  // phase-1 lane map: two adjacent features (f0, f0+1) of rows wave*16 + 2*i + (lane>>5), i = 0..7
  const float2 rc = *reinterpret_cast<const float2*>(g.r + f0);
  const float2 tc = *reinterpret_cast<const float2*>(g.t + f0);
  uint4 q0, q1, q2;                       // weight block in flight (one 16-byte chunk per plane)
  GF_WLOAD(g.w1s, 64, 128, 0, 0)          // block 0 of the first item

  for (;;) {
    // ---- publish the prefetched tile data ------------------------------------------------------
    const int q = (int)(item % Q);
    if (tid < GT) srow[tid] = n_srow;
    if (tid <= GT) rp[tid] = n_rp;
    if (tid < GT / 4) reinterpret_cast<uint32_t*>(tperm)[tid] = n_perm;
    if (tid < n_cnt) {
      ecol[tid] = n_col;
      escal[tid] = n_scal;
      if (tid + GNT < n_cnt) {
        ecol[tid + GNT] = n_col2;
        escal[tid + GNT] = n_scal2;
      }
      if (tid + 2 * GNT < n_cnt) {
        ecol[tid + 2 * GNT] = n_col3;
        escal[tid + 2 * GNT] = n_scal3;
      }
    }
    if (tid < 64) cst[768 + tid] = n_zp;
    const float gq = n_gq;                 // (a tile does not start with a global-memory round trip)
    const float2 pc = n_pc, zc = n_zc;
    const int cnt0 = n_cnt;
    const int64_t n0 = (item / Q) * GT;
    const int nrows = (int)((g.num_nodes - n0) < GT ? (g.num_nodes - n0) : GT);
    GF_WSTORE(WB0)                         // block 0 (W1, k 0..63: multiplies hh)
    __syncthreads();
    const int64_t next = (int64_t)*tick;   // (written before the previous barrier)
    const bool has_next = next < nitems;
    unsigned long long tk = 0;             // ticket of the item after `next`: in flight until the end of this item
    if (tid == 0 && has_next) tk = gridDim.x + atomicAdd(g.queue, 1ull);
    if (has_next) GF_STAGE1(next)
    GF_WLOAD(g.w1s, 64, 128, 0, 64)        // block 1 in flight while the tile is being assembled

    // ---- phase 1: h1 of the tile rows, gated neighbour sum hh -----------------------------------
    {
      // (the asm keeps the 40-odd row / plane addresses of this phase from being hoisted out of the
      // item loop, where they would sit in registers through the GEMMs)
      int lane1 = lane;
      asm volatile("" : "+v"(lane1));
      const int half1 = lane1 >> 5, f1 = 2 * (lane1 & 31);
      float2 hh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) hh[i] = make_float2(0.f, 0.f);
      const int ebeg = rp[0], eend = rp[GT];
      // pass 0: the prefetched records [ebeg, ebeg+cnt0); later passes (tiles with more than PCAP
      // neighbour records) stage ECAP records at a time from global memory
      int base = ebeg, cnt = cnt0;
      for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = tperm[wave * 16 + 2 * i + half1];
          const int node = (int)n0 + row;
          int lo = rp[row] - base, hi = rp[row + 1] - base;
          lo = lo < 0 ? 0 : lo;
          hi = hi > cnt ? cnt : hi;
          float2 a = hh[i];
          for (int e = lo; e < hi; ++e) {
            const float4 sj = escal[e];
            // (three dependent FMAs, z first: also keeps hipcc from selecting a packed multiply that takes the record's
            //  odd dword through OP_SEL on src1 -- rule PK-OPSEL of tools/check_isa.py)
            float hx = __builtin_fmaf(sj.x, pc.x, __builtin_fmaf(sj.y, rc.x, __builtin_fmaf(sj.w, tc.x, zc.x)));
            float hy = __builtin_fmaf(sj.x, pc.y, __builtin_fmaf(sj.y, rc.y, __builtin_fmaf(sj.w, tc.y, zc.y)));
            hx = hx > 0.f ? hx : 0.f;
            hy = hy > 0.f ? hy : 0.f;
            const float gt = ecol[e] < node ? gq : 1.f - gq;
            a.x += gt * hx;
            a.y += gt * hy;
          }
          hh[i] = a;
        }
        base += cnt;
        if (base >= eend) break;
        __syncthreads();          // everyone is done with the staged records
        cnt = (eend - base) < ECAP ? (eend - base) : ECAP;
        for (int e = tid; e < cnt; e += GNT) {
          const int j = g.col[base + e];
          ecol[e] = j;
          escal[e] = g.scal[(int64_t)j * Q + q];
        }
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = tperm[wave * 16 + 2 * i + half1];
        const float4 si = srow[row];
        float hx = __builtin_fmaf(si.x, pc.x, __builtin_fmaf(si.y, rc.x, __builtin_fmaf(si.w, tc.x, zc.x)));
        float hy = __builtin_fmaf(si.x, pc.y, __builtin_fmaf(si.y, rc.y, __builtin_fmaf(si.w, tc.y, zc.y)));
        hx = hx > 0.f ? hx : 0.f;
        hy = hy > 0.f ? hy : 0.f;
        uint32_t h_, m_, l_;
        split2_bf16x3(hx, hy, h_, m_, l_);
        const int o = gf_pidx(row, f1);
        *reinterpret_cast<uint32_t*>(I0 + o) = h_;
        *reinterpret_cast<uint32_t*>(I0 + PLN + o) = m_;
        *reinterpret_cast<uint32_t*>(I0 + 2 * PLN + o) = l_;
        split2_bf16x3(hh[i].x, hh[i].y, h_, m_, l_);
        *reinterpret_cast<uint32_t*>(I1 + o) = h_;
        *reinterpret_cast<uint32_t*>(I1 + PLN + o) = m_;
        *reinterpret_cast<uint32_t*>(I1 + 2 * PLN + o) = l_;
      }
    }
    __syncthreads();

    f32x4 acc00, acc01, acc10, acc11;      // [feature tile][node tile]
    const float4 sx0 = srow[xrow], sx1 = srow[xrow + 16];   // (a0, b0, a1, x) of this lane's two nodes
    int xrow_e = xrow, fq_e = fq;          // epilogue addressing, not hoisted out of the item loop
    asm volatile("" : "+v"(xrow_e), "+v"(fq_e));
    // ---- blocks 0, 1: h2 = relu([hh|h1] W1 + a1*u + d1) -> I1 -----------------------------------
    // (the affine terms seed the accumulators, so an epilogue is activation + split only)
    GF_SEED2(sx0.z, sx1.z, cst, cst + 64)
    GF_MFMA_BLOCK(WB0, I1)
    GF_WSTORE(WB1)                         // block 1 (staging is dead)
    GF_WLOAD(g.wps, 64, 128, 0, 0)
    __syncthreads();
    if (has_next) GF_STAGE2()
    GF_MFMA_BLOCK(WB1, I0)
    GF_WSTORE(WB0)                         // block 2
    GF_WLOAD(g.wps, 64, 128, 0, 64)
    GF_EPI(I1, GF_RELU)
    __syncthreads();
    // ---- blocks 2, 3: y1 = leaky([h1|h2] Wp + x*tp + zp_q, 0.1) -> I0 ---------------------------
    GF_SEED2(sx0.w, sx1.w, cst + 128, cst + 768)
    GF_MFMA_BLOCK(WB0, I0)
    GF_WSTORE(WB1)                         // block 3
    GF_WLOAD(g.w3s, 64, 64, 0, 0)
    __syncthreads();
    if (has_next) GF_STAGE3(next)
    GF_MFMA_BLOCK(WB1, I1)
    GF_WSTORE(WB0)                         // block 4
    GF_WLOAD(g.w5s, 256, 64, 0, 0)
    GF_EPI(I0, GF_LEAKY01)
    __syncthreads();
    // ---- block 4: y2 = relu(y1 W3 + b3) -> I1 ------------------------------------------------------
    GF_SEED1(cst + 192)
    GF_MFMA_BLOCK(WB0, I0)
    GF_WSTORE(WB1)                         // block 5 (W5 column group 0)
    GF_WLOAD(g.w5s, 256, 64, 64, 0)
    GF_EPI(I1, GF_RELU)
    __syncthreads();
    // ---- blocks 5..8: head partial  sum_c relu(y2 W5 + b5)[c] * w7[c], 4 column groups of 64 -------
    float part0 = 0.f, part1 = 0.f;        // this lane's two nodes
#define GF_HEAD(cg_)                                                                        \
  {                                                                                         \
    const float4 wa_ = *reinterpret_cast<const float4*>(cst + 512 + 64 * (cg_) + fq_e);      \
    const float4 wb_ = *reinterpret_cast<const float4*>(cst + 512 + 64 * (cg_) + fq_e + 16); \
    part0 += GF_RELU(acc00[0]) * wa_.x + GF_RELU(acc00[1]) * wa_.y + GF_RELU(acc00[2]) * wa_.z + \
             GF_RELU(acc00[3]) * wa_.w + GF_RELU(acc10[0]) * wb_.x + GF_RELU(acc10[1]) * wb_.y + \
             GF_RELU(acc10[2]) * wb_.z + GF_RELU(acc10[3]) * wb_.w;                           \
    part1 += GF_RELU(acc01[0]) * wa_.x + GF_RELU(acc01[1]) * wa_.y + GF_RELU(acc01[2]) * wa_.z + \
             GF_RELU(acc01[3]) * wa_.w + GF_RELU(acc11[0]) * wb_.x + GF_RELU(acc11[1]) * wb_.y + \
             GF_RELU(acc11[2]) * wb_.z + GF_RELU(acc11[3]) * wb_.w;                           \
  }
    GF_SEED1(cst + 256)
    GF_MFMA_BLOCK(WB1, I1)
    GF_WSTORE(WB0)                         // block 6
    GF_WLOAD(g.w5s, 256, 64, 128, 0)
    GF_HEAD(0)
    __syncthreads();
    GF_SEED1(cst + 256 + 64)
    GF_MFMA_BLOCK(WB0, I1)
    GF_WSTORE(WB1)                         // block 7
    GF_WLOAD(g.w5s, 256, 64, 192, 0)
    GF_HEAD(1)
    __syncthreads();
    GF_SEED1(cst + 256 + 128)
    GF_MFMA_BLOCK(WB1, I1)
    GF_WSTORE(WB0)                         // block 8
    GF_WLOAD(g.w1s, 64, 128, 0, 0)         // block 0 of the next item
    GF_HEAD(2)
    __syncthreads();
    GF_SEED1(cst + 256 + 192)
    GF_MFMA_BLOCK(WB0, I1)
    GF_HEAD(3)
#undef GF_HEAD
    // fold the four lane quarters (features 4 q4 .. 4 q4 + 3 of every tile), then the two feature groups wn
    part0 += __shfl_xor(part0, 16, 64);
    part1 += __shfl_xor(part1, 16, 64);
    part0 += __shfl_xor(part0, 32, 64);
    part1 += __shfl_xor(part1, 32, 64);
    if (lane < 16) {
      red[wn * GT + xrow] = part0;
      red[wn * GT + xrow + 16] = part1;
    }
    __syncthreads();
    if (tid < nrows) g.out[(n0 + tid) * Q + q] = red[tid] + red[GT + tid] + g.b7 + srow[tid].w;
    if (!has_next) break;
    item = next;
    if (tid == 0) *tick = tk;
    // No barrier here: what the next item's publish overwrites is either read by the SAME thread above (srow[tid]) or was
    // last read before the barrier in front of block 8 (rp, tperm, the staging area in WB1, zp_q) or before the one above
    // (WB0: block 8); red is not written again before ten more barriers, and the ticket is read behind the next one.
#ifdef GF_END_BARRIER
    __syncthreads();
#endif
  }
  if (tid == 0 && atomicAdd(g.queue + 1, 1ull) == gridDim.x - 1) {      // last block out: leave the slot clean
    g.queue[0] = 0;
    g.queue[1] = 0;
  }
#undef GF_STAGE1
#undef GF_STAGE2
#undef GF_STAGE3
}

#undef GF_WLOAD
#undef GF_WSTORE
#undef GF_M16
#undef GF_MM
#undef GF_LDX
#undef GF_LDW
#undef GF_MFMA_BLOCK
#undef GF_SEED1
#undef GF_SEED2
#undef GF_PUT4
#undef GF_EPI
#undef GF_RELU
#undef GF_LEAKY01

}  // namespace desco

using namespace desco;

extern "C" int desco_gossip_scalars_f32(const float* x, int64_t ldx, const int32_t* rowptr,
                                        const int32_t* col, int64_t num_nodes, int num_q,
                                        const float* g0, const float* g1, float* scal4,
                                        desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  if (!x || !rowptr || !g0 || !g1 || !scal4 || num_nodes < 0 || num_q < 1 || num_q > 64 ||
      (reinterpret_cast<uintptr_t>(scal4) & 15))
    return fail(DESCO_EINVAL, "desco_gossip_scalars_f32: bad argument (1 <= num_q <= 64)");
  const int64_t per_block = num_q <= 32 ? 8 : 4;          // (a half wave per node when the queries fit one)
  const int64_t blocks = (num_nodes + per_block - 1) / per_block;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_scalars_f32: too many nodes");
  hipLaunchKernelGGL(gossip_scalars_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, rowptr, col, num_nodes, num_q, g0, g1,
                     reinterpret_cast<float4*>(scal4));
  return launch_status("desco_gossip_scalars_f32");
}

extern "C" int desco_gossip_tile_order(const int32_t* rowptr, int64_t num_nodes, uint8_t* perm,
                                       desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  if (!rowptr || !perm || num_nodes < 0) return fail(DESCO_EINVAL, "desco_gossip_tile_order: bad argument");
  const int64_t tiles = (num_nodes + GT - 1) / GT;
  if (tiles > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_tile_order: too many nodes");
  hipLaunchKernelGGL(gossip_tile_order_kernel, dim3((unsigned)tiles), dim3(128), 0, (hipStream_t)stream, rowptr,
                     num_nodes, perm);
  return launch_status("desco_gossip_tile_order");
}

extern "C" int desco_gossip_fused_f32(const float* scal4, const int32_t* rowptr, const int32_t* col,
                                      int64_t num_nodes, int num_q, const float* g1, const float* p,
                                      const float* z, const float* zp, const float* r,
                                      const float* t, const float* u, const float* tp,
                                      const float* d1, const int16_t* w1_planes,
                                      const int16_t* wp_planes, const int16_t* w3_planes,
                                      const float* b3, const int16_t* w5_planes, const float* b5,
                                      const float* w7, float b7, float* out, const uint8_t* tile_perm,
                                      desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  auto mis16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) != 0; };
  auto mis8 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 7) != 0; };
  if (!scal4 || !rowptr || !g1 || !p || !z || !zp || !r || !t || !u || !tp || !d1 || !w1_planes ||
      !wp_planes || !w3_planes || !b3 || !w5_planes || !b5 || !w7 || !out || num_nodes < 0 ||
      num_q < 1 || num_q > 65535 || mis16(scal4) || mis16(w1_planes) || mis16(wp_planes) ||
      mis16(w3_planes) || mis16(w5_planes) || mis8(p) || mis8(z) || mis8(r) || mis8(t) ||
      (reinterpret_cast<uintptr_t>(tile_perm) & 3))
    return fail(DESCO_EINVAL, "desco_gossip_fused_f32: bad argument");
  const int64_t bx = (num_nodes + GT - 1) / GT;
  if (bx > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_fused_f32: too many nodes");
  GossipFusedArgs a{reinterpret_cast<const float4*>(scal4), rowptr, col, num_nodes, num_q, g1, p, z,
                    zp, r, t, u, tp, d1,
                    reinterpret_cast<const short*>(w1_planes), reinterpret_cast<const short*>(wp_planes),
                    reinterpret_cast<const short*>(w3_planes), b3,
                    reinterpret_cast<const short*>(w5_planes), b5, w7, b7, out, tile_perm, nullptr};
  {
    static std::atomic<unsigned> seq{0};
    unsigned long long* base = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(desco::gf_queue));
    if (e != hipSuccess || !base) return fail((int)e, "desco_gossip_fused_f32: no work-queue symbol");
    a.queue = base + 2 * (seq.fetch_add(1) % GF_QSLOTS);
  }
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gossip_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)GOSSIP_LDS_BYTES);
    if (e != hipSuccess) return fail((int)e, "desco_gossip_fused_f32: cannot size LDS");
    attr_once.mark();
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  const int64_t nitems = bx * num_q;
  const unsigned grid = (unsigned)(nitems < (int64_t)cus ? nitems : (int64_t)cus);
  hipLaunchKernelGGL(gossip_fused_kernel, dim3(grid), dim3(GNT), GOSSIP_LDS_BYTES, (hipStream_t)stream,
                     a, bx);
  return launch_status("desco_gossip_fused_f32");
}

