// Fused gossip stage for gfx950, three-product fp16 form ("f16x3", common_device.hpp) -- the product path's gossip
// kernel since round 4.  Same algebra and contract as gossip_fused.hip (reference: BaseGNN gossip path
// gnn_model.py:58-103, 230-260, 303-350 looped over queries in lightning_model.py:613-628):
//
//   h1   = relu(a0*p_q + b0*r + x*t + z_q)                           (layer 0, closed form)
//   hh   = sum_j (j<i ? g1 : 1-g1) * h1_j                            (h1_j recomputed from j's scalar record)
//   h2   = relu([hh|h1] W1 + a1*u + d1)                              (layer 1, K=128)      MFMA
//   y1   = leaky([h1|h2] Wp + x*tp + zp_q, 0.1)                      (post_mp.0, K=128)    MFMA
//   y2   = relu(y1 W3 + b3)                                          (post_mp.3, K=64)     MFMA
//   out  = x + b7 + sum_c relu(y2 W5 + b5)[c] * w7[c]                (post_mp.5/.7, N=256) MFMA
//
// What changed against the six-product bf16 kernel (gossip_fused.hip, kept as the cross-check), and why:
//   * Arithmetic: x s = hi + lo in fp16 (22 bits), hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16: 3 instead of 6
//     MFMAs per tile step, 2 instead of 3 operand planes, 3 instead of 9 VALU per split pair.  fp16's 5-bit exponent is
//     carried by power-of-two scales: one per weight matrix, one PER NODE per activation vector (largest |v| of the
//     node's 64 features -> [2^14, 2^15)), undone exactly in the next epilogue.
//   * A wave owns 16 NODES x all 64 features (4 feature tiles x 1 node tile): the per-node maximum is a wave-local
//     reduction (two lane-quarter swaps), and in the transposed MFMA form (A = weight rows, B = activation rows) the C/D
//     layout of one GEMM -- lane = node, registers = features 16 i + 4 q + e -- IS the B layout of the next one under a
//     fixed permutation of k (baked into the weight stream, desco_gossip_f16_stream): h2, y1 and y2 never leave the
//     registers.
//   * WAVE-AUTONOMOUS.  With two planes ALL nine 64 x 64 weight blocks fit in LDS at once (147 456 B), and the neighbour
//     sums are computed directly in the B layout of the first GEMM (lane = node, 16 features per lane), so nothing is
//     shared between the waves of a workgroup but those read-only weights: no activation images, no weight ring, NO
//     barrier in the work loop.  Every wave carries its own 16 nodes through the whole network for a chunk of 5 queries
//     (work unit, drawn per wave from the caller's queue), and the eight waves of a CU drift apart freely -- one wave's
//     neighbour sums, epilogues and record loads run under the other waves' MFMAs.  The first version of this file kept
//     the bf16 kernel's block structure (128-node items, lock-stepped waves, weights through a ring of four buffers, six
//     barriers per item): 1.53 ms per 3.95 M (node, query) rows against 0.98 for this form on the same box.
//   * Everything a query needs from memory (its records, p_q / z_q / zp_q, the first four neighbour records of every
//     node) is requested one query ahead, in front of the previous query's GEMM chain.
//   * Round 5: the weight fragments travel through a ring of four register sets, three (tile, k step) ahead of the
//     MFMAs that use them, across block boundaries, epilogues and queries (the 72 steps of a query repeat for the next
//     one): round 4's block issued each fragment read one step ahead and waited out the LDS latency in front of every
//     MFMA triple.  Packed fp32 selection is back on and the end-of-block accumulator drain is gone: the wrong-result
//     mode of round 4 was one operand selection of the packed instructions (OP_SEL on src1/src2), which this source
//     does not produce and tools/check_isa.py refuses in the linked library (profiles/r5_a_gossip_f16_hazard.md).
#include "tu_no_packed_f32_begin.hpp"
#include "common_device.hpp"

namespace desco {

namespace gf16 {

constexpr int GT = 128;            // nodes per tile of the optional degree order (desco_gossip_tile_order)
constexpr int GNT = 512;           // threads per block: 8 waves, one workgroup per CU
constexpr int WPL = 64 * 64;       // halves per weight-block plane
constexpr int WBLK = 2 * WPL;      // halves per weight block (hi, lo)

struct Args {
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
  const short* wstream;     // [9][2][64*64] fp16: the nine 64x64 weight blocks in LDS image order (desco_gossip_f16_stream)
  const float* winv;        // [4] 1 / scale of W1, Wp, W3, W5
  const float* b3;          // [64]
  const float* b5;          // [256]
  const float* w7;          // [256]
  float b7;
  float* out;               // [N,Q]
  const uint8_t* tperm;     // [tiles*128] phase-1 slot -> row of the tile, or null
  unsigned long long* queue;  // {next ticket, finished blocks}: zero before the first launch that uses it, left zero
  unsigned grid_blocks;     // gridDim.x (set by the launcher)
};

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// max / sum over the four lanes (r, r+16, r+32, r+48) that share a node in the MFMA layouts
__device__ __forceinline__ float quarters_max(const float f) {      // f >= 0
  uint32_t v = __float_as_uint(f);
  u32x2 w = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = w[0] > w[1] ? w[0] : w[1];
  w = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(w[0] > w[1] ? w[0] : w[1]);
}
__device__ __forceinline__ float quarters_sum(float v) {
  u32x2 w = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(w[0]) + __uint_as_float(w[1]);
  w = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(w[0]) + __uint_as_float(w[1]);
}

// The activation vector of a node as the next GEMM's B operand.  A lane holds v[i][e] = feature 16 i + 4 q + e
// (i = feature tile, q = lane quarter); k slot (t, q, j) of the permuted weight blocks is feature
// 16 (2 t + (j >> 2)) + 4 q + (j & 3), so the fragment of k step t is (v[2t][0..3], v[2t+1][0..3]).
struct Frag {
  f16x8 h0, l0, h1, l1;     // k steps 0, 1; hi and lo planes
};
__device__ __forceinline__ void make_frag(const f32x4 v0, const f32x4 v1, const f32x4 v2, const f32x4 v3, const float s,
                                          Frag& f) {
  uint32_t h[8], l[8];
  const f32x4 w0 = v0 * s, w1 = v1 * s, w2 = v2 * s, w3 = v3 * s;
  split2_f16x2(w0[0], w0[1], h[0], l[0]);
  split2_f16x2(w0[2], w0[3], h[1], l[1]);
  split2_f16x2(w1[0], w1[1], h[2], l[2]);
  split2_f16x2(w1[2], w1[3], h[3], l[3]);
  split2_f16x2(w2[0], w2[1], h[4], l[4]);
  split2_f16x2(w2[2], w2[3], h[5], l[5]);
  split2_f16x2(w3[0], w3[1], h[6], l[6]);
  split2_f16x2(w3[2], w3[3], h[7], l[7]);
  f.h0 = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
  f.l0 = __builtin_bit_cast(f16x8, u32x4{l[0], l[1], l[2], l[3]});
  f.h1 = __builtin_bit_cast(f16x8, u32x4{h[4], h[5], h[6], h[7]});
  f.l1 = __builtin_bit_cast(f16x8, u32x4{l[4], l[5], l[6], l[7]});
}
// a * b + c on four features, one rounding each (explicit: the contraction of `a * b + c` differs between hipcc's
// scalar and packed selections, and the packed forms must not depend on the SLP vectoriser finding them)
__device__ __forceinline__ f32x4 fma4(const float a, const f32x4 b, const f32x4 c) {
  return __builtin_elementwise_fma(f32x4{a, a, a, a}, b, c);
}
__device__ __forceinline__ float absmax16(const f32x4 a, const f32x4 b, const f32x4 c, const f32x4 d) {
  const float m0 = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
  const float m1 = fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3])));
  const float m2 = fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3])));
  const float m3 = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3])));
  return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

#define GF16_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0);
// the three products (smallest first) of weight fragment (wh_, wl_) with activation fragment (xh_, xl_)
#define GF16_MM(c_, wh_, wl_, xh_, xl_) GF16_MFMA(wl_, xh_, c_) GF16_MFMA(wh_, xl_, c_) GF16_MFMA(wh_, xh_, c_)
// Weight fragment ring.  A query's 9 blocks are 36 PAIR STEPS: pair step d = (block d / 4, tile pair (d % 4) / 2, k step
// d % 2) multiplies the k-step fragments of feature tiles 2 p and 2 p + 1 into their two accumulators -- six MFMAs, the
// two dependent chains interleaved.  The hi / lo fragments of pair step d (four ds_read_b128, 16 registers) live in ring
// slot d % 3 and are requested in front of the MFMAs of pair step d - 2: two pair steps = twelve MFMAs = 192 matrix-pipe
// cycles of this wave ahead, the SIMD's other wave on top -- more than the LDS round trip under the eight waves' read
// traffic.  The stream is the same for every query and node, so the ring runs on across block
// boundaries, epilogues, queries and work units (round 4 requested each fragment one step ahead and waited out the LDS
// latency in front of every MFMA triple: hipcc sinks a plain C++ LDS read to just above its use).
// The reads and the waits are inline asm so that their ISSUE POINTS are fixed: volatile asm statements keep their program
// order (also against the compiler's own LDS accesses), the fragments are outputs of the read and in/outputs of the wait,
// so the MFMAs that use them come after it.  The counted wait is safe whatever else the compiler puts in between: LDS
// operations of a wave return in order, a fragment read is complete once at most as many LGKM operations are outstanding
// as were issued after it, and lgkmcnt(4) -- the four reads of the one younger pair step -- is at most that number.
// Hazards the compiler does not see (its hazard recogniser and waitcnt insertion do not look inside the asm):
//   * RAW on a fragment: covered by GF16_WAIT's counted wait (above).
//   * WAW against an MFMA result: excluded by construction -- the only MFMA destinations are the accumulators acc0..acc3,
//     which are live across every REQ / WAIT of a block, and a ring slot is a live asm OUTPUT from its REQ to its last
//     MFMA: two simultaneously live values never share a register.
//   * WAR against an MFMA that still reads the slot (a slot re-requested while the MFMAs of its previous contents are in
//     flight): slot d % 3 is re-requested by REQ(d + 3), which is issued in front of the MFMAs of pair step d + 1 -- the
//     six MFMAs of pair step d have all been ISSUED by then, an MFMA reads its A / B operands in its first passes, and
//     the read's data returns an LDS round trip (> 64 cycles, > the 16 cycles of one 16x16x32 MFMA) later.
//   tests/test_model_gpu.py holds this to bit-identical repeats on molecule, Syn_1827, MSRC-21 + IMDB and hub / ragged
//   shapes (200 / 100 launches each).
struct WFrag { f16x8 h0, l0, h1, l1; };          // tiles 2 p (0) and 2 p + 1 (1)
#define GF16_RD_(dst_, base_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(base_), "n"(off_));
#define GF16_REQ(d_)                                                                                    \
  {                                                                                                     \
    constexpr int dd_ = (d_) % 36, byte_ = (dd_ / 4) * (WBLK * 2) + ((dd_ % 4) / 2) * (2 * 16 * 64 * 2); \
    const uint32_t b_ = (dd_ % 2 ? wa1 : wa0) + (uint32_t)(byte_ & ~0xffff);                            \
    GF16_RD_(WF[(d_) % 3].h0, b_, byte_ & 0xffff)                                                       \
    GF16_RD_(WF[(d_) % 3].l0, b_, (byte_ & 0xffff) + WPL * 2)                                           \
    GF16_RD_(WF[(d_) % 3].h1, b_, (byte_ & 0xffff) + 16 * 64 * 2)                                       \
    GF16_RD_(WF[(d_) % 3].l1, b_, (byte_ & 0xffff) + 16 * 64 * 2 + WPL * 2)                             \
  }
#define GF16_WAIT(d_)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(WF[(d_) % 3].h0), "+v"(WF[(d_) % 3].l0), "+v"(WF[(d_) % 3].h1), \
               "+v"(WF[(d_) % 3].l1));
// pair step d_: (accA_, accB_) += W[tiles 2p, 2p+1][k step] X^T, three products each, smallest first, chains interleaved
#define GF16_PAIR(d_, accA_, accB_, xh_, xl_)                                                           \
  {                                                                                                     \
    GF16_WAIT(d_)                                                                                       \
    GF16_REQ((d_) + 2)                                                                                  \
    GF16_MFMA(WF[(d_) % 3].l0, xh_, accA_) GF16_MFMA(WF[(d_) % 3].l1, xh_, accB_)                       \
    GF16_MFMA(WF[(d_) % 3].h0, xl_, accA_) GF16_MFMA(WF[(d_) % 3].h1, xl_, accB_)                       \
    GF16_MFMA(WF[(d_) % 3].h0, xh_, accA_) GF16_MFMA(WF[(d_) % 3].h1, xh_, accB_)                       \
  }
// the 24 MFMAs of weight block b_ on this wave's 16 nodes: acc_i += W[16 i .. +15][:] X^T
#define GF16_BLOCK(b_, X_)                                                                              \
  GF16_PAIR(4 * (b_) + 0, acc0, acc1, X_.h0, X_.l0) GF16_PAIR(4 * (b_) + 1, acc0, acc1, X_.h1, X_.l1)   \
  GF16_PAIR(4 * (b_) + 2, acc2, acc3, X_.h0, X_.l0) GF16_PAIR(4 * (b_) + 3, acc2, acc3, X_.h1, X_.l1)
#define GF16_ZERO() { acc0 = acc1 = acc2 = acc3 = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define GF16_SCALE(f_) { acc0 *= (f_); acc1 *= (f_); acc2 *= (f_); acc3 *= (f_); }

constexpr int WQ = 5;                       // queries per work unit (29 = 6 units of <= 5: 3-10 measured, profiles/r4_j_*)
constexpr int WCOLS = 15;                   // neighbour steps whose column ids are staged per wave ([15][16] ints)
constexpr int WCST = 896;                   // u, d1, tp, b3 (64 each), b5, w7 (256 each), r, t (64 each)
constexpr size_t LDS_WAVE = (size_t)9 * WBLK * 2 + (size_t)WCST * 4 + (size_t)8 * (WCOLS * 16 + 64) * 4;
static_assert(LDS_WAVE <= 160 * 1024, "gossip_f16: LDS budget exceeded");

__global__ __launch_bounds__(GNT) void gossip_fused_f16_kernel(Args g, int64_t num_groups) {
  extern __shared__ __attribute__((aligned(16))) uint4 gf_lds[];
  short* WB = reinterpret_cast<short*>(gf_lds);                       // nine resident weight blocks
  float* cst = reinterpret_cast<float*>(WB + 9 * WBLK);
  const int tid = (int)__builtin_amdgcn_workitem_id_x(), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* ecolw = reinterpret_cast<int*>(cst + WCST) + wave * (WCOLS * 16 + 64);   // this wave's staged column ids ...
  float* zpw = reinterpret_cast<float*>(ecolw + WCOLS * 16);                     // ... and zp_q
  const int q4 = lane >> 4;
  const int wrow = lane & 15;
  const int wswz = (wrow >> 1) & 7;
  const int Q = g.Q;
  {
    const uint4* src = reinterpret_cast<const uint4*>(g.wstream);
    for (int i = tid; i < 9 * WBLK * 2 / 16; i += GNT) gf_lds[i] = src[i];
    for (int i = tid; i < 64; i += GNT) {
      cst[i] = g.u[i];
      cst[64 + i] = g.d1[i];
      cst[128 + i] = g.tp[i];
      cst[192 + i] = g.b3[i];
      cst[768 + i] = g.r[i];
      cst[832 + i] = g.t[i];
    }
    for (int i = tid; i < 256; i += GNT) {
      cst[256 + i] = g.b5[i];
      cst[512 + i] = g.w7[i];
    }
  }
  const float winv1 = g.winv[0], winvp = g.winv[1], winv3 = g.winv[2], winv5 = g.winv[3];
  __syncthreads();                                     // the only barrier: weights and constants are in place
  // LDS byte addresses of this lane's fragment chunk of k step 0 / 1 in weight block 0, tile 0, hi plane
  const uint32_t wa0 = (uint32_t)(uintptr_t)(WB + wrow * 64 + (((0 + q4) ^ wswz) << 3));
  const uint32_t wa1 = (uint32_t)(uintptr_t)(WB + wrow * 64 + (((4 + q4) ^ wswz) << 3));
  WFrag WF[3];
  GF16_REQ(0) GF16_REQ(1)                              // the ring's head start: pair steps 0 and 1 of the first query

  const int QC = (Q + WQ - 1) / WQ;
  const int64_t nunits = num_groups * QC;
  const unsigned long long nwaves = (unsigned long long)g.grid_blocks * 8;      // (= gridDim.x; passed, not queried)
  int64_t unit = (int64_t)__builtin_amdgcn_workgroup_id_x() * 8 + wave;
  const int fq = 4 * q4;
  // this lane's 16 features of the standard-order operands (hh, h1): 8 q4 .. +7 and 32 + 8 q4 .. +7
  const int fa = 8 * q4, fb = 32 + 8 * q4;
#define GW_V4(p_) (*reinterpret_cast<const f32x4*>(p_))
  while (unit < nunits) {
    // ticket of the next unit: in flight over this one
    unsigned long long tk = 0;
    if (lane == 0) tk = nwaves + atomicAdd(g.queue, 1ull);
    const int64_t grp = unit / QC;
    const int qa = (int)(unit - grp * QC) * WQ;
    const int qb = qa + WQ < Q ? qa + WQ : Q;
    // this lane's node.  With a tile order (desco_gossip_tile_order: the rows of a 128-node tile sorted by degree, pairs
    // dealt to eight waves in snake order) group gi of the tile takes sorted ranks 16 gi .. 16 gi + 15 -- rank t sits
    // at slot 16 w + 2 (t >> 4) + (t & 1) with w = its pair's snake position -- so the 16 nodes of a wave have similar
    // degrees and the neighbour loop below (as many steps as the group's largest degree) wastes few lanes.
    int64_t row_raw = grp * 16 + wrow;
    if (g.tperm) {
      const int gi = (int)(grp & 7), sl = wrow >> 1;
      const int64_t t0 = (grp >> 3) * GT;
      row_raw = t0 + g.tperm[t0 + 16 * ((gi & 1) ? 7 - sl : sl) + 2 * gi + (wrow & 1)];
    }
    const bool valid = row_raw < g.num_nodes;
    const int64_t row = valid ? row_raw : g.num_nodes - 1;
    const int e0 = g.rowptr[row];
    const int deg = valid ? g.rowptr[row + 1] - e0 : 0;
    int maxdeg = deg;
    for (int m = 1; m < 16; m <<= 1) {
      const int o = __shfl_xor(maxdeg, m, 64);
      maxdeg = maxdeg > o ? maxdeg : o;
    }
    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
    const int nst = maxdeg < WCOLS ? maxdeg : WCOLS;
    for (int i = q4; i < nst; i += 4)
      if (i < deg) ecolw[i * 16 + wrow] = g.col[e0 + i];
    // bit i: neighbour i of this node has the smaller id (gate g1 instead of 1 - g1); the same for every query
    uint32_t lt = 0;
    for (int i = 0; i < nst; ++i)
      if (i < deg && ecolw[i * 16 + wrow] < (int)row) lt |= 1u << i;

    // Everything query q needs from memory travels one query ahead: issued in front of the GEMM chain of query q - 1
    // (in front of the loop for the first one), consumed at the top of query q.
    float nzp, ngq;
    float4 nsi, nr0, nr1, nr2, nr3;
    f32x4 np0, np1, np2, np3, nz0, nz1, nz2, nz3;
    nr0 = nr1 = nr2 = nr3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define GW_NREC(i_) g.scal[(int64_t)((i_) < deg ? ecolw[(i_) * 16 + wrow] : (int)row) * Q + qn_]
#define GW_PREFETCH(q_)                                                                        \
  {                                                                                            \
    const int qn_ = (q_);                                                                      \
    nzp = g.zp[qn_ * 64 + lane];                                                               \
    ngq = g.g1[qn_];                                                                           \
    nsi = g.scal[row * Q + qn_];                                                               \
    np0 = GW_V4(g.p + qn_ * 64 + fa); np1 = GW_V4(g.p + qn_ * 64 + fa + 4);                    \
    np2 = GW_V4(g.p + qn_ * 64 + fb); np3 = GW_V4(g.p + qn_ * 64 + fb + 4);                    \
    nz0 = GW_V4(g.z + qn_ * 64 + fa); nz1 = GW_V4(g.z + qn_ * 64 + fa + 4);                    \
    nz2 = GW_V4(g.z + qn_ * 64 + fb); nz3 = GW_V4(g.z + qn_ * 64 + fb + 4);                    \
    if (maxdeg > 0) nr0 = GW_NREC(0);                                                          \
    if (maxdeg > 1) nr1 = GW_NREC(1);                                                          \
    if (maxdeg > 2) nr2 = GW_NREC(2);                                                          \
    if (maxdeg > 3) nr3 = GW_NREC(3);                                                          \
  }
    GW_PREFETCH(qa)

    for (int q = qa; q < qb; ++q) {
      zpw[lane] = nzp;
      const float gq = ngq;
      const float4 si = nsi;                           // (a0, b0, a1, x)
      f32x4 acc0, acc1, acc2, acc3;
      Frag XH, X1;
      float s_a;
      {
        // ---- neighbour sum and own h1 in the B layout of the first GEMM ----------------------------------------------
        const f32x4 p0 = np0, p1 = np1, p2 = np2, p3 = np3, z0 = nz0, z1 = nz1, z2 = nz2, z3 = nz3;
        const float4 c0 = nr0, c1 = nr1, c2 = nr2, c3 = nr3;
        const f32x4 r0 = GW_V4(cst + 768 + fa), r1 = GW_V4(cst + 768 + fa + 4), r2 = GW_V4(cst + 768 + fb),
                    r3 = GW_V4(cst + 768 + fb + 4);
        const f32x4 t0 = GW_V4(cst + 832 + fa), t1 = GW_V4(cst + 832 + fa + 4), t2 = GW_V4(cst + 832 + fb),
                    t3 = GW_V4(cst + 832 + fb + 4);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 h0 = zero4, h1 = zero4, h2 = zero4, h3 = zero4;
// (three dependent FMAs per feature, z first: as a sum of products hipcc emits mul + 2 fma + add)
#define GW_H1(s_, c_) __builtin_elementwise_max(fma4((s_).x, p##c_, fma4((s_).y, r##c_, fma4((s_).w, t##c_, z##c_))), zero4)
#define GW_ADD(rec_, gt_)                                                                      \
  {                                                                                            \
    const float g_ = (gt_);                                                                    \
    h0 = fma4(g_, GW_H1(rec_, 0), h0);                                                         \
    h1 = fma4(g_, GW_H1(rec_, 1), h1);                                                         \
    h2 = fma4(g_, GW_H1(rec_, 2), h2);                                                         \
    h3 = fma4(g_, GW_H1(rec_, 3), h3);                                                         \
  }
#define GW_GATE(i_) ((i_) < deg ? (((lt >> (i_)) & 1u) ? gq : 1.f - gq) : 0.f)
        if (maxdeg > 0) GW_ADD(c0, GW_GATE(0))
        if (maxdeg > 1) GW_ADD(c1, GW_GATE(1))
        if (maxdeg > 2) GW_ADD(c2, GW_GATE(2))
        if (maxdeg > 3) GW_ADD(c3, GW_GATE(3))
        for (int i = 4; i < maxdeg; i += 4) {          // beyond the prefetched four: four records in flight per step
          int j0 = (int)row, j1 = j0, j2 = j0, j3 = j0;
#define GW_COL(j_, k_) if (i + (k_) < deg) j_ = i + (k_) < WCOLS ? ecolw[(i + (k_)) * 16 + wrow] : g.col[e0 + i + (k_)];
          GW_COL(j0, 0) GW_COL(j1, 1) GW_COL(j2, 2) GW_COL(j3, 3)
#undef GW_COL
          const float4 a0 = g.scal[(int64_t)j0 * Q + q];
          float4 a1 = a0, a2 = a0, a3 = a0;
          if (i + 1 < maxdeg) a1 = g.scal[(int64_t)j1 * Q + q];
          if (i + 2 < maxdeg) a2 = g.scal[(int64_t)j2 * Q + q];
          if (i + 3 < maxdeg) a3 = g.scal[(int64_t)j3 * Q + q];
#define GW_GT(j_, k_) (i + (k_) < deg ? ((j_) < (int)row ? gq : 1.f - gq) : 0.f)
          GW_ADD(a0, GW_GT(j0, 0))
          if (i + 1 < maxdeg) GW_ADD(a1, GW_GT(j1, 1))
          if (i + 2 < maxdeg) GW_ADD(a2, GW_GT(j2, 2))
          if (i + 3 < maxdeg) GW_ADD(a3, GW_GT(j3, 3))
#undef GW_GT
        }
        const f32x4 s0 = GW_H1(si, 0), s1 = GW_H1(si, 1), s2 = GW_H1(si, 2), s3 = GW_H1(si, 3);
#undef GW_GATE
#undef GW_ADD
#undef GW_H1
        // one power of two for the node's h1 AND hh (they meet in one accumulator)
        const float m = quarters_max(fmaxf(absmax16(h0, h1, h2, h3), absmax16(s0, s1, s2, s3)));
        s_a = f16_scale_for(m);
        make_frag(h0, h1, h2, h3, s_a, XH);
        make_frag(s0, s1, s2, s3, s_a, X1);
      }
      if (q + 1 < qb) GW_PREFETCH(q + 1)
      Frag XC;
      float s_c;
      // ---- blocks 0, 1: h2 = relu([hh|h1] W1 + a1*u + d1) ------------------------------------------------------------
      GF16_ZERO()
      GF16_BLOCK(0, XH)
      GF16_BLOCK(1, X1)
      {
        const float f = pow2_inverse(s_a) * winv1;
        const float* u_ = cst + fq;
        const float* d_ = cst + 64 + fq;
#define GW_EPI1(a_, i_) a_ = __builtin_elementwise_max(fma4(f, a_, fma4(si.z, GW_V4(u_ + 16 * (i_)), GW_V4(d_ + 16 * (i_)))), f32x4{0.f, 0.f, 0.f, 0.f});
        GW_EPI1(acc0, 0) GW_EPI1(acc1, 1) GW_EPI1(acc2, 2) GW_EPI1(acc3, 3)
#undef GW_EPI1
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- blocks 2, 3: y1 = leaky([h1|h2] Wp + x*tp + zp_q, 0.1) ------------------------------------------------------
      GF16_ZERO()
      GF16_BLOCK(2, X1)
      {
        const float rs = s_c * pow2_inverse(s_a);
        GF16_SCALE(rs)
      }
      GF16_BLOCK(3, XC)
      {
        const float f = pow2_inverse(s_c) * winvp;
        const float* t_ = cst + 128 + fq;
        const float* z_ = zpw + fq;
#define GW_EPI2(a_, i_)                                                                                       \
  {                                                                                                           \
    const f32x4 v_ = fma4(f, a_, fma4(si.w, GW_V4(t_ + 16 * (i_)), GW_V4(z_ + 16 * (i_))));                    \
    a_ = __builtin_elementwise_max(v_, v_ * 0.1f);                                                            \
  }
        GW_EPI2(acc0, 0) GW_EPI2(acc1, 1) GW_EPI2(acc2, 2) GW_EPI2(acc3, 3)
#undef GW_EPI2
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- block 4: y2 = relu(y1 W3 + b3) ----------------------------------------------------------------------------
      GF16_ZERO()
      GF16_BLOCK(4, XC)
      {
        const float f = pow2_inverse(s_c) * winv3;
        const float* b_ = cst + 192 + fq;
#define GW_EPI3(a_, i_) a_ = __builtin_elementwise_max(fma4(f, a_, GW_V4(b_ + 16 * (i_))), f32x4{0.f, 0.f, 0.f, 0.f});
        GW_EPI3(acc0, 0) GW_EPI3(acc1, 1) GW_EPI3(acc2, 2) GW_EPI3(acc3, 3)
#undef GW_EPI3
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- blocks 5..8: head partial  sum_c relu(y2 W5 + b5)[c] * w7[c] ------------------------------------------------
      f32x4 hp = {0.f, 0.f, 0.f, 0.f};                 // four running sums per lane, folded once after the last block
      const float fh = pow2_inverse(s_c) * winv5;
#define GW_HEAD1(a_, cg_, i_)                                                                                 \
  hp = __builtin_elementwise_fma(__builtin_elementwise_max(fma4(fh, a_, GW_V4(cst + 256 + 64 * (cg_) + 16 * (i_) + fq)), \
                                                           f32x4{0.f, 0.f, 0.f, 0.f}),                        \
                                 GW_V4(cst + 512 + 64 * (cg_) + 16 * (i_) + fq), hp);
#define GW_HEAD(cg_)                                                                                         \
  GF16_ZERO()                                                                                                \
  GF16_BLOCK(5 + (cg_), XC)                                                                    \
  GW_HEAD1(acc0, cg_, 0) GW_HEAD1(acc1, cg_, 1) GW_HEAD1(acc2, cg_, 2) GW_HEAD1(acc3, cg_, 3)
      GW_HEAD(0) GW_HEAD(1) GW_HEAD(2) GW_HEAD(3)
#undef GW_HEAD
#undef GW_HEAD1
      const float part = quarters_sum((hp[0] + hp[1]) + (hp[2] + hp[3]));
      if (lane < 16 && valid) g.out[row * Q + q] = part + g.b7 + si.w;
    }
    unit = (int64_t)__builtin_amdgcn_readfirstlane((int)(tk & 0xffffffffull)) |
           ((int64_t)__builtin_amdgcn_readfirstlane((int)(tk >> 32)) << 32);
  }
#undef GW_PREFETCH
#undef GW_NREC
#undef GW_V4
  if (lane == 0 && atomicAdd(g.queue + 1, 1ull) == nwaves - 1) {        // last wave out: leave the queue clean
    g.queue[0] = 0;
    g.queue[1] = 0;
  }
}

// The weight stream: block b of the nine 64 x 64 blocks (W1[:, 0:64], W1[:, 64:128], Wp[:, 0:64], Wp[:, 64:128], W3,
// W5[0:64], W5[64:128], W5[128:192], W5[192:256]) as the LDS image the kernel copies linearly: plane-major, rows of
// 64 halves with the 16-byte chunk index XOR (row >> 1) & 7, and for the blocks whose input arrives in registers
// (3..8) the k slots permuted to the C/D layout of the producing GEMM.  One thread per (block, row, k slot).
__global__ __launch_bounds__(256) void gossip_f16_stream_kernel(const short* __restrict__ w1, const short* __restrict__ wp,
                                                                const short* __restrict__ w3, const short* __restrict__ w5,
                                                                short* __restrict__ stream) {
  const int idx = (int)(__builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x());       // [9][64 n][64 k slot]
  if (idx >= 9 * 4096) return;
  const int b = idx >> 12, n = (idx >> 6) & 63, ks = idx & 63;
  const short* src;
  int ld, rows, n0 = 0, k0 = 0;
  if (b < 2) { src = w1; ld = 128; rows = 64; k0 = 64 * b; }
  else if (b < 4) { src = wp; ld = 128; rows = 64; k0 = 64 * (b - 2); }
  else if (b == 4) { src = w3; ld = 64; rows = 64; }
  else { src = w5; ld = 64; rows = 256; n0 = 64 * (b - 5); }
  const int t = ks >> 5, q = (ks >> 3) & 3, j = ks & 7;
  const int k = b >= 3 ? 16 * (2 * t + (j >> 2)) + 4 * q + (j & 3) : ks;
  const int dst = n * 64 + ((((ks >> 3) ^ (n >> 1)) & 7) << 3) + (ks & 7);
  const int64_t s = (int64_t)(n0 + n) * ld + k0 + k;
  stream[b * WBLK + dst] = src[s];                                  // hi plane
  stream[b * WBLK + WPL + dst] = src[(int64_t)rows * ld + s];       // lo plane
}

}  // namespace gf16
}  // namespace desco

using namespace desco;

extern "C" int desco_gossip_f16_stream(const int16_t* w1_planes, const int16_t* wp_planes, const int16_t* w3_planes,
                                       const int16_t* w5_planes, int16_t* stream, desco_stream_t st) {
  if (!w1_planes || !wp_planes || !w3_planes || !w5_planes || !stream)
    return fail(DESCO_EINVAL, "desco_gossip_f16_stream: bad argument");
  hipLaunchKernelGGL(gf16::gossip_f16_stream_kernel, dim3(9 * 4096 / 256), dim3(256), 0, (hipStream_t)st,
                     reinterpret_cast<const short*>(w1_planes), reinterpret_cast<const short*>(wp_planes),
                     reinterpret_cast<const short*>(w3_planes), reinterpret_cast<const short*>(w5_planes),
                     reinterpret_cast<short*>(stream));
  return launch_status("desco_gossip_f16_stream");
}

extern "C" int desco_gossip_fused_f16x3_f32(const float* scal4, const int32_t* rowptr, const int32_t* col,
                                            int64_t num_nodes, int num_q, const float* g1, const float* p,
                                            const float* z, const float* zp, const float* r, const float* t,
                                            const float* u, const float* tp, const float* d1, const int16_t* wstream,
                                            const float* winv, const float* b3, const float* b5, const float* w7,
                                            float b7, float* out, const uint8_t* tile_perm, uint64_t* queue,
                                            desco_stream_t stream) {
  using namespace gf16;
  if (num_nodes == 0) return 0;
  auto mis16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) != 0; };
  auto mis8 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 7) != 0; };
  if (!scal4 || !rowptr || !g1 || !p || !z || !zp || !r || !t || !u || !tp || !d1 || !wstream || !winv || !b3 || !b5 ||
      !w7 || !out || !queue || num_nodes < 0 || num_q < 1 || num_q > 65535 || mis16(scal4) || mis16(wstream) ||
      mis16(p) || mis16(z) || mis16(r) || mis16(t) || mis8(queue))
    return fail(DESCO_EINVAL, "desco_gossip_fused_f16x3_f32: bad argument");
  // (with a tile order the groups are the eighths of whole 128-node tiles: a ragged last tile has empty ranks)
  const int64_t groups = tile_perm ? (num_nodes + GT - 1) / GT * 8 : (num_nodes + 15) / 16;
  Args a{reinterpret_cast<const float4*>(scal4), rowptr, col, num_nodes, num_q, g1, p, z, zp, r, t, u, tp, d1,
         reinterpret_cast<const short*>(wstream), winv, b3, b5, w7, b7, out, tile_perm,
         reinterpret_cast<unsigned long long*>(queue)};
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gossip_fused_f16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_WAVE);
    if (e != hipSuccess) return fail((int)e, "desco_gossip_fused_f16x3_f32: cannot size LDS");
    attr_once.mark();
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int64_t units = groups * ((num_q + WQ - 1) / WQ);
  const int64_t blocks = (units + 7) / 8;
  const unsigned grid = (unsigned)(blocks < (int64_t)cus ? blocks : (int64_t)cus);
  a.grid_blocks = grid;
  hipLaunchKernelGGL(gossip_fused_f16_kernel, dim3(grid), dim3(GNT), LDS_WAVE, (hipStream_t)stream, a, groups);
  return launch_status("desco_gossip_fused_f16x3_f32");
}

#include "tu_no_packed_f32_end.hpp"
