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
// What changed against the six-product bf16 kernel, and why:
//   * Arithmetic: x s = hi + lo in fp16 (22 bits), hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16: 3 instead of 6
//     MFMAs per tile step, 2 instead of 3 operand planes, 3-5 instead of 9 VALU per split pair.  fp16's 5-bit exponent is
//     carried by power-of-two scales: one per weight matrix (host), one PER NODE per activation vector (largest |v| of
//     the node's 64 features -> [2^14, 2^15)), undone exactly in the next epilogue.
//   * A wave owns 16 NODES x all 64 features of a block (4 feature tiles x 1 node tile) instead of 32 x 32: the per-node
//     maximum is then a wave-local reduction (two lane-quarter swaps), and in the transposed MFMA form (A = weight rows,
//     B = activation rows) the C/D layout of one GEMM -- lane = node, registers = features 16 i + 4 q + e -- IS the B
//     layout of the next one under a fixed permutation of k (baked into the weight stream on the host): h2, y1 and y2
//     never leave the registers.  No activation images for them, no epilogue LDS writes, no fragment reads of
//     activations for 6 of the 9 weight blocks, and no barrier on their account.
//   * h1 / hh (written by the neighbour-sum phase in its own lane map) still pass through LDS images; a wave reads back
//     exactly the 16 rows it wrote (wave-private).  One block barrier separates the phase from the GEMMs all the same
//     (see there).
//   * The only block-wide data left are the tile's scalar records (published once per item) and the weight blocks, which
//     stream through a RING of four 16 KB buffers, loaded four and stored two blocks ahead of their use: one barrier per
//     TWO blocks.  6 barriers per item instead of 10.
// Block = 8 waves = one CU, persistent over (tile, query) items drawn from a ticket queue the CALLER provides.
#include "common_device.hpp"

namespace desco {

namespace gf16 {

constexpr int GT = 128;            // rows (nodes) per tile
constexpr int GNT = 512;           // threads per block
constexpr int PLN = GT * 64;       // halves per activation plane
constexpr int WPL = 64 * 64;       // halves per weight-block plane
constexpr int WBLK = 2 * WPL;      // halves per weight block (hi, lo)
constexpr int PCAP = 1216;         // neighbour records prefetched for the next tile (up to three per thread)
constexpr int ECAP = 1216;         // neighbour records staged per pass
constexpr int CST = 832;           // u, d1, tp, b3 (64 each), b5, w7 (256 each), zp_q (64)
constexpr size_t LDS_BYTES = (size_t)2 * 2 * PLN * 2 + (size_t)4 * WBLK * 2 + (size_t)ECAP * 20 + GT * 16 + 132 * 4 +
                             CST * 4 + GT * 4 + GT + 16;
static_assert(LDS_BYTES <= 160 * 1024, "gossip_f16: LDS budget exceeded");

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
};

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// element offset of (row, k) inside a [rows][64] fp16 plane with swizzled 16-byte chunks
__device__ __forceinline__ int pidx(const int row, const int k) {
  return row * 64 + ((((k >> 3) ^ (row >> 1)) & 7) << 3) + (k & 7);
}

// max over the 32 lanes of a half wave (both halves at once) of NON-NEGATIVE floats, compared as unsigned integers
// (same order, no NaN canonicalisation, v_max_u32_dpp): four DPP steps inside the 16-lane rows, one row swap
__device__ __forceinline__ float half_wave_max(const float f) {
  uint32_t v = __float_as_uint(f);
#define GF16_DPP_MAX(ctrl_) { const uint32_t o_ = __builtin_amdgcn_update_dpp(0u, v, (ctrl_), 0xf, 0xf, true); v = v > o_ ? v : o_; }
  GF16_DPP_MAX(0xB1)     // quad_perm [1,0,3,2]
  GF16_DPP_MAX(0x4E)     // quad_perm [2,3,0,1]
  GF16_DPP_MAX(0x141)    // row_half_mirror
  GF16_DPP_MAX(0x140)    // row_mirror
#undef GF16_DPP_MAX
  const u32x2 w = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return __uint_as_float(w[0] > w[1] ? w[0] : w[1]);
}
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
  split2_f16x2(v0[0] * s, v0[1] * s, h[0], l[0]);
  split2_f16x2(v0[2] * s, v0[3] * s, h[1], l[1]);
  split2_f16x2(v1[0] * s, v1[1] * s, h[2], l[2]);
  split2_f16x2(v1[2] * s, v1[3] * s, h[3], l[3]);
  split2_f16x2(v2[0] * s, v2[1] * s, h[4], l[4]);
  split2_f16x2(v2[2] * s, v2[3] * s, h[5], l[5]);
  split2_f16x2(v3[0] * s, v3[1] * s, h[6], l[6]);
  split2_f16x2(v3[2] * s, v3[3] * s, h[7], l[7]);
  f.h0 = __builtin_bit_cast(f16x8, u32x4{h[0], h[1], h[2], h[3]});
  f.l0 = __builtin_bit_cast(f16x8, u32x4{l[0], l[1], l[2], l[3]});
  f.h1 = __builtin_bit_cast(f16x8, u32x4{h[4], h[5], h[6], h[7]});
  f.l1 = __builtin_bit_cast(f16x8, u32x4{l[4], l[5], l[6], l[7]});
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
// weight fragments of (feature tile i_, k step t_) from ring buffer wb_ into (h_, l_)
#define GF16_LDW(h_, l_, wb_, i_, t_)                                                        \
  {                                                                                          \
    const short* a_ = (wb_) + (i_) * 16 * 64 + (((4 * (t_) + q4) ^ wswz) << 3);              \
    h_ = *reinterpret_cast<const f16x8*>(a_);                                                \
    l_ = *reinterpret_cast<const f16x8*>(a_ + WPL);                                          \
  }
// 24 MFMAs of one 64x64 weight block on this wave's 16 nodes: acc_i += W[16 i .. +15][:] X^T.  The fragments of the
// next (tile, k step) are read while the MFMAs of the current one run.
#define GF16_BLOCK(wb_, X_)                                                                  \
  {                                                                                          \
    const short* w_ = (wb_) + wrow * 64;                                                     \
    f16x8 ah_, al_, bh_, bl_;                                                                \
    GF16_LDW(ah_, al_, w_, 0, 0)                                                             \
    GF16_LDW(bh_, bl_, w_, 0, 1)                                                             \
    GF16_MM(acc0, ah_, al_, X_.h0, X_.l0)                                                    \
    GF16_LDW(ah_, al_, w_, 1, 0)                                                             \
    GF16_MM(acc0, bh_, bl_, X_.h1, X_.l1)                                                    \
    GF16_LDW(bh_, bl_, w_, 1, 1)                                                             \
    GF16_MM(acc1, ah_, al_, X_.h0, X_.l0)                                                    \
    GF16_LDW(ah_, al_, w_, 2, 0)                                                             \
    GF16_MM(acc1, bh_, bl_, X_.h1, X_.l1)                                                    \
    GF16_LDW(bh_, bl_, w_, 2, 1)                                                             \
    GF16_MM(acc2, ah_, al_, X_.h0, X_.l0)                                                    \
    GF16_LDW(ah_, al_, w_, 3, 0)                                                             \
    GF16_MM(acc2, bh_, bl_, X_.h1, X_.l1)                                                    \
    GF16_LDW(bh_, bl_, w_, 3, 1)                                                             \
    GF16_MM(acc3, ah_, al_, X_.h0, X_.l0)                                                    \
    GF16_MM(acc3, bh_, bl_, X_.h1, X_.l1)                                                    \
    GF16_DRAIN()                                                                             \
  }
// End of a block: one VALU read of every accumulator chain's last result, fenced for the scheduler, BEFORE any later load
// is issued.  An MFMA is issued in order but retires later (dependent chains of three queue up behind each other and
// behind the SIMD's other wave), and the register allocator is free to hand a chain's dying intermediate register -- or
// one the chain still has to write -- to the next LDS load (it did: the h1 fragments of block 1 landed in a register
// that block 0's last MFMAs had not written yet; the load returned first, the MFMA result then overwrote it.  Found
// as a run-to-run difference on a handful of nodes, tools/debug/gf16_variants.sh).  A VALU read of an MFMA result
// is interlocked, a returning load is not.
#define GF16_DRAIN()                                                                         \
  {                                                                                          \
    const float t_ = (acc0[3] + acc1[3]) + (acc2[3] + acc3[3]);                              \
    asm volatile("" :: "v"(t_));                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  }
#define GF16_ZERO() { acc0 = acc1 = acc2 = acc3 = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define GF16_SCALE(f_) { acc0 *= (f_); acc1 *= (f_); acc2 *= (f_); acc3 *= (f_); }

// weight stream: block b_ (0..8) global -> register set, register set -> ring slot
// (global base pointer + 32-bit offset: the block offset is an opaque scalar so that the compiler neither folds it into
// per-block 64-bit vector addresses -- 36 registers it then spilled -- nor loses the global address space)
#define GF16_WLOAD(q_, b_)                                                                   \
  {                                                                                          \
    uint32_t o_ = (uint32_t)(b_) * (WBLK * 2);                                               \
    asm volatile("" : "+s"(o_));                                                             \
    const char* s_ = reinterpret_cast<const char*>(g.wstream);                               \
    q_##0 = *reinterpret_cast<const uint4*>(s_ + (o_ + woff));                               \
    q_##1 = *reinterpret_cast<const uint4*>(s_ + (o_ + woff + WPL * 2));                     \
  }
#define GF16_WSTORE(q_, slot_)                                                               \
  {                                                                                          \
    short* d_ = WB + ((((slot_) + wbase) & 3) * WBLK) + 8 * tid;                             \
    *reinterpret_cast<uint4*>(d_) = q_##0;                                                   \
    *reinterpret_cast<uint4*>(d_ + WPL) = q_##1;                                             \
  }
#define GF16_RING(k_) (WB + ((((k_) + wbase) & 3) * WBLK))

__global__ __launch_bounds__(GNT) void gossip_fused_f16_kernel(Args g, int64_t num_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint4 gf_lds[];
  short* I0 = reinterpret_cast<short*>(gf_lds);          // h1 planes (hi, lo)
  short* I1 = I0 + 2 * PLN;                              // hh planes
  short* WB = I1 + 2 * PLN;                              // ring of four weight blocks
  int* ecol = reinterpret_cast<int*>(WB + 4 * WBLK);     // [ECAP] neighbour records: column ...
  float4* escal = reinterpret_cast<float4*>(ecol + ECAP);   // ... and scalar record
  float4* srow = escal + ECAP;                           // scalars of the tile rows
  int* rp = reinterpret_cast<int*>(srow + GT);           // rowptr[n0 .. n0+128]
  float* cst = reinterpret_cast<float*>(rp + 132);       // bias vectors (see CST)
  float* sA = cst + CST;                                 // [128] per-node scale of (h1, hh)
  uint8_t* tperm = reinterpret_cast<uint8_t*>(sA + GT);  // [128] phase-1 slot -> row
  unsigned long long* tick = reinterpret_cast<unsigned long long*>(tperm + GT);   // the item after the current one

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4;                 // k chunk within a 32-deep step / feature quad of a 16x16 tile
  const int wrow = lane & 15;               // this lane's weight row inside a feature tile (A operand)
  const int wswz = (wrow >> 1) & 7;         // chunk swizzle of weight rows 16 i + wrow
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
  const float winv1 = g.winv[0], winvp = g.winv[1], winv3 = g.winv[2], winv5 = g.winv[3];

  // The (rowptr -> col -> scalar record) chain of the NEXT item is fetched into these registers in three stages spread
  // over the current item and published to LDS when the current item is done.
  float4 n_srow = make_float4(0.f, 0.f, 0.f, 0.f), n_scal = n_srow, n_scal2 = n_srow, n_scal3 = n_srow;
  int n_rp = 0, n_col = 0, n_col2 = 0, n_col3 = 0, n_ebeg = 0, n_cnt = 0;
  uint32_t n_perm = 0x03020100u + 0x04040404u * (uint32_t)(tid & 31);   // identity slots 4 tid .. 4 tid + 3
  float n_zp = 0.f, n_gq = 0.f;
  float2 n_pc = make_float2(0.f, 0.f), n_zc = n_pc;
  const int f0 = 2 * (lane & 31);          // phase-1 lane map: features (f0, f0+1)
#define GF16_STAGE1(it_)                                                                   \
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
#define GF16_STAGE2() \
  {                                                          \
    if (tid < n_cnt) n_col = g.col[n_ebeg + tid];            \
    if (tid + GNT < n_cnt) n_col2 = g.col[n_ebeg + tid + GNT]; \
    if (tid + 2 * GNT < n_cnt) n_col3 = g.col[n_ebeg + tid + 2 * GNT]; \
  }
#define GF16_STAGE3(it_) \
  {                                                                                          \
    if (tid < n_cnt) n_scal = g.scal[(int64_t)n_col * Q + (int)((it_) % Q)];                 \
    if (tid + GNT < n_cnt) n_scal2 = g.scal[(int64_t)n_col2 * Q + (int)((it_) % Q)];         \
    if (tid + 2 * GNT < n_cnt) n_scal3 = g.scal[(int64_t)n_col3 * Q + (int)((it_) % Q)];     \
  }

  if (tid == 0) *tick = gridDim.x + atomicAdd(g.queue, 1ull);
  GF16_STAGE1(item)
  GF16_STAGE2()
  GF16_STAGE3(item)
  const float2 rc = *reinterpret_cast<const float2*>(g.r + f0);
  const float2 tc = *reinterpret_cast<const float2*>(g.t + f0);
  // Weight ring.  An item is ten slots (nine blocks + one empty), so slot parity = block parity: block k of an item is
  // stored at step k - 2 from register set (k & 1), loaded at step k - 4, and sits in ring buffer (k + wbase) & 3
  // with wbase = 0, 2, 0, 2, ... over this block's items.  A barrier in front of every EVEN step orders both hazards
  // (stored two steps ahead -> visible; overwritten buffer last read four steps ago -> done).
  uint4 qa0, qa1, qb0, qb1;
  int wbase = 0;
  const uint32_t woff = 16u * (uint32_t)tid;
  GF16_WLOAD(qa, 0)
  GF16_WLOAD(qb, 1)
  GF16_WSTORE(qa, 0)
  GF16_WSTORE(qb, 1)
  GF16_WLOAD(qa, 2)
  GF16_WLOAD(qb, 3)

  for (;;) {
    // ---- publish the prefetched tile data ------------------------------------------------------------------------
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
    const float gq = n_gq;
    const float2 pc = n_pc, zc = n_zc;
    const int cnt0 = n_cnt;
    const int64_t n0 = (item / Q) * GT;
    const int nrows = (int)((g.num_nodes - n0) < GT ? (g.num_nodes - n0) : GT);
    __syncthreads();
    const int64_t next = (int64_t)*tick;   // (written before the barrier)
    const bool has_next = next < nitems;
    unsigned long long tk = 0;             // ticket of the item after `next`: in flight until the end of this item
    if (tid == 0 && has_next) tk = gridDim.x + atomicAdd(g.queue, 1ull);
    if (has_next) GF16_STAGE1(next)

    // ---- phase 1: h1 of the tile rows, gated neighbour sum hh; rows tperm[16 wave ..] belong to this wave ---------
    {
      int lane1 = lane;
      asm volatile("" : "+v"(lane1));      // keeps this phase's addresses out of the registers of the GEMM phase
      const int half1 = lane1 >> 5, f1 = 2 * (lane1 & 31);
      float2 hh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) hh[i] = make_float2(0.f, 0.f);
      const int ebeg = rp[0], eend = rp[GT];
      // pass 0: the prefetched records [ebeg, ebeg+cnt0); later passes (tiles with more than PCAP neighbour records)
      // stage ECAP records at a time from global memory
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
            float hx = sj.x * pc.x + sj.y * rc.x + sj.w * tc.x + zc.x;
            float hy = sj.x * pc.y + sj.y * rc.y + sj.w * tc.y + zc.y;
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
      // (two passes: the loads of all eight rows first -- the LDS stores of a row would fence the next row's loads)
      int rows8[8];
      float2 hs[8];
      const uint4 tp16 = *reinterpret_cast<const uint4*>(tperm + wave * 16);      // this wave's 16 slot -> row bytes
      const uint32_t tpw[4] = {tp16.x, tp16.y, tp16.z, tp16.w};
      float4 sis[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        rows8[i] = (int)((tpw[i >> 1] >> (16 * (i & 1) + 8 * half1)) & 0xffu);           // byte 2 i + half1
        sis[i] = srow[rows8[i]];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 si = sis[i];
        float hx = si.x * pc.x + si.y * rc.x + si.w * tc.x + zc.x;
        float hy = si.x * pc.y + si.y * rc.y + si.w * tc.y + zc.y;
        hs[i] = make_float2(hx > 0.f ? hx : 0.f, hy > 0.f ? hy : 0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = rows8[i];
        const float hx = hs[i].x, hy = hs[i].y;
        // one power of two for the node's h1 AND hh (they meet in one accumulator): the largest of the 128 values
        const float m = half_wave_max(fmaxf(fmaxf(hx, hy), fmaxf(fabsf(hh[i].x), fabsf(hh[i].y))));
        const float s = f16_scale_for(m);
        if (f1 == 0) sA[row] = s;
        uint32_t h_, l_;
        const int o = pidx(row, f1);
        split2_f16x2(hx * s, hy * s, h_, l_);
        *reinterpret_cast<uint32_t*>(I0 + o) = h_;
        *reinterpret_cast<uint32_t*>(I0 + PLN + o) = l_;
        split2_f16x2(hh[i].x * s, hh[i].y * s, h_, l_);
        *reinterpret_cast<uint32_t*>(I1 + o) = h_;
        *reinterpret_cast<uint32_t*>(I1 + PLN + o) = l_;
      }
    }
    // The rows this wave reads below are the rows it wrote above, so the images would need no block barrier -- but the
    // phase above must not run beside another wave's GEMM phase: without this barrier a handful of (node, query) results
    // differed from run to run.  Traced (round 4, profiles/r4_b_gossip_f16_race.md) to the self term h1 coming out
    // wrong in lanes 48-63, low half of the packed fp32 operations only, with every ingredient (dumped from the same
    // registers) right -- only when a slower wave was still in phase 1 while others issued MFMAs; neither full waits
    // for LDS / vector memory, nor replacing the lane swaps and byte reads, nor fencing the MFMA chains removed it,
    // the barrier does (84 of 84 repeat runs identical, tools/debug/gossip_f16_probe2.py).  The compiler fence keeps
    // the fragment reads (fp16 vectors) behind the image stores (32-bit words) under type-based aliasing.
    asm volatile("" ::: "memory");
    __syncthreads();
    // ---- GEMM chain on this wave's 16 nodes ------------------------------------------------------------------------
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    const int nrow = tperm[wave * 16 + (lane2 & 15)];      // this lane's node (row of the tile)
    const float4 sx = srow[nrow];                          // (a0, b0, a1, x)
    const float s_a = sA[nrow];
    const int fq = 4 * (lane2 >> 4);                        // first feature of this lane inside a feature tile
    Frag XH, X1;                                           // hh, h1 as B fragments (standard k order)
    const int xsw = (nrow >> 1) & 7;
    const int xc0 = ((q4 ^ xsw) & 7) << 3, xc1 = (((4 + q4) ^ xsw) & 7) << 3;
#define GF16_LDX(X_, img_)                                                   \
  {                                                                          \
    const short* a_ = (img_) + nrow * 64;                                    \
    X_.h0 = *reinterpret_cast<const f16x8*>(a_ + xc0);                       \
    X_.l0 = *reinterpret_cast<const f16x8*>(a_ + PLN + xc0);                 \
    X_.h1 = *reinterpret_cast<const f16x8*>(a_ + xc1);                       \
    X_.l1 = *reinterpret_cast<const f16x8*>(a_ + PLN + xc1);                 \
  }
    GF16_LDX(XH, I1)
    f32x4 acc0, acc1, acc2, acc3;          // feature tiles 0..3 of this lane's node
    Frag XC;                               // h2, then y1, then y2 (permuted k order)
    float s_c;                             // its scale
#define GF16_V4(p_) (*reinterpret_cast<const f32x4*>(p_))
    // ---- blocks 0, 1: h2 = relu([hh|h1] W1 + a1*u + d1) ------------------------------------------------------------
    GF16_WSTORE(qa, 2)
    GF16_WLOAD(qa, 4)
    GF16_ZERO()
    GF16_BLOCK(GF16_RING(0), XH)
    GF16_WSTORE(qb, 3)
    GF16_WLOAD(qb, 5)
    GF16_LDX(X1, I0)
    GF16_BLOCK(GF16_RING(1), X1)
    {
      const float f = pow2_inverse(s_a) * winv1;
      const float* u_ = cst + fq;
      const float* d_ = cst + 64 + fq;
#define GF16_EPI1(a_, i_) a_ = __builtin_elementwise_max(a_ * f + (GF16_V4(u_ + 16 * (i_)) * sx.z + GF16_V4(d_ + 16 * (i_))), f32x4{0.f, 0.f, 0.f, 0.f});
      GF16_EPI1(acc0, 0) GF16_EPI1(acc1, 1) GF16_EPI1(acc2, 2) GF16_EPI1(acc3, 3)
#undef GF16_EPI1
      s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
      make_frag(acc0, acc1, acc2, acc3, s_c, XC);
    }
    // ---- blocks 2, 3: y1 = leaky([h1|h2] Wp + x*tp + zp_q, 0.1) ------------------------------------------------------
    __syncthreads();
    if (has_next) GF16_STAGE2()
    GF16_WSTORE(qa, 4)
    GF16_WLOAD(qa, 6)
    GF16_ZERO()
    GF16_BLOCK(GF16_RING(2), X1)
    GF16_WSTORE(qb, 5)
    GF16_WLOAD(qb, 7)
    {
      const float rs = s_c * pow2_inverse(s_a);       // units of h1's scale -> units of h2's scale (exact)
      GF16_SCALE(rs)
    }
    GF16_BLOCK(GF16_RING(3), XC)
    {
      const float f = pow2_inverse(s_c) * winvp;
      const float* t_ = cst + 128 + fq;
      const float* z_ = cst + 768 + fq;
#define GF16_EPI2(a_, i_)                                                                                     \
  {                                                                                                           \
    const f32x4 v_ = a_ * f + (GF16_V4(t_ + 16 * (i_)) * sx.w + GF16_V4(z_ + 16 * (i_)));                      \
    a_ = __builtin_elementwise_max(v_, v_ * 0.1f);                                                            \
  }
      GF16_EPI2(acc0, 0) GF16_EPI2(acc1, 1) GF16_EPI2(acc2, 2) GF16_EPI2(acc3, 3)
#undef GF16_EPI2
      s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
      make_frag(acc0, acc1, acc2, acc3, s_c, XC);
    }
    // ---- block 4: y2 = relu(y1 W3 + b3) ----------------------------------------------------------------------------
    __syncthreads();
    if (has_next) GF16_STAGE3(next)
    GF16_WSTORE(qa, 6)
    GF16_WLOAD(qa, 8)
    GF16_ZERO()
    GF16_BLOCK(GF16_RING(4), XC)
    {
      const float f = pow2_inverse(s_c) * winv3;
      const float* b_ = cst + 192 + fq;
#define GF16_EPI3(a_, i_) a_ = __builtin_elementwise_max(a_ * f + GF16_V4(b_ + 16 * (i_)), f32x4{0.f, 0.f, 0.f, 0.f});
      GF16_EPI3(acc0, 0) GF16_EPI3(acc1, 1) GF16_EPI3(acc2, 2) GF16_EPI3(acc3, 3)
#undef GF16_EPI3
      s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
      make_frag(acc0, acc1, acc2, acc3, s_c, XC);
    }
    // ---- blocks 5..8: head partial  sum_c relu(y2 W5 + b5)[c] * w7[c], 4 column groups of 64 ------------------------
    float part = 0.f;
    const float fh = pow2_inverse(s_c) * winv5;
#define GF16_HEAD1(a_, cg_, i_)                                                                               \
  {                                                                                                           \
    const f32x4 v_ = __builtin_elementwise_max(a_ * fh + GF16_V4(cst + 256 + 64 * (cg_) + 16 * (i_) + fq),     \
                                               f32x4{0.f, 0.f, 0.f, 0.f}) *                                   \
                     GF16_V4(cst + 512 + 64 * (cg_) + 16 * (i_) + fq);                                        \
    part += (v_[0] + v_[1]) + (v_[2] + v_[3]);                                                                \
  }
// (fenced for the scheduler: left alone it overlaps the four blocks' bias / w7 reads and epilogues -- 213 live registers
// there, and the allocator then parks the next item's prefetched records in scratch)
#define GF16_HEAD(cg_)                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                         \
  GF16_HEAD1(acc0, cg_, 0) GF16_HEAD1(acc1, cg_, 1) GF16_HEAD1(acc2, cg_, 2) GF16_HEAD1(acc3, cg_, 3)        \
  asm volatile("" : "+v"(part));   /* the sum is DONE here (pure arithmetic is otherwise sunk to its use) */  \
  __builtin_amdgcn_sched_barrier(0);
    GF16_WSTORE(qb, 7)
    if (has_next) GF16_WLOAD(qb, 1)        // (step 5 would load the empty slot 9; step 7's load is taken here)
    GF16_ZERO()
    GF16_BLOCK(GF16_RING(5), XC)
    GF16_HEAD(0)
    __syncthreads();
    GF16_WSTORE(qa, 8)
    if (has_next) GF16_WLOAD(qa, 0)        // block 0 of the next item (slot 10)
    GF16_ZERO()
    GF16_BLOCK(GF16_RING(6), XC)
    GF16_HEAD(1)
    GF16_ZERO()                            // step 7: nothing to store (slot 9 is empty); its load went out at step 5
    GF16_BLOCK(GF16_RING(7), XC)
    GF16_HEAD(2)
    __syncthreads();
    if (has_next) {                        // step 8: block 0 of the next item into slot 10
      GF16_WSTORE(qa, 10)
      GF16_WLOAD(qa, 2)
    }
    GF16_ZERO()
    GF16_BLOCK(GF16_RING(8), XC)
    GF16_HEAD(3)
#undef GF16_HEAD
#undef GF16_HEAD1
    if (has_next) {                        // step 9 (no block): block 1 of the next item into slot 11
      GF16_WSTORE(qb, 11)
      GF16_WLOAD(qb, 3)
    }
    part = quarters_sum(part);
    if (lane2 < 16 && nrow < nrows) g.out[(n0 + nrow) * Q + q] = part + g.b7 + sx.w;
    if (!has_next) break;
    item = next;
    wbase ^= 2;
    if (tid == 0) *tick = tk;
    // No barrier here: everything the next publish overwrites (srow, rp, tperm, the records, zp_q, sA is written in
    // phase 1) was last read in front of the barrier before block 8 or is read by the writing thread only (tick).
  }
  if (tid == 0 && atomicAdd(g.queue + 1, 1ull) == gridDim.x - 1) {      // last block out: leave the queue clean
    g.queue[0] = 0;
    g.queue[1] = 0;
  }
#undef GF16_STAGE1
#undef GF16_STAGE2
#undef GF16_STAGE3
#undef GF16_V4
#undef GF16_LDX
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-autonomous form (round 4, second half).  With two fp16 planes ALL nine weight blocks fit in LDS at once
// (147 456 B), and with the neighbour-sum phase computed directly in the MFMA B layout (lane = node, 16 features per
// lane) nothing is shared between the waves of a workgroup but those read-only weights: no activation images, no weight
// ring, NO barrier inside the work loop.  Every wave carries its own 16 nodes through the whole network for a chunk of
// queries, and the eight waves of a CU drift apart freely -- one wave's neighbour sums, epilogues and record loads run
// under the other waves' MFMAs, which the lock-stepped block form could not do (phase 1 and six barriers per item were
// ~40 % of its time with no MFMA in flight).  Work unit = (16-node group, WQ queries), drawn per wave from the caller's
// queue.  The GEMM chain, scales, k permutation and weight images are those of the block form above.
constexpr int WQ = 8;                       // queries per work unit
constexpr int WCOLS = 15;                   // neighbour steps whose column ids are staged per wave ([15][16] ints)
constexpr int WCST = 896;                   // u, d1, tp, b3 (64 each), b5, w7 (256 each), r, t (64 each)
constexpr size_t LDS_WAVE = (size_t)9 * WBLK * 2 + (size_t)WCST * 4 + (size_t)8 * (WCOLS * 16 + 64) * 4;
static_assert(LDS_WAVE <= 160 * 1024, "gossip_wave_f16: LDS budget exceeded");

__global__ __launch_bounds__(GNT) DESCO_NO_PACKED_F32 void gossip_wave_f16_kernel(Args g, int64_t num_groups) {
  extern __shared__ __attribute__((aligned(16))) uint4 gf_lds[];
  short* WB = reinterpret_cast<short*>(gf_lds);                       // nine resident weight blocks
  float* cst = reinterpret_cast<float*>(WB + 9 * WBLK);
  const int tid = threadIdx.x, lane = tid & 63;
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

  const int QC = (Q + WQ - 1) / WQ;
  const int64_t nunits = num_groups * QC;
  const unsigned long long nwaves = (unsigned long long)gridDim.x * 8;
  int64_t unit = (int64_t)blockIdx.x * 8 + wave;
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
    const int64_t row_raw = grp * 16 + wrow;
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

    for (int q = qa; q < qb; ++q) {
      zpw[lane] = g.zp[q * 64 + lane];
      const float gq = g.g1[q];
      const float4 si = g.scal[row * Q + q];           // (a0, b0, a1, x)
      f32x4 acc0, acc1, acc2, acc3;
      Frag XH, X1;
      float s_a;
      {
        // ---- neighbour sum and own h1 in the B layout of the first GEMM ----------------------------------------------
        const f32x4 p0 = GW_V4(g.p + q * 64 + fa), p1 = GW_V4(g.p + q * 64 + fa + 4), p2 = GW_V4(g.p + q * 64 + fb),
                    p3 = GW_V4(g.p + q * 64 + fb + 4);
        const f32x4 z0 = GW_V4(g.z + q * 64 + fa), z1 = GW_V4(g.z + q * 64 + fa + 4), z2 = GW_V4(g.z + q * 64 + fb),
                    z3 = GW_V4(g.z + q * 64 + fb + 4);
        const f32x4 r0 = GW_V4(cst + 768 + fa), r1 = GW_V4(cst + 768 + fa + 4), r2 = GW_V4(cst + 768 + fb),
                    r3 = GW_V4(cst + 768 + fb + 4);
        const f32x4 t0 = GW_V4(cst + 832 + fa), t1 = GW_V4(cst + 832 + fa + 4), t2 = GW_V4(cst + 832 + fb),
                    t3 = GW_V4(cst + 832 + fb + 4);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 h0 = zero4, h1 = zero4, h2 = zero4, h3 = zero4;
#define GW_H1(s_, c_) __builtin_elementwise_max((s_).x * p##c_ + (s_).y * r##c_ + (s_).w * t##c_ + z##c_, zero4)
        for (int i = 0; i < maxdeg; ++i) {
          const bool on = i < deg;
          int j = (int)row;
          if (on) j = i < WCOLS ? ecolw[i * 16 + wrow] : g.col[e0 + i];
          const float4 sj = g.scal[(int64_t)j * Q + q];
          const float gt = on ? (j < (int)row ? gq : 1.f - gq) : 0.f;
          h0 += gt * GW_H1(sj, 0);
          h1 += gt * GW_H1(sj, 1);
          h2 += gt * GW_H1(sj, 2);
          h3 += gt * GW_H1(sj, 3);
        }
        const f32x4 s0 = GW_H1(si, 0), s1 = GW_H1(si, 1), s2 = GW_H1(si, 2), s3 = GW_H1(si, 3);
#undef GW_H1
        // one power of two for the node's h1 AND hh (they meet in one accumulator)
        const float m = quarters_max(fmaxf(absmax16(h0, h1, h2, h3), absmax16(s0, s1, s2, s3)));
        s_a = f16_scale_for(m);
        make_frag(h0, h1, h2, h3, s_a, XH);
        make_frag(s0, s1, s2, s3, s_a, X1);
      }
      Frag XC;
      float s_c;
      // ---- blocks 0, 1: h2 = relu([hh|h1] W1 + a1*u + d1) ------------------------------------------------------------
      GF16_ZERO()
      GF16_BLOCK(WB + 0 * WBLK, XH)
      GF16_BLOCK(WB + 1 * WBLK, X1)
      {
        const float f = pow2_inverse(s_a) * winv1;
        const float* u_ = cst + fq;
        const float* d_ = cst + 64 + fq;
#define GW_EPI1(a_, i_) a_ = __builtin_elementwise_max(a_ * f + (GW_V4(u_ + 16 * (i_)) * si.z + GW_V4(d_ + 16 * (i_))), f32x4{0.f, 0.f, 0.f, 0.f});
        GW_EPI1(acc0, 0) GW_EPI1(acc1, 1) GW_EPI1(acc2, 2) GW_EPI1(acc3, 3)
#undef GW_EPI1
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- blocks 2, 3: y1 = leaky([h1|h2] Wp + x*tp + zp_q, 0.1) ------------------------------------------------------
      GF16_ZERO()
      GF16_BLOCK(WB + 2 * WBLK, X1)
      {
        const float rs = s_c * pow2_inverse(s_a);
        GF16_SCALE(rs)
      }
      GF16_BLOCK(WB + 3 * WBLK, XC)
      {
        const float f = pow2_inverse(s_c) * winvp;
        const float* t_ = cst + 128 + fq;
        const float* z_ = zpw + fq;
#define GW_EPI2(a_, i_)                                                                                       \
  {                                                                                                           \
    const f32x4 v_ = a_ * f + (GW_V4(t_ + 16 * (i_)) * si.w + GW_V4(z_ + 16 * (i_)));                          \
    a_ = __builtin_elementwise_max(v_, v_ * 0.1f);                                                            \
  }
        GW_EPI2(acc0, 0) GW_EPI2(acc1, 1) GW_EPI2(acc2, 2) GW_EPI2(acc3, 3)
#undef GW_EPI2
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- block 4: y2 = relu(y1 W3 + b3) ----------------------------------------------------------------------------
      GF16_ZERO()
      GF16_BLOCK(WB + 4 * WBLK, XC)
      {
        const float f = pow2_inverse(s_c) * winv3;
        const float* b_ = cst + 192 + fq;
#define GW_EPI3(a_, i_) a_ = __builtin_elementwise_max(a_ * f + GW_V4(b_ + 16 * (i_)), f32x4{0.f, 0.f, 0.f, 0.f});
        GW_EPI3(acc0, 0) GW_EPI3(acc1, 1) GW_EPI3(acc2, 2) GW_EPI3(acc3, 3)
#undef GW_EPI3
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- blocks 5..8: head partial  sum_c relu(y2 W5 + b5)[c] * w7[c] ------------------------------------------------
      float part = 0.f;
      const float fh = pow2_inverse(s_c) * winv5;
#define GW_HEAD1(a_, cg_, i_)                                                                                 \
  {                                                                                                           \
    const f32x4 v_ = __builtin_elementwise_max(a_ * fh + GW_V4(cst + 256 + 64 * (cg_) + 16 * (i_) + fq),       \
                                               f32x4{0.f, 0.f, 0.f, 0.f}) *                                   \
                     GW_V4(cst + 512 + 64 * (cg_) + 16 * (i_) + fq);                                          \
    part += (v_[0] + v_[1]) + (v_[2] + v_[3]);                                                                \
  }
#define GW_HEAD(cg_)                                                                                         \
  GF16_ZERO()                                                                                                \
  GF16_BLOCK(WB + (5 + (cg_)) * WBLK, XC)                                                                    \
  GW_HEAD1(acc0, cg_, 0) GW_HEAD1(acc1, cg_, 1) GW_HEAD1(acc2, cg_, 2) GW_HEAD1(acc3, cg_, 3)
      GW_HEAD(0) GW_HEAD(1) GW_HEAD(2) GW_HEAD(3)
#undef GW_HEAD
#undef GW_HEAD1
      part = quarters_sum(part);
      if (lane < 16 && valid) g.out[row * Q + q] = part + g.b7 + si.w;
    }
    unit = (int64_t)__builtin_amdgcn_readfirstlane((int)(tk & 0xffffffffull)) |
           ((int64_t)__builtin_amdgcn_readfirstlane((int)(tk >> 32)) << 32);
  }
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
  const int idx = blockIdx.x * 256 + threadIdx.x;       // [9][64 n][64 k slot]
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

extern "C" int desco_gossip_wave_f16x3_f32(const float* scal4, const int32_t* rowptr, const int32_t* col,
                                           int64_t num_nodes, int num_q, const float* g1, const float* p,
                                           const float* z, const float* zp, const float* r, const float* t,
                                           const float* u, const float* tp, const float* d1, const int16_t* wstream,
                                           const float* winv, const float* b3, const float* b5, const float* w7,
                                           float b7, float* out, uint64_t* queue, desco_stream_t stream) {
  using namespace gf16;
  if (num_nodes == 0) return 0;
  auto mis16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) != 0; };
  auto mis8 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 7) != 0; };
  if (!scal4 || !rowptr || !g1 || !p || !z || !zp || !r || !t || !u || !tp || !d1 || !wstream || !winv || !b3 || !b5 ||
      !w7 || !out || !queue || num_nodes < 0 || num_q < 1 || num_q > 65535 || mis16(scal4) || mis16(wstream) ||
      mis16(p) || mis16(z) || mis16(r) || mis16(t) || mis8(queue))
    return fail(DESCO_EINVAL, "desco_gossip_wave_f16x3_f32: bad argument");
  const int64_t groups = (num_nodes + 15) / 16;
  Args a{reinterpret_cast<const float4*>(scal4), rowptr, col, num_nodes, num_q, g1, p, z, zp, r, t, u, tp, d1,
         reinterpret_cast<const short*>(wstream), winv, b3, b5, w7, b7, out, nullptr,
         reinterpret_cast<unsigned long long*>(queue)};
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gossip_wave_f16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_WAVE);
    if (e != hipSuccess) return fail((int)e, "desco_gossip_wave_f16x3_f32: cannot size LDS");
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
  hipLaunchKernelGGL(gossip_wave_f16_kernel, dim3(grid), dim3(GNT), LDS_WAVE, (hipStream_t)stream, a, groups);
  return launch_status("desco_gossip_wave_f16x3_f32");
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
      mis8(p) || mis8(z) || mis8(r) || mis8(t) || mis8(queue) || (reinterpret_cast<uintptr_t>(tile_perm) & 3))
    return fail(DESCO_EINVAL, "desco_gossip_fused_f16x3_f32: bad argument");
  const int64_t bx = (num_nodes + GT - 1) / GT;
  if (bx > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_fused_f16x3_f32: too many nodes");
  Args a{reinterpret_cast<const float4*>(scal4), rowptr, col, num_nodes, num_q, g1, p, z, zp, r, t, u, tp, d1,
         reinterpret_cast<const short*>(wstream), winv, b3, b5, w7, b7, out, tile_perm,
         reinterpret_cast<unsigned long long*>(queue)};
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gossip_fused_f16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) return fail((int)e, "desco_gossip_fused_f16x3_f32: cannot size LDS");
    attr_once.mark();
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int64_t nitems = bx * num_q;
  const unsigned grid = (unsigned)(nitems < (int64_t)cus ? nitems : (int64_t)cus);
  hipLaunchKernelGGL(gossip_fused_f16_kernel, dim3(grid), dim3(GNT), LDS_BYTES, (hipStream_t)stream, a, bx);
  return launch_status("desco_gossip_fused_f16x3_f32");
}
