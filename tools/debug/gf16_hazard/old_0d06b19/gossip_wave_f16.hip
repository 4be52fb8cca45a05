// The gossip_f16.hip of commit 0d06b19 (round 4: the first wave-autonomous version, the one whose packed-fp32 build was
// wrong in 1-2 % of its outputs), kept as the reproducer (build_variants.sh): the wave-autonomous kernel only (the block-form kernel of
// that commit is cut), verbatim but for the VAR_* bisection switches:
//   VAR_PK  VAR_WAVES  VAR_FENCE  VAR_P1_SCALAR  VAR_EPI_SCALAR   as in gossip_f16_var.hip
//   VAR_H1FMA    neighbour recompute as three dependent FMAs (the form of the shipped kernel) instead of a sum of products
//   VAR_NOFLAT   the neighbour id through two typed loads + select instead of a select of pointers (flat_load)
//   VAR_VMCNT0   s_waitcnt vmcnt(0) lgkmcnt(0) in front of the neighbour arithmetic
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
#ifndef VAR_WAVES
#define VAR_WAVES 8
#endif
#if defined(VAR_FENCE)
#define VFENCE() __builtin_amdgcn_sched_barrier(0);
#else
#define VFENCE()
#endif

namespace desco {

namespace gf16 {

constexpr int GT = 128;            // rows (nodes) per tile
constexpr int GNT = 64 * VAR_WAVES;
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

// VAR_ASMNOP: the fp16 split's inline-asm instructions with 16 wait states in front of each (inside the asm statement, so
//             the compiler's schedule and register allocation are those of the failing build)
// VAR_NOASM:  the split in plain C (compiler-selected instructions, all hazards visible to hipcc)
#if defined(VAR_ASMNOP)
__device__ __forceinline__ void vsplit2(const float f0, const float f1, uint32_t& hi, uint32_t& lo) {
  const desco_f2 f = {f0, f1};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, desco_h2));
  asm("s_nop 7\ns_nop 7\nv_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(f0), "v"(hi));
  asm("s_nop 7\ns_nop 7\nv_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(f1), "v"(hi));
}
#elif defined(VAR_NOASM)
__device__ __forceinline__ void vsplit2(const float f0, const float f1, uint32_t& hi, uint32_t& lo) {
  const desco_f2 f = {f0, f1};
  const desco_h2 h = __builtin_convertvector(f, desco_h2);
  hi = __builtin_bit_cast(uint32_t, h);
  const desco_f2 d = {f0 - (float)h[0], f1 - (float)h[1]};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, desco_h2));
}
#else
#define vsplit2 split2_f16x2
#endif
struct Frag {
  f16x8 h0, l0, h1, l1;     // k steps 0, 1; hi and lo planes
};
__device__ __forceinline__ void make_frag(const f32x4 v0, const f32x4 v1, const f32x4 v2, const f32x4 v3, const float s,
                                          Frag& f) {
  uint32_t h[8], l[8];
  vsplit2(v0[0] * s, v0[1] * s, h[0], l[0]);
  vsplit2(v0[2] * s, v0[3] * s, h[1], l[1]);
  vsplit2(v1[0] * s, v1[1] * s, h[2], l[2]);
  vsplit2(v1[2] * s, v1[3] * s, h[3], l[3]);
  vsplit2(v2[0] * s, v2[1] * s, h[4], l[4]);
  vsplit2(v2[2] * s, v2[3] * s, h[5], l[5]);
  vsplit2(v3[0] * s, v3[1] * s, h[6], l[6]);
  vsplit2(v3[2] * s, v3[3] * s, h[7], l[7]);
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
#if defined(VAR_EPI_SCALAR)
#define GF16_SC1(a_, f_) { a_[0] *= (f_); a_[1] *= (f_); a_[2] *= (f_); a_[3] *= (f_); }
#define GF16_SCALE(f_) { GF16_SC1(acc0, f_) GF16_SC1(acc1, f_) GF16_SC1(acc2, f_) GF16_SC1(acc3, f_) }
#else
#define GF16_SCALE(f_) { acc0 *= (f_); acc1 *= (f_); acc2 *= (f_); acc3 *= (f_); }
#endif

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
constexpr size_t LDS_WAVE = (size_t)9 * WBLK * 2 + (size_t)WCST * 4 + (size_t)VAR_WAVES * (WCOLS * 16 + 64) * 4;
static_assert(LDS_WAVE <= 160 * 1024, "gossip_wave_f16: LDS budget exceeded");

#if defined(VAR_PK)
#define VAR_ATTR
#else
#define VAR_ATTR DESCO_NO_PACKED_F32
#endif
__global__ __launch_bounds__(GNT) VAR_ATTR void gossip_wave_f16_kernel(Args g, int64_t num_groups) {
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
  const unsigned long long nwaves = (unsigned long long)gridDim.x * VAR_WAVES;
  int64_t unit = (int64_t)blockIdx.x * VAR_WAVES + wave;
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
#if defined(VAR_P1_SCALAR)
#define GW_H1E(s_, c_, e_) fmaxf((s_).x * p##c_[e_] + (s_).y * r##c_[e_] + (s_).w * t##c_[e_] + z##c_[e_], 0.f)
#define GW_H1(s_, c_) f32x4{GW_H1E(s_, c_, 0), GW_H1E(s_, c_, 1), GW_H1E(s_, c_, 2), GW_H1E(s_, c_, 3)}
#define GW_ACC(h_, c_) { h_[0] += gt * GW_H1E(sj, c_, 0); h_[1] += gt * GW_H1E(sj, c_, 1); h_[2] += gt * GW_H1E(sj, c_, 2); h_[3] += gt * GW_H1E(sj, c_, 3); }
#elif defined(VAR_H1FMA)
#define GW_FMA4(a_, b_, c_) f32x4{__builtin_fmaf(a_, b_[0], c_[0]), __builtin_fmaf(a_, b_[1], c_[1]), __builtin_fmaf(a_, b_[2], c_[2]), __builtin_fmaf(a_, b_[3], c_[3])}
#define GW_H1(s_, c_) __builtin_elementwise_max(GW_FMA4((s_).x, p##c_, GW_FMA4((s_).y, r##c_, GW_FMA4((s_).w, t##c_, z##c_))), zero4)
#define GW_ACC(h_, c_) h_ += gt * GW_H1(sj, c_);
#elif defined(VAR_PKMASK)
// the sum of products term by term, each term either on the 4-vector (packed selection) or element by element
// (scalar; build with -fno-slp-vectorize).  bit 0: y*r  bit 1: + x*p  bit 2: + w*t  bit 3: + z  bit 4: h += gate * relu
#define GW_EL(expr_) f32x4{expr_(0), expr_(1), expr_(2), expr_(3)}
  auto term_h1 = [&](const float sx, const float sy, const float sw, const f32x4 pc, const f32x4 rc, const f32x4 tc,
                     const f32x4 zc) {
    f32x4 a;
#if !defined(VAR_M1FORM)
    if (VAR_PKMASK & 1) a = sy * rc; else { a = f32x4{sy * rc[0], sy * rc[1], sy * rc[2], sy * rc[3]}; }
#else
    {   // forms of the ONE packed instruction that fails (y * r as v_pk_mul_f32 r, rec op_sel:[0,1]):
      desco_f2 lo_, hi_;
      const desco_f2 r01 = {rc[0], rc[1]}, r23 = {rc[2], rc[3]};
#if VAR_M1FORM == 1      // y broadcast into its own register pair by two v_mov, plain v_pk_mul (no op_sel)
      desco_f2 yy;
      asm volatile("v_mov_b32 %0, %1" : "=v"(yy.x) : "v"(sy));
      asm volatile("v_mov_b32 %0, %1" : "=v"(yy.y) : "v"(sy));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(lo_) : "v"(r01), "v"(yy));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(hi_) : "v"(r23), "v"(yy));
#elif VAR_M1FORM == 2    // the record pair copied to fresh registers by v_mov, then the failing form on the copy
      desco_f2 cp;
      asm volatile("v_mov_b32 %0, %1" : "=v"(cp.x) : "v"(sx));
      asm volatile("v_mov_b32 %0, %1" : "=v"(cp.y) : "v"(sy));
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(lo_) : "v"(r01), "v"(cp));
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(hi_) : "v"(r23), "v"(cp));
#elif VAR_M1FORM == 3    // the record pair (as loaded) as src0: op_sel:[1,0]
      const desco_f2 rec = {sx, sy};
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(lo_) : "v"(rec), "v"(r01));
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(hi_) : "v"(rec), "v"(r23));
#elif VAR_M1FORM == 4    // the failing form itself, as asm (control: must fail like m1)
      const desco_f2 rec = {sx, sy};
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(lo_) : "v"(r01), "v"(rec));
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(hi_) : "v"(r23), "v"(rec));
#elif VAR_M1FORM == 5    // the failing form with s_nop 4 in front of each
      const desco_f2 rec = {sx, sy};
      asm volatile("s_nop 4\nv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(lo_) : "v"(r01), "v"(rec));
      asm volatile("s_nop 4\nv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(hi_) : "v"(r23), "v"(rec));
#endif
      a = f32x4{lo_.x, lo_.y, hi_.x, hi_.y};
    }
#endif
    if (VAR_PKMASK & 2) a = sx * pc + a; else { a = f32x4{__builtin_fmaf(sx, pc[0], a[0]), __builtin_fmaf(sx, pc[1], a[1]), __builtin_fmaf(sx, pc[2], a[2]), __builtin_fmaf(sx, pc[3], a[3])}; }
    if (VAR_PKMASK & 4) a = sw * tc + a; else { a = f32x4{__builtin_fmaf(sw, tc[0], a[0]), __builtin_fmaf(sw, tc[1], a[1]), __builtin_fmaf(sw, tc[2], a[2]), __builtin_fmaf(sw, tc[3], a[3])}; }
    if (VAR_PKMASK & 8) a = a + zc; else { a = f32x4{a[0] + zc[0], a[1] + zc[1], a[2] + zc[2], a[3] + zc[3]}; }
    return f32x4{fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f)};
  };
#define GW_H1(s_, c_) term_h1((s_).x, (s_).y, (s_).w, p##c_, r##c_, t##c_, z##c_)
#define GW_ACC(h_, c_)                                                                                   \
  {                                                                                                      \
    const f32x4 v_ = GW_H1(sj, c_);                                                                      \
    if (VAR_PKMASK & 16) h_ += gt * v_;                                                                  \
    else h_ = f32x4{__builtin_fmaf(gt, v_[0], h_[0]), __builtin_fmaf(gt, v_[1], h_[1]), __builtin_fmaf(gt, v_[2], h_[2]), __builtin_fmaf(gt, v_[3], h_[3])}; \
  }
#else
#define GW_H1(s_, c_) __builtin_elementwise_max((s_).x * p##c_ + (s_).y * r##c_ + (s_).w * t##c_ + z##c_, zero4)
#define GW_ACC(h_, c_) h_ += gt * GW_H1(sj, c_);
#endif
        VFENCE()
        for (int i = 0; i < maxdeg; ++i) {
          const bool on = i < deg;
          int j = (int)row;
#if defined(VAR_NOFLAT)
          if (on) {
            const int jl = ecolw[(i < WCOLS ? i : 0) * 16 + wrow];
            j = i < WCOLS ? jl : g.col[e0 + i];
          }
#else
          if (on) j = i < WCOLS ? ecolw[i * 16 + wrow] : g.col[e0 + i];
#endif
          const float4 sj = g.scal[(int64_t)j * Q + q];
#if defined(VAR_VMCNT0)
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
          const float gt = on ? (j < (int)row ? gq : 1.f - gq) : 0.f;
          GW_ACC(h0, 0) GW_ACC(h1, 1) GW_ACC(h2, 2) GW_ACC(h3, 3)
        }
        const f32x4 s0 = GW_H1(si, 0), s1 = GW_H1(si, 1), s2 = GW_H1(si, 2), s3 = GW_H1(si, 3);
#undef GW_H1
        VFENCE()
        // one power of two for the node's h1 AND hh (they meet in one accumulator)
        const float m = quarters_max(fmaxf(absmax16(h0, h1, h2, h3), absmax16(s0, s1, s2, s3)));
        s_a = f16_scale_for(m);
        make_frag(h0, h1, h2, h3, s_a, XH);
        make_frag(s0, s1, s2, s3, s_a, X1);
        VFENCE()
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
#if defined(VAR_EPI_SCALAR)
#define GW_EPI1(a_, i_)                                                                                       \
  {                                                                                                           \
    const f32x4 uu_ = GW_V4(u_ + 16 * (i_)), dd_ = GW_V4(d_ + 16 * (i_));                                      \
    for (int e_ = 0; e_ < 4; ++e_) a_[e_] = fmaxf(__builtin_fmaf(a_[e_], f, __builtin_fmaf(uu_[e_], si.z, dd_[e_])), 0.f); \
  }
#else
#define GW_EPI1(a_, i_) a_ = __builtin_elementwise_max(a_ * f + (GW_V4(u_ + 16 * (i_)) * si.z + GW_V4(d_ + 16 * (i_))), f32x4{0.f, 0.f, 0.f, 0.f});
#endif
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
#if defined(VAR_EPI_SCALAR)
#define GW_EPI2(a_, i_)                                                                                       \
  {                                                                                                           \
    const f32x4 tt_ = GW_V4(t_ + 16 * (i_)), zz_ = GW_V4(z_ + 16 * (i_));                                      \
    for (int e_ = 0; e_ < 4; ++e_) {                                                                          \
      const float v_ = __builtin_fmaf(a_[e_], f, __builtin_fmaf(tt_[e_], si.w, zz_[e_]));                     \
      a_[e_] = fmaxf(v_, v_ * 0.1f);                                                                          \
    }                                                                                                         \
  }
#else
#define GW_EPI2(a_, i_)                                                                                       \
  {                                                                                                           \
    const f32x4 v_ = a_ * f + (GW_V4(t_ + 16 * (i_)) * si.w + GW_V4(z_ + 16 * (i_)));                          \
    a_ = __builtin_elementwise_max(v_, v_ * 0.1f);                                                            \
  }
#endif
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
#if defined(VAR_EPI_SCALAR)
#define GW_EPI3(a_, i_)                                                                                       \
  {                                                                                                           \
    const f32x4 bb_ = GW_V4(b_ + 16 * (i_));                                                                  \
    for (int e_ = 0; e_ < 4; ++e_) a_[e_] = fmaxf(__builtin_fmaf(a_[e_], f, bb_[e_]), 0.f);                   \
  }
#else
#define GW_EPI3(a_, i_) a_ = __builtin_elementwise_max(a_ * f + GW_V4(b_ + 16 * (i_)), f32x4{0.f, 0.f, 0.f, 0.f});
#endif
        GW_EPI3(acc0, 0) GW_EPI3(acc1, 1) GW_EPI3(acc2, 2) GW_EPI3(acc3, 3)
#undef GW_EPI3
        s_c = f16_scale_for(quarters_max(absmax16(acc0, acc1, acc2, acc3)));
        make_frag(acc0, acc1, acc2, acc3, s_c, XC);
      }
      // ---- blocks 5..8: head partial  sum_c relu(y2 W5 + b5)[c] * w7[c] ------------------------------------------------
      float part = 0.f;
      const float fh = pow2_inverse(s_c) * winv5;
#if defined(VAR_EPI_SCALAR)
#define GW_HEAD1(a_, cg_, i_)                                                                                 \
  {                                                                                                           \
    const f32x4 bb_ = GW_V4(cst + 256 + 64 * (cg_) + 16 * (i_) + fq), ww_ = GW_V4(cst + 512 + 64 * (cg_) + 16 * (i_) + fq); \
    float v_[4];                                                                                              \
    for (int e_ = 0; e_ < 4; ++e_) v_[e_] = fmaxf(__builtin_fmaf(a_[e_], fh, bb_[e_]), 0.f) * ww_[e_];        \
    part += (v_[0] + v_[1]) + (v_[2] + v_[3]);                                                                \
  }
#else
#define GW_HEAD1(a_, cg_, i_)                                                                                 \
  {                                                                                                           \
    const f32x4 v_ = __builtin_elementwise_max(a_ * fh + GW_V4(cst + 256 + 64 * (cg_) + 16 * (i_) + fq),       \
                                               f32x4{0.f, 0.f, 0.f, 0.f}) *                                   \
                     GW_V4(cst + 512 + 64 * (cg_) + 16 * (i_) + fq);                                          \
    part += (v_[0] + v_[1]) + (v_[2] + v_[3]);                                                                \
  }
#endif
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
  const int64_t blocks = (units + VAR_WAVES - 1) / VAR_WAVES;
  const unsigned grid = (unsigned)(blocks < (int64_t)cus ? blocks : (int64_t)cus);
  hipLaunchKernelGGL(gossip_wave_f16_kernel, dim3(grid), dim3(GNT), LDS_WAVE, (hipStream_t)stream, a, groups);
  return launch_status("desco_gossip_wave_f16x3_f32");
}

