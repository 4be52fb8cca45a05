// Fused gossip stage for gfx950: for a tile of 64 nodes and ONE query the whole per-query network
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
// ~2.3 KB per (node,query)); the kernel is bound by the f32 MFMA rate (288 MFMAs per wave and tile).
// Block = 4 waves, output tiles 64x64 as 2x2 wave tiles of 32x32 (v_mfma_f32_32x32x2_f32); three
// 64x68 activation images + one 64x68 weight image in LDS (71 KB -> 2 blocks per CU); the last GEMM
// (64 -> 256) runs as a 6-product bf16 split (fp32-accurate) whose planes overlay dead images; weight
// blocks are prefetched into registers under the previous block's MFMAs.  Blocks are persistent
// (2 per CU) and walk (tile, query) items; the next item's CSR slice and scalar records are
// prefetched through registers in three stages under the current item's GEMMs.
// Algebra: DESIGN.md 4.2.
#include "common_device.hpp"

namespace desco {

constexpr int GT = 64;      // rows (nodes) per tile
constexpr int GAS = 68;     // image row stride (floats): 16-B aligned rows, conflict-free b128 reads
constexpr int ECAP = 768;   // neighbour records staged per pass (aliases the third image)

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
  const float* w1;          // [64,128]  (n-major = transposed: row n holds the 128 k)
  const float* wp;          // [64,128]
  const float* w3;          // [64,64]
  const float* b3;          // [64]
  const float* w5;          // [256,64]  (kept for reference / un-split builds)
  const short* w5s;         // [3][256][64] bf16 planes (hi, mid, lo) of w5, split on the host
  const float* b5;          // [256]
  const float* w7;          // [256]
  float b7;
  float* out;               // [N,Q]
};

// scalars pre-pass: one wave per node, lane = query
__global__ __launch_bounds__(256) void gossip_scalars_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             int64_t num_nodes, int Q,
                                                             const float* __restrict__ g0,
                                                             const float* __restrict__ g1,
                                                             float4* __restrict__ scal) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= num_nodes) return;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  const int q = lane < Q ? lane : Q - 1;
  float slo = 0.f, shi = 0.f;
  int dlo = 0;
  for (int e = e0; e < e1; ++e) {
    const int64_t j = col[e];
    const float xv = x[j * ldx + q];
    if (j < i) {
      slo += xv;
      ++dlo;
    } else {
      shi += xv;
    }
  }
  const float flo = (float)dlo, fhi = (float)(e1 - e0 - dlo);
  const float a = g0[q], b = g1[q];
  if (lane < Q)
    scal[i * Q + lane] = make_float4(a * flo + (1.f - a) * fhi, a * slo + (1.f - a) * shi,
                                     b * flo + (1.f - b) * fhi, x[i * ldx + q]);
}

// weight block staging: 64 output columns n x 64 k, source is n-major (row n, leading dim ld_);
// thread t moves 4 float4: n = t>>4 (+16,+32,+48), k = 4*(t&15)
#define DESCO_WLOAD(src_, ld_)                                                          \
  {                                                                                     \
    const float* s_ = (src_) + (int64_t)(tid >> 4) * (ld_) + 4 * (tid & 15);            \
    w0 = *reinterpret_cast<const float4*>(s_);                                          \
    w1 = *reinterpret_cast<const float4*>(s_ + 16 * (int64_t)(ld_));                    \
    w2 = *reinterpret_cast<const float4*>(s_ + 32 * (int64_t)(ld_));                    \
    w3 = *reinterpret_cast<const float4*>(s_ + 48 * (int64_t)(ld_));                    \
  }
#define DESCO_WSTORE()                                                                  \
  {                                                                                     \
    float* d_ = Bs + (tid >> 4) * GAS + 4 * (tid & 15);                                 \
    *reinterpret_cast<float4*>(d_) = w0;                                                \
    *reinterpret_cast<float4*>(d_ + 16 * GAS) = w1;                                     \
    *reinterpret_cast<float4*>(d_ + 32 * GAS) = w2;                                     \
    *reinterpret_cast<float4*>(d_ + 48 * GAS) = w3;                                     \
  }
// 32 MFMAs of one 64-deep K block.  Lane (r = lane&31, h = lane>>5) owns k = 32h .. 32h+31 of
// its A row and of its B column (the sum over k is order-free, so the MFMA's k pairing can be
// chosen per lane half): 8 ds_read_b128 per operand instead of 32 ds_read_b32.
#define DESCO_MFMA_BLOCK(Aimg_)                                                                   \
  {                                                                                               \
    const float4* ap_ = reinterpret_cast<const float4*>((Aimg_) + (wr * 32 + (lane & 31)) * GAS + \
                                                        32 * (lane >> 5));                        \
    const float4* bp_ = reinterpret_cast<const float4*>(Bs + (wc * 32 + (lane & 31)) * GAS +      \
                                                        32 * (lane >> 5));                        \
    _Pragma("unroll") for (int t_ = 0; t_ < 8; ++t_) {                                            \
      const float4 a_ = ap_[t_], b_ = bp_[t_];                                                    \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.x, b_.x, acc, 0, 0, 0);                       \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.y, b_.y, acc, 0, 0, 0);                       \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.z, b_.z, acc, 0, 0, 0);                       \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.w, b_.w, acc, 0, 0, 0);                       \
    }                                                                                             \
  }
using bf16x8 = __attribute__((ext_vector_type(8))) short;
#define DESCO_ACC_ZERO() \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) acc[i_] = 0.f;

constexpr int PST = 72;     // bf16 plane row stride (144 B): 16-B aligned, conflict-free b128 reads
constexpr int PCAP = 256;   // neighbour records prefetched (one per thread) for the next tile

__global__ __launch_bounds__(256, 2) void gossip_fused_kernel(GossipFusedArgs g, int64_t num_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[3 * GT * GAS + 64 * GAS + 4 * GT + 68];
  float* A0 = lds;                      // h1, later y2
  float* A2 = lds + GT * GAS;           // neighbour staging, later h2, later head partials
  float* A1 = lds + 2 * GT * GAS;       // hh, later y1
  // post_mp.5 (64 -> 256, 44 % of the tile's MFMA work) runs fp32-accurately on the bf16 pipe
  // (bf16x6, see gemm_split.hip): y2 is written as three bf16 planes over A0 + the head of A2, the
  // pre-split W5 block planes go over the tail of A2 + A1 + the head of Bs (all dead by then).
  short* Y2P = reinterpret_cast<short*>(lds);            // [3][64][PST]
  short* W5P = Y2P + 3 * GT * PST;                       // [3][64][PST]
  float* Bs = lds + 3 * GT * GAS;       // weight block, n-major [64 n][64 k], stride GAS
  float4* srow = reinterpret_cast<float4*>(Bs + 64 * GAS);   // scalars of the tile rows
  int* rp = reinterpret_cast<int*>(Bs + 64 * GAS + 4 * GT);  // rowptr[n0 .. n0+64]
  int* ecol = reinterpret_cast<int*>(A2);                   // [ECAP]
  float4* escal = reinterpret_cast<float4*>(A2 + ECAP);     // [ECAP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int Q = g.Q;
  const int64_t nitems = num_tiles * Q;           // item = tile * Q + q: neighbours share a tile
  int64_t item = blockIdx.x;
  if (item >= nitems) return;

  // The (rowptr -> col -> scalar record) chain of the NEXT item is fetched into these registers
  // in three stages spread over the current item's GEMMs and published to LDS when the current
  // item is done: a tile no longer starts with three dependent global-memory latencies.
  float4 n_srow = make_float4(0.f, 0.f, 0.f, 0.f), n_scal = n_srow;
  int n_rp = 0, n_col = 0, n_ebeg = 0, n_cnt = 0;
#define DESCO_STAGE1(it_)                                                                  \
  {                                                                                        \
    const int q_ = (int)((it_) % Q);                                                       \
    const int64_t t0_ = ((it_) / Q) * GT;                                                  \
    const int nr_ = (int)((g.num_nodes - t0_) < GT ? (g.num_nodes - t0_) : GT);            \
    if (tid < GT) n_srow = g.scal[(t0_ + (tid < nr_ ? tid : nr_ - 1)) * Q + q_];           \
    if (tid <= GT) n_rp = g.rowptr[t0_ + (tid < nr_ ? tid : nr_)];                         \
    n_ebeg = g.rowptr[t0_];                                                                \
    n_cnt = g.rowptr[t0_ + nr_] - n_ebeg;                                                  \
    n_cnt = n_cnt < PCAP ? n_cnt : PCAP;                                                   \
  }
#define DESCO_STAGE2() \
  if (tid < n_cnt) n_col = g.col[n_ebeg + tid];
#define DESCO_STAGE3(it_) \
  if (tid < n_cnt) n_scal = g.scal[(int64_t)n_col * Q + (int)((it_) % Q)];

  DESCO_STAGE1(item)
  DESCO_STAGE2()
  DESCO_STAGE3(item)
  const float rc = g.r[lane], tc = g.t[lane];
  const int col = wc * 32 + (lane & 31);

  for (;;) {
    // ---- publish the prefetched tile data ------------------------------------------------------
    if (tid < GT) srow[tid] = n_srow;
    if (tid <= GT) rp[tid] = n_rp;
    if (tid < n_cnt) {
      ecol[tid] = n_col;
      escal[tid] = n_scal;
    }
    const int cnt0 = n_cnt;
    const int q = (int)(item % Q);
    const int64_t n0 = (item / Q) * GT;
    const int nrows = (int)((g.num_nodes - n0) < GT ? (g.num_nodes - n0) : GT);
    const int64_t next = item + gridDim.x;
    const bool has_next = next < nitems;
    __syncthreads();
    if (has_next) DESCO_STAGE1(next)

    // first weight block (W1 columns for hh) in flight while the tile is being assembled
    float4 w0, w1, w2, w3;
    DESCO_WLOAD(g.w1, 128)
    const float gq = g.g1[q];
    const float pc = g.p[q * 64 + lane], zc = g.z[q * 64 + lane];

    // ---- phase 1: h1 of the tile rows, gated neighbour sum hh -----------------------------------
    float hh[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) hh[k] = 0.f;
    const int ebeg = rp[0], eend = rp[GT];
    // pass 0: the prefetched records [ebeg, ebeg+cnt0); later passes (tiles with more than PCAP
    // neighbour records) stage ECAP records at a time from global memory
    int base = ebeg, cnt = cnt0;
    for (;;) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = wave * 16 + k;
        const int node = (int)n0 + row;
        int lo = rp[row] - base, hi = rp[row + 1] - base;
        lo = lo < 0 ? 0 : lo;
        hi = hi > cnt ? cnt : hi;
        float a = hh[k];
        for (int e = lo; e < hi; ++e) {
          const float4 sj = escal[e];
          float h = sj.x * pc + sj.y * rc + sj.w * tc + zc;
          h = h > 0.f ? h : 0.f;
          a += (ecol[e] < node ? gq : 1.f - gq) * h;
        }
        hh[k] = a;
      }
      base += cnt;
      if (base >= eend) break;
      __syncthreads();          // everyone is done with the staged records
      cnt = (eend - base) < ECAP ? (eend - base) : ECAP;
      for (int e = tid; e < cnt; e += 256) {
        const int j = g.col[base + e];
        ecol[e] = j;
        escal[e] = g.scal[(int64_t)j * Q + q];
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = wave * 16 + k;
      const float4 si = srow[row];
      float h = si.x * pc + si.y * rc + si.w * tc + zc;
      A0[row * GAS + lane] = h > 0.f ? h : 0.f;
      A1[row * GAS + lane] = hh[k];
    }

    f32x16 acc;

  // ---- G1: h2 = relu([hh|h1] W1 + a1*u + d1) -> A2 -----------------------------------------
  DESCO_ACC_ZERO()
  DESCO_WSTORE()
  __syncthreads();
  DESCO_WLOAD(g.w1 + 64, 128)
  DESCO_MFMA_BLOCK(A1)
  __syncthreads();
  DESCO_WSTORE()
  __syncthreads();
  DESCO_WLOAD(g.wp, 128)
  DESCO_MFMA_BLOCK(A0)
  {
    const float uc = g.u[col], dc = g.d1[col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const float v = acc[reg] + srow[row].z * uc + dc;
      A2[row * GAS + col] = v > 0.f ? v : 0.f;
    }
  }
  __syncthreads();
  if (has_next) DESCO_STAGE2()
  // ---- G2: y1 = leaky([h1|h2] Wp + x*tp + zp_q, 0.1) -> A1 ----------------------------------
  DESCO_ACC_ZERO()
  DESCO_WSTORE()
  __syncthreads();
  DESCO_WLOAD(g.wp + 64, 128)
  DESCO_MFMA_BLOCK(A0)
  __syncthreads();
  DESCO_WSTORE()
  __syncthreads();
  DESCO_WLOAD(g.w3, 64)
  DESCO_MFMA_BLOCK(A2)
  {
    const float tpc = g.tp[col], zpc = g.zp[q * 64 + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const float v = acc[reg] + srow[row].w * tpc + zpc;
      A1[row * GAS + col] = v > 0.f ? v : 0.1f * v;
    }
  }
  __syncthreads();
  if (has_next) DESCO_STAGE3(next)
  // ---- G3: y2 = relu(y1 W3 + b3) -> three bf16 planes (hi, mid, lo) -------------------------
  DESCO_ACC_ZERO()
  DESCO_WSTORE()
  __syncthreads();
  // first W5 column group (64 n x 64 k, three planes): thread t moves 16 shorts of row n = t>>2
  uint4 q0, q1, q2, q3, q4, q5;
#define DESCO_W5LOAD(cg_)                                                                     \
  {                                                                                           \
    const short* s_ = g.w5s + ((int64_t)(64 * (cg_) + (tid >> 2))) * 64 + 16 * (tid & 3);     \
    q0 = *reinterpret_cast<const uint4*>(s_);                                                 \
    q1 = *reinterpret_cast<const uint4*>(s_ + 8);                                             \
    q2 = *reinterpret_cast<const uint4*>(s_ + 256 * 64);                                      \
    q3 = *reinterpret_cast<const uint4*>(s_ + 256 * 64 + 8);                                  \
    q4 = *reinterpret_cast<const uint4*>(s_ + 2 * 256 * 64);                                  \
    q5 = *reinterpret_cast<const uint4*>(s_ + 2 * 256 * 64 + 8);                              \
  }
#define DESCO_W5STORE()                                                                       \
  {                                                                                           \
    short* d_ = W5P + (tid >> 2) * PST + 16 * (tid & 3);                                      \
    *reinterpret_cast<uint4*>(d_) = q0;                                                       \
    *reinterpret_cast<uint4*>(d_ + 8) = q1;                                                   \
    *reinterpret_cast<uint4*>(d_ + GT * PST) = q2;                                            \
    *reinterpret_cast<uint4*>(d_ + GT * PST + 8) = q3;                                        \
    *reinterpret_cast<uint4*>(d_ + 2 * GT * PST) = q4;                                        \
    *reinterpret_cast<uint4*>(d_ + 2 * GT * PST + 8) = q5;                                    \
  }
  DESCO_W5LOAD(0)
  DESCO_MFMA_BLOCK(A1)
  {
    const float bc = g.b3[col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      float v = acc[reg] + bc;
      v = v > 0.f ? v : 0.f;
      // truncation split: hi + mid + lo carries all 24 significand bits of v
      const uint32_t uh = __float_as_uint(v) & 0xffff0000u;
      const float r1 = v - __uint_as_float(uh);
      const uint32_t um = __float_as_uint(r1) & 0xffff0000u;
      const float r2 = r1 - __uint_as_float(um);
      short* d = Y2P + row * PST + col;
      d[0] = (short)(uh >> 16);
      d[GT * PST] = (short)(um >> 16);
      d[2 * GT * PST] = (short)(__float_as_uint(r2) >> 16);
    }
  }
  __syncthreads();
  // ---- G4/G5: head partials  sum_c relu(y2 W5 + b5)[c] * w7[c], 4 column groups of 64 --------
  float part[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) part[k] = 0.f;
#pragma unroll
  for (int cg = 0; cg < 4; ++cg) {
    DESCO_ACC_ZERO()
    DESCO_W5STORE()
    __syncthreads();
    if (cg < 3) DESCO_W5LOAD(cg + 1)
    {
      // lane (r = lane&31, h = lane>>5): A[row r][k = 16 s + 8 h + j], B[k = 16 s + 8 h + j][col r]
      const short* ya = Y2P + (wr * 32 + (lane & 31)) * PST + 8 * (lane >> 5);
      const short* wb = W5P + (wc * 32 + (lane & 31)) * PST + 8 * (lane >> 5);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ya + 16 * s4);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(ya + GT * PST + 16 * s4);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ya + 2 * GT * PST + 16 * s4);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wb + 16 * s4);
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(wb + GT * PST + 16 * s4);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wb + 2 * GT * PST + 16 * s4);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      }
    }
    const float bc = g.b5[cg * 64 + col], wv = g.w7[cg * 64 + col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const float v = acc[reg] + bc;
      part[reg] += (v > 0.f ? v : 0.f) * wv;
    }
    __syncthreads();
  }
#undef DESCO_W5LOAD
#undef DESCO_W5STORE
  // reduce the 64 column partials of every row through the (now free) A2 image
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    A2[row * GAS + col] = part[reg];
  }
  __syncthreads();
  {
    const int row = tid >> 2, qt = tid & 3;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += A2[row * GAS + qt * 16 + c];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (qt == 0 && row < nrows) g.out[(n0 + row) * Q + q] = s + g.b7 + srow[row].w;
  }
    if (!has_next) break;
    item = next;
    __syncthreads();            // A2 / srow / rp are free again
  }
#undef DESCO_STAGE1
#undef DESCO_STAGE2
#undef DESCO_STAGE3
}

#undef DESCO_WLOAD
#undef DESCO_WSTORE
#undef DESCO_MFMA_BLOCK
#undef DESCO_ACC_ZERO

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
  const int64_t blocks = (num_nodes + 3) / 4;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_scalars_f32: too many nodes");
  hipLaunchKernelGGL(gossip_scalars_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, rowptr, col, num_nodes, num_q, g0, g1,
                     reinterpret_cast<float4*>(scal4));
  return launch_status("desco_gossip_scalars_f32");
}

extern "C" int desco_gossip_fused_f32(const float* scal4, const int32_t* rowptr, const int32_t* col,
                                      int64_t num_nodes, int num_q, const float* g1, const float* p,
                                      const float* z, const float* zp, const float* r,
                                      const float* t, const float* u, const float* tp,
                                      const float* d1, const float* w1, const float* wp,
                                      const float* w3, const float* b3, const float* w5,
                                      const int16_t* w5_planes, const float* b5, const float* w7,
                                      float b7, float* out, desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  auto mis16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) != 0; };
  if (!scal4 || !rowptr || !g1 || !p || !z || !zp || !r || !t || !u || !tp || !d1 || !w1 || !wp ||
      !w3 || !b3 || !w5 || !w5_planes || !b5 || !w7 || !out || num_nodes < 0 || num_q < 1 || num_q > 65535 ||
      mis16(scal4) || mis16(w1) || mis16(wp) || mis16(w3) || mis16(w5) || mis16(w5_planes))
    return fail(DESCO_EINVAL, "desco_gossip_fused_f32: bad argument");
  const int64_t bx = (num_nodes + GT - 1) / GT;
  if (bx > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_fused_f32: too many nodes");
  GossipFusedArgs a{reinterpret_cast<const float4*>(scal4), rowptr, col, num_nodes, num_q, g1, p, z,
                    zp, r, t, u, tp, d1, w1, wp, w3, b3, w5, reinterpret_cast<const short*>(w5_planes), b5, w7, b7, out};
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  const int64_t nitems = bx * num_q;
  const unsigned grid = (unsigned)(nitems < 2 * (int64_t)cus ? nitems : 2 * (int64_t)cus);
  hipLaunchKernelGGL(gossip_fused_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, bx);
  return launch_status("desco_gossip_fused_f32");
}
