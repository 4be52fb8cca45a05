// fp32-accurate GEMM on the bf16 matrix pipe ("bf16x6"): every fp32 operand is split into three
// bf16 terms (truncation: hi + mid + lo reproduces all 24 significand bits) and the six products
// whose weight is >= 2^-16 are accumulated in fp32:
//     a*b ~= hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)            (error ~2^-23 |a||b|)
// v_mfma_f32_32x32x16_bf16 runs at 16x the f32 MFMA rate, so six of them per 16-deep step cost
// 192 cycles against 512 for eight v_mfma_f32_32x32x2_f32 -- and the operand fragments are 16-byte
// LDS reads.  Same contract as desco_gemm_f32 except for the weight operand: it is passed already
// split, as three n-major bf16 planes w_planes[3][n][k] (desco_split_bf16x3_f32 makes them once per
// weight version), so only the activation operand is split in the kernel.
//
// The split is VALU work (~6 instructions per float) and VALU does not overlap MFMA on a SIMD, so
// the tile is made wide: 256 threads = 2x2 waves, block tile 128 rows x 64*WN cols (WN = 3, 2, 1 ->
// 192, 128, 64), wave tile 64 x 32*WN = 2 x WN accumulators; one split of an A chunk then feeds
// 6*WN MFMAs per 16-deep step.  K chunks of 32: A global float4 -> split in registers -> three bf16
// planes in LDS; W planes are copied global -> registers -> LDS; rows are 80 bytes ([32+8] bf16):
// conflict-free ds_read_b128.  Block ids are laid out XCD-aware with the column tile fastest, so the
// n-tiles that share an A row panel run back to back on the same XCD and re-read it from its L2.
#include "common_device.hpp"

namespace desco {

struct GemmSplitArgs {
  const float* a1;
  int64_t lda1;
  int k1;
  const float* a2;
  int64_t lda2;
  int k2;
  const short* w;     // planes [3][n][k1+k2] (hi, mid, lo)
  int n;
  const float* bias;
  int bias_rows;
  const float* s;
  int ns;
  const float* ws;
  int act;
  float slope;
  float* c;
  int64_t ldc;
  int64_t m;
  // POOLA instantiation (desco_pool_post_bf16x6_f32): the A operand is never materialised -- row b, column 64 l + c is
  //   a1[b, 64 l + c]  +  (l == 0 ? rows(b) * x0[c] : sum over the <= 3 tiles of segment b of part[l][slot][c])
  // i.e. the anchor rows plus the fused pooling's partial sums (desco_pool_reduce_f32's arithmetic, same order)
  const int32_t* seg_ptr;
  const uint32_t* pool_bits;
  const int32_t* pool_slot;
  const float* part[9];     // [1..8]: the layers' partial arrays [slots][64]
  const float* x0;          // [64]: the constant first block's row
  // training backward (desco_gemm_bf16x6_desc_f32; desco_gemm_desc's meaning): c = v * dropout factor * act'(gate)
  const float* gate;
  int64_t ldg;
  int gate_act;
  float gate_slope;
  DropArgs drop;
  int ws_rows;              // > 1: the scalar tail's matrix is per row class, ws[row % ws_rows][ns][n] (the gossip step's
                            // per-query tables: desco_affine_rows_f32 fused into the product's epilogue)
};

__device__ __attribute__((aligned(16))) float gs_zero_row[64] = {};

using bf16x8 = __attribute__((ext_vector_type(8))) short;

#define GS_SPLIT(a_, b_, h_, m_, l_) split2_bf16x3(a_, b_, h_, m_, l_)

constexpr int SBK = 32, SST = 32;              // SST: plane row stride in bf16 (64 B, no padding)
// 16-byte chunk c (0..3) of plane row r sits at chunk c ^ ((r >> 3) & 3): conflict-free for the fragment
// reads (ds_read_b128 lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}: their rows differ in r & 3 or in
// (r >> 3) & 3) and for the staging stores (two whole 64-B rows per 128-B write window; the 80-B padded
// rows were 2-way conflicted on every store: tools/micro/lds_banks.py, SQ_LDS_BANK_CONFLICT 0.33)
__device__ __forceinline__ int gs_chunk(const int row, const int c) { return ((c ^ (row >> 3)) & 3) << 3; }

// NP = 3: hi/mid/lo planes and six products (fp32-accurate); NP = 1: one round-to-nearest bf16 plane
// and one product (plain bf16 MFMA with fp32 accumulation, the bf16 training mode).
// BM = rows per block tile (128: 4 waves, two blocks per CU).  The A chunk AFTER the next one is kept
// in flight too: one chunk of MFMAs (2 304 cycles) does not cover an HBM round trip under load, two do
// (144 -> 156 TF/s fp32-equivalent on the 576^2 anchor GEMM).
// (the kernel's body: one block tile of problem g; `id` = the block's index among the problem's blocks, a multiple of 8
//  blocks per problem so that id & 7 is the XCD the hardware dispatched the block to)
template <int WN, int NP, int BM, bool POOLA = false>
__device__ __forceinline__ void gemm_split_body(const GemmSplitArgs& g, const int64_t gm, const int ny, const int64_t id) {
  constexpr int NT = 2 * BM;                   // threads
  constexpr int BN = 64 * WN, BPLANE = BN * SST, APLANE = BM * SST;
  constexpr int AR = BM / 4;                   // A staging: row step between a thread's 4 rows
  constexpr int BR = NT / 4;                   // W staging: rows covered by one pass of the block
  constexpr int BJ = (BN + BR - 1) / BR;       // passes over the BN weight rows
  extern __shared__ __attribute__((aligned(16))) short lds[];
  short* Ap = lds;                  // planes hi, mid, lo of the A chunk [BM][40]
  short* Bp = lds + NP * APLANE;    // planes hi, mid, lo of the W chunk [BN][40]

  // block id -> (m tile, n tile): ids id, id+8, id+16, ... share an XCD (round-robin dispatch);
  // within an XCD the n tile runs fastest so an A row panel is fetched from HBM once per XCD.
  const int64_t local = id >> 3;
  const int64_t mt = (local / ny) * 8 + (id & 7);
  if (mt >= gm) return;
  const int n0 = (int)(local % ny) * BN;
  const int64_t m0 = mt * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int K = g.k1 + g.k2;
  const int nchunks = K / SBK;

  // staging maps: A BM rows x 8 float4 -> 4 per thread; W planes BN rows x 4 uint4 -> BJ per plane
  const int arow = tid >> 3, ac4 = tid & 7;
  const int64_t mlast = g.m - 1;
  int64_t r0 = m0 + arow, r1 = r0 + AR, r2 = r0 + 2 * AR, r3 = r0 + 3 * AR;
  r0 = r0 < g.m ? r0 : mlast;
  r1 = r1 < g.m ? r1 : mlast;
  r2 = r2 < g.m ? r2 : mlast;
  r3 = r3 < g.m ? r3 : mlast;
  // (named scalars, not arrays: hipcc parks indexed register arrays in scratch)
  const float* p10 = g.a1 + r0 * g.lda1 + 4 * ac4;
  const float* p11 = g.a1 + r1 * g.lda1 + 4 * ac4;
  const float* p12 = g.a1 + r2 * g.lda1 + 4 * ac4;
  const float* p13 = g.a1 + r3 * g.lda1 + 4 * ac4;
  const float* p20 = g.k2 ? g.a2 + r0 * g.lda2 + 4 * ac4 - g.k1 : p10;
  const float* p21 = g.k2 ? g.a2 + r1 * g.lda2 + 4 * ac4 - g.k1 : p11;
  const float* p22 = g.k2 ? g.a2 + r2 * g.lda2 + 4 * ac4 - g.k1 : p12;
  const float* p23 = g.k2 ? g.a2 + r3 * g.lda2 + 4 * ac4 - g.k1 : p13;
  const int brow = tid >> 2, bpart = tid & 3;
  const short* pw = g.w + (int64_t)(n0 + brow) * K + 8 * bpart;
  const int64_t wplane = (int64_t)g.n * K;
  const int64_t wj = (int64_t)BR * K;
  // weight rows brow + j*BR, j < BJ; the last pass may cover only part of the block's threads
  const bool v0 = BR <= BN || brow < BN;
  const bool v1 = BJ > 1 && brow + BR < BN;
  const bool v2 = BJ > 2 && brow + 2 * BR < BN;

  float4 ra0, ra1, ra2, ra3;                   // A chunk about to be stored
  float4 rn0, rn1, rn2, rn3;                   // A chunk after it (two chunks of latency cover)
  // POOLA: the segment's partial rows of the chunk's layer, three per row (absent ones read a row of zeros), for both
  // chunks in flight; per row the offsets of its (at most three) slots and its row count
  float4 qa00, qa01, qa02, qa10, qa11, qa12, qa20, qa21, qa22, qa30, qa31, qa32;
  float4 qn00, qn01, qn02, qn10, qn11, qn12, qn20, qn21, qn22, qn30, qn31, qn32;
  int so00 = 0, so01 = -1, so02 = -1, so10 = 0, so11 = -1, so12 = -1, so20 = 0, so21 = -1, so22 = -1, so30 = 0,
      so31 = -1, so32 = -1;                    // (float offsets slot * 64: the entry point checks they fit 31 bits)
  float nb0 = 0.f, nb1 = 0.f, nb2 = 0.f, nb3 = 0.f;
  float4 x0a = make_float4(0.f, 0.f, 0.f, 0.f), x0b = x0a;
  if constexpr (POOLA) {
#define DESCO_POOL_ROW(j_, r_)                                                                                  \
  {                                                                                                             \
    const int a_ = g.seg_ptr[r_], e_ = g.seg_ptr[(r_) + 1];                                                     \
    const int t0_ = a_ >> 4, t1_ = (e_ - 1) >> 4, f_ = a_ - (t0_ << 4);                                         \
    so##j_##0 = (g.pool_slot[t0_] + __popc(g.pool_bits[t0_] & ((1u << f_) - 1u))) * 64;                         \
    so##j_##1 = t1_ > t0_ ? g.pool_slot[t0_ + 1] * 64 : -1;                                                     \
    so##j_##2 = t1_ > t0_ + 1 ? g.pool_slot[t0_ + 2] * 64 : -1;                                                 \
    nb##j_ = (float)(e_ - a_);                                                                                  \
  }
    DESCO_POOL_ROW(0, r0) DESCO_POOL_ROW(1, r1) DESCO_POOL_ROW(2, r2) DESCO_POOL_ROW(3, r3)
#undef DESCO_POOL_ROW
    x0a = *reinterpret_cast<const float4*>(g.x0 + 4 * ac4);
    x0b = *reinterpret_cast<const float4*>(g.x0 + 32 + 4 * ac4);
  }
  (void)qa00; (void)qn00; (void)nb0; (void)x0a; (void)x0b;
  uint4 rb00, rb01, rb02, rb10, rb11, rb12, rb20, rb21, rb22;   // W planes of the next chunk
  rb00 = rb01 = rb02 = rb10 = rb11 = rb12 = rb20 = rb21 = rb22 = make_uint4(0, 0, 0, 0);
#define DESCO_LOAD_A(d_, kk_)                                                                 \
  {                                                                                           \
    const int k_ = (kk_);                                                                     \
    const bool s1_ = k_ < g.k1;                                                               \
    d_##0 = *reinterpret_cast<const float4*>((s1_ ? p10 : p20) + k_);                         \
    d_##1 = *reinterpret_cast<const float4*>((s1_ ? p11 : p21) + k_);                         \
    d_##2 = *reinterpret_cast<const float4*>((s1_ ? p12 : p22) + k_);                         \
    d_##3 = *reinterpret_cast<const float4*>((s1_ ? p13 : p23) + k_);                         \
  }
// POOLA: the partial rows of chunk kk_ (layer l = kk_ / 64; block 0 has none: its term is rows(b) * x0)
#define DESCO_LOAD_Q1(d_, j_, P_)                                                             \
  {                                                                                           \
    d_##j_##0 = *reinterpret_cast<const float4*>((P_) ? (P_) + so##j_##0 : zr_);              \
    d_##j_##1 = *reinterpret_cast<const float4*>((P_) && so##j_##1 >= 0 ? (P_) + so##j_##1 : zr_);  \
    d_##j_##2 = *reinterpret_cast<const float4*>((P_) && so##j_##2 >= 0 ? (P_) + so##j_##2 : zr_);  \
  }
#define DESCO_LOAD_Q(d_, kk_)                                                                 \
  if constexpr (POOLA) {                                                                      \
    const int kq_ = (kk_);                                                                    \
    const int lq_ = kq_ >> 6;                                                                 \
    const float* zr_ = gs_zero_row + 4 * ac4;                                                 \
    const float* pq_ = lq_ > 0 ? g.part[lq_] + (kq_ & 63) + 4 * ac4 : nullptr;                \
    DESCO_LOAD_Q1(d_, 0, pq_) DESCO_LOAD_Q1(d_, 1, pq_) DESCO_LOAD_Q1(d_, 2, pq_) DESCO_LOAD_Q1(d_, 3, pq_)  \
  }
// pooled row = ((p0 + p1) + p2) + anchor row (desco_pool_reduce_f32's order); block 0: rows(b) * x0 + anchor row, the
// product rounded on its own (as desco_degree_affine_f32 forms it)
#define DESCO_POOL_SUM(v_, j_, xq_)                                                           \
  {                                                                                           \
    float4 t_;                                                                                \
    t_.x = (((qa##j_##0).x + (qa##j_##1).x) + (qa##j_##2).x) + __fmul_rn(nb##j_, (xq_).x);          \
    t_.y = (((qa##j_##0).y + (qa##j_##1).y) + (qa##j_##2).y) + __fmul_rn(nb##j_, (xq_).y);          \
    t_.z = (((qa##j_##0).z + (qa##j_##1).z) + (qa##j_##2).z) + __fmul_rn(nb##j_, (xq_).z);          \
    t_.w = (((qa##j_##0).w + (qa##j_##1).w) + (qa##j_##2).w) + __fmul_rn(nb##j_, (xq_).w);          \
    v_.x = t_.x + v_.x;                                                                       \
    v_.y = t_.y + v_.y;                                                                       \
    v_.z = t_.z + v_.z;                                                                       \
    v_.w = t_.w + v_.w;                                                                       \
  }
#define DESCO_LOAD_WJ(j_, v_)                                                                 \
  if (BJ > (j_) && (v_)) {                                                                    \
    rb##j_##0 = *reinterpret_cast<const uint4*>(w_ + (j_) * wj);                              \
    if constexpr (NP == 3) {                                                                  \
      rb##j_##1 = *reinterpret_cast<const uint4*>(w_ + (j_) * wj + wplane);                   \
      rb##j_##2 = *reinterpret_cast<const uint4*>(w_ + (j_) * wj + 2 * wplane);               \
    }                                                                                         \
  }
#define DESCO_LOAD_W(kk_)                                                                     \
  {                                                                                           \
    const short* w_ = pw + (kk_);                                                             \
    DESCO_LOAD_WJ(0, v0) DESCO_LOAD_WJ(1, v1) DESCO_LOAD_WJ(2, v2)                            \
  }
#define DESCO_PUT(row_, v_)                                                                   \
  {                                                                                           \
    short* d_ = Ap + (row_)*SST + gs_chunk((row_), ac4 >> 1) + 4 * (ac4 & 1);                 \
    if constexpr (NP == 3) {                                                                  \
      uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                                  \
      GS_SPLIT(v_.x, v_.y, h0_, m0_, l0_);                                                    \
      GS_SPLIT(v_.z, v_.w, h1_, m1_, l1_);                                                    \
      *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                                   \
      *reinterpret_cast<uint2*>(d_ + APLANE) = make_uint2(m0_, m1_);                          \
      *reinterpret_cast<uint2*>(d_ + 2 * APLANE) = make_uint2(l0_, l1_);                      \
    } else {                                                                                  \
      *reinterpret_cast<uint2*>(d_) = make_uint2(pack2_bf16_rne(v_.x, v_.y),                  \
                                                 pack2_bf16_rne(v_.z, v_.w));                 \
    }                                                                                         \
  }
#define DESCO_STORE_WJ(j_, v_)                                                                \
  if (BJ > (j_) && (v_)) {                                                                    \
    short* bj_ = Bp + (brow + (j_) * BR) * SST + gs_chunk(brow + (j_) * BR, bpart);           \
    *reinterpret_cast<uint4*>(bj_) = rb##j_##0;                                               \
    if constexpr (NP == 3) {                                                                  \
      *reinterpret_cast<uint4*>(bj_ + BPLANE) = rb##j_##1;                                    \
      *reinterpret_cast<uint4*>(bj_ + 2 * BPLANE) = rb##j_##2;                                \
    }                                                                                         \
  }
#define DESCO_STORE_CHUNK()                                                                   \
  {                                                                                           \
    DESCO_PUT(arow, ra0)                                                                      \
    DESCO_PUT(arow + AR, ra1)                                                                 \
    DESCO_PUT(arow + 2 * AR, ra2)                                                             \
    DESCO_PUT(arow + 3 * AR, ra3)                                                             \
    DESCO_STORE_WJ(0, v0) DESCO_STORE_WJ(1, v1) DESCO_STORE_WJ(2, v2)                         \
  }

  f32x16 acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  DESCO_LOAD_A(ra, 0)
  DESCO_LOAD_Q(qa, 0)
  DESCO_LOAD_W(0)
  DESCO_LOAD_A(rn, (nchunks > 1 ? 1 : 0) * SBK)
  DESCO_LOAD_Q(qn, (nchunks > 1 ? 1 : 0) * SBK)
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch > 0) __syncthreads();          // previous chunk's fragments have been read
    if constexpr (POOLA) {
      const float4 zq_ = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 xq_ = ch == 0 ? x0a : ch == 1 ? x0b : zq_;
      DESCO_POOL_SUM(ra0, 0, xq_) DESCO_POOL_SUM(ra1, 1, xq_) DESCO_POOL_SUM(ra2, 2, xq_) DESCO_POOL_SUM(ra3, 3, xq_)
    }
    DESCO_STORE_CHUNK()
    __syncthreads();
    const int chn = ch + 1 < nchunks ? ch + 1 : ch;
    const int chnn = ch + 2 < nchunks ? ch + 2 : chn;
    ra0 = rn0; ra1 = rn1; ra2 = rn2; ra3 = rn3;
    if constexpr (POOLA) {
      qa00 = qn00; qa01 = qn01; qa02 = qn02; qa10 = qn10; qa11 = qn11; qa12 = qn12;
      qa20 = qn20; qa21 = qn21; qa22 = qn22; qa30 = qn30; qa31 = qn31; qa32 = qn32;
    }
    DESCO_LOAD_W(chn * SBK)                // in flight under the MFMAs
    DESCO_LOAD_A(rn, chnn * SBK)           // two chunks ahead (HBM latency)
    DESCO_LOAD_Q(qn, chnn * SBK)
    // lane (r = lane&31, h = lane>>5): A[row r][k = 16 s + 8 h + j], B[k = 16 s + 8 h + j][col r]
    // (row + 32 i / 32 j keeps (row >> 3) & 3, so one swizzle per lane serves every tile)
    const int fsw = (lane >> 3) & 3, fh = lane >> 5;
    const short* ap = Ap + (wr * 64 + (lane & 31)) * SST;
    const short* bp = Bp + (wc * 32 * WN + (lane & 31)) * SST;
    {
#pragma unroll
    for (int s = 0; s < SBK / 16; ++s) {
      bf16x8 ah[2], am[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int co = (((2 * s + fh) ^ fsw) & 3) << 3;
        ah[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * SST + co);
        if constexpr (NP == 3) {
          am[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * SST + APLANE + co);
          al[i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * SST + 2 * APLANE + co);
        }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const short* bt = bp + j * 32 * SST + ((((2 * s + fh) ^ fsw) & 3) << 3);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bt);
        if constexpr (NP == 3) {
          const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bt + BPLANE);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bt + 2 * BPLANE);
          // smallest terms first; the two row tiles alternate so dependent MFMAs are never adjacent
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bh, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1], bh, acc[1][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bl, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bl, acc[1][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], bm, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], bm, acc[1][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], bh, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], bh, acc[1][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bm, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bm, acc[1][j], 0, 0, 0);
        }
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bh, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bh, acc[1][j], 0, 0, 0);
      }
    }
    }
  }
#undef DESCO_LOAD_A
#undef DESCO_LOAD_Q
#undef DESCO_LOAD_Q1
#undef DESCO_POOL_SUM
#undef DESCO_LOAD_W
#undef DESCO_LOAD_WJ
#undef DESCO_PUT
#undef DESCO_STORE_CHUNK
#undef DESCO_STORE_WJ

  // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), i.e. a
  // lane holds single columns: stored as they sit, a wave instruction writes 2 x 128 B and the 96 of them
  // per wave made the stores (not the MFMAs) the longest part of the anchor GEMM (no-MFMA build 3.7 ms with
  // and 1.75 ms without them, of 4.8).  So each 32-row half of the wave tile goes through a wave-private
  // [32][32 WN] fp32 image in the (now idle) staging LDS and leaves as 16 bytes per lane: one instruction
  // covers 1 KB of whole 128-B lines.
  __syncthreads();                       // every wave is done with the last chunk's fragments
  constexpr int EW = 32 * WN;            // floats per staged row
  float* st = reinterpret_cast<float*>(lds) + wave * (32 * EW);
  const int col = lane & 31;
  const bool wide = ((reinterpret_cast<uintptr_t>(g.c) & 15) == 0) && ((g.ldc & 3) == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t grow0 = m0 + wr * 64 + 32 * i;
    // row class of this lane's first row (ws_rows > 1): its other 15 rows are 1..27 rows further on -- one division per
    // lane and half tile, then add and wrap
    const uint32_t wsr = g.ws_rows > 1 ? (uint32_t)g.ws_rows : 1u;
    const uint32_t qbase = g.ws_rows > 1 ? (uint32_t)(grow0 + 4 * (lane >> 5)) % wsr : 0u;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int gcol = n0 + wc * EW + 32 * j + col;
      float wsv[4] = {0.f, 0.f, 0.f, 0.f};
      if (g.ws_rows <= 1)
        for (int q = 0; q < g.ns; ++q) wsv[q] = g.ws[(int64_t)q * g.n + gcol];
      const float b_single = (g.bias && g.bias_rows == 1) ? g.bias[gcol] : 0.f;
      // registers 4q..4q+3 of a lane are four consecutive rows (aligned to 4) of one column: one Philox call per quad
      float fac[16];
      if (g.drop.key) {
        const uint64_t seed = g.drop.key[0], step = g.drop.key[1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t r4 = (grow0 + 8 * q + 4 * (lane >> 5)) >> 2;
          const PhiloxOut o = dropout_bits4(g.drop, seed, step, (uint32_t)r4, (uint32_t)gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) fac[4 * q + e] = o.w[e] < g.drop.threshold ? 0.f : g.drop.scale;
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int64_t grow = grow0 + row < g.m ? grow0 + row : g.m - 1;
        float v = acc[i][j][reg];
        if (g.bias) {
          if (g.bias_rows == 1)
            v += b_single;
          else
            v += g.bias[(grow % g.bias_rows) * g.n + gcol];
        }
        if (g.ws_rows > 1) {
          uint32_t qc_ = qbase + (uint32_t)((reg & 3) + 8 * (reg >> 2));
          while (qc_ >= wsr) qc_ -= wsr;
          const float* wr_ = g.ws + (int64_t)qc_ * g.ns * g.n + gcol;
          for (int q = 0; q < g.ns; ++q) v = fmaf(g.s[grow * g.ns + q], wr_[(int64_t)q * g.n], v);
        } else {
          for (int q = 0; q < g.ns; ++q) v += g.s[grow * g.ns + q] * wsv[q];
        }
        v = apply_act(v, g.act, g.slope);
        if (g.drop.key) v *= fac[reg];
        if (g.gate) {
          const float o_ = g.gate[grow * g.ldg + gcol];
          v = o_ > 0.f ? v : (g.gate_act == DESCO_ACT_RELU ? 0.f : g.gate_act == DESCO_ACT_LEAKY ? v * g.gate_slope : v);
        }
        st[row * EW + 32 * j + col] = v;
      }
    }
    __syncthreads();
    float* crow = g.c + n0 + wc * EW;
#pragma unroll
    for (int p = 0; p < 4 * WN; ++p) {
      const int idx = lane + 64 * p;               // float4 index in the image: row idx / (8 WN)
      const int row = idx / (8 * WN), c4 = idx % (8 * WN);
      const float4 v = *reinterpret_cast<const float4*>(st + 4 * idx);
      const int64_t grow = grow0 + row;
      if (grow < g.m) {
        float* o = crow + grow * g.ldc + 4 * c4;
        if (wide) {
          __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(o));
        } else {
          o[0] = v.x;
          o[1] = v.y;
          o[2] = v.z;
          o[3] = v.w;
        }
      }
    }
    if (i == 0) __syncthreads();
  }
}

template <int WN, int NP, int BM, bool POOLA = false>
__global__ __launch_bounds__(2 * BM) __attribute__((amdgpu_waves_per_eu(2))) void gemm_split_kernel(GemmSplitArgs g, int64_t gm, int ny) {
  gemm_split_body<WN, NP, BM, POOLA>(g, gm, ny, blockIdx.x);
}

// Up to four INDEPENDENT products in one launch (desco_gemm_bf16x6_multi_f32: the count-row and canonical-row halves of a
// training layer), 128 x 64 tiles for every problem; workgroups [blk_end[i-1], blk_end[i]) belong to problem i.
constexpr int kSplitMulti = 4;
struct GemmSplitMulti {
  const float* a1[kSplitMulti];
  const float* a2[kSplitMulti];
  const short* w[kSplitMulti];
  const float* bias[kSplitMulti];
  const float* gate[kSplitMulti];
  float* c[kSplitMulti];
  int64_t lda1[kSplitMulti], lda2[kSplitMulti], ldc[kSplitMulti], ldg[kSplitMulti], m[kSplitMulti], gm[kSplitMulti];
  int k1[kSplitMulti], k2[kSplitMulti], n[kSplitMulti], act[kSplitMulti], gate_act[kSplitMulti], ny[kSplitMulti];
  float slope[kSplitMulti], gate_slope[kSplitMulti];
  DropArgs drop[kSplitMulti];
  int blk_end[kSplitMulti];
  int num;
};

template <int NP, int BM>
__global__ __launch_bounds__(2 * BM) __attribute__((amdgpu_waves_per_eu(2))) void gemm_split_multi_kernel(const GemmSplitMulti q) {
  int b = blockIdx.x, i = 0;
  while (i < q.num - 1 && b >= q.blk_end[i]) ++i;
  b -= i ? q.blk_end[i - 1] : 0;
  const GemmSplitArgs g{q.a1[i], q.lda1[i], q.k1[i], q.a2[i], q.lda2[i], q.k2[i], q.w[i], q.n[i], q.bias[i], 1, nullptr, 0,
                        nullptr, q.act[i], q.slope[i], q.c[i], q.ldc[i], q.m[i], nullptr, nullptr, nullptr, {}, nullptr,
                        q.gate[i], q.ldg[i], q.gate_act[i], q.gate_slope[i], q.drop[i], 1};
  gemm_split_body<1, NP, BM, false>(g, q.gm[i], q.ny[i], b);
}

// w[count] -> round-to-nearest-even bf16 bit patterns
__global__ __launch_bounds__(256) void round_bf16_kernel(const float* __restrict__ w, int64_t count,
                                                         short* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  out[i] = (short)(pack2_bf16_rne(w[i], 0.f) & 0xffffu);
}

// w[count] -> planes[3][count] (hi, mid, lo bf16 bit patterns of the truncation split)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ w, int64_t count,
                                                            short* __restrict__ planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float f = w[i];
  const uint32_t uh = __float_as_uint(f) & 0xffff0000u;
  const float r1 = f - __uint_as_float(uh);
  const uint32_t um = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(um);
  planes[i] = (short)(uh >> 16);
  planes[count + i] = (short)(um >> 16);
  planes[2 * count + i] = (short)(__float_as_uint(r2) >> 16);
}

template <int WN, int NP, int BM>
static int launch_gemm_split_bm(const GemmSplitArgs& g, hipStream_t stream) {
  constexpr int BN = 64 * WN;
  constexpr size_t stage_bytes = (size_t)(NP * BM * SST + NP * BN * SST) * sizeof(short);
  constexpr size_t epi_bytes = (size_t)(BM / 32) * 32 * 32 * WN * sizeof(float);   // one [32][32 WN] image per wave
  constexpr size_t lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<WN, NP, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "desco_gemm_bf16x6_f32: cannot size LDS");
    attr_once.mark();
  }
  const int64_t gm = (g.m + BM - 1) / BM;
  const int ny = g.n / BN;
  const int64_t blocks = ((gm + 7) / 8) * 8 * ny;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gemm_bf16x6_f32: m too large");
  hipLaunchKernelGGL((gemm_split_kernel<WN, NP, BM>), dim3((unsigned)blocks), dim3(2 * BM), lds_bytes,
                     stream, g, gm, ny);
  return launch_status("desco_gemm_bf16x6_f32");
}

// BM = 256 (8 waves, half the weight-plane traffic per flop) measured 3 % SLOWER than BM = 128 on the
// 576^2 anchor GEMM (152 vs 156 TF/s fp32-equivalent, same box): the weight stream from L2 is not
// what bounds the kernel, so every shape takes the 128-row tile (two blocks per CU).
template <int WN, int NP>
static int launch_gemm_split(const GemmSplitArgs& g, hipStream_t stream) {
  return launch_gemm_split_bm<WN, NP, 128>(g, stream);
}

}  // namespace desco

static int gemm_planes(const char* who, int np, const float* a1, int64_t lda1, int k1, const float* a2,
                       int64_t lda2, int k2, const int16_t* w, int n, const float* bias,
                       int bias_rows, const float* s, int ns, const float* ws, int act, float slope,
                       float* c, int64_t ldc, int64_t m, desco_stream_t stream, const float* gate = nullptr,
                       int64_t ldg = 0, int gate_act = 0, float gate_slope = 0.f,
                       desco::DropArgs drop = desco::DropArgs{nullptr, 0u, 0u, 1.f}, int ws_rows = 1) {
  using namespace desco;
  if (m == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (m < 0 || !a1 || !w || !c || k1 <= 0 || k1 % SBK || k2 < 0 || k2 % SBK || n <= 0 ||
      n % 64 || (k2 > 0 && !a2) || ns < 0 || ns > 4 || (ns > 0 && (!s || !ws)) ||
      (bias && bias_rows < 1) || lda1 % 4 || (k2 > 0 && lda2 % 4) || mis16(a1) ||
      (k2 > 0 && mis16(a2)) || mis16(w))
    return fail(DESCO_EINVAL, who);
  GemmSplitArgs g{a1, lda1, k1, a2, lda2, k2, reinterpret_cast<const short*>(w), n, bias,
                  bias ? bias_rows : 1, s, ns, ws, act, slope, c, ldc, m, nullptr, nullptr, nullptr, {}, nullptr,
                  gate, ldg, gate_act, gate_slope, drop, ws_rows};
  hipStream_t st = (hipStream_t)stream;
  if (np == 3) {
    if (n % 192 == 0) return launch_gemm_split<3, 3>(g, st);
    if (n % 128 == 0) return launch_gemm_split<2, 3>(g, st);
    return launch_gemm_split<1, 3>(g, st);
  }
  if (n % 192 == 0) return launch_gemm_split<3, 1>(g, st);
  if (n % 128 == 0) return launch_gemm_split<2, 1>(g, st);
  return launch_gemm_split<1, 1>(g, st);
}

extern "C" int desco_gemm_bf16x6_f32(const float* a1, int64_t lda1, int k1, const float* a2,
                                     int64_t lda2, int k2, const int16_t* w_planes, int n,
                                     const float* bias, int bias_rows, const float* s, int ns,
                                     const float* ws, int act, float slope, float* c, int64_t ldc,
                                     int64_t m, desco_stream_t stream) {
  return gemm_planes("desco_gemm_bf16x6_f32: bad argument (k%32, n%64, 16-byte alignment)", 3, a1, lda1,
                     k1, a2, lda2, k2, w_planes, n, bias, bias_rows, s, ns, ws, act, slope, c, ldc, m,
                     stream);
}

// One descriptor of desco_gemm_f32_multi's form on the bf16x6 pipe (training: the gossip step's forward and input-gradient
// products, with the activation-derivative gate and the dropout factor in the epilogue)
extern "C" int desco_gemm_bf16x6_desc_f32(const desco_gemm_desc* d, const int16_t* w_planes, int ws_rows,
                                          desco_stream_t stream) {
  if (!d || !w_planes) return desco::fail(DESCO_EINVAL, "desco_gemm_bf16x6_desc_f32: null descriptor / planes");
  if (d->accum || (d->gate && (d->ldg < d->n)) || ws_rows < 1 || d->m >= ((int64_t)1 << 31))
    return desco::fail(DESCO_EINVAL, "desco_gemm_bf16x6_desc_f32: accum is not supported; gate rows shorter than n; "
                                     "ws_rows >= 1; m < 2^31");
  const desco::DropArgs drop{d->drop.key, d->drop.site, d->drop.threshold, d->drop.scale};
  return gemm_planes("desco_gemm_bf16x6_desc_f32: bad argument (k%32, n%64, 16-byte alignment)", 3, d->a1, d->lda1,
                     d->k1, d->a2, d->lda2, d->k2, w_planes, d->n, d->bias, d->bias_rows, d->s, d->ns, d->ws, d->act,
                     d->slope, d->c, d->ldc, d->m, stream, d->gate, d->ldg, d->gate_act, d->gate_slope, drop, ws_rows);
}

// planes[3][n][k] of the TRANSPOSE of w [k, n] (row stride ldw): the n-major operand of the bf16x6 products from a weight
// kept as [in, out] (the training step's folded weights) or, for an input-gradient product, from torch's [out, in]
namespace desco {
__global__ __launch_bounds__(256) void split_bf16x3_t_kernel(const float* __restrict__ w, int k, int n, int64_t ldw,
                                                             short* __restrict__ planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // output index: row nn, column kk
  const int64_t count = (int64_t)k * n;
  if (i >= count) return;
  const int nn = (int)(i / k), kk = (int)(i % k);
  const float f = w[(int64_t)kk * ldw + nn];
  const uint32_t uh = __float_as_uint(f) & 0xffff0000u;
  const float r1 = f - __uint_as_float(uh);
  const uint32_t um = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(um);
  planes[i] = (short)(uh >> 16);
  planes[count + i] = (short)(um >> 16);
  planes[2 * count + i] = (short)(__float_as_uint(r2) >> 16);
}
}  // namespace desco

extern "C" int desco_split_bf16x3_t_f32(const float* w, int k, int n, int64_t ldw, int16_t* planes,
                                        desco_stream_t stream) {
  if (!w || !planes || k <= 0 || n <= 0 || ldw < n)
    return desco::fail(DESCO_EINVAL, "desco_split_bf16x3_t_f32: bad argument");
  const int64_t count = (int64_t)k * n;
  hipLaunchKernelGGL(desco::split_bf16x3_t_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     k, n, ldw, reinterpret_cast<short*>(planes));
  return desco::launch_status("desco_split_bf16x3_t_f32");
}

// nb matrices [rows][cols] -> planes [nb][3][..]: transpose == 0: [rows][cols] as they are (n-major planes of a matrix kept
// as [n, k]); transpose != 0: [cols][rows] (n-major planes of a matrix kept as [k, n]).  One launch for the stacked
// weights of a training trunk.
namespace desco {
__global__ __launch_bounds__(256) void split_bf16x3_batch_kernel(const float* __restrict__ w, int64_t per, int rows,
                                                                 int cols, int transpose, int64_t total, int np,
                                                                 short* __restrict__ planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // output element: matrix b, position o
  if (i >= total) return;
  const int64_t b = i / per, o = i % per;
  int64_t src = o;
  if (transpose) {
    const int64_t rr = o / rows, cc = o % rows;                   // output row rr (a source column), column cc
    src = cc * cols + rr;
  }
  const float f = w[b * per + src];
  if (np == 1) {           // (the bf16 training mode's operand: one plane, round to nearest even)
    planes[b * per + o] = (short)(pack2_bf16_rne(f, 0.f) & 0xffffu);
    return;
  }
  const uint32_t uh = __float_as_uint(f) & 0xffff0000u;
  const float r1 = f - __uint_as_float(uh);
  const uint32_t um = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(um);
  short* pb = planes + b * 3 * per;
  pb[o] = (short)(uh >> 16);
  pb[per + o] = (short)(um >> 16);
  pb[2 * per + o] = (short)(__float_as_uint(r2) >> 16);
}
}  // namespace desco

extern "C" int desco_split_bf16x3_batch_f32(const float* w, int64_t num, int rows, int cols, int transpose,
                                            int num_planes, int16_t* planes, desco_stream_t stream) {
  if (num == 0) return 0;
  if (!w || !planes || num < 0 || rows <= 0 || cols <= 0 || (num_planes != 3 && num_planes != 1))
    return desco::fail(DESCO_EINVAL, "desco_split_bf16x3_batch_f32: bad argument (num_planes 3 or 1)");
  const int64_t per = (int64_t)rows * cols, total = per * num;
  hipLaunchKernelGGL(desco::split_bf16x3_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, per, rows, cols, transpose, total, num_planes,
                     reinterpret_cast<short*>(planes));
  return desco::launch_status("desco_split_bf16x3_batch_f32");
}

static int gemm_planes_multi(int np, int num, const desco_gemm_desc* descs, const int16_t* const* planes,
                             desco_stream_t stream);
extern "C" int desco_gemm_bf16x6_multi_f32(int num, const desco_gemm_desc* descs, const int16_t* const* planes,
                                           desco_stream_t stream) {
  return gemm_planes_multi(3, num, descs, planes, stream);
}
// ... and with ONE plane per weight (round-to-nearest bf16, desco_split_bf16x3_batch_f32 with num_planes = 1) and A rounded
// in the kernel: the bf16 training mode's products (desco_gemm_bf16_f32's arithmetic), several problems per launch
extern "C" int desco_gemm_bf16_multi_f32(int num, const desco_gemm_desc* descs, const int16_t* const* planes,
                                         desco_stream_t stream) {
  return gemm_planes_multi(1, num, descs, planes, stream);
}
static int gemm_planes_multi(int np, int num, const desco_gemm_desc* descs, const int16_t* const* planes,
                             desco_stream_t stream) {
  using namespace desco;
  if (num == 0) return 0;
  if (num < 0 || num > kSplitMulti || !descs || !planes)
    return fail(DESCO_EINVAL, "desco_gemm_bf16x6_multi_f32: 1..4 descriptors");
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  GemmSplitMulti q{};
  int nq = 0, blocks = 0;
  for (int i = 0; i < num; ++i) {
    const desco_gemm_desc& d = descs[i];
    if (d.m == 0) continue;
    if (d.m < 0 || !d.a1 || !planes[i] || !d.c || d.k1 <= 0 || d.k1 % SBK || d.k2 < 0 || d.k2 % SBK || d.n <= 0 ||
        d.n % 64 || (d.k2 > 0 && !d.a2) || d.ns != 0 || d.accum || (d.bias && d.bias_rows != 1) || d.lda1 % 4 ||
        (d.k2 > 0 && d.lda2 % 4) || mis16(d.a1) || (d.k2 > 0 && mis16(d.a2)) || mis16(planes[i]) ||
        (d.gate && d.ldg < d.n))
      return fail(DESCO_EINVAL, "desco_gemm_bf16x6_multi_f32: bad descriptor (k%32, n%64, 16-byte alignment, no scalar "
                                "tail / accum / per-row bias)");
    const int64_t gm = (d.m + 127) / 128;
    const int ny = d.n / 64;
    const int64_t nb = ((gm + 7) / 8) * 8 * ny;
    if (nb + blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gemm_bf16x6_multi_f32: m too large");
    q.a1[nq] = d.a1; q.a2[nq] = d.a2; q.w[nq] = reinterpret_cast<const short*>(planes[i]); q.bias[nq] = d.bias;
    q.gate[nq] = d.gate; q.c[nq] = d.c;
    q.lda1[nq] = d.lda1; q.lda2[nq] = d.lda2; q.ldc[nq] = d.ldc; q.ldg[nq] = d.ldg; q.m[nq] = d.m; q.gm[nq] = gm;
    q.k1[nq] = d.k1; q.k2[nq] = d.k2; q.n[nq] = d.n; q.act[nq] = d.act; q.gate_act[nq] = d.gate_act; q.ny[nq] = ny;
    q.slope[nq] = d.slope; q.gate_slope[nq] = d.gate_slope;
    q.drop[nq] = DropArgs{d.drop.key, d.drop.site, d.drop.threshold, d.drop.scale};
    blocks += (int)nb;
    q.blk_end[nq] = blocks;
    ++nq;
  }
  if (nq == 0) return 0;
  q.num = nq;
  constexpr int BM = 128, BN = 64;
  constexpr size_t stage_bytes = (size_t)(3 * BM * SST + 3 * BN * SST) * sizeof(short);      // (sized for three planes)
  constexpr size_t epi_bytes = (size_t)(BM / 32) * 32 * 32 * sizeof(float);
  constexpr size_t lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_multi_kernel<3, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_multi_kernel<1, BM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "desco_gemm_bf16x6_multi_f32: cannot size LDS");
    attr_once.mark();
  }
  if (np == 3)
    hipLaunchKernelGGL((gemm_split_multi_kernel<3, BM>), dim3((unsigned)blocks), dim3(2 * BM), lds_bytes,
                       (hipStream_t)stream, q);
  else
    hipLaunchKernelGGL((gemm_split_multi_kernel<1, BM>), dim3((unsigned)blocks), dim3(2 * BM), lds_bytes,
                       (hipStream_t)stream, q);
  return launch_status("desco_gemm_bf16x6_multi_f32");
}

// post_mp.0 on the pooled embeddings WITHOUT materialising them (round 6): pooled[b] = anchor row + the fused pooling's
// partial sums was written by desco_pool_reduce_f32 and read straight back by this product -- 5.5 GB per COX2 x64 pass.
// The POOLA instantiation forms the operand's chunks in its load phase: per row the (at most three) slots of its
// segment are looked up once, a chunk of layer l is the anchor chunk + the segment's partial rows of that layer.
extern "C" int desco_pool_post_bf16x6_f32(const float* anch, int64_t lda, int num_layers, const int16_t* w_planes, int n,
                                          const float* bias, int act, float slope, float* c, int64_t ldc, int64_t m,
                                          const int32_t* seg_ptr, const uint32_t* pool_bits, const int32_t* pool_slot,
                                          const float* const* parts, const float* x0, int tile_rows,
                                          desco_stream_t stream) {
  using namespace desco;
  if (m == 0) return 0;
  if (m > (int64_t)(1 << 24)) return fail(DESCO_EINVAL, "desco_pool_post_bf16x6_f32: more than 2^24 segments (slot offsets are 31 bits)");
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (m < 0 || !anch || !w_planes || !c || num_layers < 1 || num_layers > 8 || n != 64 || !seg_ptr || !pool_bits ||
      !pool_slot || !parts || !x0 || tile_rows != 16 || lda % 4 || lda < 64 * (num_layers + 1) || mis16(anch) ||
      mis16(w_planes) || mis16(x0))
    return fail(DESCO_EINVAL, "desco_pool_post_bf16x6_f32: bad argument (n == 64, 1..8 pooled layers, 16-row tiles)");
  GemmSplitArgs g{anch, lda, 64 * (num_layers + 1), nullptr, 0, 0, reinterpret_cast<const short*>(w_planes), n, bias, 1,
                  nullptr, 0, nullptr, act, slope, c, ldc, m, seg_ptr, pool_bits, pool_slot, {}, x0};
  for (int l = 1; l <= num_layers; ++l) {
    if (!parts[l - 1] || mis16(parts[l - 1])) return fail(DESCO_EINVAL, "desco_pool_post_bf16x6_f32: NULL / misaligned partial array");
    g.part[l] = parts[l - 1];
  }
  constexpr int BM = 128;
  constexpr size_t stage_bytes = (size_t)(3 * BM * SST + 3 * 64 * SST) * sizeof(short);
  constexpr size_t epi_bytes = (size_t)(BM / 32) * 32 * 32 * sizeof(float);
  constexpr size_t lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<1, 3, BM, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "desco_pool_post_bf16x6_f32: cannot size LDS");
    attr_once.mark();
  }
  const int64_t gm = (m + BM - 1) / BM;
  const int64_t blocks = ((gm + 7) / 8) * 8;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_pool_post_bf16x6_f32: m too large");
  hipLaunchKernelGGL((gemm_split_kernel<1, 3, BM, true>), dim3((unsigned)blocks), dim3(2 * BM), lds_bytes,
                     (hipStream_t)stream, g, gm, 1);
  return launch_status("desco_pool_post_bf16x6_f32");
}

// Plain bf16 MFMA GEMM with fp32 accumulation and fp32 output (bf16 training mode): A is rounded
// to nearest-even bf16 in the kernel, the weight arrives rounded (desco_round_bf16_f32), n-major.
extern "C" int desco_gemm_bf16_f32(const float* a1, int64_t lda1, int k1, const float* a2,
                                   int64_t lda2, int k2, const int16_t* w_bf16, int n,
                                   const float* bias, int bias_rows, const float* s, int ns,
                                   const float* ws, int act, float slope, float* c, int64_t ldc,
                                   int64_t m, desco_stream_t stream) {
  return gemm_planes("desco_gemm_bf16_f32: bad argument (k%32, n%64, 16-byte alignment)", 1, a1, lda1,
                     k1, a2, lda2, k2, w_bf16, n, bias, bias_rows, s, ns, ws, act, slope, c, ldc, m,
                     stream);
}

extern "C" int desco_round_bf16_f32(const float* w, int64_t count, int16_t* out,
                                    desco_stream_t stream) {
  using namespace desco;
  if (count == 0) return 0;
  if (count < 0 || !w || !out) return fail(DESCO_EINVAL, "desco_round_bf16_f32: bad argument");
  const int64_t blocks = (count + 255) / 256;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_round_bf16_f32: count too large");
  hipLaunchKernelGGL(round_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                     count, reinterpret_cast<short*>(out));
  return launch_status("desco_round_bf16_f32");
}

extern "C" int desco_split_bf16x3_f32(const float* w, int64_t count, int16_t* planes,
                                      desco_stream_t stream) {
  using namespace desco;
  if (count == 0) return 0;
  if (count < 0 || !w || !planes)
    return fail(DESCO_EINVAL, "desco_split_bf16x3_f32: bad argument");
  const int64_t blocks = (count + 255) / 256;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_split_bf16x3_f32: count too large");
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     w, count, reinterpret_cast<short*>(planes));
  return launch_status("desco_split_bf16x3_f32");
}
