// fp32-accurate GEMM on the fp16 matrix pipe in THREE products ("f16x3"): every fp32 operand is scaled by a
// power of two and split into two fp16 terms, x s = hi + lo (hi = rne(x s), lo = rne(x s - hi): 22 significand
// bits), and
//     a*b ~= (hi*hi + hi*lo + lo*hi) / (s_a s_b)                      (dropped: lo*lo, 2^-22 |a||b| at worst)
// accumulates in fp32 on v_mfma_f32_32x32x16_f16.  Against the six-product bf16 form of gemm_split.hip: half the
// MFMAs, two operand planes instead of three (LDS image and fragment reads -33 %), 3 VALU per pair for the split
// instead of 9.  Measured error (tools/micro/f16x3_probe.hip, K = 576): rms 1.5e-8 of sum|a||b| against 2.4e-8 for
// v_mfma_f32_16x16x4_f32 and 2.0e-8 for bf16x6.
//
// fp16 has a 5-bit exponent, so the scales carry the range:
//   * weights: ONE power of two per matrix (largest |w| -> [2^14, 2^15)), applied when the planes are made
//     (desco_split_f16x2_f32), its inverse kept on the device next to the planes;
//   * activations: one power of two PER ROW over the whole K extent (largest |a| of the row -> [2^14, 2^15)), so a
//     row's products share one accumulator; derived from a per-row bound that either a pre-pass over A provides
//     (row_scale_kernel: HBM-bound, A once) or the kernel that wrote A, and undone in the epilogue together with the
//     weight scale (exact: powers of two).
// An element 2^-17 below its row maximum keeps 22 bits; smaller ones degrade gradually (lo subnormal) down to an
// absolute error of 2^-40 of the row maximum -- below the fp32 rounding error of the row's dot products.
//
// Same contract and tile structure as desco_gemm_bf16x6_f32 (gemm_split.hip: 128 x 64 WN block tile, 2x2 waves,
// K chunks of 32, A two chunks ahead, XCD-aware tile order, swizzled 64-byte plane rows, LDS-staged epilogue).
#include "common_device.hpp"

namespace desco {

struct GemmF16Args {
  const float* a1;
  int64_t lda1;
  int k1;
  const float* a2;
  int64_t lda2;
  int k2;
  const short* w;         // planes [2][n][k1+k2] (hi, lo) of the scaled weight
  const float* w_scale;   // device [2]: {scale, 1 / scale}
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
  const float* row_scale;  // [m] bound of each row's largest |a| (desco_row_absmax_f32 or the producer of A)
};

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int FBK = 32, FST = 32;              // K chunk; plane row stride in halves (64 B, no padding)
// same swizzle as gemm_split.hip: 16-byte chunk c of plane row r sits at chunk c ^ ((r >> 3) & 3)
__device__ __forceinline__ int gf16_chunk(const int row, const int c) { return ((c ^ (row >> 3)) & 3) << 3; }

// per-row bound of the A operand: max_k |A[i, k]| (the kernel turns it into the power of two s with s * bound in
// [2^14, 2^15)).  16 lanes per row, float4 loads.  A kernel that WRITES A can leave the same array instead (the SHMP
// layer's canonical launches do, for the anchor operand): any value >= the row's largest magnitude will do.
__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ a1, int64_t lda1, int k1,
                                                        const float* __restrict__ a2, int64_t lda2, int k2,
                                                        int64_t m, float* __restrict__ out) {
  const int lane16 = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t r = row < m ? row : m - 1;
  float mx = 0.f;
  const float* p1 = a1 + r * lda1;
  for (int k = 4 * lane16; k < k1; k += 64) {
    const float4 v = *reinterpret_cast<const float4*>(p1 + k);
    mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (k2 > 0) {
    const float* p2 = a2 + r * lda2;
    for (int k = 4 * lane16; k < k2; k += 64) {
      const float4 v = *reinterpret_cast<const float4*>(p2 + k);
      mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane16 == 0 && row < m) out[row] = mx;
}

// BM = 128 rows per block tile (4 waves, two blocks per CU), WN = 3, 2, 1 column tiles of 64
template <int WN, int BM>
__global__ __launch_bounds__(2 * BM) __attribute__((amdgpu_waves_per_eu(2))) void gemm_f16x3_kernel(GemmF16Args g, int64_t gm,
                                                                                                   int ny) {
  constexpr int NT = 2 * BM;
  constexpr int BN = 64 * WN, BPLANE = BN * FST, APLANE = BM * FST;
  constexpr int AR = BM / 4;
  constexpr int BR = NT / 4;
  constexpr int BJ = (BN + BR - 1) / BR;
  constexpr int EW = 32 * WN;
  constexpr int STAGE = 2 * APLANE + 2 * BPLANE;                  // shorts
  constexpr int RAW = BM * FBK * 2;                               // shorts: one fp32 A chunk [BM][32] as it arrives
  constexpr int EPI = (BM / 32) * 32 * EW * 2;                    // shorts (fp32 image)
  constexpr int BODY = STAGE + 2 * RAW > EPI ? STAGE + 2 * RAW : EPI;
  static_assert(BR <= BN && BN % BR == 0, "every thread stages BJ full W rows");
  extern __shared__ __attribute__((aligned(16))) short lds[];
  short* Ap = lds;                  // planes hi, lo of the A chunk [BM][32]
  short* Bp = lds + 2 * APLANE;     // planes hi, lo of the W chunk [BN][32]
  // The memory pipeline of the chunk loop is written by hand (inline asm), because it has to keep TWO A chunks in flight
  // across the loop's back edge and the compiler cannot be told to:
  //   * the fp32 A stream lands in LDS (global_load_lds_dwordx4: no registers held while in flight), two chunks in two
  //     landing buffers; a thread fetches exactly the 4 x 16 bytes it converts itself (float index 4 tid + AR*32 j of the
  //     chunk: one wave-instruction = 1 KB of contiguous LDS), so no barrier stands between arrival and use, only the
  //     thread's own vmcnt;
  //   * vector memory returns in order: per step a thread issues the W planes of the next chunk (registers), THEN the A
  //     chunk two steps ahead; the next step starts with vmcnt(4) -- everything but those four A loads has arrived.
  // Why not in C++: round 4 held the A chunks in two register sets and copied the second into the first behind the
  // barrier -- the copy needs the loads issued ONE step earlier, so a step took a memory round trip under load (3.7 us for
  // 1152 cycles of MFMAs per wave: 0.26 of the matrix rate); alternating the sets in a loop unrolled by two spills (256
  // registers), and with the LDS-direct builtin the compiler's wait insertion puts vmcnt(0) in front of the first VALU
  // instruction that touches a register an outstanding LDS-direct load used as its address (profiles/r5_g_*).
  float* Araw = reinterpret_cast<float*>(lds + STAGE);  // [2][BM][32] fp32
  float* rinv = reinterpret_cast<float*>(lds + BODY);   // [BM] 1 / (row scale * weight scale)

  const int64_t id = blockIdx.x;
  const int64_t local = id >> 3;
  const int64_t mt = (local / ny) * 8 + (id & 7);
  if (mt >= gm) return;
  const int n0 = (int)(local % ny) * BN;
  const int64_t m0 = mt * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int K = g.k1 + g.k2;
  const int nchunks = K / FBK;

  const int arow = tid >> 3, ac4 = tid & 7;
  const int64_t mlast = g.m - 1;
  int64_t r0 = m0 + arow, r1 = r0 + AR, r2 = r0 + 2 * AR, r3 = r0 + 3 * AR;
  r0 = r0 < g.m ? r0 : mlast;
  r1 = r1 < g.m ? r1 : mlast;
  r2 = r2 < g.m ? r2 : mlast;
  r3 = r3 < g.m ? r3 : mlast;
  const float sc0 = f16_scale_for(g.row_scale[r0]), sc1 = f16_scale_for(g.row_scale[r1]);
  const float sc2 = f16_scale_for(g.row_scale[r2]), sc3 = f16_scale_for(g.row_scale[r3]);
  if (tid < BM) {
    const int64_t rr = m0 + tid < g.m ? m0 + tid : mlast;
    rinv[tid] = pow2_inverse(f16_scale_for(g.row_scale[rr])) * g.w_scale[1];
  }
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
  constexpr bool v0 = true, v1 = BJ > 1, v2 = BJ > 2;   // (BN is a multiple of BR: every thread has BJ full rows)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  f32x4 ra0, ra1, ra2, ra3;                    // A chunk about to be stored (read back from its landing buffer)
  u32x4 rb00, rb01, rb10, rb11, rb20, rb21;    // W planes of the next chunk (in flight across the back edge)
  rb00 = rb01 = rb10 = rb11 = rb20 = rb21 = u32x4{0, 0, 0, 0};
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // LDS byte offsets (the low 32 bits of a generic LDS address): this thread's part of a landing buffer; this wave's KB
  const uint32_t araw_tid = (uint32_t)(uintptr_t)Araw + 16u * (uint32_t)tid;
  const uint32_t araw_wave = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)Araw + 1024u * (uint32_t)wave_u);
  // chunk kk_ of A, global -> landing buffer buf_: lane i's 16 bytes land at m0 + 16 i
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (m0 in the clobber list is "reserved": it is what the instruction reads)
#define DESCO_DMA16(src_, m0_)                                                                \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"              \
               :: "v"(src_), "s"(m0_) : "memory", "m0");
#define DESCO_FETCH_A(buf_, kk_)                                                              \
  {                                                                                           \
    const int k_ = (kk_);                                                                     \
    const bool s1_ = k_ < g.k1;                                                               \
    const uint32_t d_ = araw_wave + (uint32_t)((buf_) * (BM * FBK * 4));                      \
    DESCO_DMA16((s1_ ? p10 : p20) + k_, d_)                                                   \
    DESCO_DMA16((s1_ ? p11 : p21) + k_, d_ + AR * FBK * 4)                                    \
    DESCO_DMA16((s1_ ? p12 : p22) + k_, d_ + 2 * AR * FBK * 4)                                \
    DESCO_DMA16((s1_ ? p13 : p23) + k_, d_ + 3 * AR * FBK * 4)                                \
  }
  // Start of a step: everything but the four youngest loads (the A chunk after this one) has arrived -- this chunk's
  // part of landing buffer buf_ -> ra, and the W registers become readable (the empty asm ties them behind the wait).
#define DESCO_TAKE(buf_)                                                                      \
  {                                                                                           \
    const uint32_t s_ = araw_tid + (uint32_t)((buf_) * (BM * FBK * 4));                       \
    asm volatile("s_waitcnt vmcnt(4)\n\t"                                                     \
                 "ds_read_b128 %0, %4\n\t"                                                    \
                 "ds_read_b128 %1, %4 offset:%5\n\t"                                          \
                 "ds_read_b128 %2, %4 offset:%6\n\t"                                          \
                 "ds_read_b128 %3, %4 offset:%7\n\t"                                          \
                 "s_waitcnt lgkmcnt(0)"                                                        \
                 : "=&v"(ra0), "=&v"(ra1), "=&v"(ra2), "=&v"(ra3)                             \
                 : "v"(s_), "n"(AR * FBK * 4), "n"(2 * AR * FBK * 4), "n"(3 * AR * FBK * 4)   \
                 : "memory");                                                                 \
    asm volatile("" : "+v"(rb00), "+v"(rb01), "+v"(rb10), "+v"(rb11), "+v"(rb20), "+v"(rb21) :: "memory"); \
  }
#define DESCO_LOAD_W1(d_, p_) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d_) : "v"(p_) : "memory");
#define DESCO_LOAD_WJ(j_, v_)                                                                 \
  if constexpr (v_) {                                                                         \
    DESCO_LOAD_W1(rb##j_##0, w_ + (j_) * wj)                                                  \
    DESCO_LOAD_W1(rb##j_##1, w_ + (j_) * wj + wplane)                                         \
  }
#define DESCO_LOAD_W(kk_)                                                                     \
  {                                                                                           \
    const short* w_ = pw + (kk_);                                                             \
    DESCO_LOAD_WJ(0, v0) DESCO_LOAD_WJ(1, v1) DESCO_LOAD_WJ(2, v2)                            \
  }
#define DESCO_PUT(row_, v_, sc_)                                                              \
  {                                                                                           \
    short* d_ = Ap + (row_)*FST + gf16_chunk((row_), ac4 >> 1) + 4 * (ac4 & 1);               \
    uint32_t h0_, l0_, h1_, l1_;                                                              \
    split2_f16x2(v_.x * (sc_), v_.y * (sc_), h0_, l0_);                                       \
    split2_f16x2(v_.z * (sc_), v_.w * (sc_), h1_, l1_);                                       \
    *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                                     \
    *reinterpret_cast<uint2*>(d_ + APLANE) = make_uint2(l0_, l1_);                            \
  }
#define DESCO_STORE_WJ(j_, v_)                                                                \
  if (BJ > (j_) && (v_)) {                                                                    \
    short* bj_ = Bp + (brow + (j_) * BR) * FST + gf16_chunk(brow + (j_) * BR, bpart);         \
    *reinterpret_cast<u32x4*>(bj_) = rb##j_##0;                                               \
    *reinterpret_cast<u32x4*>(bj_ + BPLANE) = rb##j_##1;                                      \
  }
#define DESCO_STORE_CHUNK()                                                                   \
  {                                                                                           \
    DESCO_PUT(arow, ra0, sc0)                                                                 \
    DESCO_PUT(arow + AR, ra1, sc1)                                                            \
    DESCO_PUT(arow + 2 * AR, ra2, sc2)                                                        \
    DESCO_PUT(arow + 3 * AR, ra3, sc3)                                                        \
    DESCO_STORE_WJ(0, v0) DESCO_STORE_WJ(1, v1) DESCO_STORE_WJ(2, v2)                         \
  }

  f32x16 acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  DESCO_LOAD_W(0)
  DESCO_FETCH_A(0, 0)
  DESCO_FETCH_A(1, (nchunks > 1 ? 1 : 0) * FBK)
  for (int ch = 0; ch < nchunks; ++ch) {
    DESCO_TAKE(ch & 1)
    if (ch > 0) __syncthreads();          // previous chunk's fragments have been read
    DESCO_STORE_CHUNK()
    __syncthreads();
    const int chn = ch + 1 < nchunks ? ch + 1 : ch;
    const int chnn = ch + 2 < nchunks ? ch + 2 : chn;
    DESCO_LOAD_W(chn * FBK)                // in flight under the MFMAs
    DESCO_FETCH_A(ch & 1, chnn * FBK)      // two chunks ahead, into the buffer just taken (the tail re-fetches the last)
    // lane (r = lane&31, h = lane>>5): A[row r][k = 16 s + 8 h + j], B[k = 16 s + 8 h + j][col r]
    const int fsw = (lane >> 3) & 3, fh = lane >> 5;
    const short* ap = Ap + (wr * 64 + (lane & 31)) * FST;
    const short* bp = Bp + (wc * 32 * WN + (lane & 31)) * FST;
#pragma unroll
    for (int s = 0; s < FBK / 16; ++s) {
      f16x8 ah[2], al[2];
      const int co = (((2 * s + fh) ^ fsw) & 3) << 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(ap + i * 32 * FST + co);
        al[i] = *reinterpret_cast<const f16x8*>(ap + i * 32 * FST + APLANE + co);
      }
      f16x8 bh[WN], bl[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const short* bt = bp + j * 32 * FST + co;
        bh[j] = *reinterpret_cast<const f16x8*>(bt);
        bl[j] = *reinterpret_cast<const f16x8*>(bt + BPLANE);
      }
      // smallest terms first; one product of ALL 2 WN accumulators at a time: an accumulator comes round again after
      // 2 WN MFMAs, not after two (a dependent 32x32x16 MFMA issued within its predecessor's 64 cycles waits for it)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1], bh[j], acc[1][j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bl[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bl[j], acc[1][j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[j], acc[1][j], 0, 0, 0);
      }
    }
  }
#pragma clang diagnostic pop
#undef DESCO_FETCH_A
#undef DESCO_TAKE
#undef DESCO_DMA16
#undef DESCO_LOAD_W1
#undef DESCO_LOAD_W
#undef DESCO_LOAD_WJ
#undef DESCO_PUT
#undef DESCO_STORE_CHUNK
#undef DESCO_STORE_WJ

  // Epilogue (as gemm_split.hip): each 32-row half of the wave tile through a wave-private [32][32 WN] fp32 LDS
  // image -> 16-byte row-contiguous nontemporal stores.  The scales are undone here: row = (reg & 3) + 8 (reg >> 2)
  // + 4 (lane >> 5), so the four rows of a register quad are one 16-byte read of rinv.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail's re-fetches land in what becomes the image)
  __syncthreads();                       // every wave is done with the last chunk's fragments
  float* st = reinterpret_cast<float*>(lds) + wave * (32 * EW);
  const int col = lane & 31;
  const bool wide = ((reinterpret_cast<uintptr_t>(g.c) & 15) == 0) && ((g.ldc & 3) == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t grow0 = m0 + wr * 64 + 32 * i;
    const float* ri = rinv + wr * 64 + 32 * i + 4 * (lane >> 5);
    const float4 q0 = *reinterpret_cast<const float4*>(ri), q1 = *reinterpret_cast<const float4*>(ri + 8);
    const float4 q2 = *reinterpret_cast<const float4*>(ri + 16), q3 = *reinterpret_cast<const float4*>(ri + 24);
    const float rv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int gcol = n0 + wc * EW + 32 * j + col;
      float wsv[4] = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < g.ns; ++q) wsv[q] = g.ws[(int64_t)q * g.n + gcol];
      const float b_single = (g.bias && g.bias_rows == 1) ? g.bias[gcol] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int64_t grow = grow0 + row < g.m ? grow0 + row : g.m - 1;
        float v = acc[i][j][reg] * rv[reg];
        if (g.bias) {
          if (g.bias_rows == 1)
            v += b_single;
          else
            v += g.bias[(grow % g.bias_rows) * g.n + gcol];
        }
        for (int q = 0; q < g.ns; ++q) v += g.s[grow * g.ns + q] * wsv[q];
        st[row * EW + 32 * j + col] = apply_act(v, g.act, g.slope);
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

// ---- weight planes -------------------------------------------------------------------------------------------------
// one block: scale[0] = the power of two s with s * max|w| in [2^14, 2^15), scale[1] = 1 / s
__global__ __launch_bounds__(1024) void f16_weight_scale_kernel(const float* __restrict__ w, int64_t count,
                                                                float* __restrict__ scale) {
  __shared__ float part[16];
  float mx = 0.f;
  for (int64_t i = threadIdx.x; i < count; i += 1024) mx = fmaxf(mx, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, part[i]);
    const float s = f16_scale_for(mx);
    scale[0] = s;
    scale[1] = pow2_inverse(s);
  }
}

// w[count] -> planes[2][count] (hi, lo fp16 bit patterns of scale * w)
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ w, int64_t count,
                                                          const float* __restrict__ scale,
                                                          short* __restrict__ planes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  uint32_t hi, lo;
  split2_f16x2(w[i] * scale[0], 0.f, hi, lo);
  planes[i] = (short)(hi & 0xffffu);
  planes[count + i] = (short)(lo & 0xffffu);
}

template <int WN>
static int launch_gemm_f16x3(const GemmF16Args& g, hipStream_t stream) {
  constexpr int BM = 128, BN = 64 * WN;
  constexpr size_t stage_bytes = (size_t)(2 * BM * FST + 2 * BN * FST) * sizeof(short) + 2 * (size_t)BM * FBK * sizeof(float);
  constexpr size_t epi_bytes = (size_t)(BM / 32) * 32 * 32 * WN * sizeof(float);
  constexpr size_t lds_bytes = (stage_bytes > epi_bytes ? stage_bytes : epi_bytes) + BM * sizeof(float);
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_kernel<WN, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "desco_gemm_f16x3_f32: cannot size LDS");
    attr_once.mark();
  }
  const int64_t gm = (g.m + BM - 1) / BM;
  const int ny = g.n / BN;
  const int64_t blocks = ((gm + 7) / 8) * 8 * ny;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gemm_f16x3_f32: m too large");
  hipLaunchKernelGGL((gemm_f16x3_kernel<WN, BM>), dim3((unsigned)blocks), dim3(2 * BM), lds_bytes, stream, g, gm, ny);
  return launch_status("desco_gemm_f16x3_f32");
}

}  // namespace desco

using namespace desco;

extern "C" int desco_row_absmax_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                                   int64_t m, float* row_scale, desco_stream_t stream) {
  if (m == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (m < 0 || !a1 || !row_scale || k1 <= 0 || k1 % 4 || k2 < 0 || k2 % 4 || (k2 > 0 && !a2) || lda1 % 4 ||
      (k2 > 0 && lda2 % 4) || mis16(a1) || (k2 > 0 && mis16(a2)))
    return fail(DESCO_EINVAL, "desco_row_absmax_f32: bad argument (k % 4, 16-byte alignment)");
  const int64_t blocks = (m + 15) / 16;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_row_absmax_f32: m too large");
  hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a1, lda1, k1, a2,
                     lda2, k2, m, row_scale);
  return launch_status("desco_row_absmax_f32");
}

extern "C" int desco_gemm_f16x3_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2, int k2,
                                    const int16_t* w_planes, const float* w_scale, int n, const float* bias,
                                    int bias_rows, const float* s, int ns, const float* ws, int act, float slope,
                                    float* c, int64_t ldc, int64_t m, const float* row_scale,
                                    desco_stream_t stream) {
  if (m == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (m < 0 || !a1 || !w_planes || !w_scale || !row_scale || !c || k1 <= 0 || k1 % FBK || k2 < 0 || k2 % FBK ||
      n <= 0 || n % 64 || (k2 > 0 && !a2) || ns < 0 || ns > 4 || (ns > 0 && (!s || !ws)) ||
      (bias && bias_rows < 1) || lda1 % 4 || (k2 > 0 && lda2 % 4) || mis16(a1) || (k2 > 0 && mis16(a2)) ||
      mis16(w_planes))
    return fail(DESCO_EINVAL, "desco_gemm_f16x3_f32: bad argument (k%32, n%64, 16-byte alignment)");
  GemmF16Args g{a1, lda1, k1, a2, lda2, k2, reinterpret_cast<const short*>(w_planes), w_scale, n, bias,
                bias ? bias_rows : 1, s, ns, ws, act, slope, c, ldc, m, row_scale};
  hipStream_t st = (hipStream_t)stream;
  if (n % 192 == 0) return launch_gemm_f16x3<3>(g, st);
  if (n % 128 == 0) return launch_gemm_f16x3<2>(g, st);
  return launch_gemm_f16x3<1>(g, st);
}

extern "C" int desco_split_f16x2_f32(const float* w, int64_t count, int16_t* planes, float* scale,
                                     desco_stream_t stream) {
  if (count == 0) return 0;
  if (count < 0 || !w || !planes || !scale) return fail(DESCO_EINVAL, "desco_split_f16x2_f32: bad argument");
  const int64_t blocks = (count + 255) / 256;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_split_f16x2_f32: count too large");
  hipLaunchKernelGGL(f16_weight_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, count, scale);
  hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, count, scale,
                     reinterpret_cast<short*>(planes));
  return launch_status("desco_split_f16x2_f32");
}
