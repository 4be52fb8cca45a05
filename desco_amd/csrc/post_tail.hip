// post_mp.3 -> .5 -> .7 of BaseGNN (reference gnn_model.py:44-53: Linear(64, 64), ReLU, Linear(64, 256), ReLU,
// Linear(256, 64)) in ONE launch: a row of the [m, 64] input is read once and a row of the [m, 64] result written once;
// the [m, 64] and [m, 256] intermediates of the three separate launches (3.1 GB of HBM traffic per 1.2 M rows) stay in
// registers.
//
// Arithmetic: the f16x3 form of gemm_f16x3.hip (fp32 operands scaled by a power of two and split into two fp16 terms,
// hi*hi + lo*hi + hi*lo accumulated in fp32 on v_mfma_f32_32x32x16_f16).
//
// The products are formed TRANSPOSED, D^T = W X^T: the weight fragment is the MFMA's A operand (row = output feature),
// the activations its B operand (column = data row).  In the C/D layout a lane then holds, for ITS data row (column
// lane & 31), features (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of a 32-feature block -- and registers 8t .. 8t+7 are
// exactly the eight k values a lane supplies to K step t of the next product's B operand, in the order
//     k = 16 t + 4 (lane >> 5) + (j & 3) + 8 (j >> 2),   j = 0..7,
// a permutation of the MFMA's own k order (8 (lane >> 5) + j) that the next layer's weight image in LDS carries (bits 2
// and 3 of k swapped when the image is filled).  So a layer's output becomes the next layer's operand by converting
// registers in place: no LDS round trip, no shuffles.  The per-row power-of-two scale of an activation row is a per-LANE
// constant in this form.
//
// The 256 hidden features never exist at once: each block of 32 is produced (12 MFMAs), scaled / split, and consumed
// by the last layer's 12 MFMAs straight away.  Its row scale is a RUNNING one: when a block raises the row's maximum,
// the last layer's accumulators are multiplied by the (exact, power-of-two) ratio of the scales, so every block is split
// against a bound between its own and the row's final maximum -- the guarantee of a whole-row scale.
//
// One workgroup per CU (the three weight images fill the LDS: 156 KB), 8 waves of 32 rows, the next tile's rows in
// flight under the current tile's 216 MFMAs.
#include "tu_no_packed_f32_begin.hpp"
#include "common_device.hpp"

namespace desco {
namespace tail {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int D1 = 64, D2 = 256, D3 = 64;     // output widths of the three layers (input 64)
constexpr int ST64 = 72, ST256 = 264;         // LDS row strides in halves: 16-byte reads of 32 rows conflict-free
constexpr int W1_H = 2 * D1 * ST64, W2_H = 2 * D2 * ST64, W3_H = 2 * D3 * ST256;
constexpr int NWT = 8;                        // waves per workgroup: the count head (224 registers) ...
#ifndef DESCO_TAIL_WAVES
#define DESCO_TAIL_WAVES 12
#endif
constexpr int TNW = DESCO_TAIL_WAVES;         // ... and the post_mp tail (162 registers: three waves per SIMD)
constexpr size_t TAIL_LDS = (size_t)(W1_H + W2_H + W3_H) * 2 + (size_t)(D1 + D2 + D3) * 4;
static_assert(TAIL_LDS <= 160 * 1024, "post_mp tail: LDS budget exceeded");

struct TailArgs {
  const float* x;
  int64_t ldx, m;
  const short *w1, *w2, *w3;        // planes [2][64][64], [2][256][64], [2][64][256] (hi, lo) of scale * W
  const float *s1, *s2, *s3;        // device {scale, 1 / scale} of each matrix
  const float *b1, *b2, *b3;        // biases or null
  float* out;
  int64_t ldo;
  unsigned grid;                    // gridDim.x (set by the launcher: see tu_no_packed_f32_begin.hpp)
};

// row maximum over both lane halves (lanes n and n + 32 hold the two halves of data row n's features)
__device__ __forceinline__ float both_halves_max(const float v) {
  const u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

__device__ __forceinline__ f16x8 frag_of(const uint32_t a, const uint32_t b, const uint32_t c, const uint32_t d) {
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(f16x8, v);
}

// fill one weight image: [2 planes][rows][K] halves -> LDS rows of `stride` halves; `swap` = the 4-half chunks of each
// 16-k group stored in the order 0, 2, 1, 3 (bits 2 and 3 of k exchanged)
template <int K, bool SWAP, int NT = NWT * 64>
__device__ __forceinline__ void fill_image(short* dst, const short* __restrict__ src, const int rows2, const int stride,
                                           const int tid) {
  constexpr int CH = K / 4;
  for (int i = tid; i < rows2 * CH; i += NT) {
    const int row = i / CH, c = i % CH;
    const int cd = SWAP ? ((c & ~3) | ((c & 1) << 1) | ((c >> 1) & 1)) : c;
    *reinterpret_cast<uint2*>(dst + row * stride + 4 * cd) = *reinterpret_cast<const uint2*>(src + (int64_t)row * K + 4 * c);
  }
}

#define DESCO_MFMA3(acc_, wh_, wl_, bh_, bl_)                                       \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_, bh_, acc_, 0, 0, 0);          \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_, bl_, acc_, 0, 0, 0);          \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_, bh_, acc_, 0, 0, 0);

__global__ __launch_bounds__(TNW * 64) void post_tail_kernel(TailArgs g) {
  extern __shared__ __attribute__((aligned(16))) short lds_h[];
  short* W1 = lds_h;
  short* W2 = W1 + W1_H;
  short* W3 = W2 + W2_H;
  float* B1 = reinterpret_cast<float*>(W3 + W3_H);
  float* B2 = B1 + D1;
  float* B3 = B2 + D2;
  const int tid = (int)__builtin_amdgcn_workitem_id_x(), lane = tid & 63, wave = tid >> 6;
  fill_image<64, false, TNW * 64>(W1, g.w1, 2 * D1, ST64, tid);
  fill_image<64, true, TNW * 64>(W2, g.w2, 2 * D2, ST64, tid);
  fill_image<256, true, TNW * 64>(W3, g.w3, 2 * D3, ST256, tid);
  for (int i = tid; i < D1 + D2 + D3; i += TNW * 64) {
    const float* b = i < D1 ? g.b1 : (i < D1 + D2 ? g.b2 : g.b3);
    const int j = i < D1 ? i : (i < D1 + D2 ? i - D1 : i - D1 - D2);
    B1[i] = b ? b[j] : 0.f;
  }
  const float iw1 = g.s1[1], iw2 = g.s2[1], iw3 = g.s3[1];
  __syncthreads();
  const int n = lane & 31, h = lane >> 5;
  const int64_t ntiles = (g.m + 31) / 32;
  int64_t tile = (int64_t)(int)__builtin_amdgcn_workgroup_id_x() * TNW + wave;
  if (tile >= ntiles) return;
  const int64_t tstep = (int64_t)g.grid * TNW;
  // K step s of the first product: the lane supplies k = 16 s + 8 h + 0..7 of its row (the MFMA's own order)
  float4 xv[8];
#define DESCO_TAIL_LOAD(t_)                                                     \
  {                                                                             \
    const int64_t r_ = (t_) * 32 + n;                                           \
    const float* p_ = g.x + (r_ < g.m ? r_ : g.m - 1) * g.ldx + 8 * h;          \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                          \
      xv[2 * s_] = *reinterpret_cast<const float4*>(p_ + 16 * s_);              \
      xv[2 * s_ + 1] = *reinterpret_cast<const float4*>(p_ + 16 * s_ + 4);      \
    }                                                                           \
  }
  DESCO_TAIL_LOAD(tile)
  // fragment addresses of this lane (in halves): row (lane & 31) of a 32-row block, 8 halves at 8 h of a 16-k group
  const short* w1p = W1 + n * ST64 + 8 * h;
  const short* w2p = W2 + n * ST64 + 8 * h;
  const short* w3p = W3 + n * ST256 + 8 * h;
  for (;;) {
    const int64_t row = tile * 32 + n;
    // ---- input rows -> scaled fp16 (hi, lo) B fragments ----
    uint32_t xh[16], xl[16];
    float inv_sx;
    {
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(xv[q].x), fabsf(xv[q].y))), fmaxf(fabsf(xv[q].z), fabsf(xv[q].w)));
      mx = both_halves_max(mx);
      const float sx = f16_scale_for(mx);
      inv_sx = pow2_inverse(sx);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        split2_f16x2(xv[q].x * sx, xv[q].y * sx, xh[2 * q], xl[2 * q]);
        split2_f16x2(xv[q].z * sx, xv[q].w * sx, xh[2 * q + 1], xl[2 * q + 1]);
      }
    }
    const int64_t tn = tile + tstep;
    const bool has_next = tn < ntiles;
    if (has_next) DESCO_TAIL_LOAD(tn)
    // ---- layer 1: 64 -> 64, relu ----
    uint32_t h1h[16], h1l[16];
    float inv_s1;
    {
      f32x16 a0, a1;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a0[i] = 0.f;
        a1[i] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 bh = frag_of(xh[4 * s], xh[4 * s + 1], xh[4 * s + 2], xh[4 * s + 3]);
        const f16x8 bl = frag_of(xl[4 * s], xl[4 * s + 1], xl[4 * s + 2], xl[4 * s + 3]);
        const f16x8 wh0 = *reinterpret_cast<const f16x8*>(w1p + 16 * s);
        const f16x8 wl0 = *reinterpret_cast<const f16x8*>(w1p + D1 * ST64 + 16 * s);
        const f16x8 wh1 = *reinterpret_cast<const f16x8*>(w1p + 32 * ST64 + 16 * s);
        const f16x8 wl1 = *reinterpret_cast<const f16x8*>(w1p + (D1 + 32) * ST64 + 16 * s);
        DESCO_MFMA3(a0, wh0, wl0, bh, bl)
        DESCO_MFMA3(a1, wh1, wl1, bh, bl)
      }
      const float u = iw1 * inv_sx;
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 c0 = *reinterpret_cast<const float4*>(B1 + 8 * q + 4 * h);
        const float4 c1 = *reinterpret_cast<const float4*>(B1 + 32 + 8 * q + 4 * h);
        a0[4 * q] = fmaxf(fmaf(a0[4 * q], u, c0.x), 0.f);
        a0[4 * q + 1] = fmaxf(fmaf(a0[4 * q + 1], u, c0.y), 0.f);
        a0[4 * q + 2] = fmaxf(fmaf(a0[4 * q + 2], u, c0.z), 0.f);
        a0[4 * q + 3] = fmaxf(fmaf(a0[4 * q + 3], u, c0.w), 0.f);
        a1[4 * q] = fmaxf(fmaf(a1[4 * q], u, c1.x), 0.f);
        a1[4 * q + 1] = fmaxf(fmaf(a1[4 * q + 1], u, c1.y), 0.f);
        a1[4 * q + 2] = fmaxf(fmaf(a1[4 * q + 2], u, c1.z), 0.f);
        a1[4 * q + 3] = fmaxf(fmaf(a1[4 * q + 3], u, c1.w), 0.f);
        mx = fmaxf(fmaxf(mx, fmaxf(a0[4 * q], a0[4 * q + 1])), fmaxf(a0[4 * q + 2], a0[4 * q + 3]));
        mx = fmaxf(fmaxf(mx, fmaxf(a1[4 * q], a1[4 * q + 1])), fmaxf(a1[4 * q + 2], a1[4 * q + 3]));
      }
      mx = both_halves_max(mx);
      const float s1 = f16_scale_for(mx);
      inv_s1 = pow2_inverse(s1);
#pragma unroll
      for (int p = 0; p < 8; ++p) {          // pair p = registers 2p, 2p+1: K step p >> 2 of the block
        split2_f16x2(a0[2 * p] * s1, a0[2 * p + 1] * s1, h1h[p], h1l[p]);
        split2_f16x2(a1[2 * p] * s1, a1[2 * p + 1] * s1, h1h[8 + p], h1l[8 + p]);
      }
    }
    // ---- layers 2 (64 -> 256, relu) and 3 (256 -> 64), one 32-feature hidden block at a time ----
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      o0[i] = 0.f;
      o1[i] = 0.f;
    }
    float M = 0.f, S = 1.f;
    const float u2 = iw2 * inv_s1;
#pragma unroll 2
    for (int c = 0; c < D2 / 32; ++c) {
      f32x16 a;
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = 0.f;
      const short* wr = w2p + c * 32 * ST64;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 bh = frag_of(h1h[4 * s], h1h[4 * s + 1], h1h[4 * s + 2], h1h[4 * s + 3]);
        const f16x8 bl = frag_of(h1l[4 * s], h1l[4 * s + 1], h1l[4 * s + 2], h1l[4 * s + 3]);
        const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + 16 * s);
        const f16x8 wl = *reinterpret_cast<const f16x8*>(wr + D2 * ST64 + 16 * s);
        DESCO_MFMA3(a, wh, wl, bh, bl)
      }
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 cb = *reinterpret_cast<const float4*>(B2 + 32 * c + 8 * q + 4 * h);
        a[4 * q] = fmaxf(fmaf(a[4 * q], u2, cb.x), 0.f);
        a[4 * q + 1] = fmaxf(fmaf(a[4 * q + 1], u2, cb.y), 0.f);
        a[4 * q + 2] = fmaxf(fmaf(a[4 * q + 2], u2, cb.z), 0.f);
        a[4 * q + 3] = fmaxf(fmaf(a[4 * q + 3], u2, cb.w), 0.f);
        mx = fmaxf(fmaxf(mx, fmaxf(a[4 * q], a[4 * q + 1])), fmaxf(a[4 * q + 2], a[4 * q + 3]));
      }
      mx = fmaxf(both_halves_max(mx), M);
      const float Sn = f16_scale_for(mx);
      M = mx;
      if (__builtin_amdgcn_ballot_w64(Sn != S) != 0) {        // a row's scale fell: rescale its accumulators (exact)
        const float ratio = Sn * pow2_inverse(S);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o0[i] *= ratio;
          o1[i] *= ratio;
        }
      }
      S = Sn;
      uint32_t ph[8], pl[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) split2_f16x2(a[2 * p] * S, a[2 * p + 1] * S, ph[p], pl[p]);
      const short* w3r = w3p + 32 * c;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 bh = frag_of(ph[4 * t], ph[4 * t + 1], ph[4 * t + 2], ph[4 * t + 3]);
        const f16x8 bl = frag_of(pl[4 * t], pl[4 * t + 1], pl[4 * t + 2], pl[4 * t + 3]);
        const f16x8 wh0 = *reinterpret_cast<const f16x8*>(w3r + 16 * t);
        const f16x8 wl0 = *reinterpret_cast<const f16x8*>(w3r + D3 * ST256 + 16 * t);
        const f16x8 wh1 = *reinterpret_cast<const f16x8*>(w3r + 32 * ST256 + 16 * t);
        const f16x8 wl1 = *reinterpret_cast<const f16x8*>(w3r + (D3 + 32) * ST256 + 16 * t);
        DESCO_MFMA3(o0, wh0, wl0, bh, bl)
        DESCO_MFMA3(o1, wh1, wl1, bh, bl)
      }
    }
    // ---- output: row `row`, features 32 ob + 8 q + 4 h + 0..3 ----
    if (row < g.m) {
      const float u3 = iw3 * pow2_inverse(S);
      float* o = g.out + row * g.ldo + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 c0 = *reinterpret_cast<const float4*>(B3 + 8 * q + 4 * h);
        const float4 c1 = *reinterpret_cast<const float4*>(B3 + 32 + 8 * q + 4 * h);
        float4 v0, v1;
        v0.x = fmaf(o0[4 * q], u3, c0.x);
        v0.y = fmaf(o0[4 * q + 1], u3, c0.y);
        v0.z = fmaf(o0[4 * q + 2], u3, c0.z);
        v0.w = fmaf(o0[4 * q + 3], u3, c0.w);
        v1.x = fmaf(o1[4 * q], u3, c1.x);
        v1.y = fmaf(o1[4 * q + 1], u3, c1.y);
        v1.z = fmaf(o1[4 * q + 2], u3, c1.z);
        v1.w = fmaf(o1[4 * q + 3], u3, c1.w);
        *reinterpret_cast<float4*>(o + 8 * q) = v0;
        *reinterpret_cast<float4*>(o + 32 + 8 * q) = v1;
      }
    }
    if (!has_next) break;
    tile = tn;
  }
#undef DESCO_TAIL_LOAD
}

// ---- count head from the embeddings (desco_count_head_emb_f16x3_f32) ------------------------------------------------
// logit[b, q] = b2 + sum_c w2[c] leaky(T[b, c] + Qh[q, c]),  T[b, :] = Wt emb[b, :]   (lightning_model.py:127-131, 176-193:
// count_model on cat(emb_target, emb_query), in the separable form of desco_count_head_f32).  The [m, 256] tensor T --
// written by one launch and read by the next: 2.5 GB per 1.2 M rows -- is never formed: a 32-feature block of it comes
// out of 12 MFMAs in the transposed C/D layout above (a lane holds 16 features of ITS data row), is consumed by the
// head's add / max / fma over the Q queries straight from the accumulator registers, and is gone.  No LDS staging of T and
// no barriers in the row loop; the (negligible) matrix work runs under the other wave's vector work.
//   leaky(z) = slope z + (1 - slope) relu(z):  sum_c w2[c] leaky(.) = slope (w2.T[b]) + slope (w2.Qh[q]) + sum_c r[c] relu(T + Qh),
//   r = (1 - slope) w2;  w2.T[b] = (r.T[b]) / (1 - slope) rides along on the same r registers;
//   relu(T + Qh) = max(T, -Qh) + Qh takes the add out of the loop (its sum over c is a per-query constant).
// LDS: Wt planes (72 KB), Qh and r re-ordered so that a lane's 16 features of a block are contiguous:
//   index(c) = 16 * (2 * (c / 32) + ((c >> 2) & 1)) + (c & 3) + 4 * ((c >> 3) & 3).
constexpr int HQ = 29, HHID = 256;
constexpr size_t HEAD_LDS = (size_t)2 * HHID * ST64 * 2 + (size_t)(HQ + 1) * HHID * 4 + 32 * 4;
static_assert(HEAD_LDS <= 160 * 1024, "count head: LDS budget exceeded");

struct HeadArgs {
  const float* x;          // embeddings [m, 64]
  int64_t ldx, m;
  const short* wt;         // planes [2][256][64] of scale * Wt
  const float* st;         // {scale, 1 / scale}
  const float* qh;         // [HQ, 256] query half + bias
  int64_t ldq;
  const float* w2;         // [256]
  float b2;
  const float* b2_dev;
  float slope;
  int exp2m1;
  float* out;              // [m, HQ]
  int64_t ldo;
  unsigned grid;           // gridDim.x (set by the launcher)
};

// x + y of a register pair as ONE scalar add: left to itself hipcc pairs these sums into v_pk_add_f32 with OP_SEL on
// src1, the operand selection MI355X executes wrongly beside MFMAs (common_device.hpp; tools/check_isa.py refuses it)
__device__ __forceinline__ float pair_sum(const desco_f2 v) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
  return r;
}

__device__ __forceinline__ int head_index(const int c) {
  return 16 * (2 * (c >> 5) + ((c >> 2) & 1)) + (c & 3) + 4 * ((c >> 3) & 3);
}

// The head's vector loop over one 32-feature block: step I = (feature quad I / 29, query I % 29) adds the quad's
// (T + Qh, relu, times r) to the query's accumulator pair (query-minor, so that consecutive steps feed different
// accumulators: a query's eight packed fmas per block are a dependent chain).  The Qh quads come through a ring of four register quads:
// each is requested three steps (24 vector instructions of this wave, the SIMD's other wave on top) before its use.
// Reads and waits are inline asm so that their issue points are fixed -- written as plain C++ loads hipcc sinks each read
// to just above its use and the loop waits out a full LDS round trip per quad (measured: 0.98 ms instead of 0.6).
// Counted waits as in gossip_f16.hip: LDS operations of a wave return in order, quad I is complete once at most as many
// LGKM operations are outstanding as were issued after it (three ring reads; the compiler's own LDS accesses in between
// only make the wait stricter), and a ring slot is re-requested only after the vector instructions that read it have
// been issued (they read their operands at issue).
typedef float desco_f4 __attribute__((ext_vector_type(4)));
// one v_max_f32 (fmaxf on a value that comes out of an asm statement costs a second one: hipcc canonicalises operands it
// cannot prove quiet)
__device__ __forceinline__ float raw_max(const float a, const float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#ifndef DESCO_HEAD_RING
#define DESCO_HEAD_RING 4      // (6 and 8 measured 3-5 % slower: more registers, nothing left to cover)
#endif
constexpr int HSTEPS = HQ * 4, HRING = DESCO_HEAD_RING;
template <int I>
__device__ __forceinline__ void head_request(desco_f4& slot, const uint32_t qaddr) {
  constexpr int off = (I % HQ) * (HHID * 4) + (I / HQ) * 16;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(slot) : "v"(qaddr), "n"(off));
}
template <int I>
__device__ __forceinline__ void head_prime(desco_f4 (&ring)[HRING], const uint32_t qaddr) {
  if constexpr (I < HRING) {
    head_request<I>(ring[I], qaddr);
    head_prime<I + 1>(ring, qaddr);
  }
}
template <int I>
__device__ __forceinline__ void head_steps(desco_f4 (&ring)[HRING], const uint32_t qaddr, const desco_f2 (&t2)[8],
                                           const desco_f2 (&r2)[8], desco_f2 (&acc)[HQ]) {
  if constexpr (I < HSTEPS) {
    constexpr int j = I % HQ, v = I / HQ, younger = (HSTEPS - 1 - I) < (HRING - 1) ? (HSTEPS - 1 - I) : (HRING - 1);
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[I % HRING]) : "n"(younger));
    const desco_f4 qv = ring[I % HRING];
#if !defined(DESCO_HEAD_ADDFORM)
    // relu(t + q) = max(t, -q) + q on the NEGATED table (sum_c r[c] Qh[q, c] is a per-query constant, folded into SQ): four
    // v_max_f32 + two v_pk_fma_f32 per quad instead of eight instructions (0.83 -> 0.70 ms per 1.2 M rows;
    // -DDESCO_HEAD_ADDFORM keeps the add / relu / fma form: 2-4x smaller rounding error, both far inside the gates)
    const desco_f2 z01 = {raw_max(t2[2 * v].x, qv.x), raw_max(t2[2 * v].y, qv.y)};
    const desco_f2 z23 = {raw_max(t2[2 * v + 1].x, qv.z), raw_max(t2[2 * v + 1].y, qv.w)};
#else
    desco_f2 z01 = t2[2 * v] + desco_f2{qv.x, qv.y}, z23 = t2[2 * v + 1] + desco_f2{qv.z, qv.w};
    z01.x = fmaxf(z01.x, 0.f);
    z01.y = fmaxf(z01.y, 0.f);
    z23.x = fmaxf(z23.x, 0.f);
    z23.y = fmaxf(z23.y, 0.f);
#endif
    acc[j] = __builtin_elementwise_fma(z01, r2[2 * v], acc[j]);
    acc[j] = __builtin_elementwise_fma(z23, r2[2 * v + 1], acc[j]);
    if constexpr (I + HRING < HSTEPS) head_request<I + HRING>(ring[I % HRING], qaddr);
    head_steps<I + 1>(ring, qaddr, t2, r2, acc);
  }
}

__global__ __launch_bounds__(NWT * 64) void count_head_emb_kernel(HeadArgs g) {
  extern __shared__ __attribute__((aligned(16))) short lds_h[];
  short* Wt = lds_h;
  float* QT = reinterpret_cast<float*>(Wt + 2 * HHID * ST64);       // [HQ][256] re-ordered
  float* R = QT + HQ * HHID;                                         // [256] re-ordered (1 - slope) w2
  float* SQ = R + HHID;                                              // [32]: b2 + slope (w2.Qh[q])
  const int tid = (int)__builtin_amdgcn_workitem_id_x(), lane = tid & 63, wave = tid >> 6;
  fill_image<64, false>(Wt, g.wt, 2 * HHID, ST64, tid);
  for (int i = tid; i < HQ * HHID; i += NWT * 64) {
    const int q = i >> 8, c = i & 255;
#if !defined(DESCO_HEAD_ADDFORM)
    QT[q * HHID + head_index(c)] = -g.qh[(int64_t)q * g.ldq + c];
#else
    QT[q * HHID + head_index(c)] = g.qh[(int64_t)q * g.ldq + c];
#endif
  }
  for (int i = tid; i < HHID; i += NWT * 64) R[head_index(i)] = (1.f - g.slope) * g.w2[i];
  const float b2 = g.b2_dev ? *g.b2_dev : g.b2;
  if (tid < HQ) {
    float sdot = 0.f;
    for (int c = 0; c < HHID; ++c) sdot = fmaf(g.w2[c], g.qh[(int64_t)tid * g.ldq + c], sdot);
#if !defined(DESCO_HEAD_ADDFORM)
    SQ[tid] = sdot + b2;            // slope (w2.Qh) + (1 - slope) (w2.Qh)
#else
    SQ[tid] = fmaf(g.slope, sdot, b2);
#endif
  }
  const float iwt = g.st[1];
  __syncthreads();
  const int n = lane & 31, h = lane >> 5;
  const int64_t ntiles = (g.m + 31) / 32;
  int64_t tile = (int64_t)(int)__builtin_amdgcn_workgroup_id_x() * NWT + wave;
  if (tile >= ntiles) return;
  const int64_t tstep = (int64_t)g.grid * NWT;
  float4 xv[8];
#define DESCO_HEAD_LOAD(t_)                                                     \
  {                                                                             \
    const int64_t r_ = (t_) * 32 + n;                                           \
    const float* p_ = g.x + (r_ < g.m ? r_ : g.m - 1) * g.ldx + 8 * h;          \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                          \
      xv[2 * s_] = *reinterpret_cast<const float4*>(p_ + 16 * s_);              \
      xv[2 * s_ + 1] = *reinterpret_cast<const float4*>(p_ + 16 * s_ + 4);      \
    }                                                                           \
  }
  DESCO_HEAD_LOAD(tile)
  const short* wp = Wt + n * ST64 + 8 * h;
  const uint32_t qbase = (uint32_t)(uintptr_t)(QT + 16 * h);          // LDS byte address of this lane half's quads
  const float* rp = R + 16 * h;
  const float lin_scale = g.slope / (1.f - g.slope);
  for (;;) {
    const int64_t row = tile * 32 + n;
    uint32_t xh[16], xl[16];
    float u;
    {
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(xv[q].x), fabsf(xv[q].y))), fmaxf(fabsf(xv[q].z), fabsf(xv[q].w)));
      mx = both_halves_max(mx);
      const float sx = f16_scale_for(mx);
      u = iwt * pow2_inverse(sx);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        split2_f16x2(xv[q].x * sx, xv[q].y * sx, xh[2 * q], xl[2 * q]);
        split2_f16x2(xv[q].z * sx, xv[q].w * sx, xh[2 * q + 1], xl[2 * q + 1]);
      }
    }
    const int64_t tn = tile + tstep;
    const bool has_next = tn < ntiles;
    if (has_next) DESCO_HEAD_LOAD(tn)
    desco_f2 acc[HQ];
#pragma unroll
    for (int j = 0; j < HQ; ++j) acc[j] = desco_f2{0.f, 0.f};
    desco_f2 lin = {0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < HHID / 32; ++c) {
      f32x16 a;
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = 0.f;
      const short* wr = wp + c * 32 * ST64;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 bh = frag_of(xh[4 * s], xh[4 * s + 1], xh[4 * s + 2], xh[4 * s + 3]);
        const f16x8 bl = frag_of(xl[4 * s], xl[4 * s + 1], xl[4 * s + 2], xl[4 * s + 3]);
        const f16x8 wh = *reinterpret_cast<const f16x8*>(wr + 16 * s);
        const f16x8 wl = *reinterpret_cast<const f16x8*>(wr + HHID * ST64 + 16 * s);
        DESCO_MFMA3(a, wh, wl, bh, bl)
      }
      desco_f2 t2[8], r2[8];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 rv = *reinterpret_cast<const float4*>(rp + 32 * c + 4 * v);
        t2[2 * v] = desco_f2{a[4 * v] * u, a[4 * v + 1] * u};
        t2[2 * v + 1] = desco_f2{a[4 * v + 2] * u, a[4 * v + 3] * u};
        r2[2 * v] = desco_f2{rv.x, rv.y};
        r2[2 * v + 1] = desco_f2{rv.z, rv.w};
        lin = __builtin_elementwise_fma(t2[2 * v], r2[2 * v], lin);
        lin = __builtin_elementwise_fma(t2[2 * v + 1], r2[2 * v + 1], lin);
      }
      const uint32_t qaddr = qbase + (uint32_t)(32 * 4) * (uint32_t)c;
      desco_f4 ring[HRING];
      head_prime<0>(ring, qaddr);
      head_steps<0>(ring, qaddr, t2, r2, acc);
    }
    // both halves' partial sums; lanes of half 0 store queries 0..14, half 1 queries 15..28 (and 14 again)
    {
      const float lsum = pair_sum(lin);
      const u32x2 tl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
      const float st = lin_scale * (__uint_as_float(tl[0]) + __uint_as_float(tl[1]));
      float* o = g.out + row * g.ldo;
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        const int jb = j + 14;                               // half 1's query for this slot (14 is written twice)
        const float p0 = pair_sum(acc[j]), p1 = pair_sum(acc[jb]);
        const u32x2 t0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p0), false, false);
        const u32x2 t1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p1), __float_as_uint(p1), false, false);
        const float s0 = __uint_as_float(t0[0]) + __uint_as_float(t0[1]);
        const float s1 = __uint_as_float(t1[0]) + __uint_as_float(t1[1]);
        const float v = (h ? s1 : s0) + (st + SQ[h ? jb : j]);
        if (row < g.m) o[h ? jb : j] = g.exp2m1 ? exp2f(v) - 1.f : v;
      }
    }
    if (!has_next) break;
    tile = tn;
  }
#undef DESCO_HEAD_LOAD
}

#undef DESCO_MFMA3

}  // namespace tail
}  // namespace desco

using namespace desco;

extern "C" int desco_post_mp_tail_f16x3_f32(const float* x, int64_t ldx, int64_t m, const int16_t* w1_planes,
                                            const float* w1_scale, const float* b1, const int16_t* w2_planes,
                                            const float* w2_scale, const float* b2, const int16_t* w3_planes,
                                            const float* w3_scale, const float* b3, float* out, int64_t ldo,
                                            desco_stream_t stream) {
  if (m < 0 || !x || !w1_planes || !w1_scale || !w2_planes || !w2_scale || !w3_planes || !w3_scale || !out ||
      ldx < 64 || ldx % 4 || ldo < 64 || ldo % 4 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) ||
      ((uintptr_t)w1_planes & 7) || ((uintptr_t)w2_planes & 7) || ((uintptr_t)w3_planes & 7))
    return fail(DESCO_EINVAL, "desco_post_mp_tail_f16x3_f32: bad argument");
  if (m == 0) return 0;
  tail::TailArgs g{x, ldx, m, reinterpret_cast<const short*>(w1_planes), reinterpret_cast<const short*>(w2_planes),
                   reinterpret_cast<const short*>(w3_planes), w1_scale, w2_scale, w3_scale, b1, b2, b3, out, ldo};
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tail::post_tail_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  const int64_t tiles = (m + 31) / 32;
  const int64_t want = (tiles + tail::TNW - 1) / tail::TNW;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const unsigned grid = (unsigned)(want < cus ? want : cus);
  g.grid = grid;
  hipLaunchKernelGGL(tail::post_tail_kernel, dim3(grid), dim3(tail::TNW * 64), tail::TAIL_LDS, (hipStream_t)stream, g);
  return launch_status("desco_post_mp_tail_f16x3_f32");
}

extern "C" int desco_count_head_emb_f16x3_f32(const float* emb, int64_t lde, int64_t m, const int16_t* wt_planes,
                                              const float* wt_scale, const float* qh, int64_t ldq, int hid,
                                              const float* w2, float b2, const float* b2_dev, float slope,
                                              int exp2_minus_1, float* out, int64_t ldo, int num_q,
                                              desco_stream_t stream) {
  if (m < 0 || !emb || !wt_planes || !wt_scale || !qh || !w2 || !out || lde < 64 || lde % 4 || ((uintptr_t)emb & 15) ||
      ((uintptr_t)wt_planes & 7) || hid != tail::HHID || num_q != tail::HQ || ldq < hid || ldo < num_q ||
      !(slope < 1.f))
    return fail(DESCO_EINVAL, "desco_count_head_emb_f16x3_f32: bad argument (hid must be 256, num_q 29)");
  if (m == 0) return 0;
  tail::HeadArgs g{emb, lde, m, reinterpret_cast<const short*>(wt_planes), wt_scale, qh, ldq, w2, b2, b2_dev, slope,
                   exp2_minus_1, out, ldo};
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tail::count_head_emb_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  const int64_t tiles = (m + 31) / 32;
  const int64_t want = (tiles + tail::NWT - 1) / tail::NWT;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const unsigned grid = (unsigned)(want < cus ? want : cus);
  g.grid = grid;
  hipLaunchKernelGGL(tail::count_head_emb_kernel, dim3(grid), dim3(tail::NWT * 64), tail::HEAD_LDS, (hipStream_t)stream, g);
  return launch_status("desco_count_head_emb_f16x3_f32");
}

#include "tu_no_packed_f32_end.hpp"
