// fp32-accurate GEMM on the bf16 matrix pipe ("bf16x6"): every fp32 operand is split into three
// bf16 terms (truncation: hi + mid + lo reproduces all 24 significand bits) and the six products
// whose weight is >= 2^-16 are accumulated in fp32:
//     a*b ~= hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)            (error ~2^-23 |a||b|)
// v_mfma_f32_32x32x16_bf16 runs at 16x the f32 MFMA rate, so six of them per 16-deep step cost
// 192 cycles against 512 for eight v_mfma_f32_32x32x2_f32 -- and the operand fragments are 16-byte
// LDS reads.  Same contract as desco_gemm_f32 except that the weight is passed n-major
// (w[n][k], torch's native [out, in] layout).
//
// Tiling: 256 threads = 4 waves, block tile 128 rows x 64 cols, wave w owns rows 32w..32w+31 and both
// 32-wide column halves.  K chunks of 32: global float4 -> split in registers -> three bf16 planes
// per operand in LDS ([row][32+8] bf16, 80-byte rows: conflict-free ds_read_b128).
#include "common_device.hpp"

namespace desco {

struct GemmSplitArgs {
  const float* a1;
  int64_t lda1;
  int k1;
  const float* a2;
  int64_t lda2;
  int k2;
  const float* w;     // [n][k1+k2]
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
};

using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int SBM = 128, SBN = 64, SBK = 32, SST = 40;   // SST: plane row stride in bf16 (80 B)
constexpr int APLANE = SBM * SST, BPLANE = SBN * SST;    // elements per plane

// split 4 floats into three planes of 4 packed bf16 (8 bytes each)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u = __float_as_uint(f[i]);
    h[i] = u & 0xffff0000u;
    const float r1 = f[i] - __uint_as_float(h[i]);
    m[i] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m[i]);
    l[i] = __float_as_uint(r2) & 0xffff0000u;
  }
  hi = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
  mid = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
  lo = make_uint2((l[0] >> 16) | l[1], (l[2] >> 16) | l[3]);
}

__global__ __launch_bounds__(256) void gemm_split_kernel(GemmSplitArgs g) {
  __shared__ __attribute__((aligned(16))) short lds[3 * APLANE + 3 * BPLANE];
  short* Ap = lds;                  // planes hi, mid, lo of the A chunk [128][40]
  short* Bp = lds + 3 * APLANE;     // planes hi, mid, lo of the W chunk [64][40]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * SBM;
  const int n0 = blockIdx.y * SBN;
  const int K = g.k1 + g.k2;
  const int nchunks = K / SBK;

  // staging maps: A 128 rows x 8 float4 -> 4 per thread; W 64 rows x 8 float4 -> 2 per thread
  const int arow = tid >> 3, ac4 = tid & 7;
  int64_t r0 = m0 + arow, r1 = r0 + 32, r2 = r0 + 64, r3 = r0 + 96;
  const int64_t mlast = g.m - 1;
  r0 = r0 < g.m ? r0 : mlast;
  r1 = r1 < g.m ? r1 : mlast;
  r2 = r2 < g.m ? r2 : mlast;
  r3 = r3 < g.m ? r3 : mlast;
  const float* p10 = g.a1 + r0 * g.lda1 + 4 * ac4;
  const float* p11 = g.a1 + r1 * g.lda1 + 4 * ac4;
  const float* p12 = g.a1 + r2 * g.lda1 + 4 * ac4;
  const float* p13 = g.a1 + r3 * g.lda1 + 4 * ac4;
  const float* p20 = g.k2 ? g.a2 + r0 * g.lda2 + 4 * ac4 - g.k1 : p10;
  const float* p21 = g.k2 ? g.a2 + r1 * g.lda2 + 4 * ac4 - g.k1 : p11;
  const float* p22 = g.k2 ? g.a2 + r2 * g.lda2 + 4 * ac4 - g.k1 : p12;
  const float* p23 = g.k2 ? g.a2 + r3 * g.lda2 + 4 * ac4 - g.k1 : p13;
  const float* pw0 = g.w + (int64_t)(n0 + arow) * K + 4 * ac4;
  const float* pw1 = pw0 + (int64_t)32 * K;

  float4 ra0, ra1, ra2, ra3, rb0, rb1;
#define DESCO_LOAD_CHUNK(kk_)                                         \
  {                                                                   \
    const int k_ = (kk_);                                             \
    const bool s1_ = k_ < g.k1;                                       \
    ra0 = *reinterpret_cast<const float4*>((s1_ ? p10 : p20) + k_);   \
    ra1 = *reinterpret_cast<const float4*>((s1_ ? p11 : p21) + k_);   \
    ra2 = *reinterpret_cast<const float4*>((s1_ ? p12 : p22) + k_);   \
    ra3 = *reinterpret_cast<const float4*>((s1_ ? p13 : p23) + k_);   \
    rb0 = *reinterpret_cast<const float4*>(pw0 + k_);                 \
    rb1 = *reinterpret_cast<const float4*>(pw1 + k_);                 \
  }
#define DESCO_PUT(base_, plane_, row_, v_)                                                        \
  {                                                                                               \
    uint2 h_, m_, l_;                                                                             \
    split4(v_, h_, m_, l_);                                                                       \
    short* d_ = (base_) + (row_)*SST + 4 * ac4;                                                   \
    *reinterpret_cast<uint2*>(d_) = h_;                                                           \
    *reinterpret_cast<uint2*>(d_ + (plane_)) = m_;                                                \
    *reinterpret_cast<uint2*>(d_ + 2 * (plane_)) = l_;                                            \
  }
#define DESCO_STORE_CHUNK()                       \
  {                                               \
    DESCO_PUT(Ap, APLANE, arow, ra0)              \
    DESCO_PUT(Ap, APLANE, arow + 32, ra1)         \
    DESCO_PUT(Ap, APLANE, arow + 64, ra2)         \
    DESCO_PUT(Ap, APLANE, arow + 96, ra3)         \
    DESCO_PUT(Bp, BPLANE, arow, rb0)              \
    DESCO_PUT(Bp, BPLANE, arow + 32, rb1)         \
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  DESCO_LOAD_CHUNK(0)
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch > 0) __syncthreads();          // previous chunk's fragments have been read
    DESCO_STORE_CHUNK()
    __syncthreads();
    const int chn = ch + 1 < nchunks ? ch + 1 : ch;
    DESCO_LOAD_CHUNK(chn * SBK)            // in flight under the MFMAs
    // lane (r = lane&31, h = lane>>5): A[row r][k = 16 s + 8 h + j], B[k = 16 s + 8 h + j][col r]
    const short* ap = Ap + (wave * 32 + (lane & 31)) * SST + 8 * (lane >> 5);
    const short* bp = Bp + (lane & 31) * SST + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < SBK / 16; ++s) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 16 * s);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + APLANE + 16 * s);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * APLANE + 16 * s);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const short* bt = bp + t * 32 * SST + 16 * s;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bt);
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bt + BPLANE);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bt + 2 * BPLANE);
        f32x16 a = t == 0 ? acc0 : acc1;
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a, 0, 0, 0);
        if (t == 0)
          acc0 = a;
        else
          acc1 = a;
      }
    }
  }
#undef DESCO_LOAD_CHUNK
#undef DESCO_PUT
#undef DESCO_STORE_CHUNK

  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = lane & 31;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int gcol = n0 + 32 * t + col;
    float wsv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < g.ns; ++j) wsv[j] = g.ws[(int64_t)j * g.n + gcol];
    const float b_single = (g.bias && g.bias_rows == 1) ? g.bias[gcol] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int64_t grow = m0 + row;
      if (grow < g.m) {
        float v = t == 0 ? acc0[reg] : acc1[reg];
        if (g.bias) {
          if (g.bias_rows == 1)
            v += b_single;
          else
            v += g.bias[(grow % g.bias_rows) * g.n + gcol];
        }
        for (int j = 0; j < g.ns; ++j) v += g.s[grow * g.ns + j] * wsv[j];
        g.c[grow * g.ldc + gcol] = apply_act(v, g.act, g.slope);
      }
    }
  }
}

}  // namespace desco

extern "C" int desco_gemm_bf16x6_f32(const float* a1, int64_t lda1, int k1, const float* a2,
                                     int64_t lda2, int k2, const float* w, int n, const float* bias,
                                     int bias_rows, const float* s, int ns, const float* ws, int act,
                                     float slope, float* c, int64_t ldc, int64_t m,
                                     desco_stream_t stream) {
  using namespace desco;
  if (m == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (m < 0 || !a1 || !w || !c || k1 <= 0 || k1 % SBK || k2 < 0 || k2 % SBK || n <= 0 || n % SBN ||
      (k2 > 0 && !a2) || ns < 0 || ns > 4 || (ns > 0 && (!s || !ws)) || (bias && bias_rows < 1) ||
      lda1 % 4 || (k2 > 0 && lda2 % 4) || mis16(a1) || (k2 > 0 && mis16(a2)) || mis16(w))
    return fail(DESCO_EINVAL, "desco_gemm_bf16x6_f32: bad argument (k%32, n%64, 16-byte alignment)");
  GemmSplitArgs g{a1, lda1, k1, a2, lda2, k2, w, n, bias, bias ? bias_rows : 1, s, ns, ws, act, slope,
                  c, ldc, m};
  const int64_t gm = (m + SBM - 1) / SBM;
  if (gm > INT32_MAX) return fail(DESCO_EINVAL, "desco_gemm_bf16x6_f32: m too large");
  hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)gm, (unsigned)(n / SBN)), dim3(256), 0,
                     (hipStream_t)stream, g);
  return launch_status("desco_gemm_bf16x6_f32");
}
