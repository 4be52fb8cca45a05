// Dense projections of the DeSCo hot path on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain per output element, MI355X guide section 3).
//
//   C[m, n] = act( [A1 | A2][m, :] * Wt + bias[(m % bias_rows), n] + sum_j S[m, j] * Ws[j, n] )
//
// Replaces every nn.Linear call site of the path: SAGEConv.lin + updates (gnn_model.py:395, 264),
// anchor_mlp / post_mp (:40-53), count_model.0 (lightning_model.py:127-131), GossipConv
// lin_com / lin_update (gnn_model.py:291-292) -- see DESIGN.md for the algebra that folds them.
//
// Tiling: 256 threads = 4 wavefronts; block tile 128 rows x 64 cols; wave w owns rows 32w..32w+31
// and both 32-wide column halves (2 accumulators, one A fragment feeds two MFMAs).  K is walked in
// chunks of 32 through a double-buffered LDS image; the next chunk's global loads are issued
// before the current chunk's 32 MFMAs and written to LDS after them.
//   A image  [128][33] floats (pad 1: lanes 0..31 read 32 different rows at one k -> 32 banks)
//   B image  [32][64]  floats (lanes 0..31 read 32 consecutive columns of one k row)
#include "common_device.hpp"

namespace desco {

struct GemmArgs {
  const float* a1;
  int64_t lda1;
  int k1;
  const float* a2;
  int64_t lda2;
  int k2;
  const float* wt;
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
  // training backward (desco_gemm_f32_multi only): c = v * act'(gate) with gate the saved activation OUTPUT the
  // produced gradient belongs to (the act_grad pass fused into the epilogue), and c += instead of c = (accum)
  const float* gate;
  int64_t ldg;
  int gate_act;
  float gate_slope;
  int accum;
  // dropout factor of (row, col) on the stored value (desco_dropout; regenerated from the step's key, never stored)
  DropArgs drop;
};

constexpr int BM = 128, BN = 64, BK = 32, ASTR = 33;

__device__ __forceinline__ void gemm_f32_tile(const GemmArgs& g, float* lds, const int64_t m0, const int n0) {
  float* As = lds;
  float* Bs = lds + 2 * BM * ASTR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = (g.k1 + g.k2) / BK;

  const int arow = tid >> 3, ac4 = tid & 7;    // A: 8 threads x float4 cover one 32-float row
  const int brow = tid >> 4, bc4 = tid & 15;   // B: 16 threads x float4 cover one 64-float row

  // Per-thread element offsets of its 4 A rows in both K segments (kept in registers: no arrays
  // indexed at run time, no lambdas -- hipcc otherwise parks the prefetch registers in scratch).
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
  const float* pb = g.wt + (int64_t)brow * g.n + n0 + 4 * bc4;
  const int64_t bstep = (int64_t)16 * g.n;

  float4 ra0, ra1, ra2, ra3, rb0, rb1;
#define DESCO_LOAD_CHUNK(kk_)                                                     \
  {                                                                               \
    const int k_ = (kk_);                                                         \
    const bool s1_ = k_ < g.k1;                                                   \
    ra0 = *reinterpret_cast<const float4*>((s1_ ? p10 : p20) + k_);               \
    ra1 = *reinterpret_cast<const float4*>((s1_ ? p11 : p21) + k_);               \
    ra2 = *reinterpret_cast<const float4*>((s1_ ? p12 : p22) + k_);               \
    ra3 = *reinterpret_cast<const float4*>((s1_ ? p13 : p23) + k_);               \
    rb0 = *reinterpret_cast<const float4*>(pb + (int64_t)k_ * g.n);               \
    rb1 = *reinterpret_cast<const float4*>(pb + (int64_t)k_ * g.n + bstep);       \
  }
#define DESCO_STORE_A(dst_, v_)  \
  {                              \
    float* d_ = (dst_);          \
    d_[0] = (v_).x;              \
    d_[1] = (v_).y;              \
    d_[2] = (v_).z;              \
    d_[3] = (v_).w;              \
  }
#define DESCO_STORE_CHUNK(buf_)                                                        \
  {                                                                                    \
    float* a_ = As + (buf_)*BM * ASTR + arow * ASTR + 4 * ac4;                         \
    float* b_ = Bs + (buf_)*BK * BN + brow * BN + 4 * bc4;                             \
    DESCO_STORE_A(a_, ra0)                                                             \
    DESCO_STORE_A(a_ + 32 * ASTR, ra1)                                                 \
    DESCO_STORE_A(a_ + 64 * ASTR, ra2)                                                 \
    DESCO_STORE_A(a_ + 96 * ASTR, ra3)                                                 \
    *reinterpret_cast<float4*>(b_) = rb0;                                              \
    *reinterpret_cast<float4*>(b_ + 16 * BN) = rb1;                                    \
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  DESCO_LOAD_CHUNK(0)
  DESCO_STORE_CHUNK(0)
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    // unconditional prefetch of the next chunk (the last iteration re-reads its own chunk)
    const int chn = ch + 1 < nchunks ? ch + 1 : ch;
    DESCO_LOAD_CHUNK(chn * BK)
    // MFMA 32x32x2 f32 operand maps: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
    const float* as = As + buf * BM * ASTR + (wave * 32 + (lane & 31)) * ASTR + (lane >> 5);
    const float* bs = Bs + buf * BK * BN + (lane >> 5) * BN + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a = as[2 * kk];
      const float b0 = bs[2 * kk * BN];
      const float b1 = bs[2 * kk * BN + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    DESCO_STORE_CHUNK(buf ^ 1)
    __syncthreads();
  }
#undef DESCO_LOAD_CHUNK
#undef DESCO_STORE_CHUNK
#undef DESCO_STORE_A

  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = lane & 31;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int gcol = n0 + 32 * t + col;
    float wsv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < g.ns; ++j) wsv[j] = g.ws[(int64_t)j * g.n + gcol];
    const float b_single = (g.bias && g.bias_rows == 1) ? g.bias[gcol] : 0.f;
    // registers 4j..4j+3 of a lane are four consecutive rows (aligned to 4: m0 is a multiple of 128) of one column:
    // one Philox call per register quad
    float fac[16];
    if (g.drop.key) {
      const uint64_t seed = g.drop.key[0], step = g.drop.key[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r4 = (m0 + wave * 32 + 8 * j + 4 * (lane >> 5)) >> 2;
        const PhiloxOut o = dropout_bits4(g.drop, seed, step, (uint32_t)r4, (uint32_t)gcol);
#pragma unroll
        for (int i = 0; i < 4; ++i) fac[4 * j + i] = o.w[i] < g.drop.threshold ? 0.f : g.drop.scale;
      }
    }
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
        v = apply_act(v, g.act, g.slope);
        if (g.drop.key) v *= fac[reg];
        if (g.gate) {
          const float o_ = g.gate[grow * g.ldg + gcol];
          v = o_ > 0.f ? v : (g.gate_act == DESCO_ACT_RELU ? 0.f : g.gate_act == DESCO_ACT_LEAKY ? v * g.gate_slope : v);
        }
        if (g.accum) v += g.c[grow * g.ldc + gcol];
        g.c[grow * g.ldc + gcol] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float lds[2 * BM * ASTR + 2 * BK * BN];
  gemm_f32_tile(g, lds, (int64_t)blockIdx.x * BM, blockIdx.y * BN);
}

// Several independent products in ONE launch (desco_gemm_f32_multi): the count-row and canonical-row halves of a
// training layer -- different row ranges, different weights, nothing to wait for between them -- were two launches
// of a few microseconds each, forty of them per step.  Workgroups [blk_end[i-1], blk_end[i]) belong to product i.
constexpr int kGemmMulti = 4;
struct GemmMulti {
  GemmArgs g[kGemmMulti];
  int blk_end[kGemmMulti];
  int num;
};

__global__ __launch_bounds__(256) void gemm_f32_multi_kernel(GemmMulti mg) {
  __shared__ float lds[2 * BM * ASTR + 2 * BK * BN];
  int b = blockIdx.x, i = 0;
  while (i < mg.num - 1 && b >= mg.blk_end[i]) ++i;
  b -= i ? mg.blk_end[i - 1] : 0;
  const GemmArgs& g = mg.g[i];                       // (uniform index: scalar loads from the kernel arguments)
  const int gm = (int)((g.m + BM - 1) / BM);
  gemm_f32_tile(g, lds, (int64_t)(b % gm) * BM, (b / gm) * BN);
}

}  // namespace desco

namespace desco {
const char* dropout_check(const desco_dropout* d, int64_t num_rows, int64_t num_cols);   // dropout.hip
}

static const char* gemm_f32_check(const desco::GemmArgs& g) {
  using namespace desco;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (g.m < 0 || !g.a1 || !g.wt || !g.c || g.k1 <= 0 || g.k1 % BK || g.k2 < 0 || g.k2 % BK || g.n <= 0 || g.n % BN ||
      (g.k2 > 0 && !g.a2) || g.ns < 0 || g.ns > 4 || (g.ns > 0 && (!g.s || !g.ws)) || (g.bias && g.bias_rows < 1) ||
      g.lda1 % 4 || (g.k2 > 0 && g.lda2 % 4) || mis16(g.a1) || (g.k2 > 0 && mis16(g.a2)) || mis16(g.wt))
    return "bad argument (k%32, n%64, 16-byte alignment)";
  if ((g.m + BM - 1) / BM * (g.n / BN) > INT32_MAX / 2) return "m too large";
  return nullptr;
}

extern "C" int desco_gemm_f32_multi(int num, const desco_gemm_desc* d, desco_stream_t stream) {
  using namespace desco;
  if (num < 0 || num > kGemmMulti || (num > 0 && !d))
    return fail(DESCO_EINVAL, "desco_gemm_f32_multi: 0..4 products per launch");
  GemmMulti mg;
  mg.num = 0;
  int blocks = 0;
  for (int i = 0; i < num; ++i) {
    if (d[i].m == 0) continue;
    GemmArgs g{d[i].a1, d[i].lda1, d[i].k1, d[i].a2, d[i].lda2, d[i].k2, d[i].wt, d[i].n, d[i].bias,
               d[i].bias ? d[i].bias_rows : 1, d[i].s, d[i].ns, d[i].ws, d[i].act, d[i].slope, d[i].c, d[i].ldc, d[i].m,
               d[i].gate, d[i].ldg, d[i].gate_act, d[i].gate_slope, d[i].accum,
               DropArgs{d[i].drop.key, d[i].drop.site, d[i].drop.threshold, d[i].drop.scale}};
    if (d[i].drop.key) {
      if (const char* why = dropout_check(&d[i].drop, d[i].m, d[i].n)) return fail(DESCO_EINVAL, why);
    }
    if (const char* why = gemm_f32_check(g)) {
      std::string msg = std::string("desco_gemm_f32_multi: ") + why;
      return fail(DESCO_EINVAL, msg.c_str());
    }
    blocks += (int)((g.m + BM - 1) / BM) * (g.n / BN);
    mg.g[mg.num] = g;
    mg.blk_end[mg.num++] = blocks;
  }
  if (mg.num == 0) return 0;
  for (int i = mg.num; i < kGemmMulti; ++i) {
    mg.g[i] = mg.g[0];
    mg.blk_end[i] = blocks;
  }
  hipLaunchKernelGGL(gemm_f32_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mg);
  return launch_status("desco_gemm_f32_multi");
}

extern "C" int desco_gemm_f32(const float* a1, int64_t lda1, int k1, const float* a2, int64_t lda2,
                              int k2, const float* wt, int n, const float* bias, int bias_rows,
                              const float* s, int ns, const float* ws, int act, float slope,
                              float* c, int64_t ldc, int64_t m, desco_stream_t stream) {
  using namespace desco;
  if (m == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (m < 0 || !a1 || !wt || !c || k1 <= 0 || k1 % BK || k2 < 0 || k2 % BK || n <= 0 || n % BN ||
      (k2 > 0 && !a2) || ns < 0 || ns > 4 || (ns > 0 && (!s || !ws)) || (bias && bias_rows < 1) ||
      lda1 % 4 || (k2 > 0 && lda2 % 4) || mis16(a1) || (k2 > 0 && mis16(a2)) || mis16(wt))
    return fail(DESCO_EINVAL, "desco_gemm_f32: bad argument (k%32, n%64, 16-byte alignment)");
  GemmArgs g{a1, lda1, k1, a2, lda2, k2, wt, n, bias, bias ? bias_rows : 1, s, ns, ws, act, slope,
             c, ldc, m, nullptr, 0, 0, 0.f, 0, DropArgs{nullptr, 0u, 0u, 1.f}};
  const int64_t gm = (m + BM - 1) / BM;
  if (gm > INT32_MAX) return fail(DESCO_EINVAL, "desco_gemm_f32: m too large");
  dim3 grid((unsigned)gm, (unsigned)(n / BN));
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
  return launch_status("desco_gemm_f32");
}
