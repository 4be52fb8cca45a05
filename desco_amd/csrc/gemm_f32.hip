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
};

constexpr int BM = 128, BN = 64, BK = 32, ASTR = 33;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float lds[2 * BM * ASTR + 2 * BK * BN];
  float* As = lds;
  float* Bs = lds + 2 * BM * ASTR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nchunks = (g.k1 + g.k2) / BK;

  const int arow = tid >> 3, ac4 = tid & 7;    // A: 8 threads x float4 cover one 32-float row
  const int brow = tid >> 4, bc4 = tid & 15;   // B: 16 threads x float4 cover one 64-float row

  int64_t arows[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t r = m0 + arow + 32 * i;
    arows[i] = r < g.m ? r : g.m - 1;
  }

  float4 ra[4], rb[2];
  auto load_chunk = [&](int ch) {
    const int kk = ch * BK;
    const float* ab;
    int64_t lda;
    if (kk < g.k1) {
      ab = g.a1 + kk;
      lda = g.lda1;
    } else {
      ab = g.a2 + (kk - g.k1);
      lda = g.lda2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ra[i] = *reinterpret_cast<const float4*>(ab + arows[i] * lda + 4 * ac4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rb[i] = *reinterpret_cast<const float4*>(g.wt + (int64_t)(kk + brow + 16 * i) * g.n + n0 +
                                               4 * bc4);
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BM * ASTR;
    float* b = Bs + buf * BK * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* d = a + (arow + 32 * i) * ASTR + 4 * ac4;
      d[0] = ra[i].x;
      d[1] = ra[i].y;
      d[2] = ra[i].z;
      d[3] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float4*>(b + (brow + 16 * i) * BN + 4 * bc4) = rb[i];
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_chunk(ch + 1);
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
    if (ch + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

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
             c, ldc, m};
  const int64_t gm = (m + BM - 1) / BM;
  if (gm > INT32_MAX) return fail(DESCO_EINVAL, "desco_gemm_f32: m too large");
  dim3 grid((unsigned)gm, (unsigned)(n / BN));
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
  return launch_status("desco_gemm_f32");
}
