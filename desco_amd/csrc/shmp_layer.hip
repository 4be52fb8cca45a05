// Fused SHMP layer for gfx950: SAGEConv gather-aggregate for every relation slot, the per-relation
// Linear, the to_hetero sum, the update Linear and the ReLU (gnn_model.py:262-264, 273, 389-395) in
// ONE launch.  The aggregates never touch HBM: each 64-row tile gathers one relation slot at a
// time into LDS and multiplies it by that slot's folded 64x64 weight block on the f32 MFMA.
//
//   out[i] = relu( sum_{s<su} (sum_{e in vrow(i,s)} x[vcol[e]]) * Wt[s] + x[i] * Wt[su] + bias )
//
// Block = 256 threads = 4 waves, tile = 64 destination rows, output 64x64 as 2x2 wave tiles of
// 32x32 (v_mfma_f32_32x32x2_f32).  Per K-block (one slot or the self term):
//   gather : wave w fills rows 16w..16w+15; a 16-lane group (float4 per lane = one 256-B row per
//            neighbour) walks one row's edge list, 4 rows in flight per wave   -> A image [64][65]
//   weights: the slot's 64x64 block, 4 x float4 per thread                      -> B image [64][64]
//   32 MFMAs per wave.
// LDS 33 KB -> 4 blocks per CU; other blocks' MFMA phases cover a block's gather latency.
// HBM traffic per row and layer: one 256-B read of x (neighbour reads hit L2: a neighborhood's rows
// are contiguous), one 256-B write, ~20 B of indices -- vs 2816 B for the unfused gather + GEMM.
#include "common_device.hpp"

namespace desco {

constexpr int TM = 64, AS = 65;

struct ShmpArgs {
  const float* x;
  int64_t ldx;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int64_t row0, num_rows;
  int S, su;
  const float* wt;
  const float* bias;
  float* out;
  int64_t ldo;
};

__global__ __launch_bounds__(256) void shmp_layer_f32_kernel(ShmpArgs g) {
  __shared__ float lds[TM * AS + 64 * 64];
  float* As = lds;
  float* Bs = lds + TM * AS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * TM;          // first tile row (relative to row0)
  const int grp = lane >> 4, l16 = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int s = 0; s <= g.su; ++s) {
    // ---- weights of this K-block: Wt rows 64s .. 64s+63 ------------------------------------
    const float* wsrc = g.wt + (int64_t)s * 64 * 64;
    float4 w0 = *reinterpret_cast<const float4*>(wsrc + tid * 4);
    float4 w1 = *reinterpret_cast<const float4*>(wsrc + 1024 + tid * 4);
    float4 w2 = *reinterpret_cast<const float4*>(wsrc + 2048 + tid * 4);
    float4 w3 = *reinterpret_cast<const float4*>(wsrc + 3072 + tid * 4);
    // ---- gather (or self copy) into the A image --------------------------------------------
    float4 av[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = wave * 16 + it * 4 + grp;
      const int64_t lr = t0 + r;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lr < g.num_rows) {
        const int64_t grow = g.row0 + lr;
        if (s == g.su) {
          a = *reinterpret_cast<const float4*>(g.x + grow * g.ldx + 4 * l16);
        } else {
          const int64_t v = grow * g.S + s;
          const int e0 = g.vrowptr[v], e1 = g.vrowptr[v + 1];
          float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
          int e = e0;
          for (; e + 1 < e1; e += 2) {
            const int64_t j0 = g.vcol[e], j1 = g.vcol[e + 1];
            const float4 u = *reinterpret_cast<const float4*>(g.x + j0 * g.ldx + 4 * l16);
            const float4 w = *reinterpret_cast<const float4*>(g.x + j1 * g.ldx + 4 * l16);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
          }
          if (e < e1) {
            const int64_t j0 = g.vcol[e];
            const float4 u = *reinterpret_cast<const float4*>(g.x + j0 * g.ldx + 4 * l16);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
          }
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
      }
      av[it] = a;
    }
    if (s > 0) __syncthreads();   // previous K-block's MFMAs have consumed the images
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float* d = As + (wave * 16 + it * 4 + grp) * AS + 4 * l16;
      d[0] = av[it].x;
      d[1] = av[it].y;
      d[2] = av[it].z;
      d[3] = av[it].w;
    }
    *reinterpret_cast<float4*>(Bs + tid * 4) = w0;
    *reinterpret_cast<float4*>(Bs + 1024 + tid * 4) = w1;
    *reinterpret_cast<float4*>(Bs + 2048 + tid * 4) = w2;
    *reinterpret_cast<float4*>(Bs + 3072 + tid * 4) = w3;
    __syncthreads();
    // ---- 32 MFMAs: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31] ------------------
    const float* as = As + (wr * 32 + (lane & 31)) * AS + (lane >> 5);
    const float* bs = Bs + (lane >> 5) * 64 + wc * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * kk], bs[2 * kk * 64], acc, 0, 0, 0);
  }

  // ---- epilogue: bias + relu; C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  const int col = wc * 32 + (lane & 31);
  const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int64_t lr = t0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    if (lr < g.num_rows) {
      const float v = acc[reg] + bv;
      g.out[(g.row0 + lr) * g.ldo + col] = v > 0.f ? v : 0.f;
    }
  }
}

}  // namespace desco

extern "C" int desco_shmp_layer_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                    const int32_t* vcol, int64_t row0, int64_t num_rows,
                                    int slots_stored, int slots_used, const float* wt,
                                    const float* bias, float* out, int64_t ldo,
                                    desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x || !vrowptr || !wt || !out || row0 < 0 || num_rows < 0 || slots_used < 0 ||
      slots_used > slots_stored || slots_stored < 1 || ldx % 4 || mis16(x) || mis16(wt))
    return fail(DESCO_EINVAL, "desco_shmp_layer_f32: bad argument");
  const int64_t blocks = (num_rows + TM - 1) / TM;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_shmp_layer_f32: too many rows");
  ShmpArgs g{x, ldx, vrowptr, vcol, row0, num_rows, slots_stored, slots_used, wt, bias, out, ldo};
  hipLaunchKernelGGL(shmp_layer_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, g);
  return launch_status("desco_shmp_layer_f32");
}
