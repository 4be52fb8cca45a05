// Backward-pass kernels of the DeSCo hot path (gfx950).  The forward kernels are linear maps with
// pointwise non-linearities, so the backward pass needs only:
//   * weight gradients   dWt[k,n] = A[M,k]^T * dZ[M,n]          -> gemm_tn (MFMA, split over M)
//   * bias gradients     db[n]    = sum_m dZ[m,n]                -> colsum
//   * activation masks   dZ = dC * act'(C)                       -> act_grad
//   * count-head backward (lightning_model.py:176-193 in separable form)
// Input gradients reuse the forward kernels: dA = dZ * W is desco_gemm_f32 with the un-transposed
// weight, and the transpose of a CSR gather is a CSR gather over the transposed index.
// Everything is deterministic: partial sums go to a workspace and are reduced in a fixed order
// (no floating-point atomics).
#include "common_device.hpp"

namespace desco {

constexpr int TK = 64, TN = 64, TMC = 32;   // output tile 64(k) x 64(n), M chunk of 32 rows

// partial[s][k0:k0+64][n0:n0+64] = sum over the M slab s of A[m,k]^T dZ[m,n]
__global__ __launch_bounds__(256) void gemm_tn_partial_kernel(const float* __restrict__ a,
                                                              int64_t lda,
                                                              const float* __restrict__ b,
                                                              int64_t ldb, int64_t M, int K, int N,
                                                              int64_t slab,
                                                              float* __restrict__ partial) {
  __shared__ float lds[2 * TMC * 64];
  float* As = lds;              // [32 m][64 k]
  float* Bs = lds + TMC * 64;   // [32 m][64 n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int k0 = blockIdx.x * TK, n0 = blockIdx.y * TN;
  const int64_t m_beg = (int64_t)blockIdx.z * slab;
  const int64_t m_end = (m_beg + slab) < M ? (m_beg + slab) : M;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // staging map: 256 threads x 2 float4 cover a 32 x 64 chunk
  const int srow = tid >> 4, sc4 = tid & 15;
  for (int64_t m0 = m_beg; m0 < m_end; m0 += TMC) {
    float4 va0, va1, vb0, vb1;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ra = m0 + srow, rb = m0 + srow + 16;
    va0 = ra < m_end ? *reinterpret_cast<const float4*>(a + ra * lda + k0 + 4 * sc4) : zero;
    va1 = rb < m_end ? *reinterpret_cast<const float4*>(a + rb * lda + k0 + 4 * sc4) : zero;
    vb0 = ra < m_end ? *reinterpret_cast<const float4*>(b + ra * ldb + n0 + 4 * sc4) : zero;
    vb1 = rb < m_end ? *reinterpret_cast<const float4*>(b + rb * ldb + n0 + 4 * sc4) : zero;
    __syncthreads();   // previous chunk consumed
    *reinterpret_cast<float4*>(As + srow * 64 + 4 * sc4) = va0;
    *reinterpret_cast<float4*>(As + (srow + 16) * 64 + 4 * sc4) = va1;
    *reinterpret_cast<float4*>(Bs + srow * 64 + 4 * sc4) = vb0;
    *reinterpret_cast<float4*>(Bs + (srow + 16) * 64 + 4 * sc4) = vb1;
    __syncthreads();
    // MFMA: A-operand[i = k index][kk = m], B-operand[kk = m][j = n index]
    const float* as = As + (lane >> 5) * 64 + wr * 32 + (lane & 31);
    const float* bs = Bs + (lane >> 5) * 64 + wc * 32 + (lane & 31);
#pragma unroll
    for (int mm = 0; mm < TMC / 2; ++mm)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * mm * 64], bs[2 * mm * 64], acc, 0, 0, 0);
  }
  float* out = partial + ((int64_t)blockIdx.z * K + k0) * N + n0;
  const int col = wc * 32 + (lane & 31);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    out[(int64_t)row * N + col] = acc[reg];
  }
}

// Weight AND bias gradient of one Linear in one pass over the M slabs (desco_linear_bwd_w_f32):
//   partial[s][k0:k0+64][n0:n0+64] = sum over slab s of [A1 | A2][m,k]^T dZ[m,n]     (k tiles 0 .. K/64-1)
//   partial[s][K][n0:n0+64]        = sum over slab s of dZ[m,n]                      (k tile K/64: the bias row)
// A1 [M,k1] and A2 [M,k2] are the two operands of the forward GEMM (aggregate | x); the bias row rides
// in the same workspace, so ONE reduce finishes dWt and db -- six launches of the step become two.
__device__ __forceinline__ void linear_bwd_w_partial_tile(
    float* lds, const int bx, const int by, const int bz,
    const float* __restrict__ a1, int64_t lda1, int k1, const float* __restrict__ a2, int64_t lda2,
    const float* __restrict__ b, int64_t ldb, int64_t M, int K, int N, int64_t slab,
    float* __restrict__ partial, float* __restrict__ direct_dwt, int64_t lddw, float* __restrict__ direct_dbias,
    const bool want_bias = true) {
  // direct_dwt != NULL (one slab): the block's tile IS the result -- written to dwt / dbias, no reduce launch
  // want_bias == false: nobody reads the bias row (the reduce discards it) -- its workgroups leave without reading dZ
  if (!want_bias && bx * TK >= K) return;
  float* As = lds;              // [32 m][64 k]
  float* Bs = lds + TMC * 64;   // [32 m][64 n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int k0 = bx * TK, n0 = by * TN;
  const int64_t m_beg = (int64_t)bz * slab;
  const int64_t m_end = (m_beg + slab) < M ? (m_beg + slab) : M;
  const int64_t rows = (int64_t)K + 1;                      // partial rows per slab (K weight rows + bias)
  if (k0 >= K) {
    // bias row: column sums of the dZ tile over the slab (16 row groups x float4, fixed fold order)
    const int rsub = tid >> 4, c4 = 4 * (tid & 15);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t m = m_beg + rsub; m < m_end; m += 16) {
      const float4 v = *reinterpret_cast<const float4*>(b + m * ldb + n0 + c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* red = lds;                                       // [16][64]
    *reinterpret_cast<float4*>(red + rsub * 64 + c4) = s;
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 64 + tid];
      if (direct_dwt) {
        if (direct_dbias) direct_dbias[n0 + tid] = t;
      } else {
        partial[((int64_t)bz * rows + K) * N + n0 + tid] = t;
      }
    }
    return;
  }
  const float* a = k0 < k1 ? a1 + k0 : a2 + (k0 - k1);
  const int64_t lda = k0 < k1 ? lda1 : lda2;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int srow = tid >> 4, sc4 = tid & 15;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 va0, va1, vb0, vb1;
  // the next chunk's rows travel under this chunk's MFMAs (round 6: they used to be requested at the top of their own
  // iteration, a full memory round trip in front of every 16 MFMAs)
#define DESCO_BWDW_LOAD(m0_)                                                                          \
  {                                                                                                   \
    const int64_t ra_ = (m0_) + srow, rb_ = (m0_) + srow + 16;                                        \
    va0 = ra_ < m_end ? *reinterpret_cast<const float4*>(a + ra_ * lda + 4 * sc4) : zero;             \
    va1 = rb_ < m_end ? *reinterpret_cast<const float4*>(a + rb_ * lda + 4 * sc4) : zero;            \
    vb0 = ra_ < m_end ? *reinterpret_cast<const float4*>(b + ra_ * ldb + n0 + 4 * sc4) : zero;        \
    vb1 = rb_ < m_end ? *reinterpret_cast<const float4*>(b + rb_ * ldb + n0 + 4 * sc4) : zero;        \
  }
  DESCO_BWDW_LOAD(m_beg)
  int buf = 0;
  for (int64_t m0 = m_beg; m0 < m_end; m0 += TMC, buf ^= 1) {
    // two chunk images: a wave that stores chunk i + 1 has passed the barrier of chunk i, which every wave reaches only
    // after its MFMAs on chunk i - 1 (the image being overwritten) -- one barrier per chunk
    float* Ac = As + buf * (2 * TMC * 64);
    float* Bc = Bs + buf * (2 * TMC * 64);
    *reinterpret_cast<float4*>(Ac + srow * 64 + 4 * sc4) = va0;
    *reinterpret_cast<float4*>(Ac + (srow + 16) * 64 + 4 * sc4) = va1;
    *reinterpret_cast<float4*>(Bc + srow * 64 + 4 * sc4) = vb0;
    *reinterpret_cast<float4*>(Bc + (srow + 16) * 64 + 4 * sc4) = vb1;
    __syncthreads();
    if (m0 + TMC < m_end) DESCO_BWDW_LOAD(m0 + TMC)
    const float* as = Ac + (lane >> 5) * 64 + wr * 32 + (lane & 31);
    const float* bs = Bc + (lane >> 5) * 64 + wc * 32 + (lane & 31);
#pragma unroll
    for (int mm = 0; mm < TMC / 2; ++mm)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * mm * 64], bs[2 * mm * 64], acc, 0, 0, 0);
  }
#undef DESCO_BWDW_LOAD
  float* out = direct_dwt ? direct_dwt + (int64_t)k0 * lddw + n0 : partial + ((int64_t)bz * rows + k0) * N + n0;
  const int64_t ldout = direct_dwt ? lddw : N;
  const int col = wc * 32 + (lane & 31);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    out[(int64_t)row * ldout + col] = acc[reg];
  }
}

__global__ __launch_bounds__(256) void linear_bwd_w_partial_kernel(
    const float* __restrict__ a1, int64_t lda1, int k1, const float* __restrict__ a2, int64_t lda2,
    const float* __restrict__ b, int64_t ldb, int64_t M, int K, int N, int64_t slab,
    float* __restrict__ partial, float* __restrict__ direct_dwt, int64_t lddw, float* __restrict__ direct_dbias) {
  __shared__ float lds[4 * TMC * 64];          // (two chunk images: one barrier per chunk)
  linear_bwd_w_partial_tile(lds, blockIdx.x, blockIdx.y, blockIdx.z, a1, lda1, k1, a2, lda2, b, ldb, M, K, N, slab,
                            partial, direct_dwt, lddw, direct_dbias);
}

// Up to 16 independent weight/bias gradients in ONE partial launch and ONE reduce launch (desco_linear_bwd_w_multi_f32):
// the 16 (layer, row type) gradients of a training step's trunk are off its critical path -- nothing downstream waits
// for them but the optimizer -- so they are formed together after the last layer's input gradient.
constexpr int kBwdWMulti = 16;
struct BwdWMulti {
  const float* a1[kBwdWMulti];
  const float* a2[kBwdWMulti];
  const float* dz[kBwdWMulti];
  float* dwt[kBwdWMulti];
  float* dbias[kBwdWMulti];
  int64_t lda1[kBwdWMulti], lda2[kBwdWMulti], lddz[kBwdWMulti], m[kBwdWMulti], slab[kBwdWMulti];
  int64_t ws_off[kBwdWMulti];        // floats: start of the problem's partials in the workspace
  int k1[kBwdWMulti], k[kBwdWMulti], n[kBwdWMulti], splits[kBwdWMulti];
  int blk_end[kBwdWMulti];           // partial launch: running workgroup count
  int red_end[kBwdWMulti];           // reduce launch: running workgroup count
  int num;
};

__global__ __launch_bounds__(256) void linear_bwd_w_multi_partial_kernel(const BwdWMulti q, float* __restrict__ ws) {
  __shared__ float lds[4 * TMC * 64];          // (two chunk images: one barrier per chunk)
  int b = blockIdx.x, i = 0;
  while (i < q.num - 1 && b >= q.blk_end[i]) ++i;
  b -= i ? q.blk_end[i - 1] : 0;
  const int gx = q.k[i] / TK + 1, gy = q.n[i] / TN;
  const int bx = b % gx, by = (b / gx) % gy, bz = b / (gx * gy);
  const bool direct = q.splits[i] == 1;
  linear_bwd_w_partial_tile(lds, bx, by, bz, q.a1[i], q.lda1[i], q.k1[i], q.a2[i], q.lda2[i], q.dz[i], q.lddz[i],
                            q.m[i], q.k[i], q.n[i], q.slab[i], ws + q.ws_off[i], direct ? q.dwt[i] : nullptr,
                            (int64_t)q.n[i], direct ? q.dbias[i] : nullptr, q.dbias[i] != nullptr);
}

__global__ __launch_bounds__(256) void linear_bwd_w_multi_reduce_kernel(const BwdWMulti q, const float* __restrict__ ws) {
  int b = blockIdx.x, i = 0;
  while (i < q.num - 1 && b >= q.red_end[i]) ++i;
  b -= i ? q.red_end[i - 1] : 0;
  const int K = q.k[i], N = q.n[i], splits = q.splits[i];
  const int64_t e = (int64_t)b * blockDim.x + threadIdx.x;
  const int64_t count = (int64_t)(K + 1) * N;
  if (e >= count) return;
  const float* partial = ws + q.ws_off[i];
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * count + e];
  const int64_t row = e / N, col = e % N;
  if (row < K)
    q.dwt[i][row * N + col] = s;
  else if (q.dbias[i])
    q.dbias[i][col] = s;
}

// dwt[k][n] = sum_s partial[s][k][n] (k < K), dbias[n] = sum_s partial[s][K][n]   (fixed order)
__global__ __launch_bounds__(256) void linear_bwd_w_reduce_kernel(const float* __restrict__ partial, int K,
                                                                  int N, int splits,
                                                                  float* __restrict__ dwt, int64_t lddw,
                                                                  float* __restrict__ dbias) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t count = (int64_t)(K + 1) * N;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * count + i];
  const int64_t row = i / N, col = i % N;
  if (row < K)
    dwt[row * lddw + col] = s;
  else if (dbias)
    dbias[col] = s;
}

// out[i] (+)= sum_s partial[s][i]   (fixed order)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial,
                                                              int64_t count, int splits,
                                                              float* __restrict__ out, int64_t ldo,
                                                              int ncols, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * count + i];
  float* o = out + (i / ncols) * ldo + (i % ncols);
  *o = accumulate ? *o + s : s;
}

// out[i] = sum_s partial[s*stride + offset + i]   (fixed order)
__global__ __launch_bounds__(256) void reduce_strided_kernel(const float* __restrict__ partial,
                                                             int64_t stride, int64_t offset,
                                                             int64_t count, int splits,
                                                             float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * stride + offset + i];
  out[i] = s;
}

// partial[s][n] = sum over the M slab s of x[m, n]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x,
                                                             int64_t ldx, int64_t M, int N,
                                                             int64_t slab,
                                                             float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t m_beg = (int64_t)blockIdx.y * slab;
  const int64_t m_end = (m_beg + slab) < M ? (m_beg + slab) : M;
  float s = 0.f;
  if (c < N)
    for (int64_t m = m_beg + wave; m < m_end; m += 4) s += x[m * ldx + c];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < N)
    partial[(int64_t)blockIdx.y * N + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// same for 16-byte aligned rows: 16 lanes x float4 per row, 16 rows per block step, four steps in
// flight; the 16 row-group partials are folded in a fixed order
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ x,
                                                              int64_t ldx, int64_t M, int N,
                                                              int64_t slab,
                                                              float* __restrict__ partial) {
  __shared__ float red[16][64];
  const int rsub = threadIdx.x >> 4, c4 = 4 * (threadIdx.x & 15);
  const int c = blockIdx.x * 64 + c4;
  const int64_t m_beg = (int64_t)blockIdx.y * slab;
  const int64_t m_end = (m_beg + slab) < M ? (m_beg + slab) : M;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (c < N) {
    const float* xc = x + c;
    int64_t m = m_beg + rsub;
    for (; m + 48 < m_end; m += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(xc + m * ldx);
      const float4 v1 = *reinterpret_cast<const float4*>(xc + (m + 16) * ldx);
      const float4 v2 = *reinterpret_cast<const float4*>(xc + (m + 32) * ldx);
      const float4 v3 = *reinterpret_cast<const float4*>(xc + (m + 48) * ldx);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
      s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
      s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
      s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
    }
    for (; m < m_end; m += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(xc + m * ldx);
      s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    }
  }
  red[rsub][c4] = (s0.x + s1.x) + (s2.x + s3.x);
  red[rsub][c4 + 1] = (s0.y + s1.y) + (s2.y + s3.y);
  red[rsub][c4 + 2] = (s0.z + s1.z) + (s2.z + s3.z);
  red[rsub][c4 + 3] = (s0.w + s1.w) + (s2.w + s3.w);
  __syncthreads();
  const int cc = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && cc < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
    partial[(int64_t)blockIdx.y * N + cc] = t;
  }
}

__global__ __launch_bounds__(256) void act_grad_kernel(const float* __restrict__ dc,
                                                       const float* __restrict__ c, int act,
                                                       float slope, float* __restrict__ dz,
                                                       int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float g = dc[i], y = c[i];
  float d = g;
  if (act == DESCO_ACT_RELU)
    d = y > 0.f ? g : 0.f;
  else if (act == DESCO_ACT_LEAKY)
    d = y > 0.f ? g : g * slope;     // y and the pre-activation have the same sign (slope > 0)
  dz[i] = d;
}

// count head backward, part 1: dT[b,c] = w2[c] * sum_q dl[b,q] * leaky'(T[b,c] + Qh[q,c])
__global__ __launch_bounds__(256) void count_head_bwd_t_kernel(
    const float* __restrict__ t, int64_t ldt, const float* __restrict__ qh, int64_t ldq, int hid,
    const float* __restrict__ w2, float slope, const float* __restrict__ dl, int64_t lddl,
    float* __restrict__ dt, int64_t lddt, int64_t num_b, int num_q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= num_b) return;
  for (int c = lane; c < hid; c += 64) {
    const float tv = t[b * ldt + c];
    float s = 0.f;
    for (int q = 0; q < num_q; ++q) {
      const float z = tv + qh[(int64_t)q * ldq + c];
      s += dl[b * lddl + q] * (z > 0.f ? 1.f : slope);
    }
    dt[b * lddt + c] = s * w2[c];
  }
}

// part 2 (partials over b slabs): dQh[q,c] = w2[c] * sum_b dl*leaky'(z);  dw2[c] = sum_{b,q} dl*leaky(z)
// partial layout: [slab][num_q + 1][hid]  (row num_q = dw2)
__global__ __launch_bounds__(256) void count_head_bwd_q_kernel(
    const float* __restrict__ t, int64_t ldt, const float* __restrict__ qh, int64_t ldq, int hid,
    const float* __restrict__ w2, float slope, const float* __restrict__ dl, int64_t lddl,
    int64_t num_b, int num_q, int64_t slab, float* __restrict__ partial) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= hid) return;
  const int64_t b_beg = (int64_t)blockIdx.y * slab;
  const int64_t b_end = (b_beg + slab) < num_b ? (b_beg + slab) : num_b;
  float* out = partial + (int64_t)blockIdx.y * (num_q + 1) * hid;
  const float wv = w2[c];
  float dw = 0.f;
  for (int q = 0; q < num_q; ++q) {
    const float qv = qh[(int64_t)q * ldq + c];
    float s = 0.f;
    for (int64_t b = b_beg; b < b_end; ++b) {
      const float z = t[b * ldt + c] + qv;
      const float g = dl[b * lddl + q];
      s += g * (z > 0.f ? 1.f : slope);
      dw += g * (z > 0.f ? z : z * slope);
    }
    out[(int64_t)q * hid + c] = s * wv;
  }
  out[(int64_t)num_q * hid + c] = dw;
}

}  // namespace desco

using namespace desco;

extern "C" size_t desco_gemm_tn_workspace(int64_t m, int k, int n, int* splits_out) {
  // enough M slabs to fill the chip a few times over, each at least 512 rows
  const int64_t tiles = (int64_t)(k / TK) * (n / TN);
  int64_t splits = (1024 + tiles - 1) / (tiles > 0 ? tiles : 1);
  const int64_t max_splits = (m + 511) / 512;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits_out) *splits_out = (int)splits;
  return sizeof(float) * (size_t)splits * (size_t)k * (size_t)n;
}

extern "C" int desco_gemm_tn_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m,
                                 int k, int n, float* out, int64_t ldo, int accumulate,
                                 float* workspace, desco_stream_t stream) {
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!a || !b || !out || !workspace || m < 0 || k <= 0 || n <= 0 || k % TK || n % TN || lda % 4 ||
      ldb % 4 || mis16(a) || mis16(b))
    return fail(DESCO_EINVAL, "desco_gemm_tn_f32: bad argument (k%64, n%64, 16-byte alignment)");
  int splits = 1;
  desco_gemm_tn_workspace(m, k, n, &splits);
  int64_t slab = (m + splits - 1) / splits;
  slab = (slab + TMC - 1) / TMC * TMC;
  if (slab < TMC) slab = TMC;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_tn_partial_kernel, dim3(k / TK, n / TN, splits), dim3(256), 0, st, a, lda, b,
                     ldb, m, k, n, slab, workspace);
  const int64_t count = (int64_t)k * n;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                     workspace, count, splits, out, ldo, n, accumulate);
  return launch_status("desco_gemm_tn_f32");
}

// M slabs of desco_linear_bwd_w_f32 (the workspace holds splits * (k1 + k2 + 1) * n floats)
static int linear_bwd_w_splits(int64_t m, int k, int n) {
  const int64_t tiles = (int64_t)(k / TK + 1) * (n / TN);
  int64_t splits = (1024 + tiles - 1) / tiles;
  const int64_t max_splits = (m + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return (int)splits;
}

extern "C" size_t desco_linear_bwd_w_workspace(int64_t m, int k, int n) {
  return sizeof(float) * (size_t)linear_bwd_w_splits(m, k, n) * (size_t)(k + 1) * (size_t)n;
}

extern "C" int desco_linear_bwd_w_f32(const float* a1, int64_t lda1, int k1, const float* a2,
                                      int64_t lda2, int k2, const float* dz, int64_t lddz, int64_t m,
                                      int n, float* dwt, int64_t lddw, float* dbias, float* workspace,
                                      desco_stream_t stream) {
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  const int k = k1 + k2;
  if ((m > 0 && (!a1 || !dz)) || !dwt || !workspace || m < 0 || k1 <= 0 || k1 % TK || k2 < 0 || k2 % TK || n <= 0 ||
      n % TN || (k2 > 0 && ((m > 0 && !a2) || lda2 % 4 || mis16(a2))) || lda1 % 4 || lddz % 4 || mis16(a1) || mis16(dz))
    return fail(DESCO_EINVAL, "desco_linear_bwd_w_f32: bad argument (k%64, n%64, 16-byte alignment)");
  const int splits = linear_bwd_w_splits(m, k, n);
  int64_t slab = (m + splits - 1) / splits;
  slab = (slab + TMC - 1) / TMC * TMC;
  if (slab < TMC) slab = TMC;
  hipStream_t st = (hipStream_t)stream;
  if (splits == 1) {      // small M (canonical rows, query graphs, post MLP): one launch, results in place
    hipLaunchKernelGGL(linear_bwd_w_partial_kernel, dim3(k / TK + 1, n / TN, 1), dim3(256), 0, st, a1, lda1, k1,
                       a2, lda2, dz, lddz, m, k, n, slab, workspace, dwt, lddw, dbias);
    return launch_status("desco_linear_bwd_w_f32");
  }
  hipLaunchKernelGGL(linear_bwd_w_partial_kernel, dim3(k / TK + 1, n / TN, splits), dim3(256), 0, st, a1,
                     lda1, k1, a2, lda2, dz, lddz, m, k, n, slab, workspace, (float*)nullptr, (int64_t)0,
                     (float*)nullptr);
  const int64_t count = (int64_t)(k + 1) * n;
  hipLaunchKernelGGL(linear_bwd_w_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                     workspace, k, n, splits, dwt, lddw, dbias);
  return launch_status("desco_linear_bwd_w_f32");
}

extern "C" size_t desco_linear_bwd_w_multi_workspace(int num, const desco_bwd_w_desc* d) {
  size_t total = 0;
  for (int i = 0; i < num; ++i) total += desco_linear_bwd_w_workspace(d[i].m, d[i].k1 + d[i].k2, d[i].n);
  return total;
}

extern "C" int desco_linear_bwd_w_multi_f32(int num, const desco_bwd_w_desc* d, float* workspace,
                                            desco_stream_t stream) {
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (num < 0 || num > kBwdWMulti || (num > 0 && (!d || !workspace)))
    return fail(DESCO_EINVAL, "desco_linear_bwd_w_multi_f32: 0..16 problems per call, workspace required");
  BwdWMulti q;
  q.num = 0;
  int blocks = 0, rblocks = 0;
  int64_t off = 0;
  for (int i = 0; i < num; ++i) {
    const desco_bwd_w_desc& e = d[i];
    const int k = e.k1 + e.k2;
    if ((e.m > 0 && (!e.a1 || !e.dz)) || !e.dwt || e.m < 0 || e.k1 <= 0 || e.k1 % TK || e.k2 < 0 || e.k2 % TK || e.n <= 0 ||
        e.n % TN || (e.k2 > 0 && ((e.m > 0 && !e.a2) || e.lda2 % 4 || mis16(e.a2))) || e.lda1 % 4 || e.lddz % 4 || mis16(e.a1) ||
        mis16(e.dz))
      return fail(DESCO_EINVAL, "desco_linear_bwd_w_multi_f32: bad argument (k%64, n%64, 16-byte alignment)");
    const int splits = linear_bwd_w_splits(e.m, k, e.n);
    int64_t slab = (e.m + splits - 1) / splits;
    slab = (slab + TMC - 1) / TMC * TMC;
    if (slab < TMC) slab = TMC;
    const int j = q.num++;
    q.a1[j] = e.a1; q.a2[j] = e.k2 ? e.a2 : e.a1; q.dz[j] = e.dz; q.dwt[j] = e.dwt; q.dbias[j] = e.dbias;
    q.lda1[j] = e.lda1; q.lda2[j] = e.k2 ? e.lda2 : e.lda1; q.lddz[j] = e.lddz; q.m[j] = e.m; q.slab[j] = slab;
    q.ws_off[j] = off;
    q.k1[j] = e.k1; q.k[j] = k; q.n[j] = e.n; q.splits[j] = splits;
    blocks += (k / TK + 1) * (e.n / TN) * splits;
    q.blk_end[j] = blocks;
    if (splits > 1) rblocks += (int)(((int64_t)(k + 1) * e.n + 255) / 256);
    q.red_end[j] = rblocks;
    off += (int64_t)splits * (k + 1) * e.n;
  }
  if (q.num == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_bwd_w_multi_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, q, workspace);
  if (rblocks > 0)
    hipLaunchKernelGGL(linear_bwd_w_multi_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, st, q, workspace);
  return launch_status("desco_linear_bwd_w_multi_f32");
}

extern "C" int desco_colsum_f32(const float* x, int64_t ldx, int64_t m, int n, float* out,
                                int accumulate, float* workspace, desco_stream_t stream) {
  if (!x || !out || !workspace || m < 0 || n <= 0)
    return fail(DESCO_EINVAL, "desco_colsum_f32: bad argument");
  int64_t splits = (m + 255) / 256;          // enough slabs to fill the chip (workspace: 512 * n floats)
  if (splits > 512) splits = 512;
  if (splits < 1) splits = 1;
  const int64_t slab = (m + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  if (n % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3((n + 63) / 64, (unsigned)splits), dim3(256), 0, st, x,
                       ldx, m, n, slab > 0 ? slab : 1, workspace);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((n + 63) / 64, (unsigned)splits), dim3(256), 0, st, x,
                       ldx, m, n, slab > 0 ? slab : 1, workspace);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     workspace, (int64_t)n, (int)splits, out, (int64_t)n, n, accumulate);
  return launch_status("desco_colsum_f32");
}

// pre_mp's backward (desco_linear_smallk_bwd_f32): partial[s][k][n] = sum over slab s of feat[m, k] dout[m, n] (k < K),
// partial[s][K][n] = sum of dout[m, n] -- the weight rows and the bias row of one tiny-K Linear in one pass over dout
namespace desco {
__global__ __launch_bounds__(256) void smallk_bwd_partial_kernel(const float* __restrict__ feat, int64_t ldf, int K,
                                                                const float* __restrict__ dout, int64_t ldd,
                                                                int64_t M, int64_t slab,
                                                                float* __restrict__ partial) {
  __shared__ float red[16][64];
  const int rsub = threadIdx.x >> 4, c4 = 4 * (threadIdx.x & 15);
  const int64_t m_beg = (int64_t)blockIdx.x * slab;
  const int64_t m_end = (m_beg + slab) < M ? (m_beg + slab) : M;
  for (int k = 0; k <= K; ++k) {                           // (K is 1 or 2 in the reference pipeline: dout stays in L2)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t m = m_beg + rsub; m < m_end; m += 16) {
      const float f = k < K ? feat[m * ldf + k] : 1.f;
      const float4 v = *reinterpret_cast<const float4*>(dout + m * ldd + c4);
      s.x += f * v.x; s.y += f * v.y; s.z += f * v.z; s.w += f * v.w;
    }
    __syncthreads();
    *reinterpret_cast<float4*>(&red[rsub][c4]) = s;
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
      partial[((int64_t)blockIdx.x * (K + 1) + k) * 64 + threadIdx.x] = t;
    }
  }
}

}  // namespace desco

extern "C" int desco_linear_smallk_bwd_f32(const float* feat, int64_t ldf, int k, const float* dout, int64_t ldd,
                                           int64_t m, float* dwb, float* workspace, desco_stream_t stream) {
  if (!feat || !dout || !dwb || !workspace || m < 0 || k < 1 || k > 16 || ldd % 4 ||
      (reinterpret_cast<uintptr_t>(dout) & 15))
    return fail(DESCO_EINVAL, "desco_linear_smallk_bwd_f32: bad argument (1 <= k <= 16, 64 output columns)");
  int64_t splits = (m + 255) / 256;
  if (splits > 512) splits = 512;
  if (splits < 1) splits = 1;
  const int64_t slab = (m + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(smallk_bwd_partial_kernel, dim3((unsigned)splits), dim3(256), 0, st, feat, ldf, k, dout, ldd, m,
                     slab > 0 ? slab : 1, workspace);
  const int64_t count = (int64_t)(k + 1) * 64;
  hipLaunchKernelGGL(reduce_strided_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, workspace, count,
                     (int64_t)0, count, (int)splits, dwb);
  return launch_status("desco_linear_smallk_bwd_f32");
}

// dz = (dout x w) * relu'(y), per-block partials of dw = y^T dout and db = sum dout (desco_rowdot_bwd_f32)
namespace desco {
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ y, int64_t ldy, int n,
                                                        const float* __restrict__ w, const float* __restrict__ dout,
                                                        int64_t R, int64_t slab, float* __restrict__ dz, int64_t lddz,
                                                        float* __restrict__ partial) {
  __shared__ float red[256 * 4 + 64];
  const int tpr = n >> 2, rpi = 256 / tpr;                // threads per row, rows per iteration
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  const int64_t r_beg = (int64_t)blockIdx.x * slab;
  const int64_t r_end = (r_beg + slab) < R ? (r_beg + slab) : R;
  const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * tc);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  for (int64_t r = r_beg + tr; r < r_end; r += rpi) {
    const float d = dout[r];
    const float4 v = *reinterpret_cast<const float4*>(y + r * ldy + 4 * tc);
    float4 o;
    o.x = v.x > 0.f ? d * w4.x : 0.f;
    o.y = v.y > 0.f ? d * w4.y : 0.f;
    o.z = v.z > 0.f ? d * w4.z : 0.f;
    o.w = v.w > 0.f ? d * w4.w : 0.f;
    *reinterpret_cast<float4*>(dz + r * lddz + 4 * tc) = o;
    s.x += d * v.x; s.y += d * v.y; s.z += d * v.z; s.w += d * v.w;
    if (tc == 0) sb += d;
  }
  // fold the row groups in a fixed order
  *reinterpret_cast<float4*>(red + 4 * threadIdx.x) = s;
  if (tc == 0) red[1024 + tr] = sb;
  __syncthreads();
  float* out = partial + (int64_t)blockIdx.x * (n + 1);
  if (threadIdx.x < tpr) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < rpi; ++g) {
      const float4 q = *reinterpret_cast<const float4*>(red + 4 * (g * tpr + threadIdx.x));
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    float* o4 = out + 4 * threadIdx.x;       // (rows of n + 1 floats: only 4-byte aligned)
    o4[0] = t.x; o4[1] = t.y; o4[2] = t.z; o4[3] = t.w;
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int g = 0; g < rpi; ++g) t += red[1024 + g];
    out[n] = t;
  }
}

}  // namespace desco

extern "C" int desco_rowdot_bwd_f32(const float* y, int64_t ldy, int n, const float* w, const float* dout,
                                    int64_t num_rows, float* dz, int64_t lddz, float* dwb, float* workspace,
                                    desco_stream_t stream) {
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!y || !w || !dout || !dz || !dwb || !workspace || num_rows < 0 || n < 16 || n > 1024 || n % 16 || 256 % (n / 4) ||
      ldy % 4 || lddz % 4 || mis16(y) || mis16(w) || mis16(dz) || mis16(workspace))
    return fail(DESCO_EINVAL, "desco_rowdot_bwd_f32: bad argument (n = 16, 32, 64, ..., 1024: at most 64 row groups per pass; 16-byte alignment)");
  int64_t splits = (num_rows + 255) / 256;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  const int64_t slab = (num_rows + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((unsigned)splits), dim3(256), 0, st, y, ldy, n, w, dout, num_rows,
                     slab > 0 ? slab : 1, dz, lddz, workspace);
  // partial rows are n + 1 floats: columns 0..n-1 -> dw, column n -> db
  hipLaunchKernelGGL(reduce_strided_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, workspace,
                     (int64_t)(n + 1), (int64_t)0, (int64_t)(n + 1), (int)splits, dwb);
  return launch_status("desco_rowdot_bwd_f32");
}

extern "C" int desco_act_grad_f32(const float* dc, const float* c, int act, float slope, float* dz,
                                  int64_t count, desco_stream_t stream) {
  if (count == 0) return 0;
  if (!dc || !c || !dz || count < 0) return fail(DESCO_EINVAL, "desco_act_grad_f32: bad argument");
  const int64_t blocks = (count + 255) / 256;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_act_grad_f32: too many elements");
  hipLaunchKernelGGL(act_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dc, c,
                     act, slope, dz, count);
  return launch_status("desco_act_grad_f32");
}

static int64_t count_head_bwd_splits(int64_t num_b) {
  // slabs of >= 16 target rows, up to 1024 of them: a reference-size batch (512 rows) used to run on
  // 8 blocks of 64 threads with 7 400 dependent iterations each (0.9 ms of a 7 ms training step)
  int64_t splits = (num_b + 15) / 16;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  return splits;
}

extern "C" size_t desco_count_head_bwd_workspace(int64_t num_b, int num_q, int hid) {
  if (num_b < 0 || num_q < 1 || hid <= 0) return 0;
  return (size_t)count_head_bwd_splits(num_b) * (size_t)(num_q + 1) * (size_t)hid * sizeof(float);
}

extern "C" int desco_count_head_bwd_f32(const float* t, int64_t ldt, const float* qh, int64_t ldq,
                                        int hid, const float* w2, float slope, const float* dl,
                                        int64_t lddl, int64_t num_b, int num_q, float* dt,
                                        int64_t lddt, float* dqh, float* dw2, float* workspace,
                                        desco_stream_t stream) {
  if (!t || !qh || !w2 || !dl || !dt || !dqh || !dw2 || !workspace || num_b < 0 || num_q < 1 ||
      hid <= 0)
    return fail(DESCO_EINVAL, "desco_count_head_bwd_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (num_b > 0)
    hipLaunchKernelGGL(count_head_bwd_t_kernel, dim3((unsigned)((num_b + 3) / 4)), dim3(256), 0, st, t,
                       ldt, qh, ldq, hid, w2, slope, dl, lddl, dt, lddt, num_b, num_q);
  const int64_t splits = count_head_bwd_splits(num_b);
  const int64_t slab = (num_b + splits - 1) / splits;
  hipLaunchKernelGGL(count_head_bwd_q_kernel, dim3((hid + 63) / 64, (unsigned)splits), dim3(64), 0, st,
                     t, ldt, qh, ldq, hid, w2, slope, dl, lddl, num_b, num_q, slab > 0 ? slab : 1,
                     workspace);
  // partial layout [slab][num_q+1][hid]: rows 0..num_q-1 -> dQh, row num_q -> dw2
  const int64_t cq = (int64_t)num_q * hid, stride = (int64_t)(num_q + 1) * hid;
  hipLaunchKernelGGL(reduce_strided_kernel, dim3((unsigned)((cq + 255) / 256)), dim3(256), 0, st,
                     workspace, stride, (int64_t)0, cq, (int)splits, dqh);
  hipLaunchKernelGGL(reduce_strided_kernel, dim3((unsigned)((hid + 255) / 256)), dim3(256), 0, st,
                     workspace, stride, cq, (int64_t)hid, (int)splits, dw2);
  return launch_status("desco_count_head_bwd_f32");
}
