// The SHMP trunk of a SMALL batch -- the 29 query graphs of the neighborhood model, 135 rows -- in one workgroup.
//
// Reference: NeighborhoodCountingModel.train_forward recomputes the query embeddings on every training batch
// (subgraph_counting/lightning_model.py:204-207, 228-254): emb_model_query's BaseGNNCore.forward (gnn_model.py:253-277)
// over the union of the query graphs, L = 8 SAGE layers with two relation slots ("union_node" rows), global_add_pool
// per layer (:88-89, 107).  As launches of the general kernels that was 18 forward and 19 backward launches of a few
// microseconds each per training step, every one of them a fraction of one CU's work.  Here the whole trunk is one
// launch per direction: the rows live in LDS, the layers are separated by workgroup barriers.
//
//   forward   X_0 = x0;  A_l = [agg_0(X_l) | agg_1(X_l) | X_l];  X_{l+1} = relu(A_l Wt_l + b_l);
//             pooled[b, 64 l : 64 (l+1)] = sum of X_l over the rows of graph b
//   backward  G_L = pool seed;  dZ_l = G_{l+1} * (X_{l+1} > 0);  dWt_l = A_l^T dZ_l;  db_l = colsum dZ_l;
//             D_l = dZ_l Wt_l^T;  G_l[i] = D_l[i, self] + sum over the transposed index of D_l's slot blocks + pool seed
//
// LDS: rows [n][64] (X / the running gradient) + [n][192] (A, then D): 147 456 B at n = 144, the most this takes.
// Products are plain fp32 FMA chains in k order (exact fp32, like the f32 MFMA path): 1.7 MFLOP per layer.
#include "common_device.hpp"

namespace desco {
namespace small {

constexpr int NT = 1024;      // threads: 16 waves, 4 per SIMD
constexpr int NMAX = 144;     // rows
constexpr int KA = 192;       // (2 slots + self) x 64

struct FwdArgs {
  const float* x0;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int n, L;
  const float* wt;            // [L][192][64]
  const float* bias;          // [L][64]
  const int32_t* seg_ptr;     // [B + 1]
  int B;
  float* xall;                // [L][n][64]
  float* pooled;              // [B][ldp], columns 64 (L + 1)
  int64_t ldp;
};

__device__ __forceinline__ void f4fma(float4& acc, const float a, const float4 w) {
  acc.x = fmaf(a, w.x, acc.x);
  acc.y = fmaf(a, w.y, acc.y);
  acc.z = fmaf(a, w.z, acc.z);
  acc.w = fmaf(a, w.w, acc.w);
}

// A[i] = [sum of the slot-0 sources | sum of the slot-1 sources | the row itself] of X (LDS or global, rows of 64)
__device__ __forceinline__ void build_a(float* __restrict__ A, const float* X, const int32_t* __restrict__ vrowptr,
                                        const int32_t* __restrict__ vcol, const int n) {
  for (int u = threadIdx.x; u < n * 3 * 16; u += NT) {
    const int c4 = 4 * (u & 15), v = u >> 4, i = v / 3, s = v - 3 * i;
    float4 acc;
    if (s == 2) {
      acc = *reinterpret_cast<const float4*>(X + i * 64 + c4);
    } else {
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int e1 = vrowptr[i * 2 + s + 1];
      for (int e = vrowptr[i * 2 + s]; e < e1; ++e) {
        const float4 t = *reinterpret_cast<const float4*>(X + vcol[e] * 64 + c4);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
    }
    *reinterpret_cast<float4*>(A + i * KA + s * 64 + c4) = acc;
  }
}

__global__ __launch_bounds__(NT) void shmp_small_fwd_kernel(const FwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;                       // [n][64]
  float* A = lds + NMAX * 64;           // [n][192]
  const int t = threadIdx.x, n = g.n;
  for (int u = t; u < n * 16; u += NT)
    *reinterpret_cast<float4*>(X + 4 * u) = *reinterpret_cast<const float4*>(g.x0 + 4 * u);
  __syncthreads();
  const int col4 = 4 * (t & 15), rg = t >> 4;             // GEMM: 16 lanes x float4 per row, 64 row groups
  for (int l = 0;; ++l) {
    // pooled block l: one 16-lane group per graph
    for (int u = t; u < g.B * 16; u += NT) {
      const int b = u >> 4, c4 = 4 * (u & 15);
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r1 = g.seg_ptr[b + 1];
      for (int r = g.seg_ptr[b]; r < r1; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(X + r * 64 + c4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(g.pooled + b * g.ldp + l * 64 + c4) = s;
    }
    if (l == g.L) break;
    build_a(A, X, g.vrowptr, g.vcol, n);
    __syncthreads();                                      // A complete; X free to be overwritten
    const float* W = g.wt + (int64_t)l * KA * 64 + col4;
    float4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < KA; k += 4) {
      const float4 w0 = *reinterpret_cast<const float4*>(W + (k + 0) * 64);
      const float4 w1 = *reinterpret_cast<const float4*>(W + (k + 1) * 64);
      const float4 w2 = *reinterpret_cast<const float4*>(W + (k + 2) * 64);
      const float4 w3 = *reinterpret_cast<const float4*>(W + (k + 3) * 64);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int r = rg + 64 * j;
        if (r < n) {
          const float4 a = *reinterpret_cast<const float4*>(A + r * KA + k);
          f4fma(acc[j], a.x, w0);
          f4fma(acc[j], a.y, w1);
          f4fma(acc[j], a.z, w2);
          f4fma(acc[j], a.w, w3);
        }
      }
    }
    const float4 b4 = *reinterpret_cast<const float4*>(g.bias + l * 64 + col4);
    float* xo = g.xall + (int64_t)l * n * 64;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int r = rg + 64 * j;
      if (r < n) {
        float4 v;
        v.x = fmaxf(acc[j].x + b4.x, 0.f);
        v.y = fmaxf(acc[j].y + b4.y, 0.f);
        v.z = fmaxf(acc[j].z + b4.z, 0.f);
        v.w = fmaxf(acc[j].w + b4.w, 0.f);
        *reinterpret_cast<float4*>(X + r * 64 + col4) = v;
        *reinterpret_cast<float4*>(xo + r * 64 + col4) = v;
      }
    }
    __syncthreads();
  }
}

struct BwdArgs {
  const float* x0;
  const float* xall;          // [L][n][64]  X_1 .. X_L
  const int32_t* vrowptr;
  const int32_t* vcol;
  const int32_t* t_rowptr;    // [n + 1] transposed index: the virtual rows (k * 3 + s) that gathered row i
  const int32_t* t_col;
  const int32_t* seg_id;      // [n]
  int n, L;
  const float* wtT;           // [L][64][192]
  const float* dpooled;       // [B][ldp]
  int64_t ldp;
  float* dwt;                 // [L][192][64]
  float* dbias;               // [L][64]
  float* dx0;                 // [n][64]
};

__global__ __launch_bounds__(NT) void shmp_small_bwd_kernel(const BwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* G = lds;                       // [n][64]   gradient of the layer's output rows, then dZ
  float* A = lds + NMAX * 64;           // [n][192]  A_l, then D_l
  const int t = threadIdx.x, n = g.n, L = g.L;
  for (int u = t; u < n * 16; u += NT) {
    const int i = u >> 4, c4 = 4 * (u & 15);
    *reinterpret_cast<float4*>(G + 4 * u) =
        *reinterpret_cast<const float4*>(g.dpooled + g.seg_id[i] * g.ldp + L * 64 + c4);
  }
  for (int l = L - 1; l >= 0; --l) {
    // dZ = G * relu'(X_{l+1})   (each thread rewrites the elements it read: no barrier needed before)
    const float* xo = g.xall + (int64_t)l * n * 64;
    for (int u = t; u < n * 16; u += NT) {
      const float4 x = *reinterpret_cast<const float4*>(xo + 4 * u);
      float4 v = *reinterpret_cast<const float4*>(G + 4 * u);
      v.x = x.x > 0.f ? v.x : 0.f;
      v.y = x.y > 0.f ? v.y : 0.f;
      v.z = x.z > 0.f ? v.z : 0.f;
      v.w = x.w > 0.f ? v.w : 0.f;
      *reinterpret_cast<float4*>(G + 4 * u) = v;
    }
    build_a(A, l ? g.xall + (int64_t)(l - 1) * n * 64 : g.x0, g.vrowptr, g.vcol, n);
    __syncthreads();
    // dWt_l[k][:] = sum_i A[i][k] dZ[i][:]: 16 lanes x float4 per weight row, rows k = kg, kg + 64, kg + 128
    {
      const int col4 = 4 * (t & 15), kg = t >> 4;
      float4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = 0; i < n; ++i) {
        const float4 dz = *reinterpret_cast<const float4*>(G + i * 64 + col4);
#pragma unroll
        for (int j = 0; j < 3; ++j) f4fma(acc[j], A[i * KA + kg + 64 * j], dz);
      }
      float* dw = g.dwt + (int64_t)l * KA * 64 + col4;
#pragma unroll
      for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(dw + (kg + 64 * j) * 64) = acc[j];
      if (t < 64) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += G[i * 64 + t];
        g.dbias[l * 64 + t] = s;
      }
    }
    __syncthreads();                                      // A consumed: D goes into its place
    // D[i][k] = sum_c dZ[i][c] Wt_l[k][c]: 48 lane groups x float4 of k per row, rows i = rg, rg + 21, ...
    if (t < 1008) {
      const int k4 = 4 * (t % 48), rg = t / 48;
      const float* wT = g.wtT + (int64_t)l * 64 * KA + k4;
      float4 acc[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c = 0; c < 64; ++c) {
        const float4 w = *reinterpret_cast<const float4*>(wT + c * KA);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int i = rg + 21 * j;
          if (i < n) f4fma(acc[j], G[i * 64 + c], w);
        }
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int i = rg + 21 * j;
        if (i < n) *reinterpret_cast<float4*>(A + i * KA + k4) = acc[j];
      }
    }
    __syncthreads();                                      // D complete, dZ consumed
    // gradient of X_l: self block + transposed gather of the slot blocks + this layer's pool seed
    for (int u = t; u < n * 16; u += NT) {
      const int i = u >> 4, c4 = 4 * (u & 15);
      float4 acc = *reinterpret_cast<const float4*>(A + i * KA + 128 + c4);
      const int e1 = g.t_rowptr[i + 1];
      for (int e = g.t_rowptr[i]; e < e1; ++e) {
        const float4 v = *reinterpret_cast<const float4*>(A + g.t_col[e] * 64 + c4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const float4 p = *reinterpret_cast<const float4*>(g.dpooled + g.seg_id[i] * g.ldp + l * 64 + c4);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
      *reinterpret_cast<float4*>(G + 4 * u) = acc;
      if (l == 0) *reinterpret_cast<float4*>(g.dx0 + 4 * u) = acc;      // (x0 comes out of a Linear: no mask)
    }
    __syncthreads();
  }
}

constexpr size_t kShmem = sizeof(float) * (size_t)NMAX * (64 + KA);

// ---- round 6: one workgroup per GRAPH -----------------------------------------------------------------------------------
// The query graphs are independent of each other (no edge crosses a graph) and have at most 5 rows: the one-workgroup
// kernels above spend their time waiting for the weight stream (48 dependent trips to L2 per layer for 135 rows' worth of
// FMAs on ONE of 256 CUs: 0.23 ms forward + 0.49 ms backward per training step, 29 % of it).  Here graph b's rows
// (at most GMAX) live in the LDS of workgroup b; per layer a thread fetches its whole share of the 192 x 64 weight block
// in four 16-byte loads issued together, and the next layer's share is requested before this layer's barrier.
//   forward   thread (k group kg of 48, 16 column lanes): 4 weight rows x float4, partial products of all rows of the
//             graph; the 4 k groups of a wave are added with two shuffles, the 12 waves through LDS in wave order.
//   backward  D = dZ Wt^T straight from Wt (thread (k, quarter of the 64 columns): 64 contiguous bytes; the 4 quarters of
//             a k sit in adjacent lanes: two shuffles) -- no transposed weight copy; dZ_l goes to a workspace from which
//             a second launch, one workgroup per (layer, 16 weight rows), forms dWt_l = A_l^T dZ_l over ALL rows in row
//             order (the one-workgroup kernel's order) with A_l gathered on the fly, and db_l.
constexpr int GT = 768;       // threads per graph workgroup: 12 waves
constexpr int GMAX = 8;       // rows per graph

struct GFwdArgs {
  const float* x0;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int L;
  const float* wt;
  const float* bias;
  const int32_t* seg_ptr;
  float* xall;
  int64_t n;
  float* pooled;
  int64_t ldp;
  DropArgs drop;              // key == nullptr: no dropout; layer l uses site drop.site + 2 l
};

__global__ __launch_bounds__(GT) void shmp_graphs_fwd_kernel(const GFwdArgs g) {
  __shared__ __attribute__((aligned(16))) float X[GMAX * 64];
  __shared__ __attribute__((aligned(16))) float A[GMAX * KA];
  __shared__ __attribute__((aligned(16))) float P[12 * GMAX * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
  const int r0 = g.seg_ptr[b];
  int nb = g.seg_ptr[b + 1] - r0;
  nb = nb > GMAX ? GMAX : nb;
  for (int u = t; u < nb * 16; u += GT)
    *reinterpret_cast<float4*>(X + 4 * u) = *reinterpret_cast<const float4*>(g.x0 + (int64_t)r0 * 64 + 4 * u);
  const int col4 = 4 * (lane & 15), kg = wave * 4 + (lane >> 4);          // k rows 4 kg .. 4 kg + 3
  float4 w0, w1, w2, w3;
#define SG_LOADW(l_)                                                                         \
  {                                                                                          \
    const float* W_ = g.wt + ((int64_t)(l_)*KA + 4 * kg) * 64 + col4;                        \
    w0 = *reinterpret_cast<const float4*>(W_);                                               \
    w1 = *reinterpret_cast<const float4*>(W_ + 64);                                          \
    w2 = *reinterpret_cast<const float4*>(W_ + 128);                                         \
    w3 = *reinterpret_cast<const float4*>(W_ + 192);                                         \
  }
  SG_LOADW(0)
  __syncthreads();
  for (int l = 0;; ++l) {
    if (t < 16) {                                                          // pooled block l of this graph
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < nb; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(X + r * 64 + 4 * t);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(g.pooled + (int64_t)b * g.ldp + l * 64 + 4 * t) = s;
    }
    if (l == g.L) break;
    for (int u = t; u < nb * 3 * 16; u += GT) {                            // A = [agg_0 | agg_1 | self] of the graph's rows
      const int c4 = 4 * (u & 15), v = u >> 4, i = v / 3, sl = v - 3 * i;
      float4 acc;
      if (sl == 2) {
        acc = *reinterpret_cast<const float4*>(X + i * 64 + c4);
      } else {
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int e1 = g.vrowptr[(r0 + i) * 2 + sl + 1];
        for (int e = g.vrowptr[(r0 + i) * 2 + sl]; e < e1; ++e) {
          int j = g.vcol[e] - r0;
          j = (j < 0 || j >= nb) ? i : j;                                  // (a source outside the graph cannot happen)
          const float4 x = *reinterpret_cast<const float4*>(X + j * 64 + c4);
          acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
      }
      *reinterpret_cast<float4*>(A + i * KA + sl * 64 + c4) = acc;
    }
    __syncthreads();
    for (int r = 0; r < nb; ++r) {
      const float4 a = *reinterpret_cast<const float4*>(A + r * KA + 4 * kg);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      f4fma(acc, a.x, w0);
      f4fma(acc, a.y, w1);
      f4fma(acc, a.z, w2);
      f4fma(acc, a.w, w3);
      acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64);
      acc.z += __shfl_xor(acc.z, 16, 64); acc.w += __shfl_xor(acc.w, 16, 64);
      acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
      acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
      if (lane < 16) *reinterpret_cast<float4*>(P + (wave * GMAX + r) * 64 + col4) = acc;
    }
    if (l + 1 < g.L) SG_LOADW(l + 1)                                       // in flight across the barrier and the epilogue
    __syncthreads();
    if (t < nb * 16) {
      const int r = t >> 4, c4 = 4 * (t & 15);
      float4 s = *reinterpret_cast<const float4*>(g.bias + l * 64 + c4);
#pragma unroll
      for (int w = 0; w < 12; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(P + (w * GMAX + r) * 64 + c4);
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
      }
      s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
      if (g.drop.key) {                                                    // F.dropout behind the relu (gnn_model.py:274)
        DropArgs d = g.drop;
        d.site += 2 * l;
        const uint64_t seed = d.key[0], step = d.key[1];
        s.x *= dropout_factor(d, seed, step, r0 + r, c4);
        s.y *= dropout_factor(d, seed, step, r0 + r, c4 + 1);
        s.z *= dropout_factor(d, seed, step, r0 + r, c4 + 2);
        s.w *= dropout_factor(d, seed, step, r0 + r, c4 + 3);
      }
      *reinterpret_cast<float4*>(X + r * 64 + c4) = s;
      *reinterpret_cast<float4*>(g.xall + ((int64_t)l * g.n + r0 + r) * 64 + c4) = s;
    }
    __syncthreads();
  }
#undef SG_LOADW
}

struct GBwdArgs {
  const float* xall;          // [L][n][64]
  const int32_t* t_rowptr;    // [n + 1]
  const int32_t* t_col;       // virtual rows k * 3 + s (global k)
  const int32_t* seg_ptr;
  int L;
  int64_t n;
  const float* wt;            // [L][192][64]
  const float* dpooled;
  int64_t ldp;
  float* dz_all;              // [L][n][64] workspace
  float* dx0;
  float mask_scale;           // 1, or the dropout factor 1 / (1 - p) of the kept elements
};

__global__ __launch_bounds__(GT) void shmp_graphs_bwd_kernel(const GBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float G[GMAX * 64];             // gradient of the layer's output rows, then dZ
  __shared__ __attribute__((aligned(16))) float D[GMAX * KA];
  const int t = threadIdx.x, b = blockIdx.x, L = g.L;
  const int r0 = g.seg_ptr[b];
  int nb = g.seg_ptr[b + 1] - r0;
  nb = nb > GMAX ? GMAX : nb;
  const int k = t >> 2, cq = t & 3;                                        // weight row k, columns 16 cq .. 16 cq + 15
  float4 w0, w1, w2, w3;
#define SG_LOADW(l_)                                                                         \
  {                                                                                          \
    const float* W_ = g.wt + ((int64_t)(l_)*KA + k) * 64 + 16 * cq;                          \
    w0 = *reinterpret_cast<const float4*>(W_);                                               \
    w1 = *reinterpret_cast<const float4*>(W_ + 4);                                           \
    w2 = *reinterpret_cast<const float4*>(W_ + 8);                                           \
    w3 = *reinterpret_cast<const float4*>(W_ + 12);                                          \
  }
  SG_LOADW(L - 1)
  const float* seed = g.dpooled + (int64_t)b * g.ldp;
  if (t < nb * 16) *reinterpret_cast<float4*>(G + 4 * t) = *reinterpret_cast<const float4*>(seed + L * 64 + 4 * (t & 15));
  for (int l = L - 1; l >= 0; --l) {
    if (t < nb * 16) {                                                     // dZ = G * relu'(X_{l+1}), kept and stored
      const int r = t >> 4, c4 = 4 * (t & 15);
      const float4 x = *reinterpret_cast<const float4*>(g.xall + ((int64_t)l * g.n + r0 + r) * 64 + c4);
      float4 v = *reinterpret_cast<const float4*>(G + 4 * t);
      v.x = x.x > 0.f ? v.x * g.mask_scale : 0.f;
      v.y = x.y > 0.f ? v.y * g.mask_scale : 0.f;
      v.z = x.z > 0.f ? v.z * g.mask_scale : 0.f;
      v.w = x.w > 0.f ? v.w * g.mask_scale : 0.f;
      *reinterpret_cast<float4*>(G + 4 * t) = v;
      *reinterpret_cast<float4*>(g.dz_all + ((int64_t)l * g.n + r0 + r) * 64 + c4) = v;
    }
    __syncthreads();
    for (int r = 0; r < nb; ++r) {                                         // D[r][k] = sum_c dZ[r][c] Wt[k][c]
      const float* z = G + r * 64 + 16 * cq;
      const float4 z0 = *reinterpret_cast<const float4*>(z), z1 = *reinterpret_cast<const float4*>(z + 4);
      const float4 z2 = *reinterpret_cast<const float4*>(z + 8), z3 = *reinterpret_cast<const float4*>(z + 12);
      float s = 0.f;
      s = fmaf(z0.x, w0.x, s); s = fmaf(z0.y, w0.y, s); s = fmaf(z0.z, w0.z, s); s = fmaf(z0.w, w0.w, s);
      s = fmaf(z1.x, w1.x, s); s = fmaf(z1.y, w1.y, s); s = fmaf(z1.z, w1.z, s); s = fmaf(z1.w, w1.w, s);
      s = fmaf(z2.x, w2.x, s); s = fmaf(z2.y, w2.y, s); s = fmaf(z2.z, w2.z, s); s = fmaf(z2.w, w2.w, s);
      s = fmaf(z3.x, w3.x, s); s = fmaf(z3.y, w3.y, s); s = fmaf(z3.z, w3.z, s); s = fmaf(z3.w, w3.w, s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (cq == 0) D[r * KA + k] = s;
    }
    if (l > 0) SG_LOADW(l - 1)
    __syncthreads();
    if (t < nb * 16) {                                                     // gradient of X_l
      const int r = t >> 4, c4 = 4 * (t & 15);
      float4 acc = *reinterpret_cast<const float4*>(D + r * KA + 128 + c4);
      const int e1 = g.t_rowptr[r0 + r + 1];
      for (int e = g.t_rowptr[r0 + r]; e < e1; ++e) {
        int v = g.t_col[e] - 3 * r0;                                       // local virtual row
        v = (v < 0 || v >= 3 * nb) ? 3 * r + 2 : v;
        const float4 d = *reinterpret_cast<const float4*>(D + v * 64 + c4);
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
      }
      const float4 p = *reinterpret_cast<const float4*>(seed + l * 64 + c4);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
      *reinterpret_cast<float4*>(G + 4 * t) = acc;
      if (l == 0) *reinterpret_cast<float4*>(g.dx0 + ((int64_t)r0 + r) * 64 + c4) = acc;
    }
    // (the next iteration's first phase touches only the elements the same thread just wrote; D is rewritten after its barrier)
  }
#undef SG_LOADW
}

// dWt[l][16 kb + kk][:] = sum_i A_l[i][16 kb + kk] dZ_l[i][:] over all rows i in row order; db[l] = colsum dZ_l
struct GBwdWArgs {
  const float* x0;
  const float* xall;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int64_t n;
  const float* dz_all;
  float* dwt;
  float* dbias;
};

__global__ __launch_bounds__(256) void shmp_graphs_bwd_w_kernel(const GBwdWArgs g) {
  constexpr int CH = 128;
  __shared__ __attribute__((aligned(16))) float Z[CH * 64];
  __shared__ float Ab[CH * 16];
  const int t = threadIdx.x, kb = blockIdx.x, l = blockIdx.y;
  const int kk = t >> 4, col4 = 4 * (t & 15);
  const int sl = kb >> 2, c0 = 16 * (kb & 3);                              // relation slot (2 = self) and column of the block
  const float* Xl = l ? g.xall + (int64_t)(l - 1) * g.n * 64 : g.x0;
  const float* dz = g.dz_all + (int64_t)l * g.n * 64;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float bs = 0.f;
  for (int64_t i0 = 0; i0 < g.n; i0 += CH) {
    const int m = (int)((g.n - i0) < CH ? (g.n - i0) : CH);
    __syncthreads();
    for (int u = t; u < m * 16; u += 256)
      *reinterpret_cast<float4*>(Z + 4 * u) = *reinterpret_cast<const float4*>(dz + i0 * 64 + 4 * u);
    for (int u = t; u < m * 16; u += 256) {
      const int i = u >> 4, c = c0 + (u & 15);
      const int64_t row = i0 + i;
      float a;
      if (sl == 2) {
        a = Xl[row * 64 + c];
      } else {
        a = 0.f;
        const int e1 = g.vrowptr[row * 2 + sl + 1];
        for (int e = g.vrowptr[row * 2 + sl]; e < e1; ++e) a += Xl[(int64_t)g.vcol[e] * 64 + c];
      }
      Ab[i * 16 + (u & 15)] = a;
    }
    __syncthreads();
    for (int i = 0; i < m; ++i) f4fma(acc, Ab[i * 16 + kk], *reinterpret_cast<const float4*>(Z + i * 64 + col4));
    if (kb == 0 && t < 64)
      for (int i = 0; i < m; ++i) bs += Z[i * 64 + t];
  }
  *reinterpret_cast<float4*>(g.dwt + ((int64_t)l * KA + 16 * kb + kk) * 64 + col4) = acc;
  if (kb == 0 && t < 64) g.dbias[l * 64 + t] = bs;
}

}  // namespace small
}  // namespace desco

extern "C" int desco_shmp_trunk_small_max_rows(void) { return desco::small::NMAX; }

extern "C" int desco_shmp_trunk_small_fwd_f32(const float* x0, const int32_t* vrowptr, const int32_t* vcol, int num_rows,
                                              int num_layers, const float* wt, const float* bias,
                                              const int32_t* seg_ptr, int num_seg, float* xall, float* pooled,
                                              int64_t ldp, desco_stream_t stream) {
  using namespace desco;
  using namespace desco::small;
  if (num_rows == 0 || num_seg == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x0 || !vrowptr || !vcol || !wt || !bias || !seg_ptr || !xall || !pooled || num_rows < 0 || num_rows > NMAX ||
      num_layers < 1 || num_seg < 0 || ldp < 64 * (num_layers + 1) || ldp % 4 || mis16(x0) || mis16(wt) ||
      mis16(bias) || mis16(xall) || mis16(pooled))
    return fail(DESCO_EINVAL, "desco_shmp_trunk_small_fwd_f32: bad argument (at most 144 rows, 16-byte alignment)");
  static DeviceOnce once;
  if (!once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_small_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShmem);
    once.mark();
  }
  FwdArgs g{x0, vrowptr, vcol, num_rows, num_layers, wt, bias, seg_ptr, num_seg, xall, pooled, ldp};
  hipLaunchKernelGGL(shmp_small_fwd_kernel, dim3(1), dim3(NT), kShmem, (hipStream_t)stream, g);
  return launch_status("desco_shmp_trunk_small_fwd_f32");
}

extern "C" int desco_shmp_trunk_small_bwd_f32(const float* x0, const float* xall, const int32_t* vrowptr,
                                              const int32_t* vcol, const int32_t* t_rowptr, const int32_t* t_col,
                                              const int32_t* seg_id, int num_rows, int num_layers, const float* wt_t,
                                              const float* dpooled, int64_t ldp, float* dwt, float* dbias, float* dx0,
                                              desco_stream_t stream) {
  using namespace desco;
  using namespace desco::small;
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x0 || !xall || !vrowptr || !vcol || !t_rowptr || !t_col || !seg_id || !wt_t || !dpooled || !dwt || !dbias ||
      !dx0 || num_rows < 0 || num_rows > NMAX || num_layers < 1 || ldp < 64 * (num_layers + 1) || ldp % 4 ||
      mis16(x0) || mis16(xall) || mis16(wt_t) || mis16(dpooled) || mis16(dwt) || mis16(dx0))
    return fail(DESCO_EINVAL, "desco_shmp_trunk_small_bwd_f32: bad argument (at most 144 rows, 16-byte alignment)");
  static DeviceOnce once;
  if (!once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_small_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShmem);
    once.mark();
  }
  BwdArgs g{x0, xall, vrowptr, vcol, t_rowptr, t_col, seg_id, num_rows, num_layers, wt_t, dpooled, ldp, dwt, dbias, dx0};
  hipLaunchKernelGGL(shmp_small_bwd_kernel, dim3(1), dim3(NT), kShmem, (hipStream_t)stream, g);
  return launch_status("desco_shmp_trunk_small_bwd_f32");
}

extern "C" int desco_shmp_trunk_graphs_max_rows(void) { return desco::small::GMAX; }

extern "C" int desco_shmp_trunk_graphs_fwd_f32(const float* x0, const int32_t* vrowptr, const int32_t* vcol,
                                               int64_t num_rows, int num_layers, const float* wt, const float* bias,
                                               const int32_t* seg_ptr, int num_seg, const desco_dropout* drop,
                                               float* xall, float* pooled, int64_t ldp, desco_stream_t stream) {
  using namespace desco;
  using namespace desco::small;
  if (num_rows == 0 || num_seg == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  DropArgs da = DropArgs{nullptr, 0u, 0u, 1.f};
  if (drop) {
    if (!drop->key || drop->site + 2u * (unsigned)num_layers >= 256u || num_rows > ((int64_t)1 << 34))
      return fail(DESCO_EINVAL, "desco_shmp_trunk_graphs_fwd_f32: dropout descriptor (key, site + 2 L < 256)");
    da = DropArgs{drop->key, drop->site, drop->threshold, drop->scale};
  }
  if (!x0 || !vrowptr || !vcol || !wt || !bias || !seg_ptr || !xall || !pooled || num_rows < 0 || num_layers < 1 ||
      num_seg < 0 || ldp < 64 * (num_layers + 1) || ldp % 4 || mis16(x0) || mis16(wt) || mis16(bias) || mis16(xall) ||
      mis16(pooled))
    return fail(DESCO_EINVAL, "desco_shmp_trunk_graphs_fwd_f32: bad argument (16-byte alignment, ldp >= 64 (L + 1))");
  GFwdArgs g{x0, vrowptr, vcol, num_layers, wt, bias, seg_ptr, xall, num_rows, pooled, ldp, da};
  hipLaunchKernelGGL(shmp_graphs_fwd_kernel, dim3((unsigned)num_seg), dim3(GT), 0, (hipStream_t)stream, g);
  return launch_status("desco_shmp_trunk_graphs_fwd_f32");
}

extern "C" int desco_shmp_trunk_graphs_bwd_f32(const float* x0, const float* xall, const int32_t* vrowptr,
                                               const int32_t* vcol, const int32_t* t_rowptr, const int32_t* t_col,
                                               const int32_t* seg_ptr, int num_seg, int64_t num_rows, int num_layers,
                                               const float* wt, const float* dpooled, int64_t ldp, float mask_scale,
                                               float* dwt, float* dbias, float* dx0, float* workspace,
                                               desco_stream_t stream) {
  using namespace desco;
  using namespace desco::small;
  if (num_rows == 0 || num_seg == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x0 || !xall || !vrowptr || !vcol || !t_rowptr || !t_col || !seg_ptr || !wt || !dpooled || !dwt || !dbias || !dx0 ||
      !workspace || num_rows < 0 || num_seg < 0 || num_layers < 1 || ldp < 64 * (num_layers + 1) || ldp % 4 ||
      mis16(x0) || mis16(xall) || mis16(wt) || mis16(dpooled) || mis16(dwt) || mis16(dx0) || mis16(workspace))
    return fail(DESCO_EINVAL, "desco_shmp_trunk_graphs_bwd_f32: bad argument (16-byte alignment, ldp >= 64 (L + 1))");
  GBwdArgs g{xall, t_rowptr, t_col, seg_ptr, num_layers, num_rows, wt, dpooled, ldp, workspace, dx0, mask_scale};
  hipLaunchKernelGGL(shmp_graphs_bwd_kernel, dim3((unsigned)num_seg), dim3(GT), 0, (hipStream_t)stream, g);
  int rc = launch_status("desco_shmp_trunk_graphs_bwd_f32");
  if (rc) return rc;
  GBwdWArgs w{x0, xall, vrowptr, vcol, num_rows, workspace, dwt, dbias};
  hipLaunchKernelGGL(shmp_graphs_bwd_w_kernel, dim3(12, (unsigned)num_layers), dim3(256), 0, (hipStream_t)stream, w);
  return launch_status("desco_shmp_trunk_graphs_bwd_f32 (weights)");
}
