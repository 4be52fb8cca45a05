// The glue of the training steps as kernels of this library (round 5): what rounds 2-4 left to torch -- the weight folding
// of the SHMP layers and its backward (three hipBLASLt batched GEMMs, stacks, cats, slices, sums), the strided copies /
// transposes between nn.Linear's [out, in] layout and the K-major operands of the GEMM kernels, the two losses and their
// gradients -- so that a training step launches nothing but desco:: kernels (tools/check_pass_is_native.sh --train).
// Reference: lightning_model.py:228-254, 285-289 (neighborhood step, smooth_l1 on log2(y + 1)), :585-608, 630-635 (gossip
// step, sum log2(|d| + 1)); gnn_model.py:253-277 (the SAGE layer whose Linear pair is folded here).
#include "common_device.hpp"

namespace desco {
namespace tn {

// ---- strided 2-D copies / transposes, many per launch ---------------------------------------------------------------
constexpr int kMaxCopies = 24;
struct CopyArgs {
  desco_copy2d_desc d[kMaxCopies];
  int first_tile[kMaxCopies + 1];        // prefix sums of 32 x 32 tiles
  int num;
};

__global__ __launch_bounds__(256) void copy2d_multi_kernel(const CopyArgs a) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int p = 0;
  while (p + 1 < a.num && b >= a.first_tile[p + 1]) ++p;
  const desco_copy2d_desc& d = a.d[p];
  const int tcols = (d.cols + 31) >> 5;
  const int t = b - a.first_tile[p];
  const int r0 = (t / tcols) << 5, c0 = (t % tcols) << 5;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (!d.transpose) {
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      if (r < d.rows && c < d.cols) {
        const float v = d.src[(int64_t)r * d.lds + c];
        float* o = d.dst + (int64_t)r * d.ldd + c;
        *o = d.accumulate ? *o + v : v;
      }
    }
    return;
  }
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < d.rows && c < d.cols) ? d.src[(int64_t)r * d.lds + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {           // dst[c][r] = src[r][c]
    const int c = c0 + i, r = r0 + tx;
    if (r < d.rows && c < d.cols) {
      float* o = d.dst + (int64_t)c * d.ldd + r;
      *o = d.accumulate ? *o + tile[tx][i] : tile[tx][i];
    }
  }
}

// ---- SHMP weight folding ------------------------------------------------------------------------------------------------
// Per layer l and row type: update Linear U [64, 128] = [U_n | U_x], bias c [64]; per relation slot s a SAGEConv Linear
// W_s [64, 64], bias b_u per DISTINCT relation (use_tconv = False ties two slots to one relation).  The fused layer
//   X' = relu([agg_0 | .. | agg_{S-1} | X] Wt + fb)    needs    Wt[s 64 + k][n] = sum_j U_n[n][j] W_s[j][k]   (s < S),
//   Wt[S 64 + k][n] = U_x[n][k],    fb[n] = sum_j U_n[n][j] (sum_u b_u[j]) + c[n].
// The parameters are read where torch keeps them, through a table of device addresses: per layer
//   [U, c, W_0 .. W_{S-1}, b_0 .. b_{NU-1}]    (int64 each; two slots of one relation carry the same W address),
// and the backward writes every parameter's gradient at its offset (a table of the same shape, in floats) of ONE flat
// buffer the caller hands out as views -- no stack / cat / slice / accumulate kernel of anybody else's.
__device__ __forceinline__ const float* tab(const int64_t* t, int i) { return reinterpret_cast<const float*>(t[i]); }

__global__ __launch_bounds__(256) void fold_shmp_fwd_kernel(const int64_t* __restrict__ table, int S, int NU,
                                                            float* __restrict__ Wt, float* __restrict__ fb) {
  __shared__ float A[64][65];      // U_n[n][j]
  __shared__ float Bm[64][65];     // W_s[j][k]   (s == S: U_x[n][k])
  const int s = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const int64_t* t = table + (int64_t)l * (2 + S + NU);
  const float* U = tab(t, 0);
  for (int i = tid; i < 64 * 64; i += 256) A[i >> 6][i & 63] = U[(i >> 6) * 128 + (i & 63)];
  float* out = Wt + ((int64_t)l * (S + 1) + s) * 64 * 64;
  if (s < S) {
    const float* W = tab(t, 2 + s);
    for (int i = tid; i < 64 * 64; i += 256) Bm[i >> 6][i & 63] = W[i];
    __syncthreads();
    const int n = tid & 63, kq = tid >> 6;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int j = 0; j < 64; ++j) {
      const float a = A[n][j];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(a, Bm[j][kq * 16 + k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) out[(kq * 16 + k) * 64 + n] = acc[k];
    return;
  }
  for (int i = tid; i < 64 * 64; i += 256) Bm[i >> 6][i & 63] = U[(i >> 6) * 128 + 64 + (i & 63)];
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) out[i] = Bm[i & 63][i >> 6];           // out[k][n] = U_x[n][k]
  if (tid < 64) {
    float acc = 0.f;
    for (int j = 0; j < 64; ++j) {
      float bs = 0.f;
      for (int u = 0; u < NU; ++u) bs += tab(t, 2 + S + u)[j];
      acc = __builtin_fmaf(A[tid][j], bs, acc);
    }
    fb[l * 64 + tid] = acc + tab(t, 1)[tid];
  }
}

__global__ __launch_bounds__(256) void fold_shmp_bwd_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ goff,
                                                            int S, int NU, const float* __restrict__ dWt,
                                                            const float* __restrict__ dfb, float* __restrict__ grads) {
  __shared__ float A[64][65];
  __shared__ float Bm[64][65];
  const int s = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const int64_t* t = table + (int64_t)l * (2 + S + NU);
  const int64_t* go = goff + (int64_t)l * (2 + S + NU);
  const float* U = tab(t, 0);
  const float* dW = dWt + (int64_t)l * (S + 1) * 64 * 64;
  if (s < S) {
    // dW_s[j][k] = sum_n U_n[n][j] dWt[s 64 + k][n], summed over the slots that share this W (first of them writes)
    for (int s1 = 0; s1 < s; ++s1)
      if (t[2 + s1] == t[2 + s]) return;
    for (int i = tid; i < 64 * 64; i += 256) A[i >> 6][i & 63] = U[(i >> 6) * 128 + (i & 63)];
    const int j = tid & 63, kq = tid >> 6;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int s2 = s; s2 < S; ++s2) {
      if (t[2 + s2] != t[2 + s]) continue;
      __syncthreads();
      for (int i = tid; i < 64 * 64; i += 256) Bm[i >> 6][i & 63] = dW[(int64_t)s2 * 4096 + i];      // [k][n]
      __syncthreads();
      for (int n = 0; n < 64; ++n) {
        const float a = A[n][j];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(a, Bm[kq * 16 + k][n], acc[k]);
      }
    }
    float* o = grads + go[2 + s];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[j * 64 + kq * 16 + k] = acc[k];
    return;
  }
  // block S of the layer: dU = [dU_n | dU_x], dc, db_u
  float* dU = grads + go[0];
  const int n = tid & 63, jq = tid >> 6;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int s2 = 0; s2 < S; ++s2) {
    const float* W = tab(t, 2 + s2);
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      A[i >> 6][i & 63] = dW[(int64_t)s2 * 4096 + i];        // [k][n]
      Bm[i >> 6][i & 63] = W[i];                             // [j][k]
    }
    __syncthreads();
    for (int k = 0; k < 64; ++k) {
      const float d = A[k][n];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_fmaf(d, Bm[jq * 16 + j][k], acc[j]);
    }
  }
  const float df = dfb[l * 64 + n];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float bs = 0.f;
    for (int u = 0; u < NU; ++u) bs += tab(t, 2 + S + u)[jq * 16 + j];
    dU[n * 128 + jq * 16 + j] = __builtin_fmaf(df, bs, acc[j]);
  }
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) A[i >> 6][i & 63] = dW[(int64_t)S * 4096 + i];              // [k][n]
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) dU[(i >> 6) * 128 + 64 + (i & 63)] = A[i & 63][i >> 6];     // dU_x[n][k]
  if (tid < 64) {
    grads[go[1] + tid] = dfb[l * 64 + tid];
    float db = 0.f;                                            // db_u[j] = sum_n U_n[n][j] dfb[n], the same for every u
    for (int m = 0; m < 64; ++m) db = __builtin_fmaf(U[m * 128 + tid], dfb[l * 64 + m], db);
    for (int u = 0; u < NU; ++u) grads[go[2 + S + u] + tid] = db;
  }
}


// ---- gossip weight folding ---------------------------------------------------------------------------------------------
// The operands of the gossip training trunk (autograd.GossipTrunk; algebra DESIGN.md 4.2, reference gnn_model.py:58-103,
// 230-260, 303-350) as functions of the raw parameters, for Q queries with embeddings E [Q, 64] (no gradient: the layer-0
// input is detached, gnn_model.py:236-240), w = pre_mp.weight[:, 0], b = pre_mp.bias (no gradient either):
//   k0 = C0b b + cb0;  a = E C0a^T + k0;  v = C0b w;  p = a D0a^T;  r = D0a v;  t = D0c w;  z = E D0b^T + D0c b + db0
//   gate_i = sigmoid(sigmoid(E G_i^T + gb_i) . g2_i + gb2_i)        (the trailing LeakyReLU is the identity on (0, 1))
//   V0[q] = [p_q, g0_q p_q, r, g0_q r, t, z_q];   wt1 = [(D1a C1)^T ; D1b^T];   u = D1a cb1;  V1[q] = [u, g1_q u, db1]
//   wtp = [P0c^T ; P0d^T];  tp = P0b w;  zp = E P0a^T + P0b b + p0;  Vp[q] = [tp, zp_q];   w3t = P3^T;  w5t = P5^T
// with C0 = [C0a | C0b] [64, 128], D0 = [D0a | D0b | D0c] [64, 192], D1 = [D1a | D1b], P0 = [P0a | P0b | P0c | P0d] [64, 256].
// One workgroup each way (the largest product is 64 x 64 x 64): exact fp32 FMA chains in a fixed order.
struct GossipFoldArgs {
  const float *E, *w, *b;                                  // [Q,64], [64], [64]
  const float *C0, *cb0, *D0, *db0, *C1, *cb1, *D1, *db1;  // conv 0 / 1: lin_com, lin_update
  const float *G0[2], *gb0[2], *g2[2], *gb2[2];            // lin_gate.0 / .2 of conv 0 / 1
  const float *P0, *p0, *P3, *P5;                          // post_mp.0 / .3 / .5 weights
  int Q;
};
struct GossipFoldOut {
  float *V0, *V1, *Vp, *wt1, *wtp, *w3t, *w5t;             // [Q,6,64] [Q,3,64] [Q,2,64] [128,64] [128,64] [64,64] [64,256]
  float *g0, *g1, *g1c;                                    // [Q] gates of conv 0 / 1, 1 - g1
  float *a, *h0, *h1;                                      // saved for the backward: a [Q,64], gate hidden layers [Q,64]
};

__device__ __forceinline__ float sigmoidf_(const float x) { return 1.f / (1.f + expf(-x)); }

// (no packed fp32 selection in the two fold kernels: hipcc broadcast the odd dword of a scalar-loaded pair through OP_SEL
//  on src1 here -- rule PK-OPSEL of tools/check_isa.py -- and nothing in them is worth a packed instruction)
__global__ __launch_bounds__(1024) DESCO_NO_PACKED_F32 void gossip_fold_fwd_kernel(const GossipFoldArgs g, const GossipFoldOut o) {
  __shared__ float vv[64], k0[64], rr[64], tt[64], uu[64], tpv[64], zc[64], zpc[64];
  const int tid = threadIdx.x, Q = g.Q;
  if (tid < 64) {
    const int n = tid;
    float sk = g.cb0[n], sv = 0.f, st = 0.f, szc = g.db0[n], stp = 0.f, szp = g.p0[n], su = 0.f;
    for (int j = 0; j < 64; ++j) {
      const float c0b = g.C0[n * 128 + 64 + j], d0c = g.D0[n * 192 + 128 + j], p0b = g.P0[n * 256 + 64 + j];
      sk = __builtin_fmaf(c0b, g.b[j], sk);
      sv = __builtin_fmaf(c0b, g.w[j], sv);
      st = __builtin_fmaf(d0c, g.w[j], st);
      szc = __builtin_fmaf(d0c, g.b[j], szc);
      stp = __builtin_fmaf(p0b, g.w[j], stp);
      szp = __builtin_fmaf(p0b, g.b[j], szp);
      su = __builtin_fmaf(g.D1[n * 128 + j], g.cb1[j], su);
    }
    k0[n] = sk; vv[n] = sv; tt[n] = st; zc[n] = szc; tpv[n] = stp; zpc[n] = szp; uu[n] = su;
  }
  __syncthreads();
  if (tid < 64) {
    float s = 0.f;
    for (int j = 0; j < 64; ++j) s = __builtin_fmaf(g.D0[tid * 192 + j], vv[j], s);
    rr[tid] = s;
  }
  // a, gate hidden layers
  for (int i = tid; i < Q * 64; i += 1024) {
    const int q = i >> 6, n = i & 63;
    float sa = k0[n], s0 = g.gb0[0][n], s1 = g.gb0[1][n];
    for (int j = 0; j < 64; ++j) {
      const float e = g.E[q * 64 + j];
      sa = __builtin_fmaf(e, g.C0[n * 128 + j], sa);
      s0 = __builtin_fmaf(e, g.G0[0][n * 64 + j], s0);
      s1 = __builtin_fmaf(e, g.G0[1][n * 64 + j], s1);
    }
    o.a[i] = sa;
    o.h0[i] = sigmoidf_(s0);
    o.h1[i] = sigmoidf_(s1);
  }
  __syncthreads();          // (o.a / o.h* are re-read below by other threads: global memory, workgroup scope)
  __threadfence_block();
  if (tid < 2 * Q) {
    const int q = tid >> 1, which = tid & 1;
    const float* h = which ? o.h1 : o.h0;
    float s = g.gb2[which][0];
    for (int j = 0; j < 64; ++j) s = __builtin_fmaf(h[q * 64 + j], g.g2[which][j], s);
    s = sigmoidf_(s);
    if (which) { o.g1[q] = s; o.g1c[q] = 1.f - s; } else o.g0[q] = s;
  }
  __syncthreads();
  __threadfence_block();
  // V0, V1, Vp
  for (int i = tid; i < Q * 64; i += 1024) {
    const int q = i >> 6, n = i & 63;
    float sp = 0.f, sz = zc[n], szp = zpc[n];
    for (int j = 0; j < 64; ++j) {
      sp = __builtin_fmaf(o.a[q * 64 + j], g.D0[n * 192 + j], sp);
      const float e = g.E[q * 64 + j];
      sz = __builtin_fmaf(e, g.D0[n * 192 + 64 + j], sz);
      szp = __builtin_fmaf(e, g.P0[n * 256 + j], szp);
    }
    const float g0 = o.g0[q], g1 = o.g1[q];
    float* v0 = o.V0 + (int64_t)q * 6 * 64 + n;
    v0[0] = sp; v0[64] = g0 * sp; v0[128] = rr[n]; v0[192] = g0 * rr[n]; v0[256] = tt[n]; v0[320] = sz;
    float* v1 = o.V1 + (int64_t)q * 3 * 64 + n;
    v1[0] = uu[n]; v1[64] = g1 * uu[n]; v1[128] = g.db1[n];
    float* vp = o.Vp + (int64_t)q * 2 * 64 + n;
    vp[0] = tpv[n]; vp[64] = szp;
  }
  // wt1 = [(D1a C1)^T ; D1b^T], wtp = [P0c^T ; P0d^T], w3t, w5t
  for (int i = tid; i < 64 * 64; i += 1024) {
    const int k = i >> 6, n = i & 63;
    float s = 0.f;
    for (int j = 0; j < 64; ++j) s = __builtin_fmaf(g.D1[n * 128 + j], g.C1[j * 64 + k], s);
    o.wt1[k * 64 + n] = s;
    o.wt1[(64 + k) * 64 + n] = g.D1[n * 128 + 64 + k];
    o.wtp[k * 64 + n] = g.P0[n * 256 + 128 + k];
    o.wtp[(64 + k) * 64 + n] = g.P0[n * 256 + 192 + k];
    o.w3t[k * 64 + n] = g.P3[n * 64 + k];
  }
  for (int i = tid; i < 64 * 256; i += 1024) {
    const int k = i >> 8, n = i & 255;            // w5t [64][256]: w5t[k][n] = P5[n][k]
    o.w5t[k * 256 + n] = g.P5[n * 64 + k];
  }
}

struct GossipFoldGrads {
  const float *dV0, *dV1, *dVp, *dwt1, *dwtp, *dw3t, *dw5t, *dg1;   // incoming (dg1 may be null)
  const float *V0, *V1, *g0, *g1, *a, *h0, *h1;                     // saved by the forward
  float *dC0, *dcb0, *dD0, *ddb0, *dC1, *dcb1, *dD1, *ddb1;         // outgoing, parameter layout
  float *dG0[2], *dgb0[2], *dg2[2], *dgb2[2];
  float *dP0, *dp0, *dP3, *dP5;
  float *scratch;                                                   // Q*64*3 + 64*8 + 2*Q floats
};

__global__ __launch_bounds__(1024) DESCO_NO_PACKED_F32 void gossip_fold_bwd_kernel(const GossipFoldArgs g, const GossipFoldGrads d) {
  __shared__ float dr[64], dt[64], du[64], dtp[64], sdz[64], sdzp[64], sda[64], dv[64], vv[64];
  __shared__ float dpre2[2][64];
  const int tid = threadIdx.x, Q = g.Q;
  float* dp = d.scratch;                  // [Q,64]
  float* da = d.scratch + Q * 64;         // [Q,64]
  float* dpre1 = d.scratch + 2 * Q * 64;  // [2][Q,64]
  // ---- per-query pieces of the V gradients ---------------------------------------------------------------------------
  for (int i = tid; i < Q * 64; i += 1024) {
    const int q = i >> 6, n = i & 63;
    const float* v0 = d.dV0 + (int64_t)q * 6 * 64 + n;
    dp[i] = v0[0] + d.g0[q] * v0[64];
  }
  if (tid < 64) {
    const int n = tid;
    float sr = 0.f, st = 0.f, su = 0.f, sb1 = 0.f, stp = 0.f, sz = 0.f, szp = 0.f;
    for (int q = 0; q < Q; ++q) {
      const float* v0 = d.dV0 + (int64_t)q * 6 * 64 + n;
      const float* v1 = d.dV1 + (int64_t)q * 3 * 64 + n;
      const float* vp = d.dVp + (int64_t)q * 2 * 64 + n;
      sr += v0[128] + d.g0[q] * v0[192];
      st += v0[256];
      sz += v0[320];
      su += v1[0] + d.g1[q] * v1[64];
      sb1 += v1[128];
      stp += vp[0];
      szp += vp[64];
    }
    dr[n] = sr; dt[n] = st; sdz[n] = sz; du[n] = su; dtp[n] = stp; sdzp[n] = szp;
    d.ddb1[n] = sb1;
    d.ddb0[n] = sz;
    d.dp0[n] = szp;
    float sv = 0.f;
    for (int j = 0; j < 64; ++j) sv = __builtin_fmaf(g.C0[n * 128 + 64 + j], g.w[j], sv);
    vv[n] = sv;
  }
  // gate gradients, first stage: dg_q -> dpre2_q = dg_q s_q (1 - s_q)
  if (tid >= 64 && tid < 64 + 2 * Q) {
    const int q = (tid - 64) >> 1, which = (tid - 64) & 1;
    float dg = 0.f;
    if (which == 0) {
      const float r_dot = 0.f;
      (void)r_dot;
      for (int n = 0; n < 64; ++n) {
        const float* v0 = d.dV0 + (int64_t)q * 6 * 64 + n;
        dg = __builtin_fmaf(v0[64], d.V0[(int64_t)q * 6 * 64 + n], dg);               // . p_q
        dg = __builtin_fmaf(v0[192], d.V0[(int64_t)q * 6 * 64 + 128 + n], dg);       // . r
      }
    } else {
      dg = d.dg1 ? d.dg1[q] : 0.f;
      for (int n = 0; n < 64; ++n)
        dg = __builtin_fmaf(d.dV1[(int64_t)q * 3 * 64 + 64 + n], d.V1[(int64_t)q * 3 * 64 + n], dg);   // . u
    }
    const float s = which ? d.g1[q] : d.g0[q];
    dpre2[which][q] = dg * s * (1.f - s);
  }
  __syncthreads();
  __threadfence_block();
  // ---- da = dp D0a;  dpre1 = dpre2 g2 h (1 - h) ---------------------------------------------------------------------------
  for (int i = tid; i < Q * 64; i += 1024) {
    const int q = i >> 6, j = i & 63;
    float s = 0.f;
    for (int n = 0; n < 64; ++n) s = __builtin_fmaf(dp[q * 64 + n], g.D0[n * 192 + j], s);
    da[i] = s;
    const float h0 = d.h0[i], h1 = d.h1[i];
    dpre1[i] = dpre2[0][q] * g.g2[0][j] * h0 * (1.f - h0);
    dpre1[Q * 64 + i] = dpre2[1][q] * g.g2[1][j] * h1 * (1.f - h1);
  }
  if (tid < 64) {      // dv = D0a^T dr
    float s = 0.f;
    for (int n = 0; n < 64; ++n) s = __builtin_fmaf(g.D0[n * 192 + tid], dr[n], s);
    dv[tid] = s;
  }
  if (tid >= 64 && tid < 192) {      // dg2[j] = sum_q dpre2_q h[q, j];  dgb2 = sum_q dpre2_q
    const int which = (tid - 64) >> 6, j = (tid - 64) & 63;
    const float* h = which ? d.h1 : d.h0;
    float s = 0.f, sb = 0.f;
    for (int q = 0; q < Q; ++q) { s = __builtin_fmaf(dpre2[which][q], h[q * 64 + j], s); sb += dpre2[which][q]; }
    d.dg2[which][j] = s;
    if (j == 0) d.dgb2[which][0] = sb;
  }
  __syncthreads();
  __threadfence_block();
  if (tid < 64) {
    float s = 0.f;
    for (int q = 0; q < Q; ++q) s += da[q * 64 + tid];
    sda[tid] = s;
    d.dcb0[tid] = s;
    float s0 = 0.f, s1 = 0.f;
    for (int q = 0; q < Q; ++q) { s0 += dpre1[q * 64 + tid]; s1 += dpre1[Q * 64 + q * 64 + tid]; }
    d.dgb0[0][tid] = s0;
    d.dgb0[1][tid] = s1;
    float sc = 0.f;                      // dcb1 = D1a^T du
    for (int n = 0; n < 64; ++n) sc = __builtin_fmaf(g.D1[n * 128 + tid], du[n], sc);
    d.dcb1[tid] = sc;
  }
  __syncthreads();
  // ---- the [64, 64] blocks --------------------------------------------------------------------------------------------------
  for (int i = tid; i < 64 * 64; i += 1024) {
    const int n = i >> 6, j = i & 63;
    float c0a = 0.f, d0a = 0.f, d0b = 0.f, p0a = 0.f, ga = 0.f, gb = 0.f;
    for (int q = 0; q < Q; ++q) {
      const float e = g.E[q * 64 + j];
      c0a = __builtin_fmaf(da[q * 64 + n], e, c0a);
      d0a = __builtin_fmaf(dp[q * 64 + n], d.a[q * 64 + j], d0a);
      d0b = __builtin_fmaf(d.dV0[(int64_t)q * 6 * 64 + 320 + n], e, d0b);
      p0a = __builtin_fmaf(d.dVp[(int64_t)q * 2 * 64 + 64 + n], e, p0a);
      ga = __builtin_fmaf(dpre1[q * 64 + n], e, ga);
      gb = __builtin_fmaf(dpre1[Q * 64 + q * 64 + n], e, gb);
    }
    d.dC0[n * 128 + j] = c0a;
    d.dC0[n * 128 + 64 + j] = __builtin_fmaf(sda[n], g.b[j], dv[n] * g.w[j]);
    d.dD0[n * 192 + j] = __builtin_fmaf(dr[n], vv[j], d0a);
    d.dD0[n * 192 + 64 + j] = d0b;
    d.dD0[n * 192 + 128 + j] = __builtin_fmaf(dt[n], g.w[j], sdz[n] * g.b[j]);
    d.dP0[n * 256 + j] = p0a;
    d.dP0[n * 256 + 64 + j] = __builtin_fmaf(dtp[n], g.w[j], sdzp[n] * g.b[j]);
    d.dP0[n * 256 + 128 + j] = d.dwtp[j * 64 + n];
    d.dP0[n * 256 + 192 + j] = d.dwtp[(64 + j) * 64 + n];
    d.dG0[0][n * 64 + j] = ga;
    d.dG0[1][n * 64 + j] = gb;
    // wt1[k][n'] = sum_j' D1a[n'][j'] C1[j'][k]:  dD1a[n][j] = sum_k dwt1[k][n] C1[j][k] + du[n] cb1[j];  dC1[n][j] as [j'][k]
    float d1a = du[n] * g.cb1[j], c1 = 0.f;
    for (int k = 0; k < 64; ++k) {
      d1a = __builtin_fmaf(d.dwt1[k * 64 + n], g.C1[j * 64 + k], d1a);
      c1 = __builtin_fmaf(g.D1[k * 128 + n], d.dwt1[j * 64 + k], c1);          // dC1[n][j] = sum_k' D1a[k'][n] dwt1[j][k']
    }
    d.dD1[n * 128 + j] = d1a;
    d.dD1[n * 128 + 64 + j] = d.dwt1[(64 + j) * 64 + n];
    d.dC1[n * 64 + j] = c1;
    d.dP3[n * 64 + j] = d.dw3t[j * 64 + n];
  }
  for (int i = tid; i < 256 * 64; i += 1024) {
    const int n = i >> 6, k = i & 63;            // dP5[n][k] = dw5t[k][n]
    d.dP5[n * 64 + k] = d.dw5t[k * 256 + n];
  }
}

// ---- losses ---------------------------------------------------------------------------------------------------------
// mode 0: mean over count of smooth_l1(pred - log2(y + 1)) (beta 1: 0.5 d^2 below 1, |d| - 0.5 above);
// mode 1: sum of log2(|pred - y| + 1).   dpred = d loss / d pred;  partial sums per block, folded in block order.
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                           int64_t count, int mode, float scale, float* __restrict__ dpred,
                                                           float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    if (mode == 0) {
      const float d = pred[i] - log2f(y[i] + 1.f);
      const float ad = fabsf(d);
      s += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
      dpred[i] = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * scale;
    } else {
      const float d = pred[i] - y[i];
      const float ad = fabsf(d);
      s += log2f(ad + 1.f);
      dpred[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (1.4426950408889634f / (ad + 1.f)) * scale;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(64) void loss_final_kernel(const float* __restrict__ partial, int n, float scale,
                                                        float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    loss[0] = s * scale;
  }
}

// out[i] = a[i] * mul[0] + add[0] + addv[i]  (each term optional): a saved gradient times the upstream gradient of its
// scalar loss; x + correction + post_mp.7's bias
__global__ __launch_bounds__(256) void affine_scalar_kernel(const float* __restrict__ a, const float* __restrict__ mul,
                                                            const float* __restrict__ add, const float* __restrict__ addv,
                                                            float* __restrict__ out, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) out[i] = a[i] * (mul ? mul[0] : 1.f) + (add ? add[0] : 0.f) + (addv ? addv[i] : 0.f);
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, float v, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) p[i] = v;
}

}  // namespace tn
}  // namespace desco

using namespace desco;

extern "C" int desco_copy2d_multi_f32(int num, const desco_copy2d_desc* descs, desco_stream_t stream) {
  if (num < 0 || (num > 0 && !descs)) return fail(DESCO_EINVAL, "desco_copy2d_multi_f32: bad argument");
  for (int i0 = 0; i0 < num; i0 += tn::kMaxCopies) {
    tn::CopyArgs a;
    a.num = 0;
    int tiles = 0;
    for (int i = i0; i < num && i < i0 + tn::kMaxCopies; ++i) {
      const desco_copy2d_desc& d = descs[i];
      if (d.rows < 0 || d.cols < 0 || ((d.rows > 0 && d.cols > 0) && (!d.src || !d.dst)))
        return fail(DESCO_EINVAL, "desco_copy2d_multi_f32: bad descriptor");
      if (d.rows == 0 || d.cols == 0) continue;
      a.d[a.num] = d;
      a.first_tile[a.num] = tiles;
      tiles += ((d.rows + 31) / 32) * ((d.cols + 31) / 32);
      ++a.num;
    }
    if (!a.num) continue;
    a.first_tile[a.num] = tiles;
    hipLaunchKernelGGL(tn::copy2d_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
  }
  return launch_status("desco_copy2d_multi_f32");
}

extern "C" int desco_fold_shmp_fwd_f32(const int64_t* table, int num_layers, int slots, int num_bias, float* wt,
                                       float* fb, desco_stream_t stream) {
  if (!table || !wt || !fb || num_layers < 1 || slots < 1 || slots > 4 || num_bias < 1 || num_bias > 4)
    return fail(DESCO_EINVAL, "desco_fold_shmp_fwd_f32: bad argument");
  hipLaunchKernelGGL(tn::fold_shmp_fwd_kernel, dim3(slots + 1, num_layers), dim3(256), 0, (hipStream_t)stream, table, slots,
                     num_bias, wt, fb);
  return launch_status("desco_fold_shmp_fwd_f32");
}

extern "C" int desco_fold_shmp_bwd_f32(const int64_t* table, const int64_t* grad_offsets, int num_layers, int slots,
                                       int num_bias, const float* dwt, const float* dfb, float* grads,
                                       desco_stream_t stream) {
  if (!table || !grad_offsets || !dwt || !dfb || !grads || num_layers < 1 || slots < 1 || slots > 4 || num_bias < 1 ||
      num_bias > 4)
    return fail(DESCO_EINVAL, "desco_fold_shmp_bwd_f32: bad argument");
  hipLaunchKernelGGL(tn::fold_shmp_bwd_kernel, dim3(slots + 1, num_layers), dim3(256), 0, (hipStream_t)stream, table,
                     grad_offsets, slots, num_bias, dwt, dfb, grads);
  return launch_status("desco_fold_shmp_bwd_f32");
}

extern "C" int desco_loss_f32(const float* pred, const float* y, int64_t count, int mode, float* loss, float* dpred,
                              float* workspace, desco_stream_t stream) {
  if (!pred || !y || !loss || !dpred || !workspace || count < 1 || (mode != 0 && mode != 1))
    return fail(DESCO_EINVAL, "desco_loss_f32: bad argument");
  int64_t blocks = (count + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  const float scale = mode == 0 ? (float)(1.0 / (double)count) : 1.f;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tn::loss_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pred, y, count, mode, scale, dpred,
                     workspace);
  hipLaunchKernelGGL(tn::loss_final_kernel, dim3(1), dim3(64), 0, st, workspace, (int)blocks, scale, loss);
  return launch_status("desco_loss_f32");
}

extern "C" int desco_affine_scalar_f32(const float* a, const float* mul, const float* add, const float* addv, float* out,
                                       int64_t count, desco_stream_t stream) {
  if (count == 0) return 0;
  if (!a || !out || count < 0) return fail(DESCO_EINVAL, "desco_affine_scalar_f32: bad argument");
  hipLaunchKernelGGL(tn::affine_scalar_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, mul, add, addv, out, count);
  return launch_status("desco_affine_scalar_f32");
}

extern "C" int desco_fill_f32(float* p, float value, int64_t count, desco_stream_t stream) {
  if (count == 0) return 0;
  if (!p || count < 0) return fail(DESCO_EINVAL, "desco_fill_f32: bad argument");
  hipLaunchKernelGGL(tn::fill_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, value,
                     count);
  return launch_status("desco_fill_f32");
}

extern "C" int desco_gossip_fold_fwd_f32(const desco_gossip_fold_params* p, const desco_gossip_fold_out* o,
                                         desco_stream_t stream) {
  if (!p || !o || p->num_q < 1 || p->num_q > 64) return fail(DESCO_EINVAL, "desco_gossip_fold_fwd_f32: bad argument (1 <= num_q <= 64)");
  tn::GossipFoldArgs a{p->E, p->w_pre, p->b_pre, p->C0, p->cb0, p->D0, p->db0, p->C1, p->cb1, p->D1, p->db1,
                       {p->G0[0], p->G0[1]}, {p->gb0[0], p->gb0[1]}, {p->g2[0], p->g2[1]}, {p->gb2[0], p->gb2[1]},
                       p->P0, p->p0, p->P3, p->P5, p->num_q};
  tn::GossipFoldOut q{o->V0, o->V1, o->Vp, o->wt1, o->wtp, o->w3t, o->w5t, o->g0, o->g1, o->g1c, o->a, o->h0, o->h1};
  hipLaunchKernelGGL(tn::gossip_fold_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, q);
  return launch_status("desco_gossip_fold_fwd_f32");
}

extern "C" int desco_gossip_fold_bwd_f32(const desco_gossip_fold_params* p, const desco_gossip_fold_out* o,
                                         const desco_gossip_fold_grads* d, desco_stream_t stream) {
  if (!p || !o || !d || p->num_q < 1 || p->num_q > 64) return fail(DESCO_EINVAL, "desco_gossip_fold_bwd_f32: bad argument (1 <= num_q <= 64)");
  tn::GossipFoldArgs a{p->E, p->w_pre, p->b_pre, p->C0, p->cb0, p->D0, p->db0, p->C1, p->cb1, p->D1, p->db1,
                       {p->G0[0], p->G0[1]}, {p->gb0[0], p->gb0[1]}, {p->g2[0], p->g2[1]}, {p->gb2[0], p->gb2[1]},
                       p->P0, p->p0, p->P3, p->P5, p->num_q};
  tn::GossipFoldGrads g{d->dV0, d->dV1, d->dVp, d->dwt1, d->dwtp, d->dw3t, d->dw5t, d->dg1,
                        o->V0, o->V1, o->g0, o->g1, o->a, o->h0, o->h1,
                        d->dC0, d->dcb0, d->dD0, d->ddb0, d->dC1, d->dcb1, d->dD1, d->ddb1,
                        {d->dG0[0], d->dG0[1]}, {d->dgb0[0], d->dgb0[1]}, {d->dg2[0], d->dg2[1]}, {d->dgb2[0], d->dgb2[1]},
                        d->dP0, d->dp0, d->dP3, d->dP5, d->scratch};
  hipLaunchKernelGGL(tn::gossip_fold_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, g);
  return launch_status("desco_gossip_fold_bwd_f32");
}
