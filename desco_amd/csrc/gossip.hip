// Gossip-propagation kernels (gfx950).  All 29 queries are carried as an extra tensor axis:
// a "row" of the gossip stage is a (node i, query q) pair, r = i*Q + q, 64 fp32 features.
//
// Replaces GossipConv.message/aggregate (gnn_model.py:335-344), the per-query pre_mp + concat of
// BaseGNNCore.forward (:231-240) and the per-call edge canonicalisation (:246-248, :315): the CSR
// handed in is already symmetric / loop-free / sorted, so edge_weight = (src < dst) is just
// "neighbour id below my id".  Algebra in DESIGN.md section 4.2.
#include "common_device.hpp"

namespace desco {

// Layer 0 in closed form.  The layer-0 input is [E_q | x[i,q]*w_pre + b_pre] (rank one in x), so
//   lin_com(h0_j) = a_q + x[j,q] * v         and the gated aggregate is  alpha*a_q + beta*v,
//   alpha = g*deg_lo + (1-g)*deg_hi,  beta = g*sum_lo x + (1-g)*sum_hi x,
// and lin_update([aggr | h0_i]) = alpha*p_q + beta*r + x[i,q]*t + z_q with host-folded p,r,t,z.
__global__ __launch_bounds__(256) void gossip_layer0_kernel(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, int64_t num_nodes, int Q, const float* __restrict__ g0,
    const float* __restrict__ g1, const float* __restrict__ p, const float* __restrict__ r,
    const float* __restrict__ t, const float* __restrict__ z, float* __restrict__ h1,
    float* __restrict__ scal) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= num_nodes) return;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  const int q = lane < Q ? lane : Q - 1;
  float slo = 0.f, shi = 0.f;
  int dlo = 0;
  for (int e = e0; e < e1; ++e) {
    const int64_t j = col[e];
    const float xv = x[j * ldx + q];
    if (j < i) {
      slo += xv;
      ++dlo;
    } else {
      shi += xv;
    }
  }
  const float flo = (float)dlo, fhi = (float)(e1 - e0 - dlo);
  const float gq0 = g0[q], gq1 = g1[q];
  const float a0 = gq0 * flo + (1.f - gq0) * fhi;
  const float b0 = gq0 * slo + (1.f - gq0) * shi;
  const float a1 = gq1 * flo + (1.f - gq1) * fhi;
  const float xi = x[i * ldx + q];
  if (lane < Q) {
    scal[(i * Q + lane) * 2 + 0] = a1;
    scal[(i * Q + lane) * 2 + 1] = xi;
  }
  const float rl = r[lane], tl = t[lane];
  for (int qq = 0; qq < Q; ++qq) {
    const float A = __shfl(a0, qq, 64), B = __shfl(b0, qq, 64), X = __shfl(xi, qq, 64);
    float h = A * p[qq * 64 + lane] + B * rl + X * tl + z[qq * 64 + lane];
    h1[(i * Q + qq) * 64 + lane] = h > 0.f ? h : 0.f;
  }
}

// Gated neighbour sum for layers >= 1 (aggregate-then-transform):
//   out[i,q,:] = sum_{j~i} (j<i ? g[q] : 1-g[q]) * h[j,q,:]
// A 16-lane group (float4 per lane) per (node, query) row, sixteen rows per workgroup: consecutive rows are consecutive
// queries of one node, so the sixteen groups read 4 KB of contiguous memory per neighbour; four neighbour rows are in
// flight per lane (absent ones re-read the row's first neighbour and are not added).  Round 4's form -- one wave per row,
// one dword per lane, one neighbour at a time -- ran at 0.107 of the HBM peak and was 56 % of the gossip training step.
// The sums run over the neighbours in CSR order, as before.
__global__ __launch_bounds__(256) void gossip_gather_kernel(const float* __restrict__ h,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            int64_t num_nodes, int Q,
                                                            const float* __restrict__ g,
                                                            int signed_mode,
                                                            float* __restrict__ out) {
  const int l = threadIdx.x & 15;
  const int64_t idx = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t i = idx / Q;
  if (i >= num_nodes) return;
  const int q = (int)(idx - i * Q);
  const float gq = signed_mode ? 0.f : g[q];
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
  const int64_t qoff = (int64_t)q * 64 + 4 * l;
  const int64_t ld = (int64_t)Q * 64;
#define GG_ADD(v_, j_)                                                        \
  if ((j_) < i) { lo.x += v_.x; lo.y += v_.y; lo.z += v_.z; lo.w += v_.w; }   \
  else { hi.x += v_.x; hi.y += v_.y; hi.z += v_.z; hi.w += v_.w; }
  for (int e = e0; e < e1; e += 4) {
    const int n = e1 - e;
    const int64_t j0 = col[e];
    const int64_t j1 = n > 1 ? col[e + 1] : j0, j2 = n > 2 ? col[e + 2] : j0, j3 = n > 3 ? col[e + 3] : j0;
    const float4 v0 = *reinterpret_cast<const float4*>(h + j0 * ld + qoff);
    const float4 v1 = *reinterpret_cast<const float4*>(h + j1 * ld + qoff);
    const float4 v2 = *reinterpret_cast<const float4*>(h + j2 * ld + qoff);
    const float4 v3 = *reinterpret_cast<const float4*>(h + j3 * ld + qoff);
    GG_ADD(v0, j0)
    if (n > 1) { GG_ADD(v1, j1) }
    if (n > 2) { GG_ADD(v2, j2) }
    if (n > 3) { GG_ADD(v3, j3) }
  }
#undef GG_ADD
  // signed_mode: lo - hi (d/dg of the gated sum), else g*lo + (1-g)*hi
  float4 r;
  if (signed_mode) r = make_float4(lo.x - hi.x, lo.y - hi.y, lo.z - hi.z, lo.w - hi.w);
  else {
    const float gh = 1.f - gq;
    r = make_float4(gq * lo.x + gh * hi.x, gq * lo.y + gh * hi.y, gq * lo.z + gh * hi.z, gq * lo.w + gh * hi.w);
  }
  *reinterpret_cast<float4*>(out + idx * 64 + 4 * l) = r;
}

// out[r,:] = act( base[r,:] + sum_{k<KS} C[r,k] * V[r % QV][k][:] )   (rank-KS per-query affine term)
__global__ __launch_bounds__(256) void affine_rows_kernel(const float* __restrict__ base,
                                                          const float* __restrict__ C, int KS,
                                                          const float* __restrict__ V, int QV,
                                                          int act, float slope, DropArgs drop,
                                                          float* __restrict__ out, int64_t R) {
  // a wave takes FOUR consecutive rows per step (aligned to 4: one Philox call gives the four rows' dropout bits of a
  // column) and walks the rows with a grid stride -- round 6: one wave per row was a million waves of one 256-byte
  // store each, and a Philox call per element
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ngroups = (R + 3) / 4;
  for (int64_t gq = (int64_t)blockIdx.x * 4 + wave; gq < ngroups; gq += (int64_t)gridDim.x * 4) {
    const int64_t r0 = 4 * gq;
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t r = r0 + e < R ? r0 + e : R - 1;
      acc[e] = base ? base[r * 64 + lane] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t r = r0 + e < R ? r0 + e : R - 1;
      const float* v = V + (int64_t)(r % QV) * KS * 64 + lane;
      for (int k = 0; k < KS; ++k) acc[e] += C[r * KS + k] * v[k * 64];
      acc[e] = apply_act(acc[e], act, slope);
    }
    // dropout behind the activation (gnn_model.py:273-274; post_mp.1 in front of its LeakyReLU is the same thing: relu
    // and leaky commute with a non-negative factor); the factor is regenerated in the backward pass, not stored
    if (drop.key) {
      const PhiloxOut o = dropout_bits4(drop, drop.key[0], drop.key[1], (uint32_t)gq, (uint32_t)lane);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] *= o.w[e] < drop.threshold ? 0.f : drop.scale;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (r0 + e < R) out[(r0 + e) * 64 + lane] = acc[e];
  }
}

// partial[slab][qv][k][c] = sum over rows r = i*QV + qv of the slab of C[r,k] * dZ[r,c]
__global__ __launch_bounds__(256) void affine_rows_bwd_kernel(const float* __restrict__ C, int KS,
                                                              const float* __restrict__ dZ, int QV,
                                                              int64_t num_i, int64_t slab,
                                                              float* __restrict__ partial) {
  __shared__ float red[4][8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qv = blockIdx.x;
  const int64_t i_beg = (int64_t)blockIdx.y * slab;
  const int64_t i_end = (i_beg + slab) < num_i ? (i_beg + slab) : num_i;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  // sixteen rows of the class in flight per wave (round 6: one row per iteration was one memory round trip per
  // iteration -- 0.25 ms per call for 0.26 GB); rows beyond the slab re-read its last row with weight 0
#ifndef DESCO_ARB_ROWS
#define DESCO_ARB_ROWS 16     // (8: 0.34 ms per step for the three calls, 16: 0.29)
#endif
  constexpr int U = DESCO_ARB_ROWS;
  for (int64_t i = i_beg + wave; i < i_end; i += 4 * U) {
    float d[U], cw[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t iu = i + 4 * u;
      cw[u] = iu < i_end ? 1.f : 0.f;
      r[u] = (iu < i_end ? iu : i_end - 1) * QV + qv;
      d[u] = dZ[r[u] * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < KS) {
        float c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = C[r[u] * KS + k];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[k] = fmaf(c[u] * cw[u], d[u], acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wave][k][lane] = acc[k];
  __syncthreads();
  if (wave == 0) {
    float* o = partial + (((int64_t)blockIdx.y * QV + qv) * KS) * 64 + lane;
    for (int k = 0; k < KS; ++k)
      o[k * 64] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ partial,
                                                           int64_t count, int splits,
                                                           float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * count + i];
  out[i] = s;
}

__global__ __launch_bounds__(256) void rowdot2_kernel(const float* __restrict__ a,
                                                      const float* __restrict__ b, int ncols,
                                                      float* __restrict__ out, int64_t R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= R) return;
  float acc = 0.f;
  for (int c = lane; c < ncols; c += 64) acc += a[r * ncols + c] * b[r * ncols + c];
  acc = wave_sum(acc);
  if (lane == 0) out[r] = acc;
}

}  // namespace desco

using namespace desco;

extern "C" int desco_gossip_layer0_f32(const float* x, int64_t ldx, const int32_t* rowptr,
                                       const int32_t* col, int64_t num_nodes, int num_q,
                                       const float* g0, const float* g1, const float* p,
                                       const float* r, const float* t, const float* z, float* h1,
                                       float* scal, desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  if (!x || !rowptr || !g0 || !g1 || !p || !r || !t || !z || !h1 || !scal || num_nodes < 0 ||
      num_q < 1 || num_q > 64)
    return fail(DESCO_EINVAL, "desco_gossip_layer0_f32: bad argument (1 <= num_q <= 64)");
  const int64_t blocks = (num_nodes + 3) / 4;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_layer0_f32: too many nodes");
  hipLaunchKernelGGL(gossip_layer0_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, rowptr, col, num_nodes, num_q, g0, g1, p, r, t, z,
                     h1, scal);
  return launch_status("desco_gossip_layer0_f32");
}

extern "C" int desco_gossip_gather_f32(const float* h, const int32_t* rowptr, const int32_t* col,
                                       int64_t num_nodes, int num_q, const float* g, float* out,
                                       desco_stream_t stream) {
  if (num_nodes == 0) return 0;
  if (!h || !rowptr || !out || num_nodes < 0 || num_q < 1 || (reinterpret_cast<uintptr_t>(h) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return fail(DESCO_EINVAL, "desco_gossip_gather_f32: bad argument (h / out 16-byte aligned)");
  const int64_t blocks = (num_nodes * num_q + 15) / 16;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_gather_f32: too many rows");
  hipLaunchKernelGGL(gossip_gather_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, h, rowptr, col, num_nodes, num_q, g, g ? 0 : 1, out);
  return launch_status("desco_gossip_gather_f32");
}

namespace desco {
const char* dropout_check(const desco_dropout* d, int64_t num_rows, int64_t num_cols);   // dropout.hip
}

static int affine_rows_launch(const char* who, const float* base, const float* c, int ks, const float* v, int qv,
                              int act, float slope, const desco_dropout* d, float* out, int64_t num_rows,
                              desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0) return 0;
  if (!c || !v || !out || ks < 1 || ks > 8 || qv < 1 || num_rows < 0) {
    std::string msg = std::string(who) + ": bad argument (1 <= ks <= 8)";
    return fail(DESCO_EINVAL, msg.c_str());
  }
  int64_t blocks = ((num_rows + 3) / 4 + 3) / 4;            // a wave per group of four rows ...
  if (blocks > 16 * 1024) blocks = 16 * 1024;                // ... and a grid-stride walk beyond 64 k waves
  DropArgs da = DropArgs{nullptr, 0u, 0u, 1.f};
  if (d) {
    if (const char* why = dropout_check(d, num_rows, 64)) return fail(DESCO_EINVAL, why);
    da = DropArgs{d->key, d->site, d->threshold, d->scale};
  }
  hipLaunchKernelGGL(affine_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     base, c, ks, v, qv, act, slope, da, out, num_rows);
  return launch_status(who);
}

extern "C" int desco_affine_rows_f32(const float* base, const float* c, int ks, const float* v,
                                     int qv, int act, float slope, float* out, int64_t num_rows,
                                     desco_stream_t stream) {
  return affine_rows_launch("desco_affine_rows_f32", base, c, ks, v, qv, act, slope, nullptr, out, num_rows, stream);
}

extern "C" int desco_affine_rows_dropout_f32(const float* base, const float* c, int ks, const float* v, int qv, int act,
                                             float slope, const desco_dropout* d, float* out, int64_t num_rows,
                                             desco_stream_t stream) {
  if (!d) return desco::fail(DESCO_EINVAL, "desco_affine_rows_dropout_f32: NULL dropout descriptor");
  return affine_rows_launch("desco_affine_rows_dropout_f32", base, c, ks, v, qv, act, slope, d, out, num_rows, stream);
}

extern "C" int desco_affine_rows_bwd_f32(const float* c, int ks, const float* dz, int qv,
                                         int64_t num_rows, float* dv, float* workspace,
                                         desco_stream_t stream) {
  if (!c || !dz || !dv || !workspace || ks < 1 || ks > 8 || qv < 1 || num_rows < 0 ||
      num_rows % qv)
    return fail(DESCO_EINVAL, "desco_affine_rows_bwd_f32: bad argument");
  const int64_t num_i = num_rows / qv;
  int64_t splits = (num_i + 1023) / 1024;
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  const int64_t slab = (num_i + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(affine_rows_bwd_kernel, dim3((unsigned)qv, (unsigned)splits), dim3(256), 0, st, c,
                     ks, dz, qv, num_i, slab > 0 ? slab : 1, workspace);
  const int64_t count = (int64_t)qv * ks * 64;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                     workspace, count, (int)splits, dv);
  return launch_status("desco_affine_rows_bwd_f32");
}

extern "C" int desco_rowdot2_f32(const float* a, const float* b, int ncols, float* out,
                                 int64_t num_rows, desco_stream_t stream) {
  if (num_rows == 0) return 0;
  if (!a || !b || !out || ncols < 1 || num_rows < 0)
    return fail(DESCO_EINVAL, "desco_rowdot2_f32: bad argument");
  const int64_t blocks = (num_rows + 3) / 4;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_rowdot2_f32: too many rows");
  hipLaunchKernelGGL(rowdot2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b,
                     ncols, out, num_rows);
  return launch_status("desco_rowdot2_f32");
}
