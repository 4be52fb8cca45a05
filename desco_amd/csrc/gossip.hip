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
__global__ __launch_bounds__(256) void gossip_gather_kernel(const float* __restrict__ h,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            int64_t num_nodes, int Q,
                                                            const float* __restrict__ g,
                                                            float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
  const int64_t i = idx / Q;
  if (i >= num_nodes) return;
  const int q = (int)(idx % Q);
  const float gq = g[q];
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  float lo = 0.f, hi = 0.f;
  for (int e = e0; e < e1; ++e) {
    const int64_t j = col[e];
    const float v = h[(j * Q + q) * 64 + lane];
    if (j < i)
      lo += v;
    else
      hi += v;
  }
  out[idx * 64 + lane] = gq * lo + (1.f - gq) * hi;
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
  if (!h || !rowptr || !g || !out || num_nodes < 0 || num_q < 1)
    return fail(DESCO_EINVAL, "desco_gossip_gather_f32: bad argument");
  const int64_t blocks = (num_nodes * num_q + 3) / 4;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_gossip_gather_f32: too many rows");
  hipLaunchKernelGGL(gossip_gather_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, h, rowptr, col, num_nodes, num_q, g, out);
  return launch_status("desco_gossip_gather_f32");
}
