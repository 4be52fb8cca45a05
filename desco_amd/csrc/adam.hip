// torch.optim.Adam's update for a whole parameter list in one or two launches.
//
// Reference: both models' configure_optimizers (subgraph_counting/lightning_model.py:160-173, 570-583) return
// torch.optim.Adam(self.parameters(), lr, weight_decay) with the default betas / eps, L2 weight decay, no amsgrad.
// torch's own implementation of that step is ~10 multi-tensor launches plus -- in the capturable form a hipGraph
// needs -- one launch PER PARAMETER for the bias corrections (~170 launches, 0.85 ms of a 5.2 ms replayed training
// step in round 4's trace).  Here: the list of (parameter, gradient) pointers goes to the kernel by value, 128
// tensors per launch; moments live in two flat buffers the caller owns, step counts per tensor on the device (a
// tensor without gradient is skipped and does not age, as in torch), the learning rate is read from device memory:
// the launch is capturable and a replay needs no host work.
//
// Arithmetic per element, as torch/optim/adam.py _single_tensor_adam:
//   g' = g + wd p;  m = m + (1-b1)(g' - m);  v = b2 v + (1-b2) g' g';
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)        (bias corrections in double)
#include "common_device.hpp"

namespace desco {

constexpr int kAdamChunk = 128;      // tensors per launch: 128 x 30 B of kernel arguments (the limit is 4 KB)
constexpr int kAdamBlock = 1024;     // elements per workgroup (256 lanes x float4)

struct AdamChunk {
  float* p[kAdamChunk];
  const float* g[kAdamChunk];
  uint32_t off[kAdamChunk];          // offset of the tensor's moments in m / v (elements)
  uint32_t n[kAdamChunk];
  int blk_end[kAdamChunk];           // running count of workgroups
  uint16_t idx[kAdamChunk];          // the tensor's index in steps / arrivals
  int count;
};

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamChunk c, float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ steps, unsigned* __restrict__ done,
                                                       const float* __restrict__ lr, double beta1, double beta2,
                                                       float eps, float wd) {
  // which tensor is this workgroup's?  (uniform: a scalar binary search over the kernel arguments)
  const int b = blockIdx.x;
  int lo = 0, hi = c.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (c.blk_end[mid] > b) hi = mid; else lo = mid + 1;
  }
  const int t = lo;
  const int b0 = t ? c.blk_end[t - 1] : 0;
  const int nb = c.blk_end[t] - b0;
  const uint32_t n = c.n[t];
  float* __restrict__ p = c.p[t];
  const float* __restrict__ g = c.g[t];
  float* __restrict__ mt = m + c.off[t];
  float* __restrict__ vt = v + c.off[t];

  __shared__ float hyp[2];
  if (threadIdx.x == 0) {
    const double s = (double)steps[c.idx[t]] + 1.0;
    hyp[0] = (float)((double)lr[0] / (1.0 - pow(beta1, s)));       // step size
    hyp[1] = (float)sqrt(1.0 - pow(beta2, s));                     // sqrt of the second bias correction
  }
  __syncthreads();
  const float step_size = hyp[0], bc2 = hyp[1];
  // (torch forms 1 - beta in double and rounds once: 1 - 0.999 is 0.001f there, not 1.f - 0.999f)
  float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), b2f = (float)beta2;
  // (hipcc converts these as one 2-vector and then broadcasts its odd dword through OP_SEL on src1 of a packed multiply,
  //  the operand selection rule PK-OPSEL of tools/check_isa.py forbids: cut the scalars loose from that register pair)
  asm volatile("" : "+v"(w1), "+v"(w2), "+v"(b2f));

  const uint32_t i0 = (uint32_t)(b - b0) * kAdamBlock + 4 * threadIdx.x;
#define DESCO_ADAM_ONE(P, G, M, V)                        \
  {                                                       \
    const float g_ = (G) + wd * (P);                      \
    const float m_ = (M) + w1 * (g_ - (M));               \
    const float v_ = (V) * b2f + w2 * g_ * g_;          \
    (M) = m_;                                             \
    (V) = v_;                                             \
    (P) = (P) - step_size * (m_ / (sqrtf(v_) / bc2 + eps)); \
  }
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(mt) |
                     reinterpret_cast<uintptr_t>(vt)) & 15u) == 0;
  if (vec && i0 + 4 <= n) {
    float4 P = *reinterpret_cast<const float4*>(p + i0);
    const float4 G = *reinterpret_cast<const float4*>(g + i0);
    float4 M = *reinterpret_cast<const float4*>(mt + i0);
    float4 V = *reinterpret_cast<const float4*>(vt + i0);
    DESCO_ADAM_ONE(P.x, G.x, M.x, V.x)
    DESCO_ADAM_ONE(P.y, G.y, M.y, V.y)
    DESCO_ADAM_ONE(P.z, G.z, M.z, V.z)
    DESCO_ADAM_ONE(P.w, G.w, M.w, V.w)
    *reinterpret_cast<float4*>(p + i0) = P;
    *reinterpret_cast<float4*>(mt + i0) = M;
    *reinterpret_cast<float4*>(vt + i0) = V;
  } else {
    for (uint32_t i = i0; i < i0 + 4 && i < n; ++i) {
      float P = p[i], M = mt[i], V = vt[i];
      DESCO_ADAM_ONE(P, g[i], M, V)
      p[i] = P;
      mt[i] = M;
      vt[i] = V;
    }
  }
#undef DESCO_ADAM_ONE
  // the tensor's last workgroup ages it (every workgroup has read steps[t] before its arrival is counted)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(done + c.idx[t], 1u) == (unsigned)(nb - 1)) {
      done[c.idx[t]] = 0;
      steps[c.idx[t]] += 1.f;
    }
  }
}

}  // namespace desco

extern "C" int desco_adam_step_f32(int num, float* const* params, const float* const* grads, const int64_t* sizes,
                                   float* m, float* v, float* steps, uint32_t* arrivals, const float* lr, double beta1,
                                   double beta2, double eps, double weight_decay, desco_stream_t stream) {
  using namespace desco;
  if (num < 0 || (num > 0 && (!params || !grads || !sizes || !m || !v || !steps || !arrivals || !lr)))
    return fail(DESCO_EINVAL, "desco_adam_step_f32: null argument");
  AdamChunk c;
  c.count = 0;
  int64_t off = 0, off0 = 0;
  int t0 = 0;
  auto flush = [&]() -> int {
    if (c.count == 0) return 0;
    hipLaunchKernelGGL(adam_step_kernel, dim3(c.blk_end[c.count - 1]), dim3(256), 0, (hipStream_t)stream, c, m + off0,
                       v + off0, steps + t0, arrivals + t0, lr, beta1, beta2, (float)eps, (float)weight_decay);
    c.count = 0;
    return launch_status("desco_adam_step_f32");
  };
  for (int t = 0; t < num; ++t) {
    const int64_t n = sizes[t];
    if (n < 0 || n > 0x7fffffff) return fail(DESCO_EINVAL, "desco_adam_step_f32: tensor of more than 2^31 elements");
    if (grads[t] && n > 0) {          // (no gradient: the tensor keeps its value, its moments and its age)
      if (!params[t]) return fail(DESCO_EINVAL, "desco_adam_step_f32: null parameter");
      // a chunk addresses moments and counters relative to its first tensor (32- and 16-bit offsets)
      if (c.count == kAdamChunk || (c.count > 0 && (off + n - off0 > 0xffffffffll || t - t0 > 0xffff)))
        if (int rc = flush()) return rc;
      if (c.count == 0) {
        off0 = off;
        t0 = t;
      }
      const int k = c.count++;
      c.p[k] = params[t];
      c.g[k] = grads[t];
      c.off[k] = (uint32_t)(off - off0);
      c.n[k] = (uint32_t)n;
      c.idx[k] = (uint16_t)(t - t0);
      c.blk_end[k] = (k ? c.blk_end[k - 1] : 0) + (int)((n + kAdamBlock - 1) / kAdamBlock);
    }
    off += n;
  }
  return flush();
}
