// Counter-based dropout of the training steps (include/desco_hip.h: desco_dropout; common_device.hpp: the Philox round
// function and the element -> counter map).  Replaces the mask tensors of F.dropout (gnn_model.py:274 of the reference,
// after every layer's relu) and nn.Dropout (post_mp.1, gnn_model.py:44-53): the epilogues that produce the dropped
// tensors multiply by a factor they compute from (seed, step, site, row, col), and the backward kernels compute it again.
//
//   rng_next_kernel       (seed, step) -> the step's key; step += 1      (one lane; captured with the step)
//   dropout_mask_kernel   the factor tensor itself (tests; the oracle takes the mask as an input)
//   act_grad_dropout      dz = dc * factor * act'(c)
#include "common_device.hpp"

namespace desco {

__global__ void rng_next_kernel(uint64_t* __restrict__ state, uint64_t* __restrict__ key_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint64_t seed = state[0], step = state[1];
    key_out[0] = seed;
    key_out[1] = step;
    state[1] = step + 1;
  }
}

// one thread per (group of four rows, column): a Philox call yields the four rows' words
__global__ __launch_bounds__(256) void dropout_mask_kernel(DropArgs d, int64_t R, int C, float* __restrict__ out,
                                                           int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r4 = i / C;
  const int c = (int)(i - r4 * C);
  if (r4 * 4 >= R) return;
  const uint64_t seed = d.key[0], step = d.key[1];
  const PhiloxOut o = dropout_bits4(d, seed, step, (uint32_t)r4, (uint32_t)c);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int64_t r = r4 * 4 + s;
    if (r < R) out[r * ldo + c] = o.w[s] < d.threshold ? 0.f : d.scale;
  }
}

__global__ __launch_bounds__(256) void act_grad_dropout_kernel(const float* __restrict__ dc, const float* __restrict__ c,
                                                               int act, float slope, DropArgs d,
                                                               float* __restrict__ dz, int64_t R, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r4 = i / C;
  const int col = (int)(i - r4 * C);
  if (r4 * 4 >= R) return;
  const uint64_t seed = d.key[0], step = d.key[1];
  const PhiloxOut o = dropout_bits4(d, seed, step, (uint32_t)r4, (uint32_t)col);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int64_t r = r4 * 4 + s;
    if (r < R) {
      const float g = dc[r * C + col], y = c[r * C + col];
      const float f = o.w[s] < d.threshold ? 0.f : d.scale;
      float v = g * f;
      if (act == DESCO_ACT_RELU)
        v = y > 0.f ? v : 0.f;
      else if (act == DESCO_ACT_LEAKY)
        v = y > 0.f ? v : v * slope;   // y = factor * leaky(z) has the sign of z where the element was kept
      dz[r * C + col] = v;
    }
  }
}

}  // namespace desco

namespace desco {
const char* dropout_check(const desco_dropout* d, int64_t num_rows, int64_t num_cols) {
  if (!d || !d->key) return "dropout descriptor without a key";
  if (d->site >= 256u) return "dropout site must be < 256";
  if (num_rows > ((int64_t)1 << 34) || num_cols >= (1 << 24)) return "dropout: rows < 2^34, cols < 2^24";
  return nullptr;
}
}  // namespace desco

extern "C" int desco_rng_next(uint64_t* state, uint64_t* key_out, desco_stream_t stream) {
  using namespace desco;
  if (!state || !key_out) return fail(DESCO_EINVAL, "desco_rng_next: NULL argument");
  hipLaunchKernelGGL(rng_next_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, key_out);
  return launch_status("desco_rng_next");
}

extern "C" int desco_dropout_mask_f32(const desco_dropout* d, int64_t num_rows, int num_cols, float* out, int64_t ldo,
                                      desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0 || num_cols == 0) return 0;
  if (!out || num_rows < 0 || num_cols < 0 || ldo < num_cols) return fail(DESCO_EINVAL, "desco_dropout_mask_f32: bad argument");
  if (const char* why = dropout_check(d, num_rows, num_cols)) return fail(DESCO_EINVAL, why);
  const int64_t items = (num_rows + 3) / 4 * num_cols, blocks = (items + 255) / 256;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_dropout_mask_f32: too many elements");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     DropArgs{d->key, d->site, d->threshold, d->scale}, num_rows, num_cols, out, ldo);
  return launch_status("desco_dropout_mask_f32");
}

extern "C" int desco_act_grad_dropout_f32(const float* dc, const float* c, int act, float slope, const desco_dropout* d,
                                          float* dz, int64_t num_rows, int num_cols, desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0 || num_cols == 0) return 0;
  if (!dc || !c || !dz || num_rows < 0 || num_cols < 0) return fail(DESCO_EINVAL, "desco_act_grad_dropout_f32: bad argument");
  if (const char* why = dropout_check(d, num_rows, num_cols)) return fail(DESCO_EINVAL, why);
  const int64_t items = (num_rows + 3) / 4 * num_cols, blocks = (items + 255) / 256;
  if (blocks > INT32_MAX) return fail(DESCO_EINVAL, "desco_act_grad_dropout_f32: too many elements");
  hipLaunchKernelGGL(act_grad_dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dc, c, act,
                     slope, DropArgs{d->key, d->site, d->threshold, d->scale}, dz, num_rows, num_cols);
  return launch_status("desco_act_grad_dropout_f32");
}
