// Same-wave interleave of bf16 MFMA and VALU: per iteration 1 v_mfma_f32_32x32x16_bf16 (a single
// accumulation chain) + K independent v_fma_f32, for K = 0..12, one or two waves per SIMD.
// Prints cycles per iteration (s_memtime): 32 = MFMA-bound; if VALU did not co-execute it would be
// 32 + 4K (one wave) .
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
template <int K>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(threadIdx.x + j); b[j] = (short)(0x3f80 + j); }
  float x[12];
  for (int j = 0; j < 12; ++j) x[j] = threadIdx.x * 1e-3f + j;
  const float m = 1.000001f, c = 1e-7f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < K; ++j) x[j] = fmaf(x[j], m, c);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = acc[0] + acc[5];
  for (int j = 0; j < 12; ++j) r += x[j];
  if (r == 123.456f) out[threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K>
void run(float* d, unsigned long long* c, int threads) {
  const int iters = 4000;
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(threads), 0, 0, d, c, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(threads), 0, 0, d, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("waves/SIMD %d  K=%2d VALU per MFMA: %.1f cycles per (MFMA + K VALU) per wave (s_memtime), %.3f ms\n",
         threads / 256, K, (double)h / (iters * 8.0), ms);
}
int main() {
  float* d; unsigned long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 64);
  for (int threads : {256, 512}) {
    run<0>(d, c, threads); run<2>(d, c, threads); run<4>(d, c, threads); run<6>(d, c, threads);
    run<8>(d, c, threads); run<12>(d, c, threads);
  }
  return 0;
}
