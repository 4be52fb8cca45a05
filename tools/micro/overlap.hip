// Do f32 MFMAs and VALU FMAs overlap on one SIMD?  Block = 512 threads = 8 waves = 2 per SIMD.
// mode 0: all waves MFMA-f32; 1: all waves VALU; 2: waves 0-3 MFMA-f32, 4-7 VALU (one of each per SIMD)
// mode 3: all waves MFMA-bf16; 4: waves 0-3 MFMA-bf16 + 4-7 VALU
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
__global__ __launch_bounds__(512) void k(float* out, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma32 = mode == 0 || (mode == 2 && wave < 4);
  const bool do_bf16 = mode == 3 || (mode == 4 && wave < 4);
  float r = 0.f;
  if (do_mfma32) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    r = acc[0] + acc[5];
  } else if (do_bf16) {
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(threadIdx.x + j); b[j] = (short)(0x3f80 + j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    r = acc[0] + acc[5];
  } else {
    float x0 = threadIdx.x * 1e-3f, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
    const float m = 1.000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
        x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
      }
    }
    r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  const int iters = 20000;
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms\n", mode, ms);
  }
  return 0;
}
