// How far apart must two MFMAs on the same accumulator be?  (developer tool)
// Bare loops like mfma_peak.hip, but the D independent accumulators of a wave are used round-robin, so a
// dependent pair is D instructions apart: D = 1, 2, 4, 8 for v_mfma_f32_16x16x32_bf16 and
// v_mfma_f32_32x32x16_bf16, one and two waves per SIMD, every CU.  Prints TFLOP/s per case.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_dep tools/micro/mfma_dep.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE, int D>
__global__ __launch_bounds__(512) void dep_loop(const short* __restrict__ src, float* __restrict__ sink, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a0 = *reinterpret_cast<const bf16x8*>(src + 8 * ((blockIdx.x * 512 + tid) & 4095));
  bf16x8 a1 = *reinterpret_cast<const bf16x8*>(src + 8 * ((blockIdx.x * 512 + tid + 977) & 4095));
  bf16x8 b0 = *reinterpret_cast<const bf16x8*>(src + 8 * ((blockIdx.x * 512 + tid + 1999) & 4095));
  bf16x8 b1 = *reinterpret_cast<const bf16x8*>(src + 8 * ((blockIdx.x * 512 + tid + 3001) & 4095));
  float out = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 c[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        c[k % D] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((k & 1) ? a1 : a0, (k & 2) ? b1 : b0, c[k % D], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
      for (int e = 0; e < 16; ++e) out += c[k][e];
  } else {
    f32x4 c[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        c[k % D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((k & 1) ? a1 : a0, (k & 2) ? b1 : b0, c[k % D], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
      for (int e = 0; e < 4; ++e) out += c[k][e];
  }
  if (out == 123.456f) sink[0] = out;
}

template <int SHAPE, int D>
static void run(int threads, int cus, const short* src, float* sink) {
  const int iters = SHAPE == 32 ? 10000 : 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  double warm = 0;
  while (warm < 400.0) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((dep_loop<SHAPE, D>), dim3(cus), dim3(threads), 0, 0, src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    warm += ms;
  }
  const double flops = (double)cus * (threads / 64) * iters * 8.0 * (SHAPE == 32 ? 32768.0 : 16384.0);
  printf("mfma %s  distance %d  %d waves/SIMD: %.1f TFLOP/s\n", SHAPE == 32 ? "32x32x16" : "16x16x32", D, threads / 256,
         flops / (ms * 1e-3) / 1e12);
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  short* src;
  float* sink;
  hipMalloc(&src, 4096 * 16);
  hipMalloc(&sink, 4);
  short h[4096 * 8];
  uint32_t s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = (short)(0x3f00 | ((s >> 16) & 0xff));          // random bf16 in [0.5, 1)
    if (s & 0x8000u) v |= (short)0x8000;
  }
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<16, 1>(threads, cus, src, sink); run<16, 2>(threads, cus, src, sink); run<16, 4>(threads, cus, src, sink); run<16, 8>(threads, cus, src, sink);
    run<32, 1>(threads, cus, src, sink); run<32, 2>(threads, cus, src, sink); run<32, 4>(threads, cus, src, sink); run<32, 8>(threads, cus, src, sink);
  }
  return 0;
}
