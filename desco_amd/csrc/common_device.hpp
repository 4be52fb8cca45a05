// Shared device-side helpers for the gfx950 kernels of libdesco_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/desco_hip.h"
#include "common_host.hpp"

namespace desco {

constexpr int kWave = 64;  // CDNA4 wavefront
constexpr int kH = DESCO_H;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail((int)e, m.c_str());
  }
  return 0;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == DESCO_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DESCO_ACT_LEAKY) return v > 0.f ? v : v * slope;
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace desco
