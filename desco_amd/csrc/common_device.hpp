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

// Compile a kernel without packed fp32 instruction selection (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): for kernels
// that are not VALU-bound and in which hipcc picks the one operand selection of those instructions that MI355X executes
// wrongly beside MFMAs (OP_SEL on src1 / src2: profiles/r5_a_gossip_f16_hazard.md; tools/check_isa.py refuses a library
// that contains it).  Note that helper functions without always_inline become real calls under a target attribute.
#if defined(__HIP_DEVICE_COMPILE__)
#define DESCO_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define DESCO_NO_PACKED_F32
#endif

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == DESCO_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DESCO_ACT_LEAKY) return v > 0.f ? v : v * slope;
  return v;
}

// Truncation split of two floats into their three bf16 terms (hi + mid + lo carries all 24
// significand bits), packed per plane as (f0 | f1 << 16): the operand format of the bf16x6 kernels.
// Written on 2-vectors so the two exact residual subtractions are one v_pk_add_f32 each.
typedef float desco_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t desco_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_bf16x3(const float f0, const float f1, uint32_t& hi,
                                              uint32_t& mid, uint32_t& lo) {
  const desco_f2 f = {f0, f1};
  const desco_u2 u = __builtin_bit_cast(desco_u2, f);
  const desco_f2 a = f - __builtin_bit_cast(desco_f2, u & 0xffff0000u);
  const desco_u2 v = __builtin_bit_cast(desco_u2, a);
  const desco_f2 c = a - __builtin_bit_cast(desco_f2, v & 0xffff0000u);
  const desco_u2 w = __builtin_bit_cast(desco_u2, c);
  // v_perm_b32: bytes {2,3} of the first float, bytes {2,3} of the second
  hi = __builtin_amdgcn_perm(u.y, u.x, 0x07060302u);
  mid = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);
  lo = __builtin_amdgcn_perm(w.y, w.x, 0x07060302u);
}

// two floats -> (bf16(f0) | bf16(f1) << 16), round to nearest even (v_cvt_pk_bf16_f32; NaN stays NaN)
typedef __bf16 desco_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16_rne(const float f0, const float f1) {
  const desco_f2 f = {f0, f1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, desco_bf2));
}

// ---- fp16 hi/lo split ("f16x3": three products hi*hi + hi*lo + lo*hi on v_mfma_f32_*_f16) -----------------------
// x (already multiplied by a power-of-two scale that puts the largest element of its row / tile into [2^14, 2^15))
// = hi + lo with hi = fp16_rne(x) and lo = fp16_rne(x - hi): 22 significand bits for every element within 2^-17 of
// the maximum, an absolute error of 2^-40 of the maximum below that (lo goes subnormal and then to zero: gradual; the
// MFMA honours fp16 subnormals, tools/micro/f16x3_probe.hip).  Three VALU per pair: v_cvt_pk_f16_f32 for the hi pair,
// v_fma_mixlo_f16 / v_fma_mixhi_f16 for the two residuals.  Packed as (f0 | f1 << 16) per plane.
typedef _Float16 desco_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_f16x2(const float f0, const float f1, uint32_t& hi, uint32_t& lo) {
  const desco_f2 f = {f0, f1};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, desco_h2));          // v_cvt_pk_f16_f32
  // lo = fp16(f - hi) straight from the packed hi halves (mixed-precision fma: f32 * 1.0 - f16 -> f16); hipcc's own
  // lowering of the C expression is v_cvt_f32_f16 x 2 + v_pk_fma_f32 + v_cvt_pk_f16_f32 (4 instructions for these 2)
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(f0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(f1), "v"(hi));
}
// the power of two s with s * mx in [2^14, 2^15) (mx >= 0; 1 for mx == 0): exponent bits only, no rounding anywhere.
// mx = 1.m * 2^(eb-127)  ->  s = 2^(141 - eb)  (biased 268 - eb, clamped to a normal number)
__device__ __forceinline__ float f16_scale_for(const float mx) {
  const int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  int sb = 268 - eb;
  sb = sb > 253 ? 253 : sb;
  return mx > 0.f ? __uint_as_float((uint32_t)sb << 23) : 1.f;
}
// 1 / s for a power of two s (exact)
__device__ __forceinline__ float pow2_inverse(const float s) {
  return __uint_as_float((254u << 23) - (__float_as_uint(s) & 0x7f800000u));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace desco
