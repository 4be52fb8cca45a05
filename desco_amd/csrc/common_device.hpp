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

// ---- counter-based dropout (round 6) -----------------------------------------------------------------------------
// F.dropout / nn.Dropout of the training steps (gnn_model.py:274, 44-53 of the reference) without a stored mask: the
// 32 random bits of element (row, col) of dropout call site `site` in optimisation step `step` are word (row & 3) of
//   Philox4x32-10( counter = { row >> 2, col | site << 24, lo32(step), hi32(step) }, key = { lo32(seed), hi32(seed) } )
// (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 round function and constants),
// so that forward and backward regenerate the same mask from (seed, step) in device memory -- which is what makes the
// step replayable from a hipGraph -- and one Philox call serves the four consecutive rows a lane holds of one column
// in the C/D layout of the 32x32 MFMAs.  The element is DROPPED iff bits < threshold (= round(p 2^32)); kept elements
// are multiplied by scale = 1 / (1 - p).  oracle/dropout.py restates this in numpy (known-answer vectors in its test).
struct PhiloxOut {
  uint32_t w[4];
};
__device__ __forceinline__ PhiloxOut philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                   uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    c0 = h1 ^ c1 ^ k0;
    c1 = l1;
    c2 = h0 ^ c3 ^ k1;
    c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return PhiloxOut{{c0, c1, c2, c3}};
}
// by-value form of desco_dropout for kernel arguments (key == nullptr: no dropout)
struct DropArgs {
  const uint64_t* key;
  uint32_t site, threshold;
  float scale;
};
__device__ __forceinline__ DropArgs no_dropout() { return DropArgs{nullptr, 0u, 0u, 1.f}; }
// random words of rows 4*(row4) .. 4*(row4)+3 of column col
__device__ __forceinline__ PhiloxOut dropout_bits4(const DropArgs& d, const uint64_t seed, const uint64_t step,
                                                   const uint32_t row4, const uint32_t col) {
  return philox4x32_10(row4, col | (d.site << 24), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}
// the factor of one element: 0 (dropped) or scale
__device__ __forceinline__ float dropout_factor(const DropArgs& d, const uint64_t seed, const uint64_t step,
                                                const int64_t row, const int col) {
  const PhiloxOut o = dropout_bits4(d, seed, step, (uint32_t)(row >> 2), (uint32_t)col);
  const int s = (int)(row & 3);
  const uint32_t bits = s == 0 ? o.w[0] : s == 1 ? o.w[1] : s == 2 ? o.w[2] : o.w[3];
  return bits < d.threshold ? 0.f : d.scale;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace desco
