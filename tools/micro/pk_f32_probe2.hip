// Round 5 follow-up of pk_f32_probe.hip: the patterns that probe did NOT cover.  The gossip kernel's packed build feeds
// v_pk_fma_f32 from SCALAR VALU results of the instruction(s) right in front of it (v_max_f32 on each half, v_cndmask
// on the low half of a pair whose high half is unrelated), inside a wave that also has its own MFMAs in flight.
// Each pattern is one asm block on fixed registers (so that the adjacency is exactly the kernel's), run against the same
// arithmetic in scalar instructions, in three settings: no MFMAs, MFMAs in the OTHER waves of the workgroup, MFMAs
// issued by the SAME wave right in front of the block.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_f32_probe2 pk_f32_probe2.hip ; run on the GPU box: ./pk_f32_probe2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define CLOB "v248", "v249", "v250", "v251"
// PAT 0: pk_fma -> v_max hi, v_max lo -> pk_fma (gate broadcast)      [the neighbour step of the kernel]
// PAT 1: the same with s_nop 0 in front of the consumer
// PAT 2: v_cndmask on the low half of the gate pair right in front of the consumer
// PAT 3: v_mul on the HIGH half right in front of a consumer that reads both halves
// PAT 4: three chained pk_fma with the kernel's operand swizzles, then v_max x2, then the gate pk_fma (full step)
template <int PAT>
__device__ __forceinline__ f2 packed(const f2 a, const f2 b, const f2 c, const f2 d, const float g0, const float g1, const int sel) {
  f2 r = d;
  if constexpr (PAT == 0)
    asm volatile("v_pk_fma_f32 v[250:251], %1, %2, %3\n"
                 "v_max_f32 v251, 0, v251\n"
                 "v_max_f32 v250, 0, v250\n"
                 "v_pk_fma_f32 %0, %4, v[250:251], %0 op_sel_hi:[0,1,1]"
                 : "+v"(r) : "v"(a), "v"(b), "v"(c), "v"(f2{g0, g1}) : CLOB);
  if constexpr (PAT == 1)
    asm volatile("v_pk_fma_f32 v[250:251], %1, %2, %3\n"
                 "v_max_f32 v251, 0, v251\n"
                 "v_max_f32 v250, 0, v250\n"
                 "s_nop 0\n"
                 "v_pk_fma_f32 %0, %4, v[250:251], %0 op_sel_hi:[0,1,1]"
                 : "+v"(r) : "v"(a), "v"(b), "v"(c), "v"(f2{g0, g1}) : CLOB);
  if constexpr (PAT == 2)
    asm volatile("v_pk_fma_f32 v[250:251], %1, %2, %3\n"
                 "v_max_f32 v251, 0, v251\n"
                 "v_max_f32 v250, 0, v250\n"
                 "v_cmp_ne_u32 vcc, 0, %6\n"
                 "v_mov_b32 v249, %7\n"
                 "v_cndmask_b32 v248, %4, %5, vcc\n"
                 "v_pk_fma_f32 %0, v[248:249], v[250:251], %0 op_sel_hi:[0,1,1]"
                 : "+v"(r) : "v"(a), "v"(b), "v"(c), "v"(g0), "v"(g1), "v"(sel), "v"(a.x) : CLOB, "vcc");
  if constexpr (PAT == 3)
    asm volatile("v_mov_b32 v250, %1\n"
                 "v_mul_f32 v251, %2, %3\n"
                 "v_pk_fma_f32 %0, v[250:251], %4, %0"
                 : "+v"(r) : "v"(a.x), "v"(a.y), "v"(g0), "v"(b) : CLOB);
  if constexpr (PAT == 4)
    asm volatile("v_pk_fma_f32 v[250:251], %1, %2, %3 op_sel_hi:[0,1,1]\n"
                 "v_pk_fma_f32 v[250:251], %1, %4, v[250:251] op_sel:[1,0,0]\n"
                 "v_pk_fma_f32 v[250:251], %5, %3, v[250:251] op_sel_hi:[0,1,1]\n"
                 "v_max_f32 v251, 0, v251\n"
                 "v_max_f32 v250, 0, v250\n"
                 "v_pk_fma_f32 %0, %6, v[250:251], %0 op_sel_hi:[0,1,1]"
                 : "+v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(f2{g1, g0}), "v"(f2{g0, g1}) : CLOB);
  return r;
}
__device__ __forceinline__ float sfma(const float a, const float b, const float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float smax0(const float a) {
  float d;
  asm volatile("v_max_f32 %0, 0, %1" : "=v"(d) : "v"(a));
  return d;
}
__device__ __forceinline__ float smul(const float a, const float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <int PAT>
__device__ __forceinline__ f2 scalar(const f2 a, const f2 b, const f2 c, const f2 d, const float g0, const float g1, const int sel) {
  f2 r = d;
  if constexpr (PAT == 0 || PAT == 1) {
    r.x = sfma(g0, smax0(sfma(a.x, b.x, c.x)), r.x);
    r.y = sfma(g0, smax0(sfma(a.y, b.y, c.y)), r.y);
  }
  if constexpr (PAT == 2) {
    const float g = sel != 0 ? g1 : g0;
    r.x = sfma(g, smax0(sfma(a.x, b.x, c.x)), r.x);
    r.y = sfma(g, smax0(sfma(a.y, b.y, c.y)), r.y);
  }
  if constexpr (PAT == 3) {
    r.x = sfma(a.x, b.x, r.x);
    r.y = sfma(smul(a.y, g0), b.y, r.y);
  }
  if constexpr (PAT == 4) {
    float tx = sfma(a.x, b.x, c.x), ty = sfma(a.x, b.y, c.y);
    tx = sfma(a.y, d.x, tx); ty = sfma(a.y, d.y, ty);
    tx = sfma(g1, c.x, tx); ty = sfma(g1, c.y, ty);
    r.x = sfma(g0, smax0(tx), r.x);
    r.y = sfma(g0, smax0(ty), r.y);
  }
  return r;
}

// mfma_mode: 0 none; 1 the other four waves of the workgroup run MFMA chains; 2 this wave issues MFMAs in front of each block
template <int PAT>
__global__ __launch_bounds__(512) void probe(const f2* __restrict__ in, int n_in, int iters, int mfma_mode, unsigned* __restrict__ bad, float* __restrict__ sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  h8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (lane + i)); hb[i] = (_Float16)(0.002f * (lane - i)); }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  if (wave < 4) {
    if (mfma_mode != 1) return;
    for (int it = 0; it < iters * 6; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[threadIdx.x] = c0[0];
    return;
  }
  unsigned nbad = 0;
  const int base = ((blockIdx.x * 4 + (wave - 4)) * 64 + lane) * 4;
  f2 accp = {0.f, 0.f}, accs = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const int o = (base + it * 977) % (n_in - 4);
    const f2 a = in[o], b = in[o + 1], c = in[o + 2], d = in[o + 3];       // fresh from memory every step
    const float g0 = a.y * 0.5f, g1 = b.x * 0.25f;
    const int sel = (lane + it) & 1;
    f2 p = d, s = d;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (mfma_mode == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
      }
      p = packed<PAT>(a, b, c, p, g0, g1, sel);
      s = scalar<PAT>(a, b, c, s, g0, g1, sel);
    }
    nbad += (__float_as_uint(p.x) != __float_as_uint(s.x)) + (__float_as_uint(p.y) != __float_as_uint(s.y));
    accp += p;
    accs += s;
  }
  if (nbad) atomicAdd(bad + (lane >> 4), nbad);             // per lane quarter
  if (accp.x + accs.x + c0[0] + c1[1] == 12345.f) sink[threadIdx.x] = accp.y;
}

template <int PAT>
int run(const f2* d_in, int n_in, unsigned* d_bad, float* d_sink, const char* name) {
  static const char* modes[3] = {"none", "other waves", "same wave"};
  for (int mode = 0; mode < 3; ++mode) {
    CHECK(hipMemset(d_bad, 0, 4 * sizeof(unsigned)));
    hipLaunchKernelGGL(probe<PAT>, dim3(1024), dim3(512), 0, 0, d_in, n_in, 2000, mode, d_bad, d_sink);
    CHECK(hipDeviceSynchronize());
    unsigned bad[4];
    CHECK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    const double total = 1024.0 * 4 * 64 * 2000 * 2;
    printf("%-52s MFMAs: %-11s mismatching results %u %u %u %u (lane quarters) of %.2e\n", name, modes[mode],
           bad[0], bad[1], bad[2], bad[3], total);
  }
  return 0;
}

int main() {
  const int n_in = 1 << 20;
  std::vector<f2> h(n_in);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.5f, 1.5f);
  for (auto& v : h) { v.x = u(rng); v.y = u(rng); }
  f2* d_in; unsigned* d_bad; float* d_sink;
  CHECK(hipMalloc(&d_in, n_in * sizeof(f2)));
  CHECK(hipMalloc(&d_bad, 4 * sizeof(unsigned)));
  CHECK(hipMalloc(&d_sink, 512 * sizeof(float)));
  CHECK(hipMemcpy(d_in, h.data(), n_in * sizeof(f2), hipMemcpyHostToDevice));
  run<0>(d_in, n_in, d_bad, d_sink, "pk_fma; v_max hi; v_max lo; pk_fma");
  run<1>(d_in, n_in, d_bad, d_sink, "pk_fma; v_max hi; v_max lo; s_nop 0; pk_fma");
  run<2>(d_in, n_in, d_bad, d_sink, "...; v_cndmask gate.lo; pk_fma gate broadcast");
  run<3>(d_in, n_in, d_bad, d_sink, "v_mov lo; v_mul hi; pk_fma");
  run<4>(d_in, n_in, d_bad, d_sink, "3 swizzled pk_fma; v_max x2; pk_fma (kernel step)");
  return 0;
}
