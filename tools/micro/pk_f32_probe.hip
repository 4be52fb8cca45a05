// Do packed fp32 VALU instructions return wrong values beside another wave's MFMAs?  (round 4: the gossip kernel's
// neighbour loop did -- profiles/r4_b_gossip_f16_race.md -- and compiling it without v_pk_*_f32 cured it.)
// Half of every workgroup's waves run a dependent-chain loop of packed instructions of ONE form on operands that have
// just come back from global memory, next to the same arithmetic done with scalar v_fma_f32 / v_mul_f32 / v_add_f32,
// and count bitwise mismatches; the other half runs MFMA chains (or idles: the control).
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_f32_probe pk_f32_probe.hip ; run on the GPU box: ./pk_f32_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// form: 0 pk_fma plain; 1 pk_mul plain; 2 pk_add plain; 3 pk_mul op_sel:[0,1] (both halves x src1.hi);
//       4 pk_fma op_sel_hi:[1,0,1] (src1.lo broadcast); 5 pk_fma op_sel_hi:[0,1,1] (src0.lo broadcast);
//       6 the gossip loop's triple: mul op_sel:[0,1], fma op_sel_hi:[1,0,1], fma op_sel_hi:[0,1,1]
template <int FORM>
__device__ __forceinline__ f2 packed(const f2 a, const f2 b, const f2 c) {
  f2 d;
  if constexpr (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if constexpr (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if constexpr (FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if constexpr (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if constexpr (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if constexpr (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if constexpr (FORM == 6) {
    f2 t;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(t) : "v"(c), "v"(b));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(b), "v"(t), "v"(c));
  }
  return d;
}
// the same arithmetic with scalar instructions (inline asm too, so that the compiler cannot pack them)
__device__ __forceinline__ float sfma(const float a, const float b, const float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float smul(const float a, const float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float sadd(const float a, const float b) {
  float d;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <int FORM>
__device__ __forceinline__ f2 scalar(const f2 a, const f2 b, const f2 c) {
  f2 d;
  if constexpr (FORM == 0) { d.x = sfma(a.x, b.x, c.x); d.y = sfma(a.y, b.y, c.y); }
  if constexpr (FORM == 1) { d.x = smul(a.x, b.x); d.y = smul(a.y, b.y); }
  if constexpr (FORM == 2) { d.x = sadd(a.x, b.x); d.y = sadd(a.y, b.y); }
  if constexpr (FORM == 3) { d.x = smul(a.x, b.y); d.y = smul(a.y, b.y); }
  if constexpr (FORM == 4) { d.x = sfma(a.x, b.x, c.x); d.y = sfma(a.y, b.x, c.y); }
  if constexpr (FORM == 5) { d.x = sfma(a.x, b.x, c.x); d.y = sfma(a.x, b.y, c.y); }
  if constexpr (FORM == 6) {
    f2 t;
    t.x = smul(a.x, b.y); t.y = smul(a.y, b.y);
    t.x = sfma(c.x, b.x, t.x); t.y = sfma(c.y, b.x, t.y);
    d.x = sfma(b.x, t.x, c.x); d.y = sfma(b.x, t.y, c.y);
  }
  return d;
}

template <int FORM>
__global__ __launch_bounds__(512) void probe(const f2* __restrict__ in, int n_in, int iters, int mfma_on, unsigned* __restrict__ bad, float* __restrict__ sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!mfma_on) return;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters * 6; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[threadIdx.x] = c0[0];
    return;
  }
  unsigned nbad = 0;
  const int base = ((blockIdx.x * 4 + (wave - 4)) * 64 + lane) * 3;
  f2 accp = {0.f, 0.f}, accs = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const int o = (base + it * 977) % (n_in - 3);
    const f2 a = in[o], b = in[o + 1], c = in[o + 2];       // fresh from memory every step
    f2 p = packed<FORM>(a, b, c), s = scalar<FORM>(a, b, c);
#pragma unroll
    for (int k = 0; k < 7; ++k) {                           // a dependent chain, as in the kernel's loop
      const f2 q = {c.y, c.x};
      p = packed<FORM>(p, b, q);
      s = scalar<FORM>(s, b, q);
    }
    nbad += (__float_as_uint(p.x) != __float_as_uint(s.x)) + (__float_as_uint(p.y) != __float_as_uint(s.y));
    accp += p;
    accs += s;
  }
  if (nbad) atomicAdd(bad + (lane >> 4), nbad);             // per lane quarter
  if (accp.x + accs.x == 12345.f) sink[threadIdx.x] = accp.y;
}

template <int FORM>
int run(const f2* d_in, int n_in, unsigned* d_bad, float* d_sink, const char* name) {
  for (int mfma_on = 1; mfma_on >= 0; --mfma_on) {
    CHECK(hipMemset(d_bad, 0, 4 * sizeof(unsigned)));
    hipLaunchKernelGGL(probe<FORM>, dim3(1024), dim3(512), 0, 0, d_in, n_in, 4000, mfma_on, d_bad, d_sink);
    CHECK(hipDeviceSynchronize());
    unsigned bad[4];
    CHECK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    const double total = 1024.0 * 4 * 64 * 4000 * 2;
    printf("%-58s MFMA waves %-3s mismatching results %u %u %u %u (lane quarters) of %.2e\n", name, mfma_on ? "on" : "off",
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
  run<0>(d_in, n_in, d_bad, d_sink, "v_pk_fma_f32 (plain)");
  run<1>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 (plain)");
  run<2>(d_in, n_in, d_bad, d_sink, "v_pk_add_f32 (plain)");
  run<3>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 op_sel:[0,1]");
  run<4>(d_in, n_in, d_bad, d_sink, "v_pk_fma_f32 op_sel_hi:[1,0,1]");
  run<5>(d_in, n_in, d_bad, d_sink, "v_pk_fma_f32 op_sel_hi:[0,1,1]");
  run<6>(d_in, n_in, d_bad, d_sink, "mul op_sel:[0,1] + fma [1,0,1] + fma [0,1,1] (gossip loop)");
  return 0;
}
