// Round 5: standalone probe of the one instruction form the gossip-kernel bisection (tools/debug/gf16_hazard,
// profiles/r5_a_gossip_f16_hazard.md) isolated:  v_pk_mul_f32 vdst, src0, src1 op_sel:[0,1]  -- OP_SEL set on SRC1 (the
// low result lane takes src1's HIGH dword) -- issued as a run of independent instructions that share src1, in waves that
// also issue MFMAs, LDS reads and global loads, eight waves per CU.  Every wave compares the packed results with the same
// products from v_mul_f32.  Forms:
//   0  v_pk_mul_f32 d, a, b op_sel:[0,1]        (src1.hi broadcast: the failing form of the kernel)
//   1  v_pk_mul_f32 d, b, a op_sel:[1,0]        (the same products with the broadcast operand as src0)
//   2  v_pk_mul_f32 d, a, bb                    (bb = {b.hi, b.hi} made by v_mov: no op_sel)
//   3  v_pk_mul_f32 d, a, b op_sel_hi:[1,0]     (src1.lo broadcast)
//   4  v_pk_fma_f32 d, a, b, 0 op_sel:[0,1,0]   (fma with the same src1 selector)
//   5  v_pk_fma_f32 d, a, 1.0, b op_sel:[0,0,1] (src2.hi into the low lane)
//   6  v_pk_add_f32 d, a, b op_sel:[0,1]
//   7  v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]   (src1 halves swapped)
// Settings: VALU only; every wave also issues MFMA bursts; + LDS reads; MFMAs ONLY in the odd waves and the packed
// instructions ONLY in the even waves (no MFMA of the checking wave itself is ever in flight); the same with 16 wait
// states (s_nop) in front of every packed run.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_f32_probe3 pk_f32_probe3.hip ; run on the GPU box: ./pk_f32_probe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// eight independent packed instructions sharing the operand b, on fixed registers v[232:247] (results) -- one asm block
#define RUN8(INSN, TAIL)                                                                        \
  asm volatile(INSN " v[232:233], %8, %16" TAIL "\n" INSN " v[234:235], %9, %16" TAIL "\n"      \
               INSN " v[236:237], %10, %16" TAIL "\n" INSN " v[238:239], %11, %16" TAIL "\n"    \
               INSN " v[240:241], %12, %16" TAIL "\n" INSN " v[242:243], %13, %16" TAIL "\n"    \
               INSN " v[244:245], %14, %16" TAIL "\n" INSN " v[246:247], %15, %16" TAIL "\n"    \
               "v_mov_b32 %0, v232\n v_mov_b32 %1, v235\n v_mov_b32 %2, v236\n v_mov_b32 %3, v239\n" \
               "v_mov_b32 %4, v240\n v_mov_b32 %5, v243\n v_mov_b32 %6, v244\n v_mov_b32 %7, v247\n" \
               : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]) \
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b) \
               : "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243",    \
                 "v244", "v245", "v246", "v247")
#define RUN8_SWAP(INSN, TAIL)                                                                   \
  asm volatile(INSN " v[232:233], %16, %8" TAIL "\n" INSN " v[234:235], %16, %9" TAIL "\n"      \
               INSN " v[236:237], %16, %10" TAIL "\n" INSN " v[238:239], %16, %11" TAIL "\n"    \
               INSN " v[240:241], %16, %12" TAIL "\n" INSN " v[242:243], %16, %13" TAIL "\n"    \
               INSN " v[244:245], %16, %14" TAIL "\n" INSN " v[246:247], %16, %15" TAIL "\n"    \
               "v_mov_b32 %0, v232\n v_mov_b32 %1, v235\n v_mov_b32 %2, v236\n v_mov_b32 %3, v239\n" \
               "v_mov_b32 %4, v240\n v_mov_b32 %5, v243\n v_mov_b32 %6, v244\n v_mov_b32 %7, v247\n" \
               : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7]) \
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b) \
               : "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243",    \
                 "v244", "v245", "v246", "v247")

__device__ __forceinline__ float smul(const float a, const float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// mode bit 0: every wave also issues MFMA bursts; bit 1: LDS reads; (global loads always: the operands)
template <int FORM>
__global__ __launch_bounds__(512) void probe(const f2* __restrict__ in, int n_in, int iters, int mode, unsigned* __restrict__ bad,
                                             float* __restrict__ sink) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 0.001f * i;
  __syncthreads();
  h8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (lane + i)); hb[i] = (_Float16)(0.002f * (lane - i)); }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  unsigned nbad = 0;
  const int base = (blockIdx.x * 512 + threadIdx.x) * 9;
  float acc = 0.f;
  const int wave = threadIdx.x >> 6;
  // split settings: the MFMA-only waves are the odd waves (mode & 4: they sit on OTHER SIMDs than the checking waves:
  // wave w runs on SIMD w % 4) or waves 0-3 (mode & 16: one MFMA wave and one checking wave on EVERY SIMD)
  if (((mode & 4) && (wave & 1)) || ((mode & 16) && wave < 4)) {
    for (int it = 0; it < iters * 3; ++it) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0);
      }
    }
    if (c0[0] + c1[1] + c2[2] == 12345.f) sink[threadIdx.x] = c0[1];
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const int o_ = (base + it * 4099) % (n_in - 9);
    f2 a[8];
    for (int k = 0; k < 8; ++k) a[k] = in[o_ + k];           // fresh from memory every step
    const f2 b = in[o_ + 8];
    float o[8];
    if (mode & 8) asm volatile("s_nop 15");
    if (mode & 32) {                                         // drain: a VALU read of every accumulator of this wave
      const float t_ = c0[3] + c1[3] + c2[3];
      asm volatile("" :: "v"(t_));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FORM == 0) RUN8("v_pk_mul_f32", " op_sel:[0,1]");
    if constexpr (FORM == 1) RUN8_SWAP("v_pk_mul_f32", " op_sel:[1,0]");
    if constexpr (FORM == 2) {
      f2 bb;
      asm volatile("v_mov_b32 %0, %1" : "=v"(bb.x) : "v"(b.y));
      asm volatile("v_mov_b32 %0, %1" : "=v"(bb.y) : "v"(b.y));
      const f2 b = bb;
      RUN8("v_pk_mul_f32", "");
    }
    float bl = b.y;                                           // the broadcast half per form
    if constexpr (FORM == 3) { RUN8("v_pk_mul_f32", " op_sel_hi:[1,0]"); bl = b.x; }
    if constexpr (FORM == 4) RUN8("v_pk_fma_f32", ", 0 op_sel:[0,1,0]");
    if constexpr (FORM == 6) RUN8("v_pk_add_f32", " op_sel:[0,1]");
    if constexpr (FORM == 7) RUN8("v_pk_mul_f32", " op_sel:[0,1] op_sel_hi:[1,0]");
    if constexpr (FORM == 5) {
      asm volatile("v_pk_fma_f32 v[232:233], %8, 1.0, %16 op_sel:[0,0,1]\n v_pk_fma_f32 v[234:235], %9, 1.0, %16 op_sel:[0,0,1]\n"
                   "v_pk_fma_f32 v[236:237], %10, 1.0, %16 op_sel:[0,0,1]\n v_pk_fma_f32 v[238:239], %11, 1.0, %16 op_sel:[0,0,1]\n"
                   "v_pk_fma_f32 v[240:241], %12, 1.0, %16 op_sel:[0,0,1]\n v_pk_fma_f32 v[242:243], %13, 1.0, %16 op_sel:[0,0,1]\n"
                   "v_pk_fma_f32 v[244:245], %14, 1.0, %16 op_sel:[0,0,1]\n v_pk_fma_f32 v[246:247], %15, 1.0, %16 op_sel:[0,0,1]\n"
                   "v_mov_b32 %0, v232\n v_mov_b32 %1, v235\n v_mov_b32 %2, v236\n v_mov_b32 %3, v239\n"
                   "v_mov_b32 %4, v240\n v_mov_b32 %5, v243\n v_mov_b32 %6, v244\n v_mov_b32 %7, v247\n"
                   : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b)
                   : "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243",
                     "v244", "v245", "v246", "v247");
    }
    // expected: results 0, 2, 4, 6 are the LOW lanes (a[k].x * b.sel), results 1, 3, 5, 7 the HIGH lanes (a[k].y * b.sel)
    for (int k = 0; k < 8; ++k) {
      const float ak = (k & 1) ? a[k].y : a[k].x;
      float e;
      if constexpr (FORM == 5 || FORM == 6) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(e) : "v"(ak), "v"(b.y)); }
      else if constexpr (FORM == 7) e = smul(ak, (k & 1) ? b.x : b.y);
      else e = smul(ak, FORM == 3 ? bl : b.y);
      nbad += __float_as_uint(e) != __float_as_uint(o[k]);
      acc += o[k];
    }
    if (mode & 1) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0);
      }
    }
    if (mode & 2) acc += lds[(lane * 17 + it * 64) & 8191] + lds[(lane * 5 + it * 32 + 4096) & 8191];
  }
  if (nbad) atomicAdd(bad + (lane >> 4), nbad);             // per lane quarter
  if (acc + c0[0] + c1[1] + c2[2] == 12345.f) sink[threadIdx.x] = acc;
}

template <int FORM>
int run(const f2* d_in, int n_in, unsigned* d_bad, float* d_sink, const char* name) {
  static const char* names[8] = {"VALU only", "+ MFMA", "+ LDS", "MFMA other SIMDs", "MFMA other wave",
                                 "other wave, nop", "+ MFMA, drained", "+ MFMA, nop 15"};
  static const int codes[8] = {0, 1, 2, 4, 16, 24, 33, 9};
  for (int mi = 0; mi < 8; ++mi) {
    const int mode = codes[mi];
    CHECK(hipMemset(d_bad, 0, 4 * sizeof(unsigned)));
    hipLaunchKernelGGL(probe<FORM>, dim3(1024), dim3(512), 0, 0, d_in, n_in, 1500, mode, d_bad, d_sink);
    CHECK(hipDeviceSynchronize());
    unsigned bad[4];
    CHECK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    const double total = 1024.0 * 512 * 1500 * 8;
    printf("%-46s %-16s mismatching results %u %u %u %u (lane quarters) of %.2e\n", name, names[mi], bad[0], bad[1], bad[2],
           bad[3], (mode & 20) ? total / 2 : total);
  }
  return 0;
}

int main() {
  const int n_in = 1 << 22;
  std::vector<f2> h(n_in);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.5f, 1.5f);
  for (auto& v : h) { v.x = u(rng); v.y = u(rng); }
  f2* d_in; unsigned* d_bad; float* d_sink;
  CHECK(hipMalloc(&d_in, n_in * sizeof(f2)));
  CHECK(hipMalloc(&d_bad, 4 * sizeof(unsigned)));
  CHECK(hipMalloc(&d_sink, 512 * sizeof(float)));
  CHECK(hipMemcpy(d_in, h.data(), n_in * sizeof(f2), hipMemcpyHostToDevice));
  run<0>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 d, a, b op_sel:[0,1] (src1.hi)");
  run<1>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 d, b, a op_sel:[1,0] (src0.hi)");
  run<2>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 d, a, {b.hi,b.hi} (no op_sel)");
  run<3>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 d, a, b op_sel_hi:[1,0] (src1.lo)");
  run<4>(d_in, n_in, d_bad, d_sink, "v_pk_fma_f32 d, a, b, 0 op_sel:[0,1,0]");
  run<5>(d_in, n_in, d_bad, d_sink, "v_pk_fma_f32 d, a, 1.0, b op_sel:[0,0,1] (src2.hi)");
  run<6>(d_in, n_in, d_bad, d_sink, "v_pk_add_f32 d, a, b op_sel:[0,1]");
  run<7>(d_in, n_in, d_bad, d_sink, "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]");
  return 0;
}
