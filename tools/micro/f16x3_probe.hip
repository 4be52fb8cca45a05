// Numerics probe for the fp16 hi/lo "three-product" form of an fp32 GEMM on the gfx950 matrix pipe
// (VERDICT r3 item 1).  x = hi + lo with hi = fp16_rne(s x), lo = fp16_rne(s x - hi) (s a power of two);
// a*b ~= hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, against
//   * the f32 MFMA (v_mfma_f32_16x16x4_f32),
//   * the bf16x6 form the library ships,
//   * the four-product form (+ lo*lo),
// all compared with an fp64 host product.  Also answers: does the MFMA honour fp16 subnormal inputs?
// Build: hipcc --offload-arch=gfx950 -O3 -o f16x3_probe f16x3_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_f16(const float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
__device__ __forceinline__ void split_bf16x3(const float f, short& h, short& m, short& l) {
  const uint32_t uh = __float_as_uint(f) & 0xffff0000u;
  const float r1 = f - __uint_as_float(uh);
  const uint32_t um = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(um);
  h = (short)(uh >> 16);
  m = (short)(um >> 16);
  l = (short)(__float_as_uint(r2) >> 16);
}

// mode 0: f32 MFMA; 1: bf16x6; 2: f16x3; 3: f16x4; 4: f16 hi only; 5: f16x3 with truncated (rtz) hi
// A [M][K] row-major, W [N][K] (n-major), C [M][N]; one wave per 16x16 tile; sa / sw: power-of-two scales
// applied before the fp16 split (row scale of A: per row max -> 2^14 when sa < 0).
__global__ void probe_kernel(const float* A, const float* W, float* C, int M, int N, int K, int mode,
                             float sa, float sw) {
  const int lane = threadIdx.x & 63;
  const int tm = blockIdx.x, tn = blockIdx.y;
  const int r = lane & 15, q = lane >> 4;
  const float* a = A + (size_t)(tm * 16 + r) * K;
  const float* w = W + (size_t)(tn * 16 + r) * K;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float rs = sa;
  if (sa < 0.f) {            // dynamic per-row scale: row max -> [2^14, 2^15)
    float mx = 0.f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(a[k]));
    int e;
    frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
    rs = mx > 0.f ? ldexpf(1.f, 15 - e) : 1.f;
  }
  if (mode == 0) {
    for (int k0 = 0; k0 < K; k0 += 4)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k0 + q], w[k0 + q], acc, 0, 0, 0);
  } else if (mode == 1) {
    for (int k0 = 0; k0 < K; k0 += 32) {
      s8 ah, am, al, bh, bm, bl;
      for (int j = 0; j < 8; ++j) {
        short h, m, l;
        split_bf16x3(a[k0 + 8 * q + j], h, m, l);
        ah[j] = h; am[j] = m; al[j] = l;
        split_bf16x3(w[k0 + 8 * q + j], h, m, l);
        bh[j] = h; bm[j] = m; bl[j] = l;
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      h8 ah, al, bh, bl;
      for (int j = 0; j < 8; ++j) {
        _Float16 h, l;
        float xa = a[k0 + 8 * q + j] * rs;
        if (mode == 5) {
          const auto t = __builtin_amdgcn_cvt_pkrtz(xa, 0.f);
          h = (_Float16)t[0];
          l = (_Float16)(xa - (float)h);
        } else {
          split_f16(xa, h, l);
        }
        ah[j] = h; al[j] = l;
        split_f16(w[k0 + 8 * q + j] * sw, h, l);
        bh[j] = h; bl[j] = l;
      }
      if (mode == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, acc, 0, 0, 0);
      if (mode != 4) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 4; ++i) {
    float inv = 1.f;
    if (mode >= 2) {         // the scale of the OUTPUT row 4 q + i (this lane split row r)
      float ors = sa;
      if (sa < 0.f) {
        const float* ao = A + (size_t)(tm * 16 + 4 * q + i) * K;
        float mx = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(ao[k]));
        int e;
        frexpf(mx, &e);
        ors = mx > 0.f ? ldexpf(1.f, 15 - e) : 1.f;
      }
      inv = 1.f / (ors * sw);
    }
    C[(size_t)(tm * 16 + 4 * q + i) * N + tn * 16 + r] = acc[i] * inv;
  }
}

// subnormal probe: A = one fp16 subnormal per row, W = 1 -> C must be that value (not 0)
__global__ void subnormal_kernel(float* out) {
  const int lane = threadIdx.x & 63;
  h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  // row r holds 2^(-15 - (r % 10)) (subnormal fp16: below 2^-14) at k = 0; B[k = 0][col] = 1
  if ((lane >> 4) == 0) {
    a[0] = (_Float16)ldexpf(1.f, -15 - ((lane & 15) % 10));
    b[0] = (_Float16)1.f;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  // and as the B operand
  f4 acc2 = {0.f, 0.f, 0.f, 0.f};
  acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc2, 0, 0, 0);
  for (int i = 0; i < 4; ++i) {
    out[(4 * (lane >> 4) + i) * 16 + (lane & 15)] = acc[i];
    out[256 + (4 * (lane >> 4) + i) * 16 + (lane & 15)] = acc2[i];
  }
}

static void run_case(const char* name, int M, int N, int K, float ascale, int adist, float wscale, float sa,
                     float sw) {
  std::mt19937_64 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::normal_distribution<float> G(0.f, 1.f);
  std::vector<float> A((size_t)M * K), W((size_t)N * K), C((size_t)M * N);
  for (int i = 0; i < M; ++i) {
    // adist 0: uniform * ascale; 1: relu(normal) * ascale * per-row log-uniform factor over 2^+-10
    const float rowf = adist == 1 ? std::ldexp(1.f, (int)(U(rng) * 10.f)) : 1.f;
    for (int k = 0; k < K; ++k) {
      float v = adist == 0 ? U(rng) : std::max(0.f, G(rng));
      A[(size_t)i * K + k] = v * ascale * rowf;
    }
  }
  for (auto& v : W) v = U(rng) * wscale;
  std::vector<double> R((size_t)M * N), Rabs((size_t)M * N);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0, sabs = 0;
      for (int k = 0; k < K; ++k) {
        const double p = (double)A[(size_t)i * K + k] * (double)W[(size_t)j * K + k];
        s += p;
        sabs += std::fabs(p);
      }
      R[(size_t)i * N + j] = s;
      Rabs[(size_t)i * N + j] = sabs;
    }
  float *dA, *dW, *dC;
  hipMalloc(&dA, A.size() * 4);
  hipMalloc(&dW, W.size() * 4);
  hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  printf("%-34s M=%d N=%d K=%d sa=%g sw=%g\n", name, M, N, K, sa, sw);
  const char* modes[] = {"f32 mfma", "bf16x6", "f16x3", "f16x4", "f16 hi only", "f16x3 rtz-hi"};
  for (int mode = 0; mode < 6; ++mode) {
    hipMemset(dC, 0, C.size() * 4);
    probe_kernel<<<dim3(M / 16, N / 16), 64>>>(dA, dW, dC, M, N, K, mode, sa, sw);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    // error relative to sum |a||w| of the element (the scale fp32 rounding errors live on): max and rms
    double mx = 0, ss = 0, mxrel = 0;
    for (size_t i = 0; i < C.size(); ++i) {
      const double e = std::fabs((double)C[i] - R[i]) / (Rabs[i] > 0 ? Rabs[i] : 1.0);
      mx = std::max(mx, e);
      ss += e * e;
      if (std::fabs(R[i]) > 0) mxrel = std::max(mxrel, std::fabs((double)C[i] - R[i]) / std::fabs(R[i]));
    }
    printf("   %-14s max err/sum|ab| %.3e  rms %.3e   max rel-to-result %.3e\n", modes[mode], mx,
           std::sqrt(ss / C.size()), mxrel);
  }
  hipFree(dA);
  hipFree(dW);
  hipFree(dC);
}

int main() {
  float* d;
  hipMalloc(&d, 512 * 4);
  subnormal_kernel<<<1, 64>>>(d);
  std::vector<float> o(512);
  hipMemcpy(o.data(), d, 512 * 4, hipMemcpyDeviceToHost);
  printf("fp16 subnormal inputs through v_mfma_f32_16x16x32_f16 (expected 2^(-15 - r%%10)):\n");
  for (int r = 0; r < 10; ++r)
    printf("   row %d: as A %.6e  as B %.6e  expected %.6e\n", r, o[r * 16], o[256 + r], std::ldexp(1.0, -15 - r));
  const float P14 = 16384.f;
  // weights as nn.Linear(576) initialises them: U(-1/24, 1/24); pre-scaled by 2^18 -> max 2^13.4
  run_case("uniform A O(1), W U(-1/24,1/24)", 256, 64, 576, 1.f, 0, 1.f / 24.f, 1.f, 1.f);
  run_case("  same, A row-scaled, W * 2^18", 256, 64, 576, 1.f, 0, 1.f / 24.f, -1.f, 262144.f);
  run_case("uniform A * 123 (K=576)", 256, 64, 576, 123.f, 0, 1.f / 24.f, -1.f, 262144.f);
  run_case("relu-normal rows over 2^+-10", 256, 64, 576, 1.f, 1, 1.f / 24.f, -1.f, 262144.f);
  run_case("  same, fixed scales (no row scale)", 256, 64, 576, 1.f, 1, 1.f / 24.f, 16.f, 262144.f);
  run_case("tiny A 1e-4, unscaled", 256, 64, 64, 1e-4f, 0, 0.125f, 1.f, 1.f);
  run_case("tiny A 1e-4, row-scaled", 256, 64, 64, 1e-4f, 0, 0.125f, -1.f, 65536.f);
  run_case("huge A 1e5, row-scaled", 256, 64, 64, 1e5f, 0, 0.125f, -1.f, 65536.f);
  run_case("K=64 O(1)", 256, 64, 64, 1.f, 1, 0.125f, -1.f, 65536.f);
  run_case("K=128 O(1)", 256, 64, 128, 1.f, 1, 0.0884f, -1.f, 65536.f);
  (void)P14;
  return 0;
}
