// Stand-alone use of the C ABI (no Python, no torch): the caller owns device memory and the stream.
//   hipcc --offload-arch=gfx950 -I include examples/capi_demo.cpp -L desco_amd -ldesco_hip \
//         -Wl,-rpath,$PWD/desco_amd -o examples/capi_demo && ./examples/capi_demo
// Computes out = relu(x W^T + b) for 10 000 rows with desco_linear64_bf16x6_f32 (weights split by
// desco_split_bf16x3_f32), builds a canonical partition of a 5-cycle with a chord on the device
// and on the host, and checks both against plain C++.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "desco_hip.h"

#define CHECK(x)                                                              \
  do {                                                                        \
    int rc_ = (x);                                                            \
    if (rc_ != 0) {                                                           \
      std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, desco_last_error()); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main() {
  std::printf("desco ABI %d, %d device(s)\n", desco_abi_version(), desco_device_count());
  const int64_t m = 10000;
  const int n = 128;
  std::vector<float> x(m * 64), w(n * 64), b(n);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : x) v = 4.f * rnd();
  for (auto& v : w) v = rnd() / 4.f;
  for (auto& v : b) v = rnd();
  float *dx, *dw, *db, *dout;
  int16_t* dplanes;
  hipMalloc(&dx, x.size() * 4);
  hipMalloc(&dw, w.size() * 4);
  hipMalloc(&db, b.size() * 4);
  hipMalloc(&dout, m * n * 4);
  hipMalloc(&dplanes, (size_t)(n / 64) * 3 * 64 * 64 * 2);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipStream_t st;
  hipStreamCreate(&st);
  for (int j = 0; j < n / 64; ++j)      // planes of every 64-row block of the [out, in] weight
    CHECK(desco_split_bf16x3_f32(dw + (size_t)j * 64 * 64, 64 * 64, dplanes + (size_t)j * 3 * 64 * 64, st));
  CHECK(desco_linear64_bf16x6_f32(dx, 64, dplanes, n / 64, db, DESCO_ACT_RELU, 0.f, dout, n, m, st));
  hipStreamSynchronize(st);
  std::vector<float> out(m * n);
  hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int64_t i = 0; i < m; i += 97)
    for (int c = 0; c < n; ++c) {
      double acc = b[c];
      for (int k = 0; k < 64; ++k) acc += (double)x[i * 64 + k] * (double)w[c * 64 + k];
      acc = acc > 0 ? acc : 0;
      worst = std::fmax(worst, std::fabs(acc - out[i * n + c]));
    }
  std::printf("linear64: max |err| vs double = %.3e\n", worst);
  if (!(worst < 1e-4)) return 2;

  // canonical partition of one 5-node graph (cycle 0-1-2-3-4 plus chord 1-3), host builder
  const int64_t graph_ptr[2] = {0, 5};
  const int64_t rowptr[6] = {0, 2, 5, 7, 10, 12};
  const int32_t col[12] = {1, 4, 0, 2, 3, 1, 3, 1, 2, 4, 0, 3};
  desco_partition* p = nullptr;
  CHECK(desco_partition_build(graph_ptr, 1, rowptr, col, 4, 0, 1, &p));
  int64_t B, Nc, E, V;
  CHECK(desco_partition_sizes(p, &B, &Nc, &E, &V));
  std::printf("partition: %lld neighborhoods, %lld count rows, %lld directed edges\n", (long long)B,
              (long long)Nc, (long long)E);
  desco_partition_free(p);
  if (B != 4 || V != 5) return 3;       // node 0 has no neighbour with a smaller id: skipped
  std::printf("ok\n");
  return 0;
}
