// Shared error plumbing of libdesco_hip.so (host side).
#pragma once
#include <string>

namespace desco {
std::string& last_error_ref();
int fail(int code, const char* msg);
}  // namespace desco

#if defined(__HIPCC__)
#include <atomic>
#include <cstdint>
#include <hip/hip_runtime.h>
namespace desco {
// "Done once PER DEVICE" flag for the hipFuncSetAttribute guards: function attributes belong to the
// device's code object, and one process may drive several devices.
struct DeviceOnce {
  std::atomic<uint64_t> mask{0};
  static int device() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : 0;
  }
  bool done() const { return (mask.load(std::memory_order_relaxed) >> device()) & 1u; }
  void mark() { mask.fetch_or(uint64_t(1) << device(), std::memory_order_relaxed); }
};
}  // namespace desco
#endif
