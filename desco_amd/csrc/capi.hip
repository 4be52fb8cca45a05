// Library-level entry points of the C ABI (include/desco_hip.h).
#include "common_device.hpp"

namespace desco {
std::string& last_error_ref() {
  static thread_local std::string s;
  return s;
}
int fail(int code, const char* msg) {
  last_error_ref() = msg ? msg : "";
  return code;
}
}  // namespace desco

// The production sources hold no timing-ablation switches (they left in round 5; tools/debug/gf16_hazard/make_ablations.py
// shows the form that replaced them: a script patches a COPY of a kernel).  A library assembled from such patched copies
// computes wrong results on purpose and is built with -DDESCO_DEBUG_ABLATION: it reports a version no product loader accepts.
#if defined(DESCO_DEBUG_ABLATION)
extern "C" int desco_abi_version(void) { return DESCO_ABI_VERSION + 1000; }
#else
extern "C" int desco_abi_version(void) { return DESCO_ABI_VERSION; }
#endif

extern "C" int desco_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return -(int)e;
  return n;
}

extern "C" const char* desco_last_error(void) { return desco::last_error_ref().c_str(); }
