// Error plumbing and ABI version of libgpmp_hip.so.
#include "common.h"
#include <cstdarg>

namespace gpmp {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return -1000 - (int)e;
}
}  // namespace gpmp

extern "C" int gpmp_hip_abi_version(void) { return 1; }
extern "C" const char* gpmp_last_error(void) { return gpmp::g_err; }
