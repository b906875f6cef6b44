// Error state and version of the C ABI (include/lasr.h).
#include "common.h"

namespace lasr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e > 0 ? (int)e : 999;
}
}  // namespace lasr

extern "C" int lasr_version(void) { return LASR_VERSION; }
extern "C" const char* lasr_last_error(void) { return lasr::g_err; }
