// ABI version + thread-local error reporting for libsegger_amd.
#include "common.h"

namespace segger {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return SEGGER_EHIP;
}

}  // namespace segger

extern "C" int segger_abi_version(void) { return SEGGER_ABI_VERSION; }
extern "C" const char* segger_last_error(void) { return segger::g_err; }
