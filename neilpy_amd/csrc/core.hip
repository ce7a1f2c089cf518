// Error reporting and library identification of libsmrf_hip.
#include <cstring>

#include "smrf_common.h"

namespace {
thread_local char g_err[512] = "";
}

int smrf_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {

int smrf_abi_version(void) { return SMRF_ABI_VERSION; }

const char* smrf_last_error(void) { return g_err; }

int smrf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
