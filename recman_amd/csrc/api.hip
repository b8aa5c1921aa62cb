// Library-level entry points: version, thread-local error text, device query.
#include <stdarg.h>

#include "rm_common.h"

namespace {
thread_local char g_err[512] = "";
}

extern "C" void rm_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *rm_last_error(void) { return g_err; }

extern "C" int rm_version(void) { return 100; }  // 0.1.0

extern "C" int rm_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RM_ELAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return RM_ELAUNCH;
  return prop.multiProcessorCount;
}
