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

extern "C" int rm_version(void) { return 200; }  // 0.2.0

namespace {
__global__ void rm_profile_marker_kernel(int tag, int *sink) {
  if (sink != nullptr && threadIdx.x == 0 && tag == -0x7fffffff) *sink = tag;  // never true: no store
}
}  // namespace

extern "C" int rm_profile_marker(int tag, rm_stream_t stream) {
  hipLaunchKernelGGL(rm_profile_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tag, (int *)nullptr);
  RM_CHECK_LAUNCH("rm_profile_marker");
  return RM_OK;
}

extern "C" int rm_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RM_ELAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return RM_ELAUNCH;
  return prop.multiProcessorCount;
}
