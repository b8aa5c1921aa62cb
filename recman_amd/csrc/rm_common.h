// Shared host-side helpers for the librecman_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/recman_hip.h"

// thread-local last-error text (defined in api.hip)
extern "C" void rm_set_error(const char *fmt, ...);

#define RM_REQUIRE(cond, ...)   \
  do {                          \
    if (!(cond)) {              \
      rm_set_error(__VA_ARGS__); \
      return RM_EINVAL;         \
    }                           \
  } while (0)

#define RM_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      rm_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return RM_ELAUNCH;                                                   \
    }                                                                      \
  } while (0)

static inline bool rm_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// MI355X: 256 CUs; memory-bound kernels cap the grid and grid-stride the rest.
static inline int rm_grid_cap(int64_t want, int cap = 256 * 8) {
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

#ifdef __HIPCC__
// 64-lane butterfly sum (all lanes get the total)
__device__ __forceinline__ float rm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum within aligned groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float rm_group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif
