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
// PredictionLayer + loss of ONE example (layers.py:796-808, utils.py:192-198): z = summed branch
// logits, t = label.  task 0: p = sigmoid(z), Keras binary_crossentropy on the probability (clip to
// [1e-7, 1 - 1e-7], epsilon inside the logs), gradient only inside the clip; task 1: MSE.
// Returns the example's loss term; *pred and *dz (dLoss_term/dz, NOT yet divided by B).
__device__ __forceinline__ float rm_loss_point(float z, float t, int task, float *pred, float *dz) {
  constexpr float kEps = 1e-7f;  // Keras backend epsilon()
  if (task == 0) {
    const float p = 1.0f / (1.0f + expf(-z));
    *pred = p;
    const float lo = kEps, hi = 1.0f - kEps;
    const float pc = fminf(fmaxf(p, lo), hi);
    const float a = pc + kEps, c = 1.0f - pc + kEps;
    const bool inside = p >= lo && p <= hi;  // clip passes the gradient only inside
    const float dp = inside ? -(t / a - (1.0f - t) / c) : 0.f;
    *dz = dp * p * (1.0f - p);
    return -(t * logf(a) + (1.0f - t) * logf(c));
  }
  *pred = z;
  const float e = z - t;
  *dz = 2.0f * e;
  return e * e;
}
// sum within aligned groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float rm_group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#endif
