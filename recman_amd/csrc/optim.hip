// Row-wise sparse optimizer step for the fused table rows (the
// step either side of fwd+bwd).  Keras semantics of create_optimizer (recman/tf/core/utils.py:
// 201-213): Adam(beta1 .9, beta2 .999, epsilon 1e-7 outside the sqrt) and Adagrad(initial
// accumulator 0.1), applied LAZILY: only rows that occur in the batch are touched (what
// Keras' sparse apply does to IndexedSlices; with the reference's new-optimizer-per-batch
// quirk, xDeepFM.py:121-126, lazy and dense coincide).
//
// Two passes over the B*F occurrences, no sort and no dense gradient buffer to clear:
//   accumulate: gbuf[row][0..D+2) += [d_rows | g_bias | g_lin]      (float atomics)
//   apply:      one occurrence per distinct row wins an atomicCAS on stamp[row] (== step),
//               reads the row's summed gradient, updates param / moments, and zeroes the
//               gbuf row again - gbuf is all-zero between steps by construction.
// Occurrences with idx < 0 are skipped (the caller hands them in by another call: the tag rows of
// a multi-valued feature arrive as an expanded one-field occurrence list, recman_amd/optim.py).
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void sparse_accumulate_kernel(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ field_off,
    const float *__restrict__ d_rows, const float *__restrict__ g_bias,
    const float *__restrict__ g_lin, const float *__restrict__ lin_mask, int64_t n, int F, int D, int LD,
    float *__restrict__ gbuf) {
  const int W = D + 2;
  const int64_t total = n * W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t o = t / W;
    const int k = (int)(t - o * W);
    const int64_t id = idx[o];
    if (id < 0) continue;  // occurrence handled elsewhere (multi-valued / value features)
    const int64_t row = field_off[o % F] + id;
    float v;
    if (k < D) v = d_rows[o * D + k];
    else if (k == D) v = g_bias ? g_bias[o / F] : 0.f;
    else v = g_lin ? g_lin[o / F] * (lin_mask ? lin_mask[o % F] : 1.f) : 0.f;  // linear_features subset
    if (v != 0.f) atomicAdd(gbuf + row * LD + k, v);
  }
}

// kind 0 = Adam, 1 = Adagrad, 2 = SGD
__global__ __launch_bounds__(kBlock) void sparse_apply_kernel(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ field_off, int64_t n, int F, int D,
    int LD, float *__restrict__ rows, float *__restrict__ m_state, float *__restrict__ v_state,
    float *__restrict__ gbuf, int *__restrict__ stamp, int step, int kind, float lr_t, float lr,
    float beta1, float beta2, float eps, int reset) {
  const int W = D + 2;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  // a wave takes 64 occurrences at a time: lane-parallel election, then the winners' rows
  // are updated one after the other by the whole wave (W <= 64 columns per row)
  for (int64_t o0 = wave * 64; o0 < n; o0 += nwaves * 64) {
    const int64_t o = o0 + lane;
    int64_t row = -1;
    bool win = false;
    if (o < n && idx[o] >= 0) {
      row = field_off[o % F] + idx[o];
      win = atomicExch(stamp + row, step) != step;  // first occurrence of this row in this step
    }
    unsigned long long mask = __ballot(win);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int64_t r = __shfl(row, src, 64);
      for (int k = lane; k < W; k += 64) {
        const int64_t a = r * LD + k;
        const float g = gbuf[a];
        gbuf[a] = 0.f;
        float p = rows[a];
        if (kind == 0) {
          float m = reset ? 0.f : m_state[a], v = reset ? 0.f : v_state[a];
          m = beta1 * m + (1.f - beta1) * g;
          v = beta2 * v + (1.f - beta2) * g * g;
          m_state[a] = m;
          v_state[a] = v;
          p -= lr_t * m / (sqrtf(v) + eps);
        } else if (kind == 1) {
          float acc = reset ? 0.1f : v_state[a];
          acc += g * g;
          v_state[a] = acc;
          p -= lr * g / (sqrtf(acc) + eps);
        } else {
          p -= lr * g;
        }
        rows[a] = p;
      }
    }
  }
}

}  // namespace

extern "C" int rm_sparse_optimizer_step(const int64_t *idx, const int64_t *field_off,
                                        const float *d_rows, const float *g_bias, const float *g_lin,
                                        int64_t B, int F, int D, int LD, float *rows, float *m_state,
                                        float *v_state, float *gbuf, int32_t *stamp, int step, int kind,
                                        float lr, float beta1, float beta2, float eps, int reset,
                                        const float *lin_field_mask, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0 && D > 0 && LD >= D + 2 && D + 2 <= 4096, "rm_sparse_optimizer_step: bad sizes");
  RM_REQUIRE(kind >= 0 && kind <= 2 && step >= 1, "rm_sparse_optimizer_step: bad kind / step");
  if (B == 0) return RM_OK;
  RM_REQUIRE(idx && field_off && d_rows && rows && gbuf && stamp, "rm_sparse_optimizer_step: NULL argument");
  RM_REQUIRE(kind == 2 || v_state, "rm_sparse_optimizer_step: state buffer missing");
  RM_REQUIRE(kind != 0 || m_state, "rm_sparse_optimizer_step: Adam needs m_state");
  const int64_t n = B * F;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = n * (D + 2);
  hipLaunchKernelGGL(sparse_accumulate_kernel, dim3(rm_grid_cap((total + kBlock - 1) / kBlock, 256 * 16)),
                     dim3(kBlock), 0, st, idx, field_off, d_rows, g_bias, g_lin, lin_field_mask, n, F, D, LD,
                     gbuf);
  float lr_t = lr;
  if (kind == 0) {
    const double t = reset ? 1.0 : (double)step;
    lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  }
  hipLaunchKernelGGL(sparse_apply_kernel, dim3(rm_grid_cap((n + 255) / 256, 256 * 8)), dim3(kBlock), 0, st,
                     idx, field_off, n, F, D, LD, rows, m_state, v_state, gbuf, stamp, step, kind, lr_t,
                     lr, beta1, beta2, eps, reset);
  RM_CHECK_LAUNCH("rm_sparse_optimizer_step");
  return RM_OK;
}
