// Branch-logit sum + PredictionLayer + loss, forward and backward in one pass.
// Replaces xDeepFM.py:99-104, layers.py:796-808, utils.py:192-198 (see recman_hip.h).
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;

__global__ __launch_bounds__(kBlock) void logit_loss_kernel(
    const float *__restrict__ la, float ca, const float *__restrict__ lb, float cb,
    const float *__restrict__ lc, float cc, const float *__restrict__ ld, float cd,
    const int64_t *__restrict__ y, const float *__restrict__ y_f, int task, int64_t B,
    float *__restrict__ logit, float *__restrict__ pred, float *__restrict__ dlogit,
    float *__restrict__ partial) {
  __shared__ float sm[kBlock / 64];
  const float invB = 1.0f / (float)B;
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
    float z = 0.f;
    if (la) z += ca * la[i];
    if (lb) z += cb * lb[i];
    if (lc) z += cc * lc[i];
    if (ld) z += cd * ld[i];
    if (logit) logit[i] = z;
    const float t = y ? (float)y[i] : (y_f ? y_f[i] : 0.f);
    float p, dz;
    acc += rm_loss_point(z, t, task, &p, &dz);
    if (pred) pred[i] = p;
    if (dlogit) dlogit[i] = dz * invB;
  }
  acc = rm_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && partial) {
    float s = 0.f;
    for (int w = 0; w < kBlock / 64; ++w) s += sm[w];
    partial[blockIdx.x] = s;
  }
}

__global__ void loss_finish_kernel(const float *__restrict__ partial, int n, float invB,
                                   float *__restrict__ loss) {
  // one wave, fixed order: deterministic
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
  acc = rm_wave_sum(acc);
  if (threadIdx.x == 0) loss[0] = acc * invB;
}

// out[b] = sum_j X[b,j]*w[j] + w0 : 16 lanes per row, float4 loads when P % 4 == 0
__global__ __launch_bounds__(kBlock) void rowdot_kernel(const float *__restrict__ X,
                                                        const float *__restrict__ w,
                                                        const float *__restrict__ w0, int64_t B,
                                                        int P, float *__restrict__ out) {
  const int lane = threadIdx.x & 63, sub = lane & 15, ex = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const float bias = w0 ? w0[0] : 0.f;
  for (int64_t b0 = wave * 4; b0 < B; b0 += nwaves * 4) {
    const int64_t b = b0 + ex;
    const int64_t bb = b < B ? b : B - 1;
    float acc = 0.f;
    if ((P & 3) == 0) {
      for (int j = sub * 4; j < P; j += 64) {
        const float4 x = *reinterpret_cast<const float4 *>(X + bb * P + j);
        const float4 ww = *reinterpret_cast<const float4 *>(w + j);
        acc += x.x * ww.x + x.y * ww.y + x.z * ww.z + x.w * ww.w;
      }
    } else {
      for (int j = sub; j < P; j += 16) acc += X[bb * P + j] * w[j];
    }
    acc = rm_group_sum<16>(acc);
    if (b < B && sub == 0) out[b] = acc + bias;
  }
}

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == RM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
__device__ __forceinline__ float act_g(float o, int act) {
  if (act == RM_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return o > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// in place: x[b, j] = act(x[b, j] + bias[j])      (the DNN layer epilogue, layers.py:593-601)
__global__ __launch_bounds__(kBlock) void bias_act_kernel(float4 *__restrict__ x,
                                                          const float *__restrict__ bias, int64_t n4,
                                                          int N4, int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
    const int j4 = (int)(t % N4);
    float4 v = x[t];
    const float4 b = bias ? *reinterpret_cast<const float4 *>(bias + 4 * j4) : make_float4(0, 0, 0, 0);
    v.x = act_f(v.x + b.x, act); v.y = act_f(v.y + b.y, act);
    v.z = act_f(v.z + b.z, act); v.w = act_f(v.w + b.w, act);
    x[t] = v;
  }
}

// in place: da[b, j] *= act'(a[b, j])  (act' read off the post-activation value)
__global__ __launch_bounds__(kBlock) void act_bwd_kernel(float4 *__restrict__ da,
                                                         const float4 *__restrict__ a, int64_t n4,
                                                         int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
    float4 d = da[t];
    const float4 o = a[t];
    d.x *= act_g(o.x, act); d.y *= act_g(o.y, act); d.z *= act_g(o.z, act); d.w *= act_g(o.w, act);
    da[t] = d;
  }
}

}  // namespace

extern "C" int rm_bias_act(float *x, const float *bias, int64_t B, int N, int act, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && N > 0 && N % 4 == 0, "rm_bias_act: N must be a positive multiple of 4");
  if (B == 0) return RM_OK;
  RM_REQUIRE(x && rm_aligned16(x) && (!bias || rm_aligned16(bias)), "rm_bias_act: NULL or unaligned");
  const int64_t n4 = B * (N / 4);
  hipLaunchKernelGGL(bias_act_kernel, dim3(rm_grid_cap((n4 + kBlock - 1) / kBlock, 256 * 16)), dim3(kBlock),
                     0, (hipStream_t)stream, (float4 *)x, bias, n4, N / 4, act);
  RM_CHECK_LAUNCH("rm_bias_act");
  return RM_OK;
}

// da[b,j] = g[b] * w[j] * act'(a[b,j]): dLoss/d(pre-activation) of the LAST hidden layer of a wide
// DNN from the logit gradient (the backward of the [H,1] output projection, layers.py:606-609, and of
// the last activation) - one pass over a instead of a K = 1 GEMM launch
__global__ __launch_bounds__(kBlock) void outer_actgrad_kernel(const float *__restrict__ g,
                                                               const float *__restrict__ w,
                                                               const float4 *__restrict__ a, int64_t B,
                                                               int N4, int act, float4 *__restrict__ da) {
  const int64_t n4 = B * N4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += stride) {
    const int64_t b = t / N4;
    const int j = (int)(t - b * N4) * 4;
    const float gb = g[b];
    float4 d = make_float4(gb * w[j], gb * w[j + 1], gb * w[j + 2], gb * w[j + 3]);
    if (a != nullptr) {
      const float4 o = a[t];
      d.x *= act_g(o.x, act); d.y *= act_g(o.y, act); d.z *= act_g(o.z, act); d.w *= act_g(o.w, act);
    }
    da[t] = d;
  }
}

extern "C" int rm_outer_actgrad(const float *g, const float *w, const float *a, int64_t B, int N, int act,
                                float *da, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && N > 0 && N % 4 == 0, "rm_outer_actgrad: N must be a multiple of 4");
  if (B == 0) return RM_OK;
  RM_REQUIRE(g && w && da && rm_aligned16(da) && (!a || rm_aligned16(a)), "rm_outer_actgrad: NULL or unaligned");
  const int64_t n4 = B * (N / 4);
  hipLaunchKernelGGL(outer_actgrad_kernel, dim3(rm_grid_cap((n4 + kBlock - 1) / kBlock, 256 * 16)),
                     dim3(kBlock), 0, (hipStream_t)stream, g, w, (const float4 *)a, B, N / 4, act,
                     (float4 *)da);
  RM_CHECK_LAUNCH("rm_outer_actgrad");
  return RM_OK;
}

// rm_outer_actgrad + the three column reductions of the same two arrays in ONE pass:
//   da[b,j] = g[b] w[j] act'(a[b,j]);  d_w[j] = sum_b g[b] a[b,j];  db[j] = sum_b da[b,j];  d_w0 = sum_b g[b].
// A block owns rows_per_block rows, a thread a float4 column group (the layout of
// colsum_w_wide_stage1): per-block partials [nblk][N+1] x 2, reduced by sums_stage2 in a fixed order.
__global__ __launch_bounds__(kBlock) void outer_actgrad_sums_kernel(
    const float *__restrict__ g, const float *__restrict__ w, const float *__restrict__ a, int64_t B,
    int N, int act, int64_t rows_per_block, float *__restrict__ da, float *__restrict__ part_w,
    float *__restrict__ part_b) {
  const int N4 = N / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < B ? r0 + rows_per_block : B;
  float *ow = part_w + (int64_t)blockIdx.x * (N + 1);
  float *ob = part_b + (int64_t)blockIdx.x * (N + 1);
  for (int c4 = threadIdx.x; c4 < N4; c4 += kBlock) {
    const float4 w4 = *reinterpret_cast<const float4 *>(w + 4 * c4);
    float4 sw = make_float4(0.f, 0.f, 0.f, 0.f), sb = sw;
#pragma unroll 4
    for (int64_t r = r0; r < r1; ++r) {
      const float gv = g[r];
      const float4 o = *reinterpret_cast<const float4 *>(a + r * N + 4 * c4);
      float4 d = make_float4(gv * w4.x, gv * w4.y, gv * w4.z, gv * w4.w);
      d.x *= act_g(o.x, act); d.y *= act_g(o.y, act); d.z *= act_g(o.z, act); d.w *= act_g(o.w, act);
      *reinterpret_cast<float4 *>(da + r * N + 4 * c4) = d;
      sw.x += gv * o.x; sw.y += gv * o.y; sw.z += gv * o.z; sw.w += gv * o.w;
      sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
    }
    ow[4 * c4 + 0] = sw.x; ow[4 * c4 + 1] = sw.y; ow[4 * c4 + 2] = sw.z; ow[4 * c4 + 3] = sw.w;
    ob[4 * c4 + 0] = sb.x; ob[4 * c4 + 1] = sb.y; ob[4 * c4 + 2] = sb.z; ob[4 * c4 + 3] = sb.w;
  }
  if (threadIdx.x == kBlock - 1) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += g[r];
    ow[N] = s;
  }
}

// column j of either partial array (blockIdx.y) summed over the blocks by four waves, fixed order
__global__ __launch_bounds__(256) void sums_stage2_kernel(const float *__restrict__ part_w,
                                                          const float *__restrict__ part_b, int nblk, int N,
                                                          float *__restrict__ d_w, float *__restrict__ d_w0,
                                                          float *__restrict__ db) {
  __shared__ float sm[4];
  const int j = blockIdx.x;
  const float *part = blockIdx.y == 0 ? part_w : part_b;
  if (blockIdx.y == 1 && j == N) return;
  float acc = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < nblk; i += 256) acc += part[(int64_t)i * (N + 1) + j];
  acc = rm_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  acc = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  if (blockIdx.y == 1) {
    if (db) db[j] = acc;
  } else if (j < N) {
    if (d_w) d_w[j] = acc;
  } else if (d_w0) {
    d_w0[0] = acc;
  }
}

static int outer_sums_blocks(int64_t B) { return rm_grid_cap((B + 63) / 64, 2048); }

extern "C" int64_t rm_outer_actgrad_sums_workspace(int64_t B, int N) {
  return 2 * (int64_t)outer_sums_blocks(B) * (N + 1);
}

extern "C" int rm_outer_actgrad_sums(const float *g, const float *w, const float *a, int64_t B, int N,
                                     int act, float *da, float *d_w, float *d_w0, float *db,
                                     float *workspace, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && N > 0 && N % 4 == 0, "rm_outer_actgrad_sums: N must be a multiple of 4");
  if (B == 0) return RM_OK;
  RM_REQUIRE(g && w && a && da && workspace && rm_aligned16(da) && rm_aligned16(a) && rm_aligned16(w),
             "rm_outer_actgrad_sums: NULL or unaligned");
  const int nblk = outer_sums_blocks(B);
  const int64_t rpb = (B + nblk - 1) / nblk;
  float *pw = workspace, *pb = workspace + (int64_t)nblk * (N + 1);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(outer_actgrad_sums_kernel, dim3(nblk), dim3(kBlock), 0, st, g, w, a, B, N, act, rpb, da,
                     pw, pb);
  hipLaunchKernelGGL(sums_stage2_kernel, dim3(N + 1, 2), dim3(256), 0, st, pw, pb, nblk, N, d_w, d_w0, db);
  RM_CHECK_LAUNCH("rm_outer_actgrad_sums");
  return RM_OK;
}

extern "C" int rm_act_bwd(float *da, const float *a, int64_t B, int N, int act, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && N > 0 && (B * N) % 4 == 0, "rm_act_bwd: B*N must be a multiple of 4");
  if (B == 0) return RM_OK;
  RM_REQUIRE(da && a && rm_aligned16(da) && rm_aligned16(a), "rm_act_bwd: NULL or unaligned");
  const int64_t n4 = B * N / 4;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(rm_grid_cap((n4 + kBlock - 1) / kBlock, 256 * 16)), dim3(kBlock),
                     0, (hipStream_t)stream, (float4 *)da, (const float4 *)a, n4, act);
  RM_CHECK_LAUNCH("rm_act_bwd");
  return RM_OK;
}

extern "C" int rm_rowdot(const float *X, const float *w, const float *w0, int64_t B, int P,
                         float *out, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && P > 0, "rm_rowdot: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(X && w && out, "rm_rowdot: NULL argument");
  RM_REQUIRE((P & 3) != 0 || (rm_aligned16(X) && rm_aligned16(w)), "rm_rowdot: unaligned");
  hipLaunchKernelGGL(rowdot_kernel, dim3(rm_grid_cap((B + 15) / 16, 256 * 8)), dim3(kBlock), 0,
                     (hipStream_t)stream, X, w, w0, B, P, out);
  RM_CHECK_LAUNCH("rm_rowdot");
  return RM_OK;
}

extern "C" int rm_logit_loss(const float *logit_a, float coef_a, const float *logit_b,
                             float coef_b, const float *logit_c, float coef_c,
                             const float *logit_d, float coef_d, const int64_t *y,
                             const float *y_f, int task, int64_t B, float *logit, float *pred,
                             float *dlogit, float *loss, float *workspace, rm_stream_t stream) {
  RM_REQUIRE(B > 0, "rm_logit_loss: B must be > 0 (got %lld)", (long long)B);
  RM_REQUIRE(task == 0 || task == 1, "rm_logit_loss: task must be 0 or 1");
  RM_REQUIRE(logit_a || logit_b || logit_c || logit_d, "rm_logit_loss: no branch logit given");
  RM_REQUIRE(!(dlogit || loss) || y || y_f, "rm_logit_loss: labels needed for loss/dlogit");
  RM_REQUIRE(!loss || workspace, "rm_logit_loss: loss needs a workspace of >= 1024 floats");
  const int nblk = rm_grid_cap((B + kBlock - 1) / kBlock, kMaxBlocks);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(logit_loss_kernel, dim3(nblk), dim3(kBlock), 0, st, logit_a, coef_a, logit_b,
                     coef_b, logit_c, coef_c, logit_d, coef_d, y, y_f, task, B, logit, pred, dlogit,
                     loss ? workspace : nullptr);
  if (loss)
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, workspace, nblk,
                       1.0f / (float)B, loss);
  RM_CHECK_LAUNCH("rm_logit_loss");
  return RM_OK;
}
