// Multi-valued (CSV tag list) features - MultiValCsvFeat (recman/tf/inputs.py:380-425).
// Replaces tf.nn.embedding_lookup_sparse(..., combiner="sqrtn") of layers.py:144-169 and
// the multi-hot linear input of utils.py:86-108, without touching the hot gather kernel:
// rm_pool_rows writes ONE pooled fused row per example into a scratch row block
//   cols 0..D-1 : sum_t emb[tag_t] / sqrt(n)     (sqrtn combiner, n = number of tags)
//   col  D      : sum_t bias[tag_t] / sqrt(n)
//   col  D+1    : sum_{t: tag_t >= 1} lin[tag_t]  (multi-hot count; slot 0 is zeroed, utils.py:108)
// and the main kernel then gathers that row like any other (its index is the scratch row).
// rm_pool_rows_bwd scatters the row gradient back to the tag rows with the same factors.
//
// With per-id weights `vals` the same two kernels serve the value-weighted lookup of
// SparseValueFeat (inputs.py:213-278; layers.py:129-142, utils.py:70-71): one id per example,
//   cols 0..D-1 : v * emb[id]     col D : bias[id] (not scaled)     col D+1 : v * lin[id]
// (no sqrtn factor, slot 0 kept).
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;

// 16 lanes per example: lane k handles columns k, k+16, ... of the fused row
__global__ __launch_bounds__(kBlock) void pool_rows_kernel(
    const float *__restrict__ rows, int64_t row0, int LD, int D, const int64_t *__restrict__ offsets,
    const int64_t *__restrict__ ids, const float *__restrict__ vals, int64_t B,
    float *__restrict__ out) {
  const int lane = threadIdx.x & 15;
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  if (b >= B) return;
  const int64_t s = offsets[b], e = offsets[b + 1];
  const float inv = e > s ? rsqrtf((float)(e - s)) : 0.f;
  for (int k = lane; k < LD; k += 16) {
    float acc = 0.f;
    if (k < D + 2) {
      for (int64_t t = s; t < e; ++t) {
        const int64_t id = ids[t];
        const float x = rows[(row0 + id) * LD + k];
        if (vals != nullptr) {
          acc += k == D ? x : vals[t] * x;
        } else if (k != D + 1 || id >= 1) {
          acc += x;
        }
      }
      if (vals == nullptr && k <= D) acc *= inv;
    }
    out[b * LD + k] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void pool_rows_bwd_kernel(
    const float *__restrict__ d_rows, int64_t dr_stride, const float *__restrict__ g_bias,
    const float *__restrict__ g_lin, int D, const int64_t *__restrict__ offsets,
    const int64_t *__restrict__ ids, const float *__restrict__ vals, int64_t B, int64_t row0,
    float *__restrict__ d_table, float *__restrict__ d_bias, float *__restrict__ d_lin) {
  const int lane = threadIdx.x & 15;
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  if (b >= B) return;
  const int64_t s = offsets[b], e = offsets[b + 1];
  if (e <= s) return;
  const float inv = rsqrtf((float)(e - s));
  for (int64_t t = s; t < e; ++t) {
    const int64_t r = row0 + ids[t];
    const float we = vals != nullptr ? vals[t] : inv;                      // embedding columns
    const float wb = vals != nullptr ? 1.f : inv;                          // bias column
    const float wl = vals != nullptr ? vals[t] : (ids[t] >= 1 ? 1.f : 0.f);  // linear column
    for (int k = lane; k < D; k += 16) atomicAdd(d_table + r * D + k, d_rows[b * dr_stride + k] * we);
    if (lane == 0) {
      if (d_bias != nullptr && g_bias != nullptr) atomicAdd(d_bias + r, g_bias[b] * wb);
      if (d_lin != nullptr && g_lin != nullptr && wl != 0.f) atomicAdd(d_lin + r, g_lin[b] * wl);
    }
  }
}


// ---- padded form (the row-sharded table's fixed-capacity exchange, recman_amd/dist.py) ----------------------
// There the tags of a feature are T extra COLUMNS of the occurrence matrix: ids[b * ids_ld + t] (-1 = no tag),
// and the row of tag (b, t) sits at position pos[b * pos_ld + t] of `rows` (the rows received from the owners).
// Static shapes: micro-batches are row slices, the step can be captured in hipGraph segments.  Same sums in the
// same order as the CSR kernels above (tags in list order; n = the number of tags present).
__device__ __forceinline__ int padded_count(const int64_t *ids, int T) {
  int n = 0;
  for (int t = 0; t < T; ++t) n += ids[t] >= 0;
  return n;
}

__global__ __launch_bounds__(kBlock) void pool_rows_padded_kernel(
    const float *__restrict__ rows, int LD, int D, const int64_t *__restrict__ pos, int64_t pos_ld,
    const int64_t *__restrict__ ids, int64_t ids_ld, const float *__restrict__ vals, int64_t vals_ld, int64_t B,
    int T, float *__restrict__ out) {
  const int lane = threadIdx.x & 15;
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  if (b >= B) return;
  const int64_t *id = ids + b * ids_ld, *ps = pos + b * pos_ld;
  const int n = padded_count(id, T);
  const float inv = n > 0 ? rsqrtf((float)n) : 0.f;
  for (int k = lane; k < LD; k += 16) {
    float acc = 0.f;
    if (k < D + 2) {
      for (int t = 0; t < T; ++t) {
        if (id[t] < 0) continue;
        const float x = rows[ps[t] * LD + k];
        if (vals != nullptr) {
          acc += k == D ? x : vals[b * vals_ld + t] * x;
        } else if (k != D + 1 || id[t] >= 1) {
          acc += x;
        }
      }
      if (vals == nullptr && k <= D) acc *= inv;
    }
    out[b * LD + k] = acc;
  }
}

// the tags' gradient rows [dE * we | g_bias * wb | g_lin * wl | 0 ..] written at their positions in the send
// buffer of the backward exchange (the factors of pool_rows_bwd_kernel; no atomics: every tag has its own slot)
__global__ __launch_bounds__(kBlock) void pack_pooled_grad_rows_kernel(
    const float *__restrict__ d_rows, int64_t dr_stride, const float *__restrict__ g_bias,
    const float *__restrict__ g_lin, int D, const int64_t *__restrict__ pos, int64_t pos_ld,
    const int64_t *__restrict__ ids, int64_t ids_ld, const float *__restrict__ vals, int64_t vals_ld, int64_t B,
    int T, int GW, float *__restrict__ out) {
  const int lane = threadIdx.x & 15;
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  if (b >= B) return;
  const int64_t *id = ids + b * ids_ld, *ps = pos + b * pos_ld;
  const int n = padded_count(id, T);
  const float inv = n > 0 ? rsqrtf((float)n) : 0.f;
  for (int t = 0; t < T; ++t) {
    if (id[t] < 0 || ps[t] < 0) continue;
    const float v = vals != nullptr ? vals[b * vals_ld + t] : 0.f;
    const float we = vals != nullptr ? v : inv;
    const float wb = vals != nullptr ? 1.f : inv;
    const float wl = vals != nullptr ? v : (id[t] >= 1 ? 1.f : 0.f);
    float *o = out + ps[t] * GW;
    for (int k = lane; k < GW; k += 16) {
      float x = 0.f;
      if (k < D) x = d_rows[b * dr_stride + k] * we;
      else if (k == D) x = g_bias != nullptr ? g_bias[b] * wb : 0.f;
      else if (k == D + 1) x = g_lin != nullptr ? g_lin[b] * wl : 0.f;
      o[k] = x;
    }
  }
}

}  // namespace

extern "C" int rm_pool_rows(const float *rows, int64_t row0, int LD, int D, const int64_t *offsets,
                            const int64_t *ids, const float *vals, int64_t B, float *out,
                            rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && D > 0 && LD >= D + 2, "rm_pool_rows: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(rows && offsets && out, "rm_pool_rows: NULL argument");
  hipLaunchKernelGGL(pool_rows_kernel, dim3((unsigned)((B * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, rows, row0, LD, D, offsets, ids, vals, B, out);
  RM_CHECK_LAUNCH("rm_pool_rows");
  return RM_OK;
}

extern "C" int rm_pool_rows_bwd(const float *d_rows, int64_t dr_stride, const float *g_bias,
                                const float *g_lin, int D, const int64_t *offsets, const int64_t *ids,
                                const float *vals, int64_t B, int64_t row0, float *d_table,
                                float *d_bias, float *d_lin, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && D > 0 && dr_stride >= D, "rm_pool_rows_bwd: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(d_rows && offsets && d_table, "rm_pool_rows_bwd: NULL argument");
  hipLaunchKernelGGL(pool_rows_bwd_kernel, dim3((unsigned)((B * 16 + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, (hipStream_t)stream, d_rows, dr_stride, g_bias, g_lin, D, offsets, ids, vals, B, row0,
                     d_table, d_bias, d_lin);
  RM_CHECK_LAUNCH("rm_pool_rows_bwd");
  return RM_OK;
}

extern "C" int rm_pool_rows_padded(const float *rows, int LD, int D, const int64_t *pos, int64_t pos_ld,
                                   const int64_t *ids, int64_t ids_ld, const float *vals, int64_t vals_ld,
                                   int64_t B, int T, float *out, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && D > 0 && LD >= D + 2 && T > 0 && pos_ld >= T && ids_ld >= T && (!vals || vals_ld >= T),
             "rm_pool_rows_padded: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(rows && pos && ids && out, "rm_pool_rows_padded: NULL argument");
  hipLaunchKernelGGL(pool_rows_padded_kernel, dim3((unsigned)((B * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, rows, LD, D, pos, pos_ld, ids, ids_ld, vals, vals_ld, B, T, out);
  RM_CHECK_LAUNCH("rm_pool_rows_padded");
  return RM_OK;
}

extern "C" int rm_pack_pooled_grad_rows(const float *d_rows, int64_t dr_stride, const float *g_bias,
                                        const float *g_lin, int D, const int64_t *pos, int64_t pos_ld,
                                        const int64_t *ids, int64_t ids_ld, const float *vals, int64_t vals_ld,
                                        int64_t B, int T, int width, float *out, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && D > 0 && dr_stride >= D && width >= D + 2 && T > 0 && pos_ld >= T && ids_ld >= T &&
                 (!vals || vals_ld >= T),
             "rm_pack_pooled_grad_rows: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(d_rows && pos && ids && out, "rm_pack_pooled_grad_rows: NULL argument");
  hipLaunchKernelGGL(pack_pooled_grad_rows_kernel, dim3((unsigned)((B * 16 + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, (hipStream_t)stream, d_rows, dr_stride, g_bias, g_lin, D, pos, pos_ld, ids, ids_ld, vals,
                     vals_ld, B, T, width, out);
  RM_CHECK_LAUNCH("rm_pack_pooled_grad_rows");
  return RM_OK;
}
