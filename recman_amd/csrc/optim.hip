// Optimizer steps (the step either side of fwd+bwd).  Keras semantics of
// create_optimizer (recman/tf/core/utils.py:201-213): Adam(beta1 .9, beta2 .999, epsilon 1e-7 OUTSIDE
// the sqrt), Adagrad(initial accumulator 0.1), SGD.
//
// rm_sparse_optimizer_step / rm_sparse_optimizer_step_rows: ROW-WISE and LAZY - only table rows that
// occur in the batch are touched.  This is LazyAdam, a deliberate deviation from Keras, whose
// Adam._resource_apply_sparse decays m and v of EVERY row and so reads and writes the whole table each
// step (1.66 GB at BASELINE configs[1], 25.6 GB at configs[4]); the two coincide under the reference's
// new-optimizer-per-batch quirk (xDeepFM.py:121-126, `reset`), and whenever every row is touched every
// step (tests/test_gpu_optim.py pins both, and the divergence for untouched rows).  DESIGN.md section 6.
//
// Store-then-sum instead of float atomics (MI355X_MICROARCH.md, Global float atomics: scattered 72-byte
// segments are the slow shape, and the sum order would depend on arrival):
//   1. keys     key[o] = global row of occurrence o (or R = "skip"), val[o] = o: iterators, never stored
//   2. sort     stable LSD radix sort of (key, val) on the bits of R (rocPRIM device radix sort - a plain
//               library primitive, like a library GEMM; the hot kernels around it are hand-written)
//   3. apply    one lane group per DISTINCT row (the first position of each run of equal keys): sums
//               the run's gradient rows in occurrence order - fixed order, bit-reproducible - and
//               updates parameter row and moments in ONE pass; nothing else is written.
//               Runs longer than kLong positions are left to a second kernel, one wave per run, whose
//               lane groups take the members round-robin and combine in a fixed order.
//
// Row layout the step works on (recman_amd/engine.py, dist.py): parameter row [ld floats] =
//   [D embedding | bias | lin | m_bias | m_lin | v_bias | v_lin | pad pad | ..]  (ld >= D + 8, ld % 4 == 0)
// and ONE moment row [2 D floats] per table row, interleaved per float4 slice:
//   [m[0:4] v[0:4] | m[4:8] v[4:8] | ...]
// a touched row costs two 128-byte lines read + written at D = 16 (round 1 kept m, v and a gradient
// buffer as three more [R, 2D] arrays: 3 lines each way + the atomics).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "rm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kLong = 32;  // runs longer than this many positions go to the one-wave-per-run kernel

struct GradSrc {
  // mode 0: d_rows [n, D] + per-example g_bias / g_lin [n / F] (+ per-field linear mask)
  // mode 1: packed rows [n, gw]: columns [0, D) embedding gradient, D: bias gradient, D+1: linear gradient
  int mode;
  const float *rows;
  int64_t ld;
  const float *g_bias, *g_lin, *lin_mask;
  int F;
};

struct OptArgs {
  int kind;  // 0 Adam, 1 Adagrad, 2 SGD
  float lr_t, lr, beta1, beta2, eps;
  int reset;
  // lazy l2: the gradient of embedding_l2_reg * 1/2 |row|^2 (resp. linear_l2_reg on the linear entry) is added
  // for the rows a batch touches, ONCE per distinct row and step (the reference's dense term, layers.py:188-193,
  // 349-354, makes every row's gradient non-zero: the whole table per step).  The FM bias entry has no l2.
  float l2_emb, l2_lin;
};

// key of occurrence o: its global table row, or R for a skipped one (sorts behind every row).
// (Handing the sort a transform iterator instead of this kernel's arrays was slower, 155 vs 138 us: every
// pass re-read the 8-byte ids.)
__global__ __launch_bounds__(kBlock) void sparse_keys_kernel(const int64_t *__restrict__ idx,
                                                            const int64_t *__restrict__ field_off, int F,
                                                            const int64_t *__restrict__ row_ids, int64_t n,
                                                            uint32_t R, uint32_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals,
                                                            uint32_t *__restrict__ long_count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) *long_count = 0;  // this step's long-run ticket counter
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += stride) {
    int64_t r;
    if (row_ids != nullptr) {
      r = row_ids[o];
    } else {
      const int64_t id = idx[o];
      r = id < 0 ? -1 : field_off[o % F] + id;
    }
    keys[o] = (r < 0 || r >= (int64_t)R) ? R : (uint32_t)r;
    vals[o] = (uint32_t)o;
  }
}

__device__ __forceinline__ void opt_update(float &p, float &m, float &v, float g, const OptArgs &a) {
  if (a.kind == 0) {
    if (a.reset) { m = 0.f; v = 0.f; }
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    p -= a.lr_t * m / (sqrtf(v) + a.eps);
  } else if (a.kind == 1) {
    if (a.reset) v = 0.1f;
    v += g * g;
    p -= a.lr * g / (sqrtf(v) + a.eps);
  } else {
    p -= a.lr * g;
  }
}

// gradient slice of occurrence o for lane `sub` of its group: sub < GE: float4 slice of the embedding
// gradient; sub == GE: (g_bias, g_lin, 0, 0)
template <int GE>
__device__ __forceinline__ float4 grad_slice(const GradSrc &s, uint32_t o, int sub, int D) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sub < GE) {
    g = *reinterpret_cast<const float4 *>(s.rows + (int64_t)o * s.ld + 4 * sub);
  } else if (sub == GE) {
    if (s.mode == 0) {
      const uint32_t b = o / (uint32_t)s.F, f = o - b * (uint32_t)s.F;
      if (s.g_bias != nullptr) g.x = s.g_bias[b];
      if (s.g_lin != nullptr) g.y = s.g_lin[b] * (s.lin_mask != nullptr ? s.lin_mask[f] : 1.f);
    } else {
      const float2 t = *reinterpret_cast<const float2 *>(s.rows + (int64_t)o * s.ld + D);
      g.x = t.x;
      g.y = t.y;
    }
  }
  return g;
}

// One table row's state as the lane group holds it.  Lanes sub < GE own a float4 slice of the embedding
// and its moments; the MOMENT ROW is interleaved [m slice 0 | v slice 0 | m slice 1 | v slice 1 ...] so
// that lane sub reads and writes 32 contiguous bytes and the group covers the whole line - every store
// is whole 32-byte sectors (a 24-byte tail store is a read-modify-write at the memory side).  Lanes GE
// and GE + 1 hold the side entries [bias lin m_b m_l] and [v_b v_l pad pad] (32 contiguous bytes too).
struct RowState {
  float4 p, m, v;
};

template <int GE>
__device__ __forceinline__ RowState load_row(const float *__restrict__ rows, int64_t ld,
                                             const float *__restrict__ mom, int D, uint32_t r, int sub,
                                             const OptArgs &a) {
  RowState s;
  s.p = s.m = s.v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sub < GE) {
    s.p = *reinterpret_cast<const float4 *>(rows + (int64_t)r * ld + 4 * sub);
    if (a.kind != 2) {
      const float4 *pm = reinterpret_cast<const float4 *>(mom + (int64_t)r * 2 * D + 8 * sub);
      s.m = pm[0];
      s.v = pm[1];
    }
  } else if (sub <= GE + 1) {
    s.p = *reinterpret_cast<const float4 *>(rows + (int64_t)r * ld + 4 * sub);  // cols D.., D+4..
  }
  return s;
}

// applies the summed gradient g (lanes sub < GE: embedding slice; lane GE: (g_bias, g_lin, -, -)) and
// stores the row back
template <int G, int GE>
__device__ __forceinline__ void apply_row(float *__restrict__ rows, int64_t ld, float *__restrict__ mom,
                                          int D, uint32_t r, int sub, float4 g, RowState s, const OptArgs &a) {
  // lane GE needs (v_b, v_l) of lane GE + 1, and hands the updated pair back
  const float vb_in = __shfl_down(s.p.x, 1, G), vl_in = __shfl_down(s.p.y, 1, G);
  float vb = vb_in, vl = vl_in;
  if (sub < GE) {
    g.x += a.l2_emb * s.p.x; g.y += a.l2_emb * s.p.y; g.z += a.l2_emb * s.p.z; g.w += a.l2_emb * s.p.w;
  } else if (sub == GE) {
    g.y += a.l2_lin * s.p.y;
  }
  if (sub < GE) {
    opt_update(s.p.x, s.m.x, s.v.x, g.x, a);
    opt_update(s.p.y, s.m.y, s.v.y, g.y, a);
    opt_update(s.p.z, s.m.z, s.v.z, g.z, a);
    opt_update(s.p.w, s.m.w, s.v.w, g.w, a);
  } else if (sub == GE) {
    opt_update(s.p.x, s.p.z, vb, g.x, a);  // bias: p.x, m = p.z, v = lane GE+1's .x
    opt_update(s.p.y, s.p.w, vl, g.y, a);  // lin:  p.y, m = p.w, v = lane GE+1's .y
  }
  const float vb_out = __shfl_up(vb, 1, G), vl_out = __shfl_up(vl, 1, G);
  if (sub == GE + 1) {
    s.p.x = vb_out;
    s.p.y = vl_out;
  }
  if (sub <= GE + 1) *reinterpret_cast<float4 *>(rows + (int64_t)r * ld + 4 * sub) = s.p;
  if (sub < GE && a.kind != 2) {
    float4 *pm = reinterpret_cast<float4 *>(mom + (int64_t)r * 2 * D + 8 * sub);
    if (a.kind == 0) pm[0] = s.m;
    pm[1] = s.v;
  }
}

// G lanes per sorted position (power of two >= GE + 2); a lane group takes kPos ADJACENT positions and
// issues each level of their loads together.  The chain keys -> {occurrence, row, moments, gradient row}
// -> update is dependent memory round trips (a wave lived 6.3 us with one position per group, 75 % of it
// in s_waitcnt: profiles/r02_optimizer.md), so the kernel lives on rows in flight: no grid-stride loop,
// several positions per group.  Position i starts a run when key[i-1] differs.
#ifndef RM_OPT_POS
#define RM_OPT_POS 2
#endif
constexpr int kPos = RM_OPT_POS;

template <int G, int GE>
__global__ __launch_bounds__(kBlock) void sparse_apply_kernel(
    const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, uint32_t R,
    GradSrc src, int D, float *__restrict__ rows, int64_t ld, float *__restrict__ mom, OptArgs a,
    uint32_t *__restrict__ long_list, uint32_t *__restrict__ long_count) {
  const int sub = threadIdx.x % G;
  const int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G) * kPos;
  if (base >= n) return;
  // keys base-1 .. base+kPos and the occurrence ids in one batch of loads (clamped at the ends)
  uint32_t k[kPos + 2], o[kPos];
#pragma unroll
  for (int u = 0; u < kPos + 2; ++u) {
    const int64_t i = base - 1 + u;
    k[u] = keys[i < 0 ? 0 : (i < n ? i : n - 1)];
  }
#pragma unroll
  for (int u = 0; u < kPos; ++u) o[u] = vals[base + u < n ? base + u : n - 1];
  bool head[kPos], more[kPos];
#pragma unroll
  for (int u = 0; u < kPos; ++u) {
    const int64_t i = base + u;
    head[u] = i < n && k[u + 1] < R && (i == 0 || k[u] != k[u + 1]);  // k >= R: skipped (sorted last)
    more[u] = head[u] && i + 1 < n && k[u + 2] == k[u + 1];
  }
  RowState st[kPos];
  float4 g[kPos];
#pragma unroll
  for (int u = 0; u < kPos; ++u) {
    g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (head[u]) {
      st[u] = load_row<GE>(rows, ld, mom, D, k[u + 1], sub, a);  // in flight beside the gradient rows
      g[u] = grad_slice<GE>(src, o[u], sub, D);
    }
  }
#pragma unroll
  for (int u = 0; u < kPos; ++u) {
    if (!more[u]) continue;
    // run length, capped: longer runs are handed to the one-block-per-run kernel
    const int64_t i = base + u;
    const uint32_t kk = k[u + 1];
    int len = 2;
    while (len <= kLong && i + len < n && keys[i + len] == kk) ++len;
    if (len > kLong) {
      if (sub == 0) {
        const uint32_t slot = atomicAdd(long_count, 1u);  // (an integer ticket: order does not matter)
        long_list[slot] = (uint32_t)i;
      }
      head[u] = false;
      continue;
    }
    for (int j = 1; j < len; ++j) {
      const float4 t = grad_slice<GE>(src, vals[i + j], sub, D);
      g[u].x += t.x; g[u].y += t.y; g[u].z += t.z; g[u].w += t.w;
    }
  }
#pragma unroll
  for (int u = 0; u < kPos; ++u)
    if (head[u]) apply_row<G, GE>(rows, ld, mom, D, k[u + 1], sub, g[u], st[u], a);
}

// one BLOCK per long run: its 256 / G lane groups take the members round-robin (each in ascending
// order, four independent gradient rows in flight per group), the group sums are then added in group
// order through LDS - a fixed order, bit-reproducible
template <int G, int GE>
__global__ __launch_bounds__(kBlock) void sparse_apply_long_kernel(
    const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, GradSrc src, int D,
    float *__restrict__ rows, int64_t ld, float *__restrict__ mom, OptArgs a,
    const uint32_t *__restrict__ long_list, const uint32_t *__restrict__ long_count) {
  constexpr int NG = kBlock / G;
  __shared__ float4 part[kBlock];
  const int sub = threadIdx.x % G, grp = threadIdx.x / G;
  const uint32_t count = *long_count;
  for (uint32_t s = blockIdx.x; s < count; s += gridDim.x) {
    const int64_t i0 = long_list[s];
    const uint32_t k = keys[i0];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t i = i0 + grp;
    while (true) {
      // up to four members of this group per round, their loads issued together
      bool ok[4];
      uint32_t o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t iu = i + (int64_t)u * NG;
        ok[u] = iu < n && keys[iu < n ? iu : n - 1] == k;
        o[u] = vals[iu < n ? iu : n - 1];
      }
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = grad_slice<GE>(src, o[u], sub, D);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) { g.x += t[u].x; g.y += t[u].y; g.z += t[u].z; g.w += t[u].w; }
      if (!ok[3]) break;
      i += 4 * NG;
    }
    part[threadIdx.x] = g;
    __syncthreads();
    if (grp == 0) {
      float4 tot = part[sub];
      for (int q = 1; q < NG; ++q) {
        const float4 t = part[q * G + sub];
        tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
      }
      const RowState st = load_row<GE>(rows, ld, mom, D, k, sub, a);
      apply_row<G, GE>(rows, ld, mom, D, k, sub, tot, st, a);
    }
    __syncthreads();
  }
}

// dense parameters: one flat buffer, one launch (torch's per-tensor foreach ops took 0.11-0.20 ms
// of launches for ~20 small tensors)
__global__ __launch_bounds__(kBlock) void dense_opt_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                          float *__restrict__ m, float *__restrict__ v,
                                                          int64_t n, OptArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = a.kind == 0 ? m[i] : 0.f, vi = a.kind != 2 ? v[i] : 0.f;
    opt_update(pi, mi, vi, g[i], a);
    p[i] = pi;
    if (a.kind == 0) m[i] = mi;
    if (a.kind != 2) v[i] = vi;
  }
}

int bits_for(uint32_t R) {  // bits needed to represent the value R itself (the "skip" key)
  int b = 1;
  while (b < 32 && (R >> b) != 0) ++b;
  return b;
}

struct WsLayout {
  size_t keys_in, vals_in, keys_out, vals_out, long_list, sort_temp, total;
  size_t sort_bytes;
};
// the first 256 bytes of the workspace: word 0 = the long-run ticket counter (cleared by the keys kernel
// of every step)
constexpr size_t kWsHeader = 256;


int ws_layout(int64_t n, WsLayout *w) {
  size_t sort_bytes = 0;
  uint32_t *nul = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, nul, nul, nul, nul, (size_t)n, 0, 32, nullptr) != hipSuccess)
    return RM_ELAUNCH;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t at = kWsHeader;
  w->keys_in = at; at += up((size_t)n * 4);
  w->vals_in = at; at += up((size_t)n * 4);
  w->keys_out = at; at += up((size_t)n * 4);
  w->vals_out = at; at += up((size_t)n * 4);
  w->long_list = at; at += up(((size_t)n / kLong + 2) * 4);
  w->sort_temp = at; at += up(sort_bytes);
  w->sort_bytes = sort_bytes;
  w->total = at;
  return RM_OK;
}

OptArgs opt_args(int step, int kind, float lr, float beta1, float beta2, float eps, int reset, float l2_emb = 0.f,
                 float l2_lin = 0.f) {
  OptArgs a;
  a.kind = kind; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.reset = reset;
  a.l2_emb = l2_emb; a.l2_lin = l2_lin;
  a.lr_t = lr;
  if (kind == 0) {
    const double t = reset ? 1.0 : (double)step;
    a.lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  }
  return a;
}

// keys + stable sort of the occurrences by table row: depends on the ids only, so a caller may issue it
// (rm_sparse_optimizer_prepare) before / beside the forward+backward pass that produces the gradients
int sparse_prepare(const int64_t *idx, const int64_t *field_off, int F, const int64_t *row_ids, int64_t n,
                   int64_t R, void *workspace, int64_t ws_bytes, hipStream_t st, const char *fn) {
  RM_REQUIRE(n > 0 && n < (int64_t)1 << 31 && R > 0 && R < ((int64_t)1 << 32) - 1, "%s: n / R out of range", fn);
  RM_REQUIRE(row_ids || (idx && field_off && F > 0), "%s: NULL ids", fn);
  WsLayout w;
  if (ws_layout(n, &w) != RM_OK) { rm_set_error("%s: rocprim size query failed", fn); return RM_ELAUNCH; }
  RM_REQUIRE(workspace && rm_aligned16(workspace) && ws_bytes >= (int64_t)w.total,
             "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)w.total);
  char *base = (char *)workspace;
  uint32_t *keys_in = (uint32_t *)(base + w.keys_in), *vals_in = (uint32_t *)(base + w.vals_in);
  uint32_t *keys = (uint32_t *)(base + w.keys_out), *vals = (uint32_t *)(base + w.vals_out);
  hipLaunchKernelGGL(sparse_keys_kernel, dim3(rm_grid_cap((n + kBlock - 1) / kBlock, 256 * 8)), dim3(kBlock), 0,
                     st, idx, field_off, F, row_ids, n, (uint32_t)R, keys_in, vals_in, (uint32_t *)base);
  size_t sort_bytes = w.sort_bytes;
  if (rocprim::radix_sort_pairs((void *)(base + w.sort_temp), sort_bytes, keys_in, keys, vals_in, vals,
                                (size_t)n, 0, bits_for((uint32_t)R), st) != hipSuccess) {
    rm_set_error("%s: radix sort failed", fn);
    return RM_ELAUNCH;
  }
  RM_CHECK_LAUNCH(fn);
  return RM_OK;
}

__global__ void opt_clear_word_kernel(uint32_t *w) { *w = 0; }

int sparse_step(const int64_t *idx, const int64_t *field_off, int F, const int64_t *row_ids, int64_t n,
                GradSrc src, int D, int64_t R, float *rows, int64_t ld, float *mom, OptArgs a, int prepared,
                void *workspace, int64_t ws_bytes, hipStream_t st, const char *fn) {
  RM_REQUIRE(n < (int64_t)1 << 31 && R > 0 && R < ((int64_t)1 << 32) - 1, "%s: n / R out of range", fn);
  RM_REQUIRE(D % 4 == 0 && D >= 8 && D <= 64 && ld % 4 == 0 && ld >= D + 8,
             "%s: need D %% 4 == 0, 8 <= D <= 64 and a row stride ld >= D + 8 floats, ld %% 4 == 0 (D=%d ld=%lld)",
             fn, D, (long long)ld);
  RM_REQUIRE(rows && rm_aligned16(rows) && (a.kind == 2 || (mom && rm_aligned16(mom))), "%s: rows / mom NULL or unaligned", fn);
  RM_REQUIRE(src.rows && rm_aligned16(src.rows) && src.ld % 4 == 0, "%s: gradient rows NULL / unaligned", fn);
  if (!prepared) {
    int rc = sparse_prepare(idx, field_off, F, row_ids, n, R, workspace, ws_bytes, st, fn);
    if (rc != RM_OK) return rc;
  } else if (workspace) {
    // a prepared sort may be consumed more than once (the same ids stepped twice): the long-run ticket counter
    // at the head of the workspace is this step's alone (the keys kernel of an unprepared step clears it)
    hipLaunchKernelGGL(opt_clear_word_kernel, dim3(1), dim3(1), 0, st, (uint32_t *)workspace);
  }
  WsLayout w;
  if (ws_layout(n, &w) != RM_OK) { rm_set_error("%s: rocprim size query failed", fn); return RM_ELAUNCH; }
  RM_REQUIRE(workspace && rm_aligned16(workspace) && ws_bytes >= (int64_t)w.total,
             "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)w.total);
  char *base = (char *)workspace;
  uint32_t *keys = (uint32_t *)(base + w.keys_out), *vals = (uint32_t *)(base + w.vals_out);
  uint32_t *long_list = (uint32_t *)(base + w.long_list);
  uint32_t *long_count = (uint32_t *)base;
  const int GE = D / 4;
  int G = 2;
  while (G < GE + 2) G <<= 1;
  RM_REQUIRE(G <= 64, "%s: D=%d unsupported (<= 248)", fn, D);
  const int64_t pos_per_block = (int64_t)(kBlock / G) * kPos;
  dim3 grid((unsigned)((n + pos_per_block - 1) / pos_per_block));  // kPos positions per lane group
#define RM_OPT_LAUNCH(G_, GE_)                                                                               \
  {                                                                                                          \
    hipLaunchKernelGGL((sparse_apply_kernel<G_, GE_>), grid, dim3(kBlock), 0, st, keys, vals, n, (uint32_t)R, \
                       src, D, rows, ld, mom, a, long_list, long_count);                                     \
    hipLaunchKernelGGL((sparse_apply_long_kernel<G_, GE_>), dim3(1024), dim3(kBlock), 0, st, keys, vals, n, src, \
                       D, rows, ld, mom, a, long_list, long_count);                                          \
  }
  switch (GE) {
    case 2: RM_OPT_LAUNCH(4, 2) break;
    case 3: RM_OPT_LAUNCH(8, 3) break;
    case 4: RM_OPT_LAUNCH(8, 4) break;
    case 6: RM_OPT_LAUNCH(8, 6) break;
    case 8: RM_OPT_LAUNCH(16, 8) break;
    case 12: RM_OPT_LAUNCH(16, 12) break;
    case 16: RM_OPT_LAUNCH(32, 16) break;
    default:
      rm_set_error("%s: D=%d unsupported by the row-wise step (D in {8,12,16,24,32,48,64})", fn, D);
      return RM_EUNSUPPORTED;
  }
#undef RM_OPT_LAUNCH
  RM_CHECK_LAUNCH(fn);
  return RM_OK;
}

}  // namespace

extern "C" int64_t rm_sparse_optimizer_workspace(int64_t n) {
  WsLayout w;
  if (n < 0 || n >= (int64_t)1 << 31 || ws_layout(n < 1 ? 1 : n, &w) != RM_OK) return -1;
  return (int64_t)w.total;
}

extern "C" int rm_sparse_optimizer_prepare(const int64_t *idx, const int64_t *field_off, const int64_t *row_ids,
                                           int64_t n, int F, int64_t R, void *workspace, int64_t ws_bytes,
                                           rm_stream_t stream) {
  if (n == 0) return RM_OK;
  return sparse_prepare(idx, field_off, F, row_ids, n, R, workspace, ws_bytes, (hipStream_t)stream,
                        "rm_sparse_optimizer_prepare");
}

extern "C" int rm_sparse_optimizer_step(const int64_t *idx, const int64_t *field_off, const float *d_rows,
                                        const float *g_bias, const float *g_lin, int64_t B, int F, int D,
                                        int64_t R, float *rows, int64_t ld, float *mom, int step, int kind,
                                        float lr, float beta1, float beta2, float eps, int reset,
                                        float l2_embedding, float l2_linear,
                                        const float *lin_field_mask, int prepared, void *workspace,
                                        int64_t ws_bytes, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0, "rm_sparse_optimizer_step: bad sizes");
  RM_REQUIRE(kind >= 0 && kind <= 2 && step >= 1, "rm_sparse_optimizer_step: bad kind / step");
  if (B == 0) return RM_OK;
  RM_REQUIRE(prepared || (idx && field_off), "rm_sparse_optimizer_step: NULL argument");
  GradSrc src = {0, d_rows, D, g_bias, g_lin, lin_field_mask, F};
  return sparse_step(idx, field_off, F, nullptr, B * F, src, D, R, rows, ld, mom,
                     opt_args(step, kind, lr, beta1, beta2, eps, reset, l2_embedding, l2_linear), prepared,
                     workspace, ws_bytes, (hipStream_t)stream, "rm_sparse_optimizer_step");
}

extern "C" int rm_sparse_optimizer_step_rows(const int64_t *row_ids, const float *grad_rows, int64_t gw,
                                             int64_t n, int D, int64_t R, float *rows, int64_t ld, float *mom,
                                             int step, int kind, float lr, float beta1, float beta2, float eps,
                                             int reset, float l2_embedding, float l2_linear, int prepared,
                                             void *workspace, int64_t ws_bytes, rm_stream_t stream) {
  RM_REQUIRE(n >= 0 && gw >= D + 2, "rm_sparse_optimizer_step_rows: gradient rows need D + 2 columns");
  RM_REQUIRE(kind >= 0 && kind <= 2 && step >= 1, "rm_sparse_optimizer_step_rows: bad kind / step");
  if (n == 0) return RM_OK;
  RM_REQUIRE(prepared || row_ids, "rm_sparse_optimizer_step_rows: NULL argument");
  GradSrc src = {1, grad_rows, gw, nullptr, nullptr, nullptr, 1};
  return sparse_step(nullptr, nullptr, 1, row_ids, n, src, D, R, rows, ld, mom,
                     opt_args(step, kind, lr, beta1, beta2, eps, reset, l2_embedding, l2_linear), prepared,
                     workspace, ws_bytes, (hipStream_t)stream, "rm_sparse_optimizer_step_rows");
}

extern "C" int rm_dense_optimizer_step(float *p, const float *g, float *m, float *v, int64_t n, int step,
                                       int kind, float lr, float beta1, float beta2, float eps, int reset,
                                       rm_stream_t stream) {
  RM_REQUIRE(n >= 0 && kind >= 0 && kind <= 2 && step >= 1, "rm_dense_optimizer_step: bad arguments");
  if (n == 0) return RM_OK;
  RM_REQUIRE(p && g && (kind == 2 || v) && (kind != 0 || m), "rm_dense_optimizer_step: NULL argument");
  hipLaunchKernelGGL(dense_opt_kernel, dim3(rm_grid_cap((n + kBlock - 1) / kBlock, 256 * 4)), dim3(kBlock), 0,
                     (hipStream_t)stream, p, g, m, v, n, opt_args(step, kind, lr, beta1, beta2, eps, reset));
  RM_CHECK_LAUNCH("rm_dense_optimizer_step");
  return RM_OK;
}
