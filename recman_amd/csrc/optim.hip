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
//   2. sort     stable LSD radix sort of (key, val): per FIELD on the field's own bits with hand-written
//               passes ("field-segmented sort" below) when the caller vouches for disjoint per-field row ranges
//               (max_field_rows > 0: the [B, F] id batches of the engines); otherwise - row_ids lists, the
//               sharded table's owner side - rocPRIM's device radix sort on the bits of R
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
// Runs of up to kLong positions are summed by their lane group in sparse_apply_kernel, one member per round trip
// (occurrence -> gradient row); longer ones go to the long-run kernels in segments of kSeg positions, one WAVE
// each with kLongFlight members in flight per lane group.  Zipf(1.05) ids at configs[1] (rows with up to 6,100
// occurrences, ~15 k runs longer than 8), sparse step in us by kLong: 2: 719, 4: 500, 8: 355, 16: 336, 32: 372,
// 64: 389 (handing a run over costs a gallop for its end and a ticket from one counter); round 2 had one BLOCK
// per run and kLong = 32: 660.
#ifndef RM_OPT_LONG
#define RM_OPT_LONG 16
#endif
constexpr int kLong = RM_OPT_LONG;
constexpr int kSeg = 128;
constexpr int kLongFlight = 8;  // members in flight per lane group of the long-run kernel
// workspace header words (cleared by the keys kernel of every step)
constexpr int kHdrSegs = 0;   // segments of long runs listed so far (a ticket counter)
constexpr int kHdrPairs = 1;  // form of the sorted list (struct Sorted)
constexpr int kHdrMulti = 2;  // runs of more than one segment listed so far
constexpr int kHdrParts = 3;  // segment sums handed out

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
                                                            uint32_t *__restrict__ hdr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[kHdrSegs] = hdr[kHdrMulti] = hdr[kHdrParts] = 0;  // this step's ticket counters
    hdr[kHdrPairs] = 0;  // the sorted list: two arrays
  }
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

// The sorted (row, occurrence) list as the apply kernels read it: two arrays (the sort over all pairs), or ONE
// array of 8-byte pairs starting at `keys` (the field-segmented sort: its last pass then scatters one store per
// element instead of two).  Word 1 of the workspace header says which (written by the keys kernel of the sort).
struct Sorted {
  const uint32_t *keys, *vals;
  bool pairs;
  __device__ __forceinline__ uint32_t key(int64_t i) const { return pairs ? keys[2 * i] : keys[i]; }
  __device__ __forceinline__ uint32_t val(int64_t i) const { return pairs ? keys[2 * i + 1] : vals[i]; }
};

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
    uint4 *__restrict__ seg_list, uint4 *__restrict__ multi_list, uint32_t *__restrict__ hdr) {
  const int sub = threadIdx.x % G;
  const int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G) * kPos;
  if (base >= n) return;
  const Sorted sd = {keys, vals, hdr[kHdrPairs] != 0};
  // keys base-1 .. base+kPos and the occurrence ids in one batch of loads (clamped at the ends)
  uint32_t k[kPos + 2], o[kPos];
#pragma unroll
  for (int u = 0; u < kPos + 2; ++u) {
    const int64_t i = base - 1 + u;
    k[u] = sd.key(i < 0 ? 0 : (i < n ? i : n - 1));
  }
#pragma unroll
  for (int u = 0; u < kPos; ++u) o[u] = sd.val(base + u < n ? base + u : n - 1);
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
    // run length, capped: longer runs are handed to the long-run kernels, in segments of kSeg positions
    const int64_t i = base + u;
    const uint32_t kk = k[u + 1];
    int len = 2;
    while (len <= kLong && i + len < n && sd.key(i + len) == kk) ++len;
    if (len > kLong) {
      // the run's end: gallop, then bisect (a run is contiguous: "key == kk" is true up to its end, then false)
      int64_t lo = i + kLong, hi, step = kLong;
      while (true) {
        const int64_t t = lo + step;
        if (t >= n) { hi = n; break; }
        if (sd.key(t) == kk) { lo = t; step <<= 1; } else { hi = t; break; }
      }
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (sd.key(mid) == kk) lo = mid; else hi = mid;
      }
      if (sub == 0) {
        // (integer tickets: the ORDER of the lists does not matter - every segment's members, every run's segment
        // order and so every sum are fixed)
        const int64_t L = hi - i;
        const uint32_t nseg = (uint32_t)((L + kSeg - 1) / kSeg);
        const uint32_t at = atomicAdd(&hdr[kHdrSegs], nseg);
        uint32_t part = 0xFFFFFFFFu;
        if (nseg > 1) {
          part = atomicAdd(&hdr[kHdrParts], nseg);
          multi_list[atomicAdd(&hdr[kHdrMulti], 1u)] = make_uint4(kk, part, nseg, 0u);
        }
        for (uint32_t q = 0; q < nseg; ++q) {
          const int64_t s0 = i + (int64_t)q * kSeg;
          seg_list[at + q] = make_uint4((uint32_t)s0, (uint32_t)(hi - s0 < kSeg ? hi - s0 : kSeg), kk,
                                        nseg > 1 ? part + q : 0xFFFFFFFFu);
        }
      }
      head[u] = false;
      continue;
    }
    for (int j = 1; j < len; ++j) {
      const float4 t = grad_slice<GE>(src, sd.val(i + j), sub, D);
      g[u].x += t.x; g[u].y += t.y; g[u].z += t.z; g[u].w += t.w;
    }
  }
#pragma unroll
  for (int u = 0; u < kPos; ++u)
    if (head[u]) apply_row<G, GE>(rows, ld, mom, D, k[u + 1], sub, g[u], st[u], a);
}

// one WAVE per SEGMENT (<= kSeg positions) of a long run: its 64 / G lane groups take the members round-robin
// (each in ascending order, kLongFlight independent gradient rows in flight per group), the group sums are then added
// in group order through LDS - a fixed order, bit-reproducible.  A run of one segment is applied here; the
// segments of a longer run (a hot row of a Zipf batch: thousands of occurrences - one block per RUN walked them
// in 48 dependent rounds) leave their sums in `partials`, combined in segment order by sparse_apply_combine_kernel.
template <int G, int GE>
__global__ __launch_bounds__(kBlock) void sparse_apply_long_kernel(
    const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, GradSrc src, int D,
    float *__restrict__ rows, int64_t ld, float *__restrict__ mom, OptArgs a,
    const uint4 *__restrict__ seg_list, float4 *__restrict__ partials, const uint32_t *__restrict__ hdr) {
  constexpr int NG = 64 / G;
  __shared__ float4 part[kBlock];
  const int lane = threadIdx.x & 63, sub = lane % G, grp = lane / G;
  float4 *wpart = part + (threadIdx.x - lane);  // this wave's 64 slots: no barrier anywhere
  const uint32_t count = hdr[kHdrSegs];
  const uint32_t nwaves = gridDim.x * (kBlock / 64);
  const Sorted sd = {keys, vals, hdr[kHdrPairs] != 0};
  for (uint32_t s = (blockIdx.x * kBlock + threadIdx.x) >> 6; s < count; s += nwaves) {
    const uint4 e = seg_list[s];
    const int64_t end = (int64_t)e.x + e.y;
    RowState st;
    if (grp == 0 && e.w == 0xFFFFFFFFu) st = load_row<GE>(rows, ld, mom, D, e.z, sub, a);  // in flight beside the members
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)e.x + grp; i < end; i += kLongFlight * NG) {
      // up to kLongFlight members of this group per round, their loads issued together
      uint32_t o[kLongFlight];
#pragma unroll
      for (int u = 0; u < kLongFlight; ++u) {
        const int64_t iu = i + (int64_t)u * NG;
        o[u] = sd.val(iu < end ? iu : end - 1);
      }
      float4 t[kLongFlight];
#pragma unroll
      for (int u = 0; u < kLongFlight; ++u) t[u] = grad_slice<GE>(src, o[u], sub, D);
#pragma unroll
      for (int u = 0; u < kLongFlight; ++u)
        if (i + (int64_t)u * NG < end) { g.x += t[u].x; g.y += t[u].y; g.z += t[u].z; g.w += t[u].w; }
    }
    wpart[lane] = g;
    __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations complete in order; this pins the program order)
    if (grp == 0) {
      float4 tot = wpart[sub];
      for (int q = 1; q < NG; ++q) {
        const float4 t = wpart[q * G + sub];
        tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
      }
      if (e.w == 0xFFFFFFFFu) {
        apply_row<G, GE>(rows, ld, mom, D, e.z, sub, tot, st, a);
      } else {
        partials[(int64_t)e.w * G + sub] = tot;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// one lane group per run of several segments: the segment sums in segment order, then the update
template <int G, int GE>
__global__ __launch_bounds__(kBlock) void sparse_apply_combine_kernel(
    int D, float *__restrict__ rows, int64_t ld, float *__restrict__ mom, OptArgs a,
    const uint4 *__restrict__ multi_list, const float4 *__restrict__ partials, const uint32_t *__restrict__ hdr) {
  const int sub = threadIdx.x % G;
  const uint32_t count = hdr[kHdrMulti];
  const uint32_t groups = gridDim.x * (kBlock / G);
  for (uint32_t m = (blockIdx.x * kBlock + threadIdx.x) / G; m < count; m += groups) {
    const uint4 e = multi_list[m];
    const RowState st = load_row<GE>(rows, ld, mom, D, e.x, sub, a);
    const float4 *p = partials + (int64_t)e.y * G + sub;
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t q = 0;
    for (; q + 4 <= e.z; q += 4) {
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = p[(int64_t)(q + u) * G];
#pragma unroll
      for (int u = 0; u < 4; ++u) { tot.x += t[u].x; tot.y += t[u].y; tot.z += t[u].z; tot.w += t[u].w; }
    }
    for (; q < e.z; ++q) {
      const float4 t = p[(int64_t)q * G];
      tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
    }
    apply_row<G, GE>(rows, ld, mom, D, e.x, sub, tot, st, a);
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

// ---- field-segmented sort ---------------------------------------------------------------------------------
// The ids of a [B, F] batch whose fields own disjoint, ascending row ranges (field_off) are F independent
// sorting problems of B keys of log2(rows per field) bits, and the occurrence number b * F + f need not be
// carried: only b.  rocPRIM's radix_sort_pairs sorts all n = B * F (key, occurrence) pairs over log2(R) bits in
// 4 passes of 7 bits (138 us for 1.7 M pairs, R = 26 M); here: keys field-major [F][Bp], then per field an LSD
// radix sort of ceil(bits / 10)-bit digits - two passes for up to 2^20 - 1 rows per field.  A pass is three
// launches: per-(field, 1024-element sub-block) digit counts, their prefix over the sub-blocks, the stable
// scatter.  One WAVE owns a sub-block: its 16 rounds of 64 elements rank themselves with one ballot per digit
// bit (lanes holding the same digit), no barrier anywhere.  The last pass writes the arrays the apply kernels
// read: global row (R for a skipped id) and occurrence number, field f at [f * B, (f + 1) * B).
constexpr int kFsSub = 1024;                 // elements per sub-block (one wave: 16 rounds of 64)
constexpr int kFsRounds = kFsSub / 64;
constexpr int kFsMaxFields = 64;
constexpr int kFsMaxDigitBits = 10;
constexpr uint32_t kFsSkip = 0xFFFFFFFFu;    // every digit of it is the largest: sorts behind the field's rows

struct FsPlan {
  int npass, dbits;
  int64_t Bp, nsb;  // examples padded to whole sub-blocks, sub-blocks per field
};

// transposes a tile of 64 examples x F fields through LDS: coalesced 8-byte reads, 256-byte runs per field out
__global__ __launch_bounds__(kBlock) void fs_keys_kernel(const int64_t *__restrict__ idx,
                                                        const int64_t *__restrict__ field_off, int F, int64_t B,
                                                        int64_t Bp, int64_t R, uint32_t max_rows,
                                                        uint32_t *__restrict__ kf,
                                                        uint32_t *__restrict__ hdr) {
  __shared__ uint32_t tile[64 * kFsMaxFields];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[kHdrSegs] = hdr[kHdrMulti] = hdr[kHdrParts] = 0;  // this step's ticket counters
    hdr[kHdrPairs] = 1;  // the sorted list: (row, occurrence) pairs
  }
  const int64_t b0 = (int64_t)blockIdx.x * 64;
  const int cnt = 64 * F;
  for (int e = threadIdx.x; e < cnt; e += kBlock) {
    const int64_t o = b0 * F + e;
    const int f = e % F;
    uint32_t k = kFsSkip;
    if (o < B * F) {
      const int64_t id = idx[o];
      const int64_t lo = field_off[f], hi = f + 1 < F ? field_off[f + 1] : R;
      int64_t rows = (hi < R ? hi : R) - lo;
      if (rows > (int64_t)max_rows) rows = max_rows;
      if (id >= 0 && id < rows) k = (uint32_t)id;
    }
    tile[e] = k;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cnt; e += kBlock) {
    const int f = e >> 6, j = e & 63;
    kf[(int64_t)f * Bp + b0 + j] = tile[j * F + f];
  }
}

// Block -> (field, group of 4 sub-blocks), XCD-aware: consecutive workgroups go round-robin to the 8 XCDs, each
// with its own L2.  All blocks of a field run on ONE XCD (field f on XCD f % 8), so the 4-byte scattered writes
// of its 64 sub-blocks into the same lines of the field's segment merge in that L2 before they reach HBM
// (spread over the XCDs every L2 wrote its own partial lines: 39 us per scatter instead of 25).
__device__ __forceinline__ bool fs_where(int F, int64_t nsb, int w, int64_t *f, int64_t *sb) {
  const int64_t nbf = (nsb + kBlock / 64 - 1) / (kBlock / 64);  // blocks per field
  const int64_t x = blockIdx.x % 8, q = blockIdx.x / 8;
  *f = (q / nbf) * 8 + x;
  *sb = (q % nbf) * (kBlock / 64) + w;
  return *f < F && *sb < nsb;
}

// lanes of the wave holding the same digit as this lane (dbits ballots)
__device__ __forceinline__ uint64_t fs_match(uint32_t d, int dbits) {
  uint64_t m = ~0ull;
  for (int b = 0; b < dbits; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t bm = __ballot(bit);
    m &= bit ? bm : ~bm;
  }
  return m;
}

// digit counts of every sub-block: cnt[f][sb][d]
// (ks: 1 = plain keys, 2 = the (key, example) pairs an earlier pass wrote)
__global__ __launch_bounds__(kBlock) void fs_count_kernel(const uint32_t *__restrict__ kin, int ks, int F,
                                                         int64_t Bp, int64_t nsb, int shift, int dbits,
                                                         uint32_t *__restrict__ cnt) {
  extern __shared__ uint32_t fs_lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t f, sb;
  if (!fs_where(F, nsb, w, &f, &sb)) return;  // (wave-uniform; the kernel has no barrier)
  const int64_t gw = f * nsb + sb;
  const int nd = 1 << dbits;
  uint32_t *run = fs_lds + w * nd;
  for (int d = lane; d < nd; d += 64) run[d] = 0;
  const uint32_t *kp = kin + (f * Bp + sb * kFsSub) * ks;
  uint32_t key[kFsRounds];
#pragma unroll
  for (int r = 0; r < kFsRounds; ++r) key[r] = kp[(r * 64 + lane) * ks];
#pragma unroll
  for (int r = 0; r < kFsRounds; ++r) atomicAdd(&run[(key[r] >> shift) & (nd - 1)], 1u);  // (integer: order-free)
  uint32_t *c = cnt + gw * nd;
  for (int d = lane; d < nd; d += 64) c[d] = run[d];
}

// per field and digit: exclusive prefix of the counts over the sub-blocks (in place) and the digit's total
__global__ __launch_bounds__(64) void fs_scan_kernel(uint32_t *__restrict__ cnt, int64_t nsb, int dbits,
                                                    uint32_t *__restrict__ tot) {
  const int nd = 1 << dbits;
  const int per = nd / 64;                       // blocks per field
  const int64_t f = blockIdx.x / per;
  const int d = (blockIdx.x % per) * 64 + threadIdx.x;
  uint32_t *c = cnt + f * nsb * nd + d;
  uint32_t run = 0;
  int64_t sb = 0;
  for (; sb + 16 <= nsb; sb += 16) {
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = c[(sb + u) * nd];
#pragma unroll
    for (int u = 0; u < 16; ++u) { c[(sb + u) * nd] = run; run += v[u]; }
  }
  for (; sb < nsb; ++sb) { const uint32_t v = c[sb * nd]; c[sb * nd] = run; run += v; }
  tot[f * nd + d] = run;
}

// the stable scatter of one pass.  Between passes an element travels as ONE 8-byte (key, example) pair (one
// scattered store instead of two); FINAL: writes the (row, occurrence) arrays of the apply kernels
template <bool FINAL>
__global__ __launch_bounds__(kBlock) void fs_scatter_kernel(
    const uint32_t *__restrict__ kin, const uint2 *__restrict__ pin, uint2 *__restrict__ pout,
    const uint32_t *__restrict__ cnt,
    const uint32_t *__restrict__ tot, int F, int64_t B, int64_t Bp, int64_t nsb, int shift, int dbits,
    const int64_t *__restrict__ field_off, uint32_t R) {
  extern __shared__ uint32_t fs_lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t f, sb;
  if (!fs_where(F, nsb, w, &f, &sb)) return;
  const int64_t gw = f * nsb + sb;
  const int nd = 1 << dbits, per = nd / 64;
  uint32_t *run = fs_lds + w * nd;
  const int64_t e0 = f * Bp + sb * kFsSub;
  uint32_t key[kFsRounds], val[kFsRounds];
  if (pin) {
#pragma unroll
    for (int r = 0; r < kFsRounds; ++r) {
      const uint2 t = pin[e0 + r * 64 + lane];
      key[r] = t.x;
      val[r] = t.y;
    }
  } else {
#pragma unroll
    for (int r = 0; r < kFsRounds; ++r) {
      key[r] = kin[e0 + r * 64 + lane];
      val[r] = (uint32_t)(sb * kFsSub + r * 64 + lane);
    }
  }
  {
    // run[d] = (elements of smaller digits in the field) + (elements of digit d in earlier sub-blocks)
    const uint32_t *t = tot + f * nd + lane * per;
    const uint32_t *c = cnt + gw * nd + lane * per;
    uint32_t sum = 0;
    for (int j = 0; j < per; ++j) sum += t[j];
    uint32_t inc = sum;  // inclusive scan of the lanes' sums
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t up = __shfl_up(inc, s);
      if (lane >= s) inc += up;
    }
    uint32_t at = inc - sum;
    for (int j = 0; j < per; ++j) { run[lane * per + j] = at + c[j]; at += t[j]; }
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint32_t off = FINAL ? (uint32_t)field_off[f] : 0u;
#pragma unroll
  for (int r = 0; r < kFsRounds; ++r) {
    const uint32_t d = (key[r] >> shift) & (nd - 1);
    const uint64_t m = fs_match(d, dbits);
    const uint32_t rank = __popcll(m & lt);
    const uint32_t base = run[d];
    if (rank == 0) run[d] = base + (uint32_t)__popcll(m);  // (a wave's LDS operations complete in order)
    const int64_t pos = base + rank;
    if (FINAL) {
      if (pos < B) {  // the padding and the skipped ids sort last; positions >= B are padding only
        const uint32_t row = key[r] == kFsSkip ? R : off + key[r];
        pout[f * B + pos] = make_uint2(row < R ? row : R, val[r] * (uint32_t)F + (uint32_t)f);
      }
    } else {
      pout[f * Bp + pos] = make_uint2(key[r], val[r]);
    }
  }
}

int bits_for(uint32_t R) {  // bits needed to represent the value R itself (the "skip" key)
  int b = 1;
  while (b < 32 && (R >> b) != 0) ++b;
  return b;
}

struct WsLayout {
  size_t keys_in, vals_in, keys_out, vals_out, seg_list, multi_list, partials, sort_temp, total;
  size_t sort_bytes;
  size_t fs_k, fs_p[2], fs_cnt, fs_tot;  // the field-segmented sort's buffers
};
// the first 256 bytes of the workspace: the header words kHdr* (ticket counters, cleared by the keys kernel of
// every step; the form of the sorted list)
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
  // long runs: a run has > kLong positions, a run of several segments > kSeg: <= n / kLong + n / kSeg segments,
  // <= n / kSeg such runs with <= 2 n / kSeg segment sums of (at most) 64 lanes x 16 bytes between them
  w->seg_list = at; at += up(((size_t)n / kLong + (size_t)n / kSeg + 2) * 16);
  w->multi_list = at; at += up(((size_t)n / kSeg + 2) * 16);
  w->partials = at; at += up((2 * (size_t)n / kSeg + 2) * 1024);
  w->sort_temp = at; at += up(sort_bytes);
  w->sort_bytes = sort_bytes;
  // field-major keys, padded per field to whole sub-blocks (<= kFsMaxFields fields); (key, example) pairs,
  // twice; one count per (sub-block, digit); one total per (field, digit)
  const size_t np = (size_t)n + (size_t)kFsMaxFields * kFsSub;
  w->fs_k = at; at += up(np * 4);
  for (int i = 0; i < 2; ++i) { w->fs_p[i] = at; at += up(np * 8); }
  w->fs_cnt = at; at += up((np / kFsSub + 1) * ((size_t)4 << kFsMaxDigitBits));
  w->fs_tot = at; at += up((size_t)kFsMaxFields * ((size_t)4 << kFsMaxDigitBits));
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

// plan of the field-segmented sort, or npass = 0 when it does not apply (then: rocPRIM over all n pairs)
FsPlan fs_plan(int F, int64_t B, int64_t max_field_rows, const int64_t *row_ids) {
  FsPlan p = {0, 0, 0, 0};
  if (row_ids || max_field_rows <= 0 || max_field_rows >= ((int64_t)1 << 30) || F > kFsMaxFields) return p;
  int bits = 1;
  while (((int64_t)1 << bits) < max_field_rows + 1) ++bits;  // 2^bits - 1 ("skip") above every id
  p.npass = (bits + kFsMaxDigitBits - 1) / kFsMaxDigitBits;
  p.dbits = (bits + p.npass - 1) / p.npass;
  if (p.dbits < 6) p.dbits = 6;  // (64 digits: one per lane in the scatter's prefix)
  p.nsb = (B + kFsSub - 1) / kFsSub;
  p.Bp = p.nsb * kFsSub;
  return p;
}

// keys + stable sort of the occurrences by table row: depends on the ids only, so a caller may issue it
// (rm_sparse_optimizer_prepare) before / beside the forward+backward pass that produces the gradients
int sparse_prepare(const int64_t *idx, const int64_t *field_off, int F, const int64_t *row_ids, int64_t n,
                   int64_t R, int64_t max_field_rows, void *workspace, int64_t ws_bytes, hipStream_t st,
                   const char *fn) {
  RM_REQUIRE(n > 0 && n < (int64_t)1 << 31 && R > 0 && R < ((int64_t)1 << 32) - 1, "%s: n / R out of range", fn);
  RM_REQUIRE(row_ids || (idx && field_off && F > 0), "%s: NULL ids", fn);
  WsLayout w;
  if (ws_layout(n, &w) != RM_OK) { rm_set_error("%s: rocprim size query failed", fn); return RM_ELAUNCH; }
  RM_REQUIRE(workspace && rm_aligned16(workspace) && ws_bytes >= (int64_t)w.total,
             "%s: workspace too small (%lld < %lld bytes)", fn, (long long)ws_bytes, (long long)w.total);
  char *base = (char *)workspace;
  uint32_t *keys_in = (uint32_t *)(base + w.keys_in), *vals_in = (uint32_t *)(base + w.vals_in);
  uint32_t *keys = (uint32_t *)(base + w.keys_out), *vals = (uint32_t *)(base + w.vals_out);
  const FsPlan p = fs_plan(F, row_ids ? 0 : n / F, max_field_rows, row_ids);
  if (p.npass > 0) {
    const int64_t B = n / F;
    uint32_t *k0 = (uint32_t *)(base + w.fs_k);
    uint2 *pr[2] = {(uint2 *)(base + w.fs_p[0]), (uint2 *)(base + w.fs_p[1])};
    uint32_t *cnt = (uint32_t *)(base + w.fs_cnt), *tot = (uint32_t *)(base + w.fs_tot);
    hipLaunchKernelGGL(fs_keys_kernel, dim3((unsigned)(p.Bp / 64)), dim3(kBlock), 0, st, idx, field_off, F, B,
                       p.Bp, R, (uint32_t)max_field_rows, k0, (uint32_t *)base);
    const int nd = 1 << p.dbits;
    const int64_t nbf = (p.nsb + kBlock / 64 - 1) / (kBlock / 64);
    const dim3 wgrid((unsigned)(((F + 7) / 8) * 8 * nbf));  // fs_where: field f on XCD f % 8
    const size_t lds = (size_t)(kBlock / 64) * nd * 4;
    for (int pass = 0; pass < p.npass; ++pass) {
      const int shift = pass * p.dbits;
      const uint2 *pin = pass == 0 ? nullptr : pr[(pass - 1) & 1];
      hipLaunchKernelGGL(fs_count_kernel, wgrid, dim3(kBlock), lds, st, pin ? (const uint32_t *)pin : k0,
                         pin ? 2 : 1, F, p.Bp, p.nsb, shift, p.dbits, cnt);
      hipLaunchKernelGGL(fs_scan_kernel, dim3((unsigned)(F * (nd / 64))), dim3(64), 0, st, cnt, p.nsb, p.dbits,
                         tot);
      if (pass + 1 == p.npass)
        // (the pairs take the place of BOTH output arrays: vals_out follows keys_out)
        hipLaunchKernelGGL((fs_scatter_kernel<true>), wgrid, dim3(kBlock), lds, st, k0, pin, (uint2 *)keys, cnt,
                           tot, F, B, p.Bp, p.nsb, shift, p.dbits, field_off, (uint32_t)R);
      else
        hipLaunchKernelGGL((fs_scatter_kernel<false>), wgrid, dim3(kBlock), lds, st, k0, pin, pr[pass & 1], cnt,
                           tot, F, B, p.Bp, p.nsb, shift, p.dbits, field_off, (uint32_t)R);
    }
    RM_CHECK_LAUNCH(fn);
    return RM_OK;
  }
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

__global__ void opt_clear_word_kernel(uint32_t *w) { w[kHdrSegs] = w[kHdrMulti] = w[kHdrParts] = 0; }

int sparse_step(const int64_t *idx, const int64_t *field_off, int F, const int64_t *row_ids, int64_t n,
                GradSrc src, int D, int64_t R, int64_t max_field_rows, float *rows, int64_t ld, float *mom,
                OptArgs a, int prepared, void *workspace, int64_t ws_bytes, hipStream_t st, const char *fn) {
  RM_REQUIRE(n < (int64_t)1 << 31 && R > 0 && R < ((int64_t)1 << 32) - 1, "%s: n / R out of range", fn);
  RM_REQUIRE(D % 4 == 0 && D >= 8 && D <= 64 && ld % 4 == 0 && ld >= D + 8,
             "%s: need D %% 4 == 0, 8 <= D <= 64 and a row stride ld >= D + 8 floats, ld %% 4 == 0 (D=%d ld=%lld)",
             fn, D, (long long)ld);
  RM_REQUIRE(rows && rm_aligned16(rows) && (a.kind == 2 || (mom && rm_aligned16(mom))), "%s: rows / mom NULL or unaligned", fn);
  RM_REQUIRE(src.rows && rm_aligned16(src.rows) && src.ld % 4 == 0, "%s: gradient rows NULL / unaligned", fn);
  if (!prepared) {
    int rc = sparse_prepare(idx, field_off, F, row_ids, n, R, max_field_rows, workspace, ws_bytes, st, fn);
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
  uint4 *seg_list = (uint4 *)(base + w.seg_list), *multi_list = (uint4 *)(base + w.multi_list);
  float4 *partials = (float4 *)(base + w.partials);
  uint32_t *hdr = (uint32_t *)base;
  const int GE = D / 4;
  int G = 2;
  while (G < GE + 2) G <<= 1;
  RM_REQUIRE(G <= 64, "%s: D=%d unsupported (<= 248)", fn, D);
  const int64_t pos_per_block = (int64_t)(kBlock / G) * kPos;
  dim3 grid((unsigned)((n + pos_per_block - 1) / pos_per_block));  // kPos positions per lane group
#define RM_OPT_LAUNCH(G_, GE_)                                                                               \
  {                                                                                                          \
    hipLaunchKernelGGL((sparse_apply_kernel<G_, GE_>), grid, dim3(kBlock), 0, st, keys, vals, n, (uint32_t)R, \
                       src, D, rows, ld, mom, a, seg_list, multi_list, hdr);                                 \
    hipLaunchKernelGGL((sparse_apply_long_kernel<G_, GE_>), dim3(8192), dim3(kBlock), 0, st, keys, vals, src, \
                       D, rows, ld, mom, a, seg_list, partials, hdr);                                        \
    hipLaunchKernelGGL((sparse_apply_combine_kernel<G_, GE_>), dim3(64), dim3(kBlock), 0, st, D, rows, ld,   \
                       mom, a, multi_list, partials, hdr);                                                   \
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
                                           int64_t n, int F, int64_t R, int64_t max_field_rows,
                                           void *workspace, int64_t ws_bytes, rm_stream_t stream) {
  if (n == 0) return RM_OK;
  RM_REQUIRE(row_ids || (F > 0 && n % F == 0), "rm_sparse_optimizer_prepare: n is not a multiple of F");
  return sparse_prepare(idx, field_off, F, row_ids, n, R, max_field_rows, workspace, ws_bytes, (hipStream_t)stream,
                        "rm_sparse_optimizer_prepare");
}

extern "C" int rm_sparse_optimizer_step(const int64_t *idx, const int64_t *field_off, const float *d_rows,
                                        const float *g_bias, const float *g_lin, int64_t B, int F, int D,
                                        int64_t R, float *rows, int64_t ld, float *mom, int step, int kind,
                                        float lr, float beta1, float beta2, float eps, int reset,
                                        float l2_embedding, float l2_linear,
                                        const float *lin_field_mask, int64_t max_field_rows, int prepared,
                                        void *workspace,
                                        int64_t ws_bytes, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0, "rm_sparse_optimizer_step: bad sizes");
  RM_REQUIRE(kind >= 0 && kind <= 2 && step >= 1, "rm_sparse_optimizer_step: bad kind / step");
  if (B == 0) return RM_OK;
  RM_REQUIRE(prepared || (idx && field_off), "rm_sparse_optimizer_step: NULL argument");
  GradSrc src = {0, d_rows, D, g_bias, g_lin, lin_field_mask, F};
  return sparse_step(idx, field_off, F, nullptr, B * F, src, D, R, max_field_rows, rows, ld, mom,
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
  return sparse_step(nullptr, nullptr, 1, row_ids, n, src, D, R, 0, rows, ld, mom,
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
