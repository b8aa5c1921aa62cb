// Embedding gather + FM + linear term (forward), their backward, and the row
// scatter/gather helpers.  gfx950 only: 64-lane waves, 16-byte lane accesses.
//
// Mapping (forward): a group of G = D/4 lanes owns ONE example and walks its F
// fields; each lane keeps a float4 slice of S = sum_f E_f and a scalar partial of
// sum_f |E_f|^2, so the FM reduction over fields needs no cross-lane traffic and
// only log2(G) shuffles at the end.  A wave therefore has 64/G examples in flight
// and issues UNR independent 16-byte row gathers per lane before it consumes any
// (memory-level parallelism is what a random 64-byte-row gather lives on).
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 8;

template <int G, bool MASK>
__global__ __launch_bounds__(kBlock) void embed_fwd_kernel(
    const int64_t *__restrict__ idx, const float *__restrict__ table, int64_t table_ld,
    const int64_t *__restrict__ field_off, const float *__restrict__ bias_table, int64_t bias_ld,
    const float *__restrict__ lin_w, int64_t lin_ld, const int64_t *__restrict__ lin_off,
    const float *__restrict__ lin_w_dense, const float *__restrict__ lin_w0,
    const float *__restrict__ dense, int Dn, const float *__restrict__ mask_b,
    const float *__restrict__ mask_e, int64_t B, int F, float *__restrict__ E,
    float *__restrict__ fm_sum, float *__restrict__ fm_logit, float *__restrict__ lin_logit) {
  constexpr int EPW = 64 / G;  // examples per wave
  constexpr int D = 4 * G;
  const int lane = threadIdx.x & 63;
  const int sub = lane % G;
  const int ex = lane / G;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const bool want_fm = fm_logit != nullptr || fm_sum != nullptr;

  for (int64_t b0 = wave * EPW; b0 < B; b0 += nwaves * EPW) {
    const int64_t b = b0 + ex;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;  // clamp: keep every lane's loads in bounds
    const int64_t *ip = idx + bb * F;
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 0.f, y1 = 0.f, lin = 0.f;

    for (int f0 = 0; f0 < F; f0 += kUnroll) {
      int64_t r[kUnroll];
      float4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int f = f0 + u < F ? f0 + u : F - 1;
        r[u] = ip[f];
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int f = f0 + u < F ? f0 + u : F - 1;
        const int64_t row = field_off[f] + r[u];
        v[u] = *reinterpret_cast<const float4 *>(table + row * table_ld + sub * 4);
      }
      if (sub == 0) {
        if (bias_table != nullptr) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const int f = f0 + u;
            if (f < F) {
              float bv = bias_table[(field_off[f] + r[u]) * bias_ld];
              if (MASK && mask_b != nullptr) bv *= mask_b[bb * F + f];
              y1 += bv;
            }
          }
        }
        if (lin_w != nullptr) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const int f = f0 + u;
            if (f < F) lin += lin_w[(lin_off[f] + r[u]) * lin_ld];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int f = f0 + u;
        if (f < F) {
          if (E != nullptr && valid)
            *reinterpret_cast<float4 *>(E + (bb * F + f) * D + sub * 4) = v[u];
          float4 m = v[u];
          if (MASK && mask_e != nullptr) {
            const float4 mk =
                *reinterpret_cast<const float4 *>(mask_e + (bb * F + f) * D + sub * 4);
            m.x *= mk.x; m.y *= mk.y; m.z *= mk.z; m.w *= mk.w;
          }
          S.x += m.x; S.y += m.y; S.z += m.z; S.w += m.w;
          ss += m.x * m.x + m.y * m.y + m.z * m.z + m.w * m.w;
        }
      }
    }

    if (want_fm) {
      if (fm_sum != nullptr && valid)
        *reinterpret_cast<float4 *>(fm_sum + bb * D + sub * 4) = S;
      float part = S.x * S.x + S.y * S.y + S.z * S.z + S.w * S.w - ss;
      part = rm_group_sum<G>(part);
      if (fm_logit != nullptr && valid && sub == 0) fm_logit[bb] = y1 + 0.5f * part;
    }
    if (lin_logit != nullptr && valid && sub == 0) {
      if (dense != nullptr) {
        for (int j = 0; j < Dn; ++j) lin += dense[bb * Dn + j] * lin_w_dense[j];
      }
      if (lin_w0 != nullptr) lin += lin_w0[0];
      lin_logit[bb] = lin;
    }
  }
}

// Fused-row variant: the table row is [D embedding floats | bias | linear weight | pad]
// with a power-of-two stride LD >= D + 4, so ONE aligned line per lookup carries all
// three values (measured, tools/bench_embed.py: separate 4-byte bias / linear gathers
// cost as much as the 64-byte row gather itself - the gather is bound by line requests,
// not bytes).  G = LD/4 lanes own an example; lanes sub < GE = D/4 hold the embedding
// slices, lane sub == GE holds (bias, lin, -, -), lanes above it issue no load.
// rows in flight per lane: 13 (= half of Criteo's 26 fields) measured 10 % faster than 8; a
// field-major lane mapping with 512-byte contiguous E stores measured 1.8x SLOWER (fewer rows
// in flight, cross-group shuffles) - tools/bench_embed.py
#ifndef RM_FUSED_UNROLL
#define RM_FUSED_UNROLL 13
#endif
// NT: the table rows are loaded NON-TEMPORALLY.  A uniformly hashed id touches its row once per batch:
// letting 218 MB of such lines allocate in L2 / the Infinity Cache only evicts E, which the MLP kernels
// read right after (measured: gather 81.8 -> 76.3 us AND the rest of the DeepFM step - 10 us; with
// Zipf(1.05) ids the hot rows want the cache and the plain loads win, 54 vs 75 us: the caller chooses).
template <int G, int GE, bool MASK, bool NT = false>
__global__ __launch_bounds__(kBlock) void embed_fwd_fused_kernel(
    const int64_t *__restrict__ idx, const float *__restrict__ table,
    const int64_t *__restrict__ field_off, const float *__restrict__ lin_w_dense,
    const float *__restrict__ lin_w0, const float *__restrict__ dense, int Dn,
    const float *__restrict__ mask_b, const float *__restrict__ mask_e, int64_t B, int F,
    int want_bias, int want_lin, float *__restrict__ E, float *__restrict__ fm_sum,
    float *__restrict__ fm_logit, float *__restrict__ lin_logit) {
  constexpr int EPW = 64 / G;
  constexpr int D = 4 * GE;
  constexpr int LD = 4 * G;
  const int lane = threadIdx.x & 63;
  const int sub = lane % G;
  const int ex = lane / G;
  const bool emb = sub < GE;
  const bool side = sub == GE && (want_bias || want_lin);
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);

  for (int64_t b0 = wave * EPW; b0 < B; b0 += nwaves * EPW) {
    const int64_t b = b0 + ex;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;
    const int64_t *ip = idx + bb * F;
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 0.f, y1 = 0.f, lin = 0.f;
    for (int f0 = 0; f0 < F; f0 += RM_FUSED_UNROLL) {
      int64_t r[RM_FUSED_UNROLL];
      float4 v[RM_FUSED_UNROLL];
#pragma unroll
      for (int u = 0; u < RM_FUSED_UNROLL; ++u) {
        const int f = f0 + u < F ? f0 + u : F - 1;
        r[u] = ip[f];
      }
#pragma unroll
      for (int u = 0; u < RM_FUSED_UNROLL; ++u) {
        const int f = f0 + u < F ? f0 + u : F - 1;
        const int64_t row = field_off[f] + r[u];
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (emb || side) {
          if constexpr (NT) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v t4 = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(table + row * LD + sub * 4));
            v[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
          } else {
            v[u] = *reinterpret_cast<const float4 *>(table + row * LD + sub * 4);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < RM_FUSED_UNROLL; ++u) {
        const int f = f0 + u;
        if (f < F) {
          if (emb) {
            if (E != nullptr && valid)
              *reinterpret_cast<float4 *>(E + (bb * F + f) * D + sub * 4) = v[u];
            float4 m = v[u];
            if (MASK && mask_e != nullptr) {
              const float4 mk =
                  *reinterpret_cast<const float4 *>(mask_e + (bb * F + f) * D + sub * 4);
              m.x *= mk.x; m.y *= mk.y; m.z *= mk.z; m.w *= mk.w;
            }
            S.x += m.x; S.y += m.y; S.z += m.z; S.w += m.w;
            ss += m.x * m.x + m.y * m.y + m.z * m.z + m.w * m.w;
          } else if (side) {
            float bv = v[u].x;
            if (MASK && mask_b != nullptr) bv *= mask_b[bb * F + f];
            y1 += bv;
            lin += v[u].y;
          }
        }
      }
    }
    if (fm_sum != nullptr && valid && emb) *reinterpret_cast<float4 *>(fm_sum + bb * D + sub * 4) = S;
    float part = emb ? S.x * S.x + S.y * S.y + S.z * S.z + S.w * S.w - ss : 0.f;
    part = rm_group_sum<G>(part);
    y1 = rm_group_sum<G>(y1);
    lin = rm_group_sum<G>(lin);
    if (valid && sub == 0) {
      if (fm_logit != nullptr) fm_logit[bb] = (want_bias ? y1 : 0.f) + 0.5f * part;
      if (lin_logit != nullptr) {
        float l = want_lin ? lin : 0.f;
        if (dense != nullptr)
          for (int j = 0; j < Dn; ++j) l += dense[bb * Dn + j] * lin_w_dense[j];
        if (lin_w0 != nullptr) l += lin_w0[0];
        lin_logit[bb] = l;
      }
    }
  }
}

// Linear term alone (the layer-callable surface, recman_amd/th/layers.py; the engines get it from the
// fused gather kernel above): out[b] = sum_f w[lin_off[f] + idx[b,f]] + sum_j dense[b,j] w_dense[j] + w0.
// A thread per example (F + Dn loads): a convenience kernel, not a hot one.
__global__ __launch_bounds__(kBlock) void linear_fwd_kernel(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ lin_off, const float *__restrict__ w,
    const float *__restrict__ dense, const float *__restrict__ w_dense, const float *__restrict__ w0,
    int64_t B, int F, int Dn, float *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride) {
    float acc = w0 != nullptr ? w0[0] : 0.f;
    for (int f = 0; f < F; ++f) acc += w[lin_off[f] + idx[b * F + f]];
    for (int j = 0; j < Dn; ++j) acc += dense[b * Dn + j] * w_dense[j];
    out[b] = acc;
  }
}

// Backward: purely elementwise over the [B, F*D/4] float4 grid once S is saved.
template <bool MASK>
__global__ __launch_bounds__(kBlock) void embed_bwd_kernel(
    const float4 *__restrict__ E, const float4 *__restrict__ fm_sum,
    const float4 *__restrict__ dE_up, const float *__restrict__ g_fm,
    const float *__restrict__ mask_b, const float4 *__restrict__ mask_e, int64_t B, int F, int G,
    float4 *__restrict__ d_rows, float *__restrict__ d_bias) {
  const int64_t per_ex = (int64_t)F * G;
  const int64_t total = B * per_ex;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / per_ex;
    const int rem = (int)(t - b * per_ex);
    const int sub = rem % G;
    float4 out = dE_up != nullptr ? dE_up[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_fm != nullptr) {
      const float g = g_fm[b];
      float4 e = E[t];
      const float4 s = fm_sum[b * G + sub];
      if (MASK && mask_e != nullptr) {
        const float4 mk = mask_e[t];
        e.x *= mk.x; e.y *= mk.y; e.z *= mk.z; e.w *= mk.w;
        out.x += g * mk.x * (s.x - e.x);
        out.y += g * mk.y * (s.y - e.y);
        out.z += g * mk.z * (s.z - e.z);
        out.w += g * mk.w * (s.w - e.w);
      } else {
        out.x += g * (s.x - e.x);
        out.y += g * (s.y - e.y);
        out.z += g * (s.z - e.z);
        out.w += g * (s.w - e.w);
      }
      if (d_bias != nullptr && sub == 0) {
        const int f = rem / G;
        float gb = g;
        if (MASK && mask_b != nullptr) gb *= mask_b[b * F + f];
        d_bias[b * F + f] = gb;
      }
    }
    d_rows[t] = out;
  }
}

__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ field_off,
    const float *__restrict__ rows, const float *__restrict__ g_row, int64_t B, int F, int width,
    int64_t ld, float *__restrict__ d_table) {
  const int64_t total = B * F * width;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t occ = t / width;
    const int k = (int)(t - occ * width);
    const int f = (int)(occ % F);
    const int64_t row = field_off[f] + idx[occ];
    const float v = g_row != nullptr ? g_row[occ / F] : rows[t];
    atomicAdd(d_table + row * ld + k, v);
  }
}

// out[j] = sum_b g[b]*X[b,j] (j < P) and out0 = sum_b g[b].  Stage 1: coalesced along
// the columns, one partial row [P+1] per block; stage 2: one wave per column sums the
// partials in a fixed order -> deterministic.
__global__ __launch_bounds__(kBlock) void colsum_w_stage1(const float *__restrict__ g,
                                                          const float *__restrict__ X, int64_t B,
                                                          int P, int Pp, float *__restrict__ partial) {
  extern __shared__ float sm[];  // [kBlock]
  const int col = threadIdx.x % Pp, rsub = threadIdx.x / Pp, rper = kBlock / Pp;
  const int64_t rows_per_block = (B + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < B ? r0 + rows_per_block : B;
  for (int c0 = 0; c0 < P + 1; c0 += Pp) {
    const int j = c0 + col;
    float acc = 0.f;
    if (j <= P) {
      for (int64_t r = r0 + rsub; r < r1; r += rper) acc += j < P ? g[r] * X[r * P + j] : g[r];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rsub == 0 && j <= P) {
      float t = 0.f;
      for (int q = 0; q < rper; ++q) t += sm[q * Pp + col];
      partial[(int64_t)blockIdx.x * (P + 1) + j] = t;
    }
    __syncthreads();
  }
}

// wide inputs (P % 4 == 0, P >= 64): float4 column groups, 128 rows per block
__global__ __launch_bounds__(kBlock) void colsum_w_wide_stage1(const float *__restrict__ g,
                                                               const float *__restrict__ X, int64_t B,
                                                               int P, int64_t rows_per_block,
                                                               float *__restrict__ partial) {
  const int P4 = P / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < B ? r0 + rows_per_block : B;
  float *out = partial + (int64_t)blockIdx.x * (P + 1);
  for (int c4 = threadIdx.x; c4 < P4; c4 += kBlock) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int64_t r = r0; r < r1; ++r) {
      const float gv = g[r];
      const float4 x = *reinterpret_cast<const float4 *>(X + r * P + 4 * c4);
      acc.x += gv * x.x; acc.y += gv * x.y; acc.z += gv * x.z; acc.w += gv * x.w;
    }
    out[4 * c4 + 0] = acc.x; out[4 * c4 + 1] = acc.y; out[4 * c4 + 2] = acc.z; out[4 * c4 + 3] = acc.w;
  }
  if (threadIdx.x == kBlock - 1) {
    float a = 0.f;
    for (int64_t r = r0; r < r1; ++r) a += g[r];
    out[P] = a;
  }
}

__global__ void colsum_w_stage2(const float *__restrict__ partial, int nblk, int P,
                                float *__restrict__ out, float *__restrict__ out0) {
  const int j = blockIdx.x;  // column, P = the sum of g
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) acc += partial[(int64_t)i * (P + 1) + j];
  acc = rm_wave_sum(acc);
  if (threadIdx.x == 0) {
    if (j < P) {
      if (out != nullptr) out[j] = acc;
    } else if (out0 != nullptr) {
      out0[0] = acc;
    }
  }
}

// (U independent (row id -> row piece) chains per thread and iteration: with one, the kernel was a dependent pair of
// loads per 16 bytes - 90 us for the 1.7 M rows of a DeepFM batch, profiles/r03_sharded_step.md)
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(
    const float4 *__restrict__ table, int64_t ld4, const int64_t *__restrict__ rows, int64_t n, int G,
    float4 *__restrict__ out) {
  constexpr int U = 8;
  const int64_t total = n * G;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < total; t0 += U * stride) {
    int64_t r[U];
    int sub[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = t0 + u * stride;
      const int64_t i = (t < total ? t : total - 1) / G;
      sub[u] = (int)((t < total ? t : total - 1) - i * G);
      r[u] = rows[i];  // r < 0: an empty slot of the fixed-capacity exchange -> zero row
    }
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = table[(r[u] >= 0 ? r[u] : 0) * ld4 + sub[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = t0 + u * stride;
      if (t < total) out[t] = r[u] >= 0 ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

__global__ __launch_bounds__(kBlock) void permute_rows_kernel(
    const float4 *__restrict__ src, const int64_t *__restrict__ slot, int64_t n, int G,
    int inverse, float4 *__restrict__ dst) {
  const int64_t total = n * G;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t i = t / G;
    const int sub = (int)(t - i * G);
    if (inverse)
      dst[slot[i] * G + sub] = src[t];
    else
      dst[t] = src[slot[i] * G + sub];
  }
}

template <int G>
int launch_embed_fwd(bool mask, dim3 grid, hipStream_t st, const int64_t *idx, const float *table,
                     int64_t table_ld, const int64_t *field_off, const float *bias_table,
                     int64_t bias_ld, const float *lin_w, int64_t lin_ld, const int64_t *lin_off,
                     const float *lin_w_dense, const float *lin_w0, const float *dense, int Dn,
                     const float *mask_b, const float *mask_e, int64_t B, int F, float *E,
                     float *fm_sum, float *fm_logit, float *lin_logit) {
  if (mask)
    hipLaunchKernelGGL((embed_fwd_kernel<G, true>), grid, dim3(kBlock), 0, st, idx, table,
                       table_ld, field_off, bias_table, bias_ld, lin_w, lin_ld, lin_off,
                       lin_w_dense, lin_w0, dense, Dn, mask_b, mask_e, B, F, E, fm_sum, fm_logit,
                       lin_logit);
  else
    hipLaunchKernelGGL((embed_fwd_kernel<G, false>), grid, dim3(kBlock), 0, st, idx, table,
                       table_ld, field_off, bias_table, bias_ld, lin_w, lin_ld, lin_off,
                       lin_w_dense, lin_w0, dense, Dn, mask_b, mask_e, B, F, E, fm_sum, fm_logit,
                       lin_logit);
  return 0;
}

}  // namespace

extern "C" int rm_embed_fwd(const int64_t *idx, const float *table, int64_t table_ld,
                            const int64_t *field_off, const float *bias_table, int64_t bias_ld,
                            const float *lin_w, int64_t lin_ld, const int64_t *lin_off,
                            const float *lin_w_dense, const float *lin_w0, const float *dense,
                            int Dn, const float *mask_b, const float *mask_e, int64_t B, int F,
                            int D, float *E, float *fm_sum, float *fm_logit, float *lin_logit,
                            int flags, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0 && D > 0, "rm_embed_fwd: bad sizes B=%lld F=%d D=%d", (long long)B, F, D);
  if (B == 0) return RM_OK;
  RM_REQUIRE(idx && table && field_off, "rm_embed_fwd: idx/table/field_off must not be NULL");
  RM_REQUIRE(D % 4 == 0 && D <= 256 && (64 % (D / 4)) == 0,
             "rm_embed_fwd: D=%d unsupported (need D in {4,8,16,32,64,128,256})", D);
  RM_REQUIRE(table_ld >= D && table_ld % 4 == 0 && rm_aligned16(table),
             "rm_embed_fwd: table must be 16-byte aligned with ld %% 4 == 0 (ld=%lld)", (long long)table_ld);
  RM_REQUIRE(!E || rm_aligned16(E), "rm_embed_fwd: E must be 16-byte aligned");
  RM_REQUIRE(!fm_sum || rm_aligned16(fm_sum), "rm_embed_fwd: fm_sum must be 16-byte aligned");
  RM_REQUIRE(!mask_e || rm_aligned16(mask_e), "rm_embed_fwd: mask_e must be 16-byte aligned");
  RM_REQUIRE(!lin_w || lin_off, "rm_embed_fwd: lin_w given without lin_off");
  RM_REQUIRE(!(lin_logit && dense) || lin_w_dense, "rm_embed_fwd: dense given without lin_w_dense");
  RM_REQUIRE(Dn >= 0, "rm_embed_fwd: Dn < 0");
  const int G = D / 4;
  hipStream_t st = (hipStream_t)stream;
  const bool mask = mask_b != nullptr || mask_e != nullptr;
  // fused-row layout: bias at column D and linear weight at column D+1 of the row itself
  {
    const bool pow2 = table_ld >= D + 4 && table_ld <= 256 && (table_ld & (table_ld - 1)) == 0;
    const bool bias_in_row = !bias_table || (bias_table == table + D && bias_ld == table_ld);
    const bool lin_in_row = !lin_w || (lin_w == table + D + 1 && lin_ld == table_ld && lin_off == field_off);
    if (pow2 && bias_in_row && lin_in_row && table_ld == 2 * D && (bias_table || lin_w)) {
      const int GF = (int)table_ld / 4;
      const int epwf = 64 / GF;
      const int64_t wavesf = (B + epwf - 1) / epwf;
      dim3 gridf(rm_grid_cap((wavesf + 3) / 4, 256 * 16));
      const int wb = bias_table != nullptr, wl = lin_w != nullptr;
#define RM_EMBED_FUSED(GF_, GE_)                                                                  \
  if (mask)                                                                                       \
    hipLaunchKernelGGL((embed_fwd_fused_kernel<GF_, GE_, true>), gridf, dim3(kBlock), 0, st, idx, \
                       table, field_off, lin_w_dense, lin_w0, dense, Dn, mask_b, mask_e, B, F, wb, \
                       wl, E, fm_sum, fm_logit, lin_logit);                                       \
  else if (flags & RM_EMBED_STREAM_ROWS)                                                          \
    hipLaunchKernelGGL((embed_fwd_fused_kernel<GF_, GE_, false, true>), gridf, dim3(kBlock), 0, st, idx, \
                       table, field_off, lin_w_dense, lin_w0, dense, Dn, mask_b, mask_e, B, F, wb, \
                       wl, E, fm_sum, fm_logit, lin_logit);                                       \
  else                                                                                            \
    hipLaunchKernelGGL((embed_fwd_fused_kernel<GF_, GE_, false>), gridf, dim3(kBlock), 0, st, idx, \
                       table, field_off, lin_w_dense, lin_w0, dense, Dn, mask_b, mask_e, B, F, wb, \
                       wl, E, fm_sum, fm_logit, lin_logit);
      bool done = true;
      if (GF == 2) { RM_EMBED_FUSED(2, 1) }
      else if (GF == 4) { RM_EMBED_FUSED(4, 2) }
      else if (GF == 8) { RM_EMBED_FUSED(8, 4) }
      else if (GF == 16) { RM_EMBED_FUSED(16, 8) }
      else if (GF == 32) { RM_EMBED_FUSED(32, 16) }
      else if (GF == 64) { RM_EMBED_FUSED(64, 32) }
      else done = false;
#undef RM_EMBED_FUSED
      if (done) {
        RM_CHECK_LAUNCH("rm_embed_fwd");
        return RM_OK;
      }
    }
  }
  const int epw = 64 / G;
  const int64_t waves = (B + epw - 1) / epw;
  dim3 grid(rm_grid_cap((waves + 3) / 4, 256 * 16));
#define RM_EMBED_CASE(g)                                                                       \
  case g:                                                                                      \
    launch_embed_fwd<g>(mask, grid, st, idx, table, table_ld, field_off, bias_table, bias_ld,  \
                        lin_w, lin_ld, lin_off, lin_w_dense, lin_w0, dense, Dn, mask_b, mask_e, \
                        B, F, E, fm_sum, fm_logit, lin_logit);                                 \
    break;
  switch (G) {
    RM_EMBED_CASE(1)
    RM_EMBED_CASE(2)
    RM_EMBED_CASE(4)
    RM_EMBED_CASE(8)
    RM_EMBED_CASE(16)
    RM_EMBED_CASE(32)
    RM_EMBED_CASE(64)
    default:
      rm_set_error("rm_embed_fwd: D=%d unsupported", D);
      return RM_EUNSUPPORTED;
  }
#undef RM_EMBED_CASE
  RM_CHECK_LAUNCH("rm_embed_fwd");
  return RM_OK;
}

extern "C" int rm_linear_fwd(const int64_t *idx, const int64_t *lin_off, const float *w, const float *dense,
                             const float *w_dense, const float *w0, int64_t B, int F, int Dn, float *out,
                             rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F >= 0 && Dn >= 0 && F + Dn > 0, "rm_linear_fwd: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(out && (F == 0 || (idx && lin_off && w)) && (Dn == 0 || (dense && w_dense)),
             "rm_linear_fwd: NULL argument");
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(rm_grid_cap((B + kBlock - 1) / kBlock, 256 * 8)), dim3(kBlock), 0,
                     (hipStream_t)stream, idx, lin_off, w, dense, w_dense, w0, B, F, Dn, out);
  RM_CHECK_LAUNCH("rm_linear_fwd");
  return RM_OK;
}

extern "C" int rm_embed_bwd(const float *E, const float *fm_sum, const float *dE_up,
                            const float *g_fm, const float *mask_b, const float *mask_e, int64_t B,
                            int F, int D, float *d_rows, float *d_bias, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0 && D > 0 && D % 4 == 0, "rm_embed_bwd: bad sizes B=%lld F=%d D=%d",
             (long long)B, F, D);
  if (B == 0) return RM_OK;
  RM_REQUIRE(d_rows && rm_aligned16(d_rows), "rm_embed_bwd: d_rows NULL or unaligned");
  RM_REQUIRE(dE_up || g_fm, "rm_embed_bwd: neither dE_up nor g_fm given");
  RM_REQUIRE(!g_fm || (E && fm_sum), "rm_embed_bwd: g_fm needs E and fm_sum");
  RM_REQUIRE((!E || rm_aligned16(E)) && (!fm_sum || rm_aligned16(fm_sum)) &&
                 (!dE_up || rm_aligned16(dE_up)) && (!mask_e || rm_aligned16(mask_e)),
             "rm_embed_bwd: 16-byte alignment required");
  const int G = D / 4;
  const int64_t total = B * F * G;
  dim3 grid(rm_grid_cap((total + kBlock - 1) / kBlock, 256 * 16));
  hipStream_t st = (hipStream_t)stream;
  if (mask_b || mask_e)
    hipLaunchKernelGGL((embed_bwd_kernel<true>), grid, dim3(kBlock), 0, st, (const float4 *)E,
                       (const float4 *)fm_sum, (const float4 *)dE_up, g_fm, mask_b,
                       (const float4 *)mask_e, B, F, G, (float4 *)d_rows, d_bias);
  else
    hipLaunchKernelGGL((embed_bwd_kernel<false>), grid, dim3(kBlock), 0, st, (const float4 *)E,
                       (const float4 *)fm_sum, (const float4 *)dE_up, g_fm, mask_b,
                       (const float4 *)mask_e, B, F, G, (float4 *)d_rows, d_bias);
  RM_CHECK_LAUNCH("rm_embed_bwd");
  return RM_OK;
}

extern "C" int rm_scatter_add_rows(const int64_t *idx, const int64_t *field_off, const float *rows,
                                   const float *g_row, int64_t B, int F, int width, int64_t ld,
                                   float *d_table, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0 && width > 0 && ld >= width, "rm_scatter_add_rows: bad sizes");
  if (B == 0) return RM_OK;
  RM_REQUIRE(idx && field_off && d_table && (rows || g_row), "rm_scatter_add_rows: NULL argument");
  RM_REQUIRE(!g_row || width == 1, "rm_scatter_add_rows: g_row needs width == 1");
  const int64_t total = B * F * width;
  dim3 grid(rm_grid_cap((total + kBlock - 1) / kBlock, 256 * 16));
  hipLaunchKernelGGL(scatter_add_rows_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, idx,
                     field_off, rows, g_row, B, F, width, ld, d_table);
  RM_CHECK_LAUNCH("rm_scatter_add_rows");
  return RM_OK;
}

extern "C" int rm_linear_dense_bwd(const float *g, const float *dense, int64_t B, int Dn,
                                   float *d_w_dense, float *d_w0, float *workspace,
                                   rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && Dn >= 0 && Dn <= 1023, "rm_linear_dense_bwd: bad sizes (Dn <= 1023)");
  RM_REQUIRE(g && workspace && (Dn == 0 || dense), "rm_linear_dense_bwd: NULL argument");
  int Pp = 1;
  while (Pp < Dn + 1 && Pp < kBlock) Pp <<= 1;
  hipStream_t st = (hipStream_t)stream;
  int nblk;
  if (Dn >= 64 && Dn % 4 == 0 && rm_aligned16(dense)) {
    // each thread streams a float4 column group over the block's rows: many short blocks
    const int64_t cap = 256 * 1024 / (Dn + 1);  // partial rows that fit the 256K-float workspace
    nblk = rm_grid_cap((B + 63) / 64, (int)(cap < 2048 ? cap : 2048));
    const int64_t rpb = (B + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_w_wide_stage1, dim3(nblk), dim3(kBlock), 0, st, g, dense, B, Dn, rpb,
                       workspace);
  } else {
    nblk = rm_grid_cap((B + 255) / 256, 256);
    hipLaunchKernelGGL(colsum_w_stage1, dim3(nblk), dim3(kBlock), kBlock * sizeof(float), st, g,
                       dense, B, Dn, Pp, workspace);
  }
  hipLaunchKernelGGL(colsum_w_stage2, dim3(Dn + 1), dim3(64), 0, st, workspace, nblk, Dn,
                     d_w_dense, d_w0);
  RM_CHECK_LAUNCH("rm_linear_dense_bwd");
  return RM_OK;
}

extern "C" int rm_gather_rows(const float *table, int64_t table_ld, const int64_t *rows, int64_t n, int width,
                              float *rows_out, rm_stream_t stream) {
  RM_REQUIRE(n >= 0 && width > 0 && width % 4 == 0 && table_ld >= width && table_ld % 4 == 0,
             "rm_gather_rows: bad sizes (width and table_ld multiples of 4, table_ld >= width)");
  if (n == 0) return RM_OK;
  RM_REQUIRE(table && rows && rows_out && rm_aligned16(table) && rm_aligned16(rows_out),
             "rm_gather_rows: NULL or unaligned argument");
  const int G = width / 4;
  const int64_t total = n * G;
  dim3 grid(rm_grid_cap((total + kBlock - 1) / kBlock, 256 * 16));
  hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream,
                     (const float4 *)table, table_ld / 4, rows, n, G, (float4 *)rows_out);
  RM_CHECK_LAUNCH("rm_gather_rows");
  return RM_OK;
}

extern "C" int rm_permute_rows(const float *src, const int64_t *slot, int64_t n, int width,
                               int inverse, float *dst, rm_stream_t stream) {
  RM_REQUIRE(n >= 0 && width > 0 && width % 4 == 0, "rm_permute_rows: bad sizes");
  if (n == 0) return RM_OK;
  RM_REQUIRE(src && slot && dst && rm_aligned16(src) && rm_aligned16(dst),
             "rm_permute_rows: NULL or unaligned argument");
  const int G = width / 4;
  const int64_t total = n * G;
  dim3 grid(rm_grid_cap((total + kBlock - 1) / kBlock, 256 * 16));
  hipLaunchKernelGGL(permute_rows_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream,
                     (const float4 *)src, slot, n, G, inverse, (float4 *)dst);
  RM_CHECK_LAUNCH("rm_permute_rows");
  return RM_OK;
}
