// Skinny MLP ("DNN", recman/tf/core/layers.py:576-609) for hidden widths <= 32, fused
// on the f32-input MFMA.  hipBLASLt's kernels for N = 32 outputs ran the three GEMMs of
// the reference-default (32,32) MLP at ~800 us per step (profiles/r01_p1); the work is
// ~2 GFLOP and one pass over x, i.e. HBM-bound at ~25 us.
//
// Orientation: everything is computed TRANSPOSED, h^T[unit][example], so that a wave's
// 32x32 accumulator has the example on the lane and the 16 units u(r,h) = (r&3)+8(r>>2)+4h
// in the registers.  That accumulator is then directly the B operand of the next layer's
// MFMA (k index = unit): layers 1.. and the whole dh chain of the backward never leave
// registers.  x = [xe | xd] (never concatenated in memory) is staged per wave through a
// private LDS chunk of 32 examples x 64 k (full 256-byte lines from HBM); W0 sits in LDS.
#include "rm_common.h"
#include "mlp_internal.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kLDX = 68;   // x chunk row stride (64 + 4 pad: conflict-free b128 reads)
constexpr int kMaxNL = 3;

__device__ __forceinline__ int unit_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float actf(float v, int act) {
  if (act == RM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
__device__ __forceinline__ float actg(float o, int act) {
  if (act == RM_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return o > 0.f ? 1.f : 0.2f;
  return 1.f;
}

#ifndef RM_MLP_NT
#define RM_MLP_NT 2  // bit 1: non-temporal x loads in mlp_bwd (last use of E in the MLP).  (Bit 0, non-temporal d_rows stores, is now the CALLER's flag RM_MLP_STREAM_DROWS: -1.3 % on a bare fwd+bwd step together with bit 1, but the optimizer step that follows in training gathers d_rows again and paid +0.05 ms for it: profiles/r01_p11)
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_stream(float *p, const float4 &v) {
  f4v t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
}
__device__ __forceinline__ float4 load4_stream(const float *p) {
  const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}

struct MlpW {
  const float *W[kMaxNL];  // W[0] [K,H0], W[l] [H_{l-1}, H_l]
  const float *b[kMaxNL];
  int H[kMaxNL];
};

// One float4 of x = [xe | xd] at element k (k % 4 == 0) of example b, WITHOUT per-lane
// branches: unconditional loads from clamped addresses + selects.  (A per-lane
// `if (k + 3 < FD) float4 else scalar tail` made hipcc emit a divergent branch per load with
// s_waitcnt vmcnt(0) in between - the 8 "independent" loads of a chunk ran one after the
// other, and loads and MFMAs did not overlap at all: rocprof ablation, profiles/r01_p7_loader_ablation.md.)
__device__ __forceinline__ float4 load_x4_tail(const float *__restrict__ xe,
                                               const float *__restrict__ xd, int FD, int Dn,
                                               int64_t b, int k) {
  const bool in_e = k + 3 < FD;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FD > 0) v = *reinterpret_cast<const float4 *>(xe + b * FD + (in_e ? k : 0));
  float t[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kk = k + e - FD;
    const bool ok = kk >= 0 && kk < Dn;
    const float x = Dn > 0 ? xd[b * Dn + (ok ? kk : 0)] : 0.f;
    t[e] = ok ? x : 0.f;
  }
  return in_e ? v : make_float4(t[0], t[1], t[2], t[3]);
}

// loads the lane's 8 float4 of a 32-example x 64-k chunk of x = [xe | xd]
__device__ __forceinline__ void load_chunk(float4 (&v)[8], const float *__restrict__ xe,
                                           const float *__restrict__ xd, int FD, int Dn, int64_t B,
                                           int64_t ex0, int k0, int lane) {
  const int c4 = lane & 15;
  const int k = k0 + 4 * c4;
  if (k0 + 64 <= FD) {  // wave-uniform: the whole chunk lies in xe
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int64_t b = ex0 + (lane >> 4) + 4 * q;
      b = b < B ? b : B - 1;
      v[q] = *reinterpret_cast<const float4 *>(xe + b * FD + k);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int64_t b = ex0 + (lane >> 4) + 4 * q;
      b = b < B ? b : B - 1;
      v[q] = load_x4_tail(xe, xd, FD, Dn, b, k);
    }
  }
}

__device__ __forceinline__ void store_chunk(float *xs, const float4 (&v)[8], int lane) {
  const int c4 = lane & 15;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = (lane >> 4) + 4 * q;
    *reinterpret_cast<float4 *>(xs + row * kLDX + 4 * c4) = v[q];
  }
}

// Stages W0 [K,H0] into LDS, zero-padded to [Kp][32] logical elements.  All of a thread's
// global loads are issued before its first LDS write (a load->store loop serialises ~28 L2
// round trips per thread: 20 us of prologue at the benchmark shape).
// TRANSPOSED: dst[u*ld + k]; otherwise dst[k*ld + u].
// Split form: ALL of a thread's W0 values in registers (Kp <= 448: at most 28 with 512 threads), so
// that the caller can put the first x loads between the W0 loads and their first use - vmcnt
// retires in order, and x loads issued BEFORE a staging batch would have to land before that batch
// could be stored (the staging then waits a full HBM round trip instead of an L2 hit).
constexpr int kW0Regs = 448 * 32 / 512;
template <int NTHREADS>
__device__ __forceinline__ void stage_w0_load(float (&v)[kW0Regs], const float *__restrict__ W0, int K,
                                              int Kp, int H0, int tid) {
  static_assert(NTHREADS == 512, "kW0Regs assumes 512 threads");
#pragma unroll
  for (int q = 0; q < kW0Regs; ++q) {
    const int t = q * NTHREADS + tid;
    const int k = t >> 5, u = t & 31;
    const bool ok = t < Kp * 32 && k < K && u < H0;
    const float x = W0[ok ? (int64_t)k * H0 + u : 0];
    v[q] = ok ? x : 0.f;
  }
}
template <bool TRANSPOSED, int NTHREADS>
__device__ __forceinline__ void stage_w0_store(float *dst, int ld, const float (&v)[kW0Regs], int Kp,
                                               int tid) {
#pragma unroll
  for (int q = 0; q < kW0Regs; ++q) {
    const int t = q * NTHREADS + tid;
    const int k = t >> 5, u = t & 31;
    if (t < Kp * 32) dst[TRANSPOSED ? u * ld + k : k * ld + u] = v[q];
  }
}

template <bool TRANSPOSED, int NTHREADS>
__device__ __forceinline__ void stage_w0(float *dst, int ld, const float *__restrict__ W0, int K,
                                         int Kp, int H0, int tid) {
  constexpr int PER = 8;
  for (int base = 0; base < Kp * 32; base += NTHREADS * PER) {
    float v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int t = base + q * NTHREADS + tid;
      const int k = t >> 5, u = t & 31;
      v[q] = (t < Kp * 32 && k < K && u < H0) ? W0[(int64_t)k * H0 + u] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int t = base + q * NTHREADS + tid;
      const int k = t >> 5, u = t & 31;
      if (t < Kp * 32) dst[TRANSPOSED ? u * ld + k : k * ld + u] = v[q];
    }
  }
}

// ---------------------------------------------------------------------------
// forward: all layers + output projection in one kernel
// ---------------------------------------------------------------------------
#ifndef RM_MLP_FWD_WAVES
#define RM_MLP_FWD_WAVES 8
#endif
// TAIL: the fused training head (rm_mlp_tail, recman_hip.h) in the epilogue - final logit, prediction,
// loss term, dLoss/dlogit and the dh chain, while h_l are still in registers.
template <int NL, bool TAIL>
__global__ __launch_bounds__(64 * RM_MLP_FWD_WAVES) void mlp_fwd_kernel(
    const float *__restrict__ xe, const float *__restrict__ xd, int FD, int Dn, MlpW w,
    const float *__restrict__ w_out, const float *__restrict__ w0_out, int act, int64_t B,
    float *__restrict__ h0, float *__restrict__ h1, float *__restrict__ h2,
    float *__restrict__ logit, rm_mlp_tail tl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = FD + Dn;
  const int Kp = ((K + 63) / 64) * 64;
  const int LDW = Kp + 4;
  float *W0t = smem;                         // [32][LDW]: W0t[u][k] = W0[k][u]
  float *WA = W0t + 32 * LDW;                // [NL-1][16][2][32]: W_l[u(s,h)][c]
  float *bs = WA + (NL - 1) * 1024;          // [NL][32] biases, then [32] w_out
  float *WBt = bs + (NL + 1) * 32;           // TAIL: [NL-1][16][2][32]: W_l[c][u(s,h)] (the chain's operand)
  float *xs_all = WBt + (TAIL ? (NL - 1) * 1024 : 0);  // [waves][32][kLDX]
  constexpr int NW = RM_MLP_FWD_WAVES, NTHR = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;

  static_assert(NTHR == 512, "the staging below is written for 8 waves");
  // Program order of the prologue's loads: parameters first (L2 hits), then the first two x chunks
  // of this wave's first tile (HBM), and only then the parameters' LDS stores - the x stream starts
  // at t = 0 and the staging runs under it.  (With the x loads first, vmcnt's in-order retirement
  // made the staging wait for them; with the staging first, HBM idled during the prologue: the
  // kernel ran 15 us over its 22 us of HBM time.)
  float w0v[kW0Regs];
  stage_w0_load<NTHR>(w0v, w.W[0], K, Kp, w.H[0], tid);
  float wav[(NL > 1 ? NL - 1 : 1) * 2];
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = tid + i * NTHR;
      const int cc = t & 31, hh = (t >> 5) & 1, s = t >> 6;
      const int ku = unit_of(s, hh);
      const bool ok = ku < w.H[l - 1] && cc < w.H[l];
      const float x = w.W[l][ok ? ku * w.H[l] + cc : 0];
      wav[(l - 1) * 2 + i] = ok ? x : 0.f;
    }
  // biases / w_out: static layer index and unconditional clamped loads (a dynamic `w.b[tid >> 5]`
  // made hipcc index the kernel-argument struct through memory, three dependent round trips)
  float wbv[(NL > 1 ? NL - 1 : 1) * 2];
  if constexpr (TAIL) {
#pragma unroll
    for (int l = 1; l < NL; ++l)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = tid + i * NTHR;
        const int cc = t & 31, hh = (t >> 5) & 1, s = t >> 6;
        const int ku = unit_of(s, hh);  // unit of layer l (the chain's reduction index)
        const bool ok = cc < w.H[l - 1] && ku < w.H[l];
        const float x = w.W[l][ok ? cc * w.H[l] + ku : 0];
        wbv[(l - 1) * 2 + i] = ok ? x : 0.f;
      }
  }
  float bsv = 0.f;
  {
    const int u = tid & 31, lsel = tid >> 5;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const bool ok = lsel == l && u < w.H[l];
      const float x = w.b[l][ok ? u : 0];
      bsv = ok ? x : bsv;
    }
    const bool oko = lsel == NL && u < w.H[NL - 1];
    const float xo = w_out[oko ? u : 0];
    bsv = oko ? xo : bsv;
  }
  const int nch = Kp / 64;
  const int64_t ntiles = (B + 31) / 32;
  const int64_t tile0 = (int64_t)blockIdx.x * NW + wave;
  float4 pfa[8], pfb[8];
  {
    const int64_t e0 = (tile0 < ntiles ? tile0 : 0) * 32;  // (a wave without a tile loads tile 0: unused)
    load_chunk(pfa, xe, xd, FD, Dn, B, e0, 0, lane);
    load_chunk(pfb, xe, xd, FD, Dn, B, e0, nch > 1 ? 64 : 0, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  stage_w0_store<true, NTHR>(W0t, LDW, w0v, Kp, tid);
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      WA[(l - 1) * 1024 + tid + i * NTHR] = wav[(l - 1) * 2 + i];
      if constexpr (TAIL) WBt[(l - 1) * 1024 + tid + i * NTHR] = wbv[(l - 1) * 2 + i];
    }
  if (tid < (NL + 1) * 32) bs[tid] = bsv;
  __syncthreads();

  float *xs = xs_all + wave * 32 * kLDX;
  for (int64_t tile = tile0; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t ex0 = tile * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // two chunks in flight per wave (the 7-deep dependent load chain is what bounds this kernel);
    // the first tile's were issued in the prologue
    if (tile != tile0) {
      load_chunk(pfa, xe, xd, FD, Dn, B, ex0, 0, lane);
      if (nch > 1) load_chunk(pfb, xe, xd, FD, Dn, B, ex0, 64, lane);
    }
    for (int ch = 0; ch < nch; ch += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int cc = ch + half;
        if (cc >= nch) break;
        if (half == 0) {
          store_chunk(xs, pfa, lane);
          if (cc + 2 < nch) load_chunk(pfa, xe, xd, FD, Dn, B, ex0, (cc + 2) * 64, lane);
        } else {
          store_chunk(xs, pfb, lane);
          if (cc + 2 < nch) load_chunk(pfb, xe, xd, FD, Dn, B, ex0, (cc + 2) * 64, lane);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 a4 = *reinterpret_cast<const float4 *>(W0t + c * LDW + cc * 64 + 8 * u + 4 * h);
          const float4 b4 = *reinterpret_cast<const float4 *>(xs + c * kLDX + 8 * u + 4 * h);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
      }
    }
    const int64_t b = ex0 + c;
    const bool valid = b < B;
    // TAIL: the head's per-example inputs are requested HERE, behind the main loop (its registers are
    // free now; under it they cost the loop 3 VGPRs it did not have) and ahead of the layer
    // epilogue, which covers most of their latency
    float ta = 0.f, tb = 0.f, ty = 0.f;
    if constexpr (TAIL) {
      const int64_t bq = valid ? b : B - 1;
      ta = tl.logit_a ? tl.logit_a[bq] : 0.f;
      tb = tl.logit_b ? tl.logit_b[bq] : 0.f;
      ty = tl.y ? (float)tl.y[bq] : tl.y_f[bq];
    }
    float hv[16];
    float hl[TAIL ? NL : 1][16];  // TAIL: every layer's post-activation values, for the chain
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (l > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WA[(l - 1) * 1024 + (s * 2 + h) * 32 + c],
                                                     hv[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[r] = actf(acc[r] + bs[l * 32 + unit_of(r, h)], act);
      if constexpr (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hl[l][r] = hv[r];
      }
      float *hp = l == 0 ? h0 : (l == 1 ? h1 : h2);
      if (valid && hp != nullptr) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<float4 *>(hp + b * 32 + 8 * gq + 4 * h) =
              make_float4(hv[4 * gq], hv[4 * gq + 1], hv[4 * gq + 2], hv[4 * gq + 3]);
      }
    }
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += hv[r] * bs[NL * 32 + unit_of(r, h)];
    part += __shfl_xor(part, 32, 64);
    const float dnn = part + (w0_out ? w0_out[0] : 0.f);
    if (valid && h == 0) logit[b] = dnn;
    if constexpr (TAIL) {
      // ---- final logit, PredictionLayer, loss term, dLoss/dlogit (rm_logit_loss's arithmetic, same
      // order of the branch sum), then the dh chain; both lane halves compute the example's scalars
      float z = 0.f;
      if (tl.logit_a) z += tl.coef_a * ta;
      if (tl.logit_b) z += tl.coef_b * tb;
      z += tl.coef_mlp * dnn;
      const float t = ty;
      float p, dz;
      const float lt = rm_loss_point(z, t, tl.task, &p, &dz);
      float gb = dz * (1.0f / (float)B);
      gb *= tl.grad_scale;
      if (valid && h == 0) {
        if (tl.logit) tl.logit[b] = z;
        if (tl.pred) tl.pred[b] = p;
        tl.dlogit[b] = gb;
      }
      const float ls = rm_wave_sum((valid && h == 0) ? lt : 0.f);
      if (lane == 0) tl.loss_partial[tile] = ls;
      float dh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dh[r] = gb * bs[NL * 32 + unit_of(r, h)] * actg(hl[NL - 1][r], act);
#pragma unroll
      for (int l = NL - 1; l >= 1; --l) {
        if (valid) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4 *>(tl.dh[l] + b * 32 + 8 * gq + 4 * h) =
                make_float4(dh[4 * gq], dh[4 * gq + 1], dh[4 * gq + 2], dh[4 * gq + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WBt[(l - 1) * 1024 + (s * 2 + h) * 32 + c], dh[s],
                                                     acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = acc[r] * actg(hl[l - 1][r], act);
      }
      if (valid) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<float4 *>(tl.dh[0] + b * 32 + 8 * gq + 4 * h) =
              make_float4(dh[4 * gq], dh[4 * gq + 1], dh[4 * gq + 2], dh[4 * gq + 3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// forward of the whole DeepFM-style front: gather + FM + linear + skinny MLP (+ training head) in ONE kernel
// (rm_embed_mlp_fwd).  rm_embed_fwd writes E [B, F*D] and rm_mlp_fwd reads it back 40 us later through a
// 7-deep chain of dependent loads per wave (37 us for 109 MB); here a wave gathers the rows of its 32-example
// tile four fields (= one 64-k chunk of x) at a time, stores them to E (the backward needs it), adds them
// into the FM sums and drops them into its private LDS chunk - the layer-0 MFMAs of that chunk run while
// the other waves' row loads are in flight.  Fixed layout: D = 16, table rows of 32 floats
// [16 emb | bias | lin | ...] (8 lanes per row: 4 embedding slices + 1 side lane), Dn <= 16.
// Lane roles: gather phase (ex_l = lane >> 3, sub = lane & 7) owns examples 8 eg + ex_l, eg = 0..3;
// MFMA phase (c = lane & 31, h = lane >> 5) as in mlp_fwd_kernel.  Prologue and tile epilogue are
// mlp_fwd_kernel's (kept textually parallel: a change there belongs here too).
// ---------------------------------------------------------------------------
// diagnostic build (-DRM_EMF_STAMP): lane 0 of every wave of embed_mlp_fwd_kernel records s_memrealtime at start /
// after the prologue / after the chunk loop of its (first) tile / at the end, and the ticks it spent waiting for row
// loads at the top of each half chunk; rm_debug_emf_stamps reads them (tools/probe/emf_stamps.py).
#ifdef RM_EMF_STAMP
__device__ unsigned long long rm_emf_stamp_buf[8 * 8 * 256];
#endif
#ifndef RM_EMF_ABL
#define RM_EMF_ABL 0  // ablation builds (wrong results): 1 no layer-0 MFMAs, 2 no E stores, 4 no x writes to LDS
#endif
struct EmbFront {
  const int64_t *idx;        // [B, F]
  const float *table;        // fused rows [16 emb | bias | lin | ...], stride table_ld floats (32, or 20 for exchanged rows)
  int table_ld;
  const int64_t *field_off;  // [F]
  const float *lin_w_dense, *lin_w0, *dense;  // linear term's dense part (NULL: none); dense = xd
  int F, want_bias, want_lin;
  float *E, *fm_sum, *fm_logit, *lin_logit;   // fm_sum / fm_logit / lin_logit may be NULL
};

template <int NL, bool TAIL, bool NT>
__global__ __launch_bounds__(64 * RM_MLP_FWD_WAVES) void embed_mlp_fwd_kernel(
    EmbFront ef, const float *__restrict__ xd, int Dn, MlpW w, const float *__restrict__ w_out,
    const float *__restrict__ w0_out, int act, int64_t B, float *__restrict__ h0, float *__restrict__ h1,
    float *__restrict__ h2, float *__restrict__ logit, rm_mlp_tail tl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int F = ef.F, FD = 16 * F;
  const int K = FD + Dn;
  const int Kp = ((K + 63) / 64) * 64;
  const int LDW = Kp + 4;
  float *W0t = smem;                         // [32][LDW]: W0t[u][k] = W0[k][u]
  float *WA = W0t + 32 * LDW;                // [NL-1][16][2][32]: W_l[u(s,h)][c]
  float *bs = WA + (NL - 1) * 1024;          // [NL][32] biases, then [32] w_out
  float *WBt = bs + (NL + 1) * 32;           // TAIL: [NL-1][16][2][32]: W_l[c][u(s,h)] (the chain's operand)
  float *xs_all = WBt + (TAIL ? (NL - 1) * 1024 : 0);  // [waves][32][kLDX]
  constexpr int NW = RM_MLP_FWD_WAVES, NTHR = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int ex_l = lane >> 3, sub = lane & 7;
  const bool emb = sub < 4, side = sub == 4 && (ef.want_bias || ef.want_lin);
  static_assert(NTHR == 512, "the staging below is written for 8 waves");
#ifdef RM_EMF_STAMP
  const unsigned long long et0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long et1 = 0, et2 = 0, et_wait = 0;
#endif

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tile0 = (int64_t)blockIdx.x * NW + wave;
  // ---- prologue: parameter loads (L2 hits), then this wave's first ids, then the parameters' LDS stores
  float w0v[kW0Regs];
  stage_w0_load<NTHR>(w0v, w.W[0], K, Kp, w.H[0], tid);
  float wav[(NL > 1 ? NL - 1 : 1) * 2];
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = tid + i * NTHR;
      const int cc = t & 31, hh = (t >> 5) & 1, s = t >> 6;
      const int ku = unit_of(s, hh);
      const bool ok = ku < w.H[l - 1] && cc < w.H[l];
      const float x = w.W[l][ok ? ku * w.H[l] + cc : 0];
      wav[(l - 1) * 2 + i] = ok ? x : 0.f;
    }
  float wbv[(NL > 1 ? NL - 1 : 1) * 2];
  if constexpr (TAIL) {
#pragma unroll
    for (int l = 1; l < NL; ++l)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = tid + i * NTHR;
        const int cc = t & 31, hh = (t >> 5) & 1, s = t >> 6;
        const int ku = unit_of(s, hh);
        const bool ok = cc < w.H[l - 1] && ku < w.H[l];
        const float x = w.W[l][ok ? cc * w.H[l] + ku : 0];
        wbv[(l - 1) * 2 + i] = ok ? x : 0.f;
      }
  }
  float bsv = 0.f;
  {
    const int u = tid & 31, lsel = tid >> 5;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const bool ok = lsel == l && u < w.H[l];
      const float x = w.b[l][ok ? u : 0];
      bsv = ok ? x : bsv;
    }
    const bool oko = lsel == NL && u < w.H[NL - 1];
    const float xo = w_out[oko ? u : 0];
    bsv = oko ? xo : bsv;
  }
  // The chunk loop works on HALF chunks of 32 k = 2 fields (8 row loads per lane: 4 example groups x 2
  // fields) with two register buffers: the rows of half chunk i + 1 are requested before half chunk i is
  // consumed.  (A store holds its data registers until it is acknowledged, and vmcnt retires loads and
  // stores in one order: re-loading into the registers the E stores have just read - a one-buffer pipeline
  // - made every load wait for those stores: 96 -> 111 us.  With two buffers the stores of half chunk i
  // have a whole iteration to drain before their registers are loaded again.)
  // row numbers (field offset + id) of one half chunk; fields past F repeat F - 1
  auto load_rows = [&](int64_t ex0, int hc, int64_t (&r)[8]) {
#pragma unroll
    for (int eg = 0; eg < 4; ++eg) {
      int64_t b = ex0 + 8 * eg + ex_l;
      b = b < B ? b : B - 1;
#pragma unroll
      for (int fl = 0; fl < 2; ++fl) {
        int f = 2 * hc + fl;
        f = f < F ? f : F - 1;
        r[eg * 2 + fl] = ef.idx[b * F + f] + ef.field_off[f];
      }
    }
  };
  const int nhc = Kp / 32;
  int64_t rows[8];
  load_rows((tile0 < ntiles ? tile0 : 0) * 32, 0, rows);
  __builtin_amdgcn_sched_barrier(0);
  stage_w0_store<true, NTHR>(W0t, LDW, w0v, Kp, tid);
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      WA[(l - 1) * 1024 + tid + i * NTHR] = wav[(l - 1) * 2 + i];
      if constexpr (TAIL) WBt[(l - 1) * 1024 + tid + i * NTHR] = wbv[(l - 1) * 2 + i];
    }
  if (tid < (NL + 1) * 32) bs[tid] = bsv;
  __syncthreads();

#ifdef RM_EMF_STAMP
  et1 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int kLDH = 36;  // row stride of the half-chunk x tile (32 + 4: conflict-free b128 reads)
  float *xs = xs_all + wave * 32 * kLDX;
  // linear term: this lane's two dense weights (columns sub and sub + 8; 0 past Dn)
  float wd0 = 0.f, wd1 = 0.f;
  if (ef.lin_w_dense != nullptr) {
    const float a0 = ef.lin_w_dense[sub < Dn ? sub : 0], a1 = ef.lin_w_dense[sub + 8 < Dn ? sub + 8 : 0];
    wd0 = sub < Dn ? a0 : 0.f;
    wd1 = sub + 8 < Dn ? a1 : 0.f;
  }
  for (int64_t tile = tile0; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t ex0 = tile * 32;
    if (tile != tile0) load_rows(ex0, 0, rows);
    // the head's label of example c (MFMA-phase lane role), requested here: behind the chunk loop it was an
    // exposed round trip in every wave's epilogue
    float ty_pf = 0.f;
    if constexpr (TAIL) {
      int64_t bq = ex0 + c;
      bq = bq < B ? bq : B - 1;
      ty_pf = tl.y ? (float)tl.y[bq] : tl.y_f[bq];
    }
    // dense inputs of this lane's examples: columns FD + sub + 8 i (i = 0, 1) of example 8 eg + ex_l
    float dv[4][2];
#pragma unroll
    for (int eg = 0; eg < 4; ++eg) {
      int64_t b = ex0 + 8 * eg + ex_l;
      b = b < B ? b : B - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = sub + 8 * i;
        const float x = Dn > 0 ? xd[b * Dn + (j < Dn ? j : 0)] : 0.f;
        dv[eg][i] = j < Dn ? x : 0.f;
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 S[4];
    float ss[4], y1[4], lin[4];
#pragma unroll
    for (int eg = 0; eg < 4; ++eg) {
      S[eg] = make_float4(0.f, 0.f, 0.f, 0.f);
      ss[eg] = 0.f; y1[eg] = 0.f; lin[eg] = 0.f;
    }
    const int64_t rows_t = B - ex0 < 32 ? B - ex0 : 32;
    const rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(ef.E + ex0 * F * 16, 0, (int)(rows_t * F * 64), 0x00020000);
    auto load_v = [&](float4 (&v)[8]) {
      if (emb || side) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float *p = ef.table + rows[q] * ef.table_ld + sub * 4;
          v[q] = NT ? load4_stream(p) : *reinterpret_cast<const float4 *>(p);
        }
      }
    };
    // one half chunk: request the next one's rows into `vn`, then consume `vc`
    auto half = [&](int hc, float4 (&vc)[8], float4 (&vn)[8]) {
      if (hc + 1 < nhc) {  // wave-uniform
        load_v(vn);
        load_rows(ex0, hc + 2 < nhc ? hc + 2 : hc + 1, rows);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef RM_EMF_STAMP
      {
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_nop 0" ::"v"(vc[0].x), "v"(vc[7].w));  // wait for this half chunk's rows
        et_wait += __builtin_amdgcn_s_memrealtime() - w0;
      }
#endif
      // FM / linear sums, E stores, x values to LDS - without per-element lane predicates (`if (f < F) /
      // if (emb) / if (valid)` per value compiled to three exec-mask branches per row): every lane adds every
      // value (only the embedding lanes' S / ss and the side lane's y1 / lin are ever read), E is stored
      // through a tile descriptor with out-of-range offsets for the lanes that must not store (examples past
      // B fall outside num_records), the LDS writes sit in ONE `if (emb)` region
      const int nf = F - 2 * hc;  // fields of this half chunk that exist (wave-uniform)
#pragma unroll
      for (int eg = 0; eg < 4; ++eg)
#pragma unroll
        for (int fl = 0; fl < 2; ++fl)
          if (fl < nf) {
            const float4 t = vc[eg * 2 + fl];
            S[eg].x += t.x; S[eg].y += t.y; S[eg].z += t.z; S[eg].w += t.w;
            ss[eg] += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
            y1[eg] += t.x;
            lin[eg] += t.y;
          }
#ifndef RM_EMF_ESTORE
#define RM_EMF_ESTORE 0  // 0: one 64-byte store per (example, field); 1: two fields per store = whole 128-byte lines
#endif
#ifndef RM_EMF_EAUX
#define RM_EMF_EAUX 0    // aux bits of the E stores (2 = nt)
#endif
#pragma unroll
      for (int eg = 0; eg < 4; ++eg) {
        if (RM_EMF_ESTORE == 1 && nf >= 2) {
          // lanes 4..7 of an example's group take field 2 hc + 1's four slices from lanes 0..3 (DPP row_shr:4
          // into banks 1 and 3): the group's 8 lanes x 16 bytes are the 128 contiguous bytes of both fields
          const float4 t0 = vc[eg * 2], t1 = vc[eg * 2 + 1];
          u32x4v tv;
          tv.x = __builtin_amdgcn_update_dpp(__builtin_bit_cast(unsigned, t0.x), __builtin_bit_cast(unsigned, t1.x), 0x114, 0xf, 0xa, false);
          tv.y = __builtin_amdgcn_update_dpp(__builtin_bit_cast(unsigned, t0.y), __builtin_bit_cast(unsigned, t1.y), 0x114, 0xf, 0xa, false);
          tv.z = __builtin_amdgcn_update_dpp(__builtin_bit_cast(unsigned, t0.z), __builtin_bit_cast(unsigned, t1.z), 0x114, 0xf, 0xa, false);
          tv.w = __builtin_amdgcn_update_dpp(__builtin_bit_cast(unsigned, t0.w), __builtin_bit_cast(unsigned, t1.w), 0x114, 0xf, 0xa, false);
          const int o = ((8 * eg + ex_l) * F + 2 * hc) * 64 + sub * 16;
          if (!(RM_EMF_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(tv, re, o, 0, RM_EMF_EAUX);
        } else {
          const int o = emb ? ((8 * eg + ex_l) * F + 2 * hc) * 64 + sub * 16 : 0x7ffffff0;
#pragma unroll
          for (int fl = 0; fl < 2; ++fl)
            if (fl < nf) {
              const float4 t = vc[eg * 2 + fl];
              f4v tv;
              tv.x = t.x; tv.y = t.y; tv.z = t.z; tv.w = t.w;
              if (!(RM_EMF_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, tv), re, o + 64 * fl, 0, RM_EMF_EAUX);
            }
        }
      }
      if (emb && !(RM_EMF_ABL & 4)) {
#pragma unroll
        for (int eg = 0; eg < 4; ++eg)
#pragma unroll
          for (int fl = 0; fl < 2; ++fl)
            *reinterpret_cast<float4 *>(xs + (8 * eg + ex_l) * kLDH + fl * 16 + sub * 4) =
                fl < nf ? vc[eg * 2 + fl] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (32 * hc + 32 > FD) {  // wave-uniform: this half chunk holds dense columns / zero padding
#pragma unroll
        for (int eg = 0; eg < 4; ++eg)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * hc + sub + 8 * q;  // this lane's columns of the half chunk
            const int j = col - FD;                 // dense index; FD % 8 == 0, so j & 7 == sub: dv[eg][j >> 3]
            const float x = (j >= 0 && j < 8) ? dv[eg][0] : ((j >= 8 && j < 16) ? dv[eg][1] : 0.f);
            if (col >= FD) xs[(8 * eg + ex_l) * kLDH + sub + 8 * q] = x;
          }
      }
#pragma unroll
      for (int u = 0; u < ((RM_EMF_ABL & 1) ? 0 : 4); ++u) {
        const float4 a4 = *reinterpret_cast<const float4 *>(W0t + c * LDW + hc * 32 + 8 * u + 4 * h);
        const float4 b4 = *reinterpret_cast<const float4 *>(xs + c * kLDH + 8 * u + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
    };
    float4 vA[8], vB[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) vA[q] = vB[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_v(vA);
    load_rows(ex0, nhc > 1 ? 1 : 0, rows);
    for (int hc = 0; hc < nhc; hc += 2) {  // nhc is even (Kp % 64 == 0)
      half(hc, vA, vB);
      half(hc + 1, vB, vA);
    }
#ifdef RM_EMF_STAMP
    if (tile == tile0) et2 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- FM second order + bias sum, linear term (rm_embed_fwd's arithmetic and order)
#pragma unroll
    for (int eg = 0; eg < 4; ++eg) {
      const int64_t b = ex0 + 8 * eg + ex_l;
      const bool valid = b < B;
      if (ef.fm_sum != nullptr && valid && emb) *reinterpret_cast<float4 *>(ef.fm_sum + b * 16 + sub * 4) = S[eg];
      float part = emb ? S[eg].x * S[eg].x + S[eg].y * S[eg].y + S[eg].z * S[eg].z + S[eg].w * S[eg].w - ss[eg] : 0.f;
      part = rm_group_sum<8>(part);
      const float ys = rm_group_sum<8>(side ? y1[eg] : 0.f);
      float ls = rm_group_sum<8>(side ? lin[eg] : 0.f);
      const float fmv = (ef.want_bias ? ys : 0.f) + 0.5f * part;
      float lv = ef.want_lin ? ls : 0.f;
      // the linear term's dense part from the dense values this lane already holds (columns sub and sub + 8)
      // and a group sum.  (rm_embed_fwd walks j = 0 .. Dn - 1 serially with two loads per step; done here
      // that way - for bit-identical sums - it cost ~8 of the 14 us every wave spends behind its chunk loop,
      // tools/probe/emf_stamps.py.  The order of the adds differs: lin_logit agrees to 1e-6, not bitwise.)
      if (ef.lin_w_dense != nullptr) lv += rm_group_sum<8>(dv[eg][0] * wd0 + dv[eg][1] * wd1);
      if (ef.lin_w0 != nullptr) lv += ef.lin_w0[0];
      if (valid && sub == 0) {
        if (ef.fm_logit != nullptr) ef.fm_logit[b] = fmv;
        if (ef.lin_logit != nullptr) ef.lin_logit[b] = lv;
      }
      if (sub == 0) {  // for the head: example 8 eg + ex_l -> xs[2 e], xs[2 e + 1] (the chunk loop is done with xs)
        xs[2 * (8 * eg + ex_l)] = lv;
        xs[2 * (8 * eg + ex_l) + 1] = fmv;
      }
    }
    const int64_t b = ex0 + c;
    const bool valid = b < B;
    float ta = 0.f, tb = 0.f, ty = 0.f;
    if constexpr (TAIL) {
      // logit_a / logit_b name the buffers this kernel has just written (lin_logit / fm_logit, in either
      // order) or are NULL: take the values from LDS instead of reading them back
      ta = tl.logit_a == nullptr ? 0.f : (tl.logit_a == ef.lin_logit ? xs[2 * c] : xs[2 * c + 1]);
      tb = tl.logit_b == nullptr ? 0.f : (tl.logit_b == ef.lin_logit ? xs[2 * c] : xs[2 * c + 1]);
      ty = ty_pf;
    }
    float hv[16];
    float hl[TAIL ? NL : 1][16];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (l > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WA[(l - 1) * 1024 + (s * 2 + h) * 32 + c],
                                                     hv[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[r] = actf(acc[r] + bs[l * 32 + unit_of(r, h)], act);
      if constexpr (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hl[l][r] = hv[r];
      }
      float *hp = l == 0 ? h0 : (l == 1 ? h1 : h2);
      if (valid && hp != nullptr) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<float4 *>(hp + b * 32 + 8 * gq + 4 * h) =
              make_float4(hv[4 * gq], hv[4 * gq + 1], hv[4 * gq + 2], hv[4 * gq + 3]);
      }
    }
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += hv[r] * bs[NL * 32 + unit_of(r, h)];
    part += __shfl_xor(part, 32, 64);
    const float dnn = part + (w0_out ? w0_out[0] : 0.f);
    if (valid && h == 0) logit[b] = dnn;
    if constexpr (TAIL) {
      float z = 0.f;
      if (tl.logit_a) z += tl.coef_a * ta;
      if (tl.logit_b) z += tl.coef_b * tb;
      z += tl.coef_mlp * dnn;
      const float t = ty;
      float p, dz;
      const float lt = rm_loss_point(z, t, tl.task, &p, &dz);
      float gb = dz * (1.0f / (float)B);
      gb *= tl.grad_scale;
      if (valid && h == 0) {
        if (tl.logit) tl.logit[b] = z;
        if (tl.pred) tl.pred[b] = p;
        tl.dlogit[b] = gb;
      }
      const float ls = rm_wave_sum((valid && h == 0) ? lt : 0.f);
      if (lane == 0) tl.loss_partial[tile] = ls;
      float dh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dh[r] = gb * bs[NL * 32 + unit_of(r, h)] * actg(hl[NL - 1][r], act);
#pragma unroll
      for (int l = NL - 1; l >= 1; --l) {
        if (valid) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4 *>(tl.dh[l] + b * 32 + 8 * gq + 4 * h) =
                make_float4(dh[4 * gq], dh[4 * gq + 1], dh[4 * gq + 2], dh[4 * gq + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WBt[(l - 1) * 1024 + (s * 2 + h) * 32 + c], dh[s],
                                                     acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = acc[r] * actg(hl[l - 1][r], act);
      }
      if (valid) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<float4 *>(tl.dh[0] + b * 32 + 8 * gq + 4 * h) =
              make_float4(dh[4 * gq], dh[4 * gq + 1], dh[4 * gq + 2], dh[4 * gq + 3]);
      }
    }
  }
#ifdef RM_EMF_STAMP
  if (lane == 0 && blockIdx.x < 256) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long *o = rm_emf_stamp_buf + 8 * (blockIdx.x * 8 + wave);
    o[0] = et1 - et0; o[1] = et2 - et1; o[2] = __builtin_amdgcn_s_memrealtime() - et2; o[3] = et_wait; o[4] = et0;
  }
#endif
}
#ifdef RM_EMF_STAMP
}  // namespace
extern "C" int rm_debug_emf_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_emf_stamp_buf), sizeof(unsigned long long) * n);
}
namespace {
#endif

// ---------------------------------------------------------------------------
// backward: dh chain in registers, dX (+ FM term) -> d_rows, dW0 accumulated on the MFMA.
// The 8 waves of a block share one 32-example tile and split its k-tiles (wave w owns
// k-tiles w and w+8): each wave loads only its own 32 x 32 slices of x (private LDS, no
// block barrier in the loop), keeps 2 dW0 accumulators (32 registers instead of 224) and
// owns disjoint rows of the block's dW0 partial (no end-of-kernel reduction).  The cheap
// dh chain is recomputed per wave.
// ---------------------------------------------------------------------------
constexpr int kLDT = 36;  // k-tile slice row stride in LDS (32 + 4 pad)
constexpr int kWT = 2;    // k-tiles per wave: Kp <= 448 -> 14 tiles over the block's 8 waves

// A k-tile that lies entirely inside xe (wave-uniform test; every tile but the last for the
// Criteo shape) is four unconditional float4 loads issued back to back; a tile entirely behind xe
// takes raw buffer loads.  Only a tile that STRADDLES the xe / xd boundary (FD % 32 == 16) takes the
// per-lane-branch form below - hipcc serialises those loads
// (branch + s_waitcnt vmcnt(0) per load), but it keeps their register footprint small, and the
// kernel is register-bound (a branch-free tail cost it spills: profiles/r01_p7_loader_ablation.md).
__device__ __forceinline__ void load_ktile(float4 (&v)[4], const float *__restrict__ xe,
                                           const float *__restrict__ xd, int FD, int Dn, int64_t B,
                                           int64_t ex0, int kb, int lane) {
  const int c4 = lane & 7;
  const int k = kb + 4 * c4;
  if (kb + 32 <= FD) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int64_t b = ex0 + (lane >> 3) + 8 * q;
      b = b < B ? b : B - 1;
#if RM_MLP_NT & 2
      v[q] = load4_stream(xe + b * FD + k);
#else
      v[q] = *reinterpret_cast<const float4 *>(xe + b * FD + k);
#endif
    }
    return;
  }
  if (kb >= FD) {
    // wave-uniform: the tile holds only dense columns and padding (the last k-tile of the Criteo shape:
    // FD = 416 = 13 x 32).  Raw buffer loads on a tile descriptor: columns past Dn get an out-of-range
    // offset and rows past B fall outside num_records - both read as 0, every load is unconditional.  (This
    // tile used to take the per-lane-branch form below: 16 serialised loads per example tile on ONE wave of
    // the block - in-kernel timestamps, tools/probe/bwd_stamps.py: that wave finished at 68 us, the others at
    // 46-61, and the block waits for it: 73 -> 62 us.)
    const int64_t rows_t = B - ex0 < 32 ? B - ex0 : 32;
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Dn > 0 ? xd + ex0 * Dn : xe), 0,
                                                        Dn > 0 ? (int)(rows_t * Dn * 4) : 0, 0x00020000);
    // (four per-lane offsets, one per element of the float4; the row term 8 q Dn is a scalar offset: the
    // 16 loads must not cost 16 address registers - the kernel sits at 239 VGPRs)
    int vo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e - FD;
      vo[e] = kk < Dn ? ((lane >> 3) * Dn + kk) * 4 : 0x7ffffff0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int so = __builtin_amdgcn_readfirstlane(8 * q * Dn * 4);
      v[q] = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[0], so, 0)),
                         __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[1], so, 0)),
                         __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[2], so, 0)),
                         __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[3], so, 0)));
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (lane >> 3) + 8 * q;
    int64_t b = ex0 + row;
    b = b < B ? b : B - 1;
    if (k + 3 < FD) {
      v[q] = *reinterpret_cast<const float4 *>(xe + b * FD + k);
    } else {
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k + e;
        t[e] = (kk >= FD && kk < FD + Dn) ? xd[b * Dn + (kk - FD)] : 0.f;
      }
      v[q] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
}

__device__ __forceinline__ void store_ktile(float *xs, const float4 (&v)[4], int lane) {
  const int c4 = lane & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (lane >> 3) + 8 * q;
    *reinterpret_cast<float4 *>(xs + row * kLDT + 4 * c4) = v[q];
  }
}

// dh chain of one 32-example tile per wave: dh_last = g w_out * act'(h_last), dh_{l-1} =
// (dh_l W_l^T) * act'(h_{l-1}); everything transposed as in mlp_fwd_kernel (lane (c, h) holds 16 units
// of example c), so the layers chain through registers.  Writes dh_l [B,32] for every layer: the
// inputs of mlp_bwd_kernel (dh0) and mlp_small_grads_mfma (all of them).
template <int NL>
__global__ __launch_bounds__(256) void mlp_dh_chain_kernel(
    MlpW w, const float *__restrict__ w_out, int act, int64_t B, const float *__restrict__ g,
    const float *__restrict__ h0, const float *__restrict__ h1, const float *__restrict__ h2,
    float *__restrict__ dh0, float *__restrict__ dh1, float *__restrict__ dh2) {
  __shared__ float WB[(NL > 1 ? NL - 1 : 1) * 1024];  // [NL-1][16][2][32]: W_l[c][u(s,h)]
  __shared__ float wo[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int64_t ntiles = (B + 31) / 32;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  // this wave's inputs first (HBM / L2), the parameters under them
  const float *hptr[3] = {h0, h1, h2};
  float4 hc[NL][4];
  int64_t b = (tile < ntiles ? tile : 0) * 32 + c;
  const bool valid = tile < ntiles && b < B;
  b = b < B ? b : B - 1;
  const float gb = g[b];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      hc[l][gq] = *reinterpret_cast<const float4 *>(hptr[l] + b * 32 + 8 * gq + 4 * h);
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = tid + i * 256;
      const int cc = t & 31, hh = (t >> 5) & 1, s = t >> 6;
      const int ku = unit_of(s, hh);  // unit of layer l (the reduction index)
      const bool ok = cc < w.H[l - 1] && ku < w.H[l];
      const float x = w.W[l][ok ? cc * w.H[l] + ku : 0];
      WB[(l - 1) * 1024 + t] = ok ? x : 0.f;
    }
  if (tid < 32) wo[tid] = tid < w.H[NL - 1] ? w_out[tid] : 0.f;
  __syncthreads();

  float dh[16];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const float4 hv = hc[NL - 1][gq];
    dh[4 * gq + 0] = gb * wo[8 * gq + 4 * h + 0] * actg(hv.x, act);
    dh[4 * gq + 1] = gb * wo[8 * gq + 4 * h + 1] * actg(hv.y, act);
    dh[4 * gq + 2] = gb * wo[8 * gq + 4 * h + 2] * actg(hv.z, act);
    dh[4 * gq + 3] = gb * wo[8 * gq + 4 * h + 3] * actg(hv.w, act);
  }
#pragma unroll
  for (int l = NL - 1; l >= 1; --l) {
    float *dout = l == 1 ? dh1 : dh2;
    if (valid) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4 *>(dout + b * 32 + 8 * gq + 4 * h) =
            make_float4(dh[4 * gq], dh[4 * gq + 1], dh[4 * gq + 2], dh[4 * gq + 3]);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WB[(l - 1) * 1024 + (s * 2 + h) * 32 + c], dh[s], acc,
                                                 0, 0, 0);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 hv = hc[l - 1][gq];
      dh[4 * gq + 0] = acc[4 * gq + 0] * actg(hv.x, act);
      dh[4 * gq + 1] = acc[4 * gq + 1] * actg(hv.y, act);
      dh[4 * gq + 2] = acc[4 * gq + 2] * actg(hv.z, act);
      dh[4 * gq + 3] = acc[4 * gq + 3] * actg(hv.w, act);
    }
  }
  if (valid) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      *reinterpret_cast<float4 *>(dh0 + b * 32 + 8 * gq + 4 * h) =
          make_float4(dh[4 * gq], dh[4 * gq + 1], dh[4 * gq + 2], dh[4 * gq + 3]);
  }
}

// diagnostic build (-DRM_BWD_STAMP): thread 0 of wave 0 of every mlp_bwd block records s_memrealtime (100 MHz) at
// start / after the prologue / after its tile loop / at the end, plus the ticks it spent in the per-tile staging
// (dT, g*S) and in its k-tile loops; rm_debug_bwd_stamps reads them (tools/probe/bwd_stamps.py).
#ifdef RM_BWD_STAMP
__device__ unsigned long long rm_bwd_stamp_buf[8 * 8 * 256];
#define RM_BT() __builtin_amdgcn_s_memrealtime()
#endif
#ifndef RM_BWD_ABL
#define RM_BWD_ABL 0  // ablation builds (wrong results): 1 no d_rows stores, 2 no dW0 MFMAs, 4 no dX MFMAs, 8 no x prefetch
#endif
template <bool NT_OUT>
__global__ __launch_bounds__(512) void mlp_bwd_kernel(
    const float *__restrict__ xe, const float *__restrict__ xd, int FD, int Dn,
    const float *__restrict__ W0, int H0, int64_t B, const float *__restrict__ g,
    const float *__restrict__ dh0, const float *__restrict__ fm_sum, int D,
    float *__restrict__ d_rows, float *__restrict__ dW0_part, int s_lds) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = FD + Dn;
  const int Kp = ((K + 63) / 64) * 64;
  const int nkt = Kp / 32;
  constexpr int LDR = 36;
  float *W0r = smem;                         // [Kp][36]: W0r[k][u] = W0[k][u]
  float *xs_all = W0r + Kp * LDR;            // [8 waves][32][kLDT]
  float *dT_all = xs_all + 8 * 32 * kLDT;    // [8 waves][32][33]: dh0 as [example][unit]
  float *gS_all = dT_all + 8 * 32 * 33;      // [8 waves][32][D]: g[b] * S[b][:] of the wave's tile (s_lds)
  // The 8 waves share one 32-example tile; wave w owns k-tiles w and w+8: 2 dW0 accumulators
  // (32 registers).  With two groups of 4 waves and 4 accumulators each the kernel sat at 256
  // VGPRs and every change to its epilogue spilled (profiles/r01_p9_mlp_bwd.md).
  const int tid = threadIdx.x, lane = tid & 63, wv8 = tid >> 6, wave = wv8;
  const int h = lane >> 5, c = lane & 31;
#ifdef RM_BWD_STAMP
  const unsigned long long bt0 = RM_BT();
  unsigned long long bt_stage = 0, bt_kt = 0;
#endif

  // prologue loads in the order parameters (L2) -> first tile's x / g / dh0 (HBM) -> parameters' LDS
  // stores, as in mlp_fwd_kernel: the HBM stream starts at t = 0 and the staging runs under it
  float w0v[kW0Regs];
  stage_w0_load<512>(w0v, W0, K, Kp, H0, tid);

  float *xs = xs_all + wv8 * 32 * kLDT;
  float *dT = dT_all + wv8 * 32 * 33;
  float *gS = gS_all + wv8 * 32 * D;
  f32x16 accw[kWT];
#pragma unroll
  for (int t = 0; t < kWT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[t][r] = 0.f;

  const int64_t ntiles = (B + 31) / 32;
  // x slices are prefetched ONE EXAMPLE TILE ahead: pf[j] always holds (or has in flight) the
  // wave's j-th k-tile of the tile about to be processed; it is re-issued for the next example
  // tile as soon as it has been copied to LDS (unconditionally - the last tile re-loads itself)
  // k-tile slots of this wave: slot j = tile wave + 8 j, both products (dX and dW0) - except at 14 k-tiles
  // (the Criteo shape), where that leaves SIMD 0 (waves 0 and 4) with four full tiles and SIMDs 2 and 3 with
  // three (in-kernel timestamps: wave 4 finished at 61 us, waves 6 and 7 at 46; the block waits for the last):
  // tile 12 is SPLIT - wave 4 keeps its dX, wave 6 (one tile so far) takes its dW0 - 3.5 / 3.5 / 3.5 / 3 tiles
  // per SIMD.  Both waves stage the tile's x slices (the second read hits L2).
  const bool split14 = nkt == 14;
  auto slot_kt = [&](int j) { return (split14 && wave == 6 && j == 1) ? 12 : wave + 8 * j; };
  auto slot_dw = [&](int j) { return !(split14 && wave == 4 && j == 1); };
  auto slot_dx = [&](int j) { return !(split14 && wave == 6 && j == 1); };
  float4 pf[kWT][4];
  {
    const int64_t e0 = (int64_t)(blockIdx.x < ntiles ? blockIdx.x : 0) * 32;
#pragma unroll
    for (int j = 0; j < kWT; ++j)
      if (slot_kt(j) < nkt) load_ktile(pf[j], xe, xd, FD, Dn, B, e0, slot_kt(j) * 32, lane);
  }
  // ... and so are the tile's g and dh0 values (written by mlp_dh_chain_kernel, the launch before):
  // lane (c = example, h) holds dh0[example][8 gq + 4 h + e] - the B operand of the dX product as it
  // stands.  (The chain used to be recomputed here by every wave; with every load and MFMA compiled
  // out the kernel still took 40 of its 102 us - g, h -> dh -> 16 DEPENDENT MFMAs -> dh0 -> LDS sat
  // on every tile's critical path: profiles/r01_p9_mlp_bwd.md.)
  float4 dn[4];
  float gn;
  // D == 16 (block-uniform): the tile's S rows are prefetched with them - lane (c, h) holds S[c][4h .. 4h+3]
  // and S[c][8+4h .. 8+4h+3], exactly the columns (k mod 16) of the 16 dX values it gets from the MFMA - and the
  // FM term g (S - E) is applied in the accumulator layout, from registers: no g*S image in LDS, no second
  // LDS read per stored piece (the epilogue was the largest part of a k-tile: 15 of a wave's 57 us,
  // tools/probe/bwd_stamps.py).  (The generic staging loop below loads g and S and stores g*S inside the
  // tile: two exposed round trips per tile on every wave.)
  const bool s_pf = s_lds && D == 16;
  float4 sn0 = make_float4(0.f, 0.f, 0.f, 0.f), sn1 = sn0;
  {
    int64_t bn = (int64_t)(blockIdx.x < ntiles ? blockIdx.x : 0) * 32 + c;
    bn = bn < B ? bn : B - 1;
    gn = g[bn];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      dn[gq] = *reinterpret_cast<const float4 *>(dh0 + bn * 32 + 8 * gq + 4 * h);
    if (s_pf) {
      sn0 = *reinterpret_cast<const float4 *>(fm_sum + bn * 16 + 4 * h);
      sn1 = *reinterpret_cast<const float4 *>(fm_sum + bn * 16 + 8 + 4 * h);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  stage_w0_store<false, 512>(W0r, LDR, w0v, Kp, tid);
  __syncthreads();
#ifdef RM_BWD_STAMP
  const unsigned long long bt1 = RM_BT();
#endif
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef RM_BWD_STAMP
    const unsigned long long bta = RM_BT();
#endif
    const int64_t ex0 = tile * 32;
    const int64_t ex_next = (tile + gridDim.x < ntiles ? tile + gridDim.x : tile) * 32;
    const int64_t b = ex0 + c;
    const bool valid = b < B;
    const float gb = valid ? gn : 0.f;
    float dh[16];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      dh[4 * gq + 0] = dn[gq].x; dh[4 * gq + 1] = dn[gq].y;
      dh[4 * gq + 2] = dn[gq].z; dh[4 * gq + 3] = dn[gq].w;
    }
    const float4 sa = sn0, sb = sn1;  // this tile's S pieces (the next tile's are requested below)
    {  // next tile's values (the last tile re-loads its own)
      int64_t bn = ex_next + c;
      bn = bn < B ? bn : B - 1;
      gn = g[bn];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        dn[gq] = *reinterpret_cast<const float4 *>(dh0 + bn * 32 + 8 * gq + 4 * h);
      if (s_pf) {
        sn0 = *reinterpret_cast<const float4 *>(fm_sum + bn * 16 + 4 * h);
        sn1 = *reinterpret_cast<const float4 *>(fm_sum + bn * 16 + 8 + 4 * h);
      }
    }
    if (s_lds && !s_pf) {
      // g[b] * S[b][:] of this tile, once per tile: every k-tile's epilogue adds the same 16-byte
      // piece of it (k % D repeats), read back row-major from LDS instead of from global memory
      const int D4 = D >> 2;
      for (int f = lane; f < 32 * D4; f += 64) {
        const int row = f / D4, p4 = f - row * D4;
        int64_t br = ex0 + row;
        br = br < B ? br : B - 1;
        const float gr = g[br];
        const float4 s4 = *reinterpret_cast<const float4 *>(fm_sum + br * D + 4 * p4);
        *reinterpret_cast<float4 *>(gS + row * D + 4 * p4) =
            make_float4(gr * s4.x, gr * s4.y, gr * s4.z, gr * s4.w);
      }
    }
    // dh0 as [example][unit] for the dW0 product (B operand: lane = unit)
#pragma unroll
    for (int r = 0; r < 16; ++r) dT[c * 33 + unit_of(r, h)] = valid ? dh[r] : 0.f;

#ifdef RM_BWD_STAMP
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the staging's LDS writes have left
    const unsigned long long btb = RM_BT();
    bt_stage += btb - bta;
#endif
    // ---- this wave's k-tiles: dX tile (-> d_rows) and dW0 tile ----
#pragma unroll
    for (int j = 0; j < kWT; ++j) {
      const int kt = slot_kt(j);
      if (kt >= nkt) break;
      const int kb = kt * 32;
      const bool do_dw = slot_dw(j);  // wave-uniform
      float *xb = xs;  // single buffer: LDS ops of one wave execute in order
      store_ktile(xb, pf[j], lane);
      if (!(RM_BWD_ABL & 8)) load_ktile(pf[j], xe, xd, FD, Dn, B, ex_next, kb, lane);
      // dW0[kb + c'][unit] += sum_ex x[ex][kb + c'] * dh0[ex][unit]   (first: needs the x tile intact)
      if (do_dw) {
#pragma unroll
        for (int s = 0; s < ((RM_BWD_ABL & 2) ? 0 : 16); ++s)
          accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[(2 * s + h) * kLDT + c],
                                                        dT[(2 * s + h) * 33 + c], accw[j], 0, 0, 0);
      }
      const bool has_dx = kb < FD && slot_dx(j);  // dX only for the embedding part of x (wave-uniform)
      f32x16 acc;
      if (has_dx) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int gq = 0; gq < ((RM_BWD_ABL & 4) ? 0 : 4); ++gq) {
          const float4 a4 = *reinterpret_cast<const float4 *>(W0r + (kb + c) * LDR + 8 * gq + 4 * h);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, dh[4 * gq + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, dh[4 * gq + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, dh[4 * gq + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, dh[4 * gq + 3], acc, 0, 0, 0);
        }
        // acc[r] = dX[example c][k = kb + u(r,h)]
      }
      if (has_dx) {
        // dX (- g*E for the FM term) goes back INTO the x tile, in place, and leaves it row-major:
        // every lane then writes whole 16-byte pieces of 128-byte row segments (8 lanes per row,
        // 8 rows per store instruction) and reads S the same way.  Storing straight from the
        // MFMA layout touched 32 rows x 32 bytes per instruction: the epilogue was 42 of the
        // kernel's 109 us (ablation, profiles/r01_p9).
        const bool fm = fm_sum != nullptr;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float4 *p4 = reinterpret_cast<float4 *>(xb + c * kLDT + 8 * gq + 4 * h);
          float4 o = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
          if (fm) {
            const float4 e = *p4;
            if (s_pf) {  // + g (S - E): column (8 gq + 4 h + e) mod 16 of S
              const float4 sv = (gq & 1) ? sb : sa;
              o.x += gb * (sv.x - e.x); o.y += gb * (sv.y - e.y); o.z += gb * (sv.z - e.z); o.w += gb * (sv.w - e.w);
            } else {
              o.x -= gb * e.x; o.y -= gb * e.y; o.z -= gb * e.z; o.w -= gb * e.w;
            }
          }
          *p4 = o;
        }
        const int piece = lane & 7;
        const int k = kb + 4 * piece;
        if (fm && s_lds) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = (lane >> 3) + 8 * q;
            const int64_t br = ex0 + row;
            float4 o = *reinterpret_cast<const float4 *>(xb + row * kLDT + 4 * piece);
            if (!s_pf) {
              const float4 s4 = *reinterpret_cast<const float4 *>(gS + row * D + (k % D));
              o.x += s4.x; o.y += s4.y; o.z += s4.z; o.w += s4.w;
            }
            if (NT_OUT) {
              if (br < B && k < FD) store4_stream(d_rows + br * FD + k, o);
            } else {
              if (br < B && k < FD && (!(RM_BWD_ABL & 1) || o.x == 1234.5f)) *reinterpret_cast<float4 *>(d_rows + br * FD + k) = o;
            }
          }
        } else {
#pragma unroll 1  // (register-bound kernel: the unrolled form with its global loads spilled 56 VGPRs)
          for (int q = 0; q < 4; ++q) {
            const int row = (lane >> 3) + 8 * q;
            const int64_t br = ex0 + row;
            if (br < B && k < FD) {
              float4 o = *reinterpret_cast<const float4 *>(xb + row * kLDT + 4 * piece);
              if (fm) {
                const float gr = g[br];
                const float4 s4 = *reinterpret_cast<const float4 *>(fm_sum + br * D + (k % D));
                o.x += gr * s4.x; o.y += gr * s4.y; o.z += gr * s4.z; o.w += gr * s4.w;
              }
              *reinterpret_cast<float4 *>(d_rows + br * FD + k) = o;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the unrolled k-tiles from being interleaved
    }
#ifdef RM_BWD_STAMP
    bt_kt += RM_BT() - btb;
#endif
  }
#ifdef RM_BWD_STAMP
  const unsigned long long bt2 = RM_BT();
#endif

  // ---- the block's dW0 partial: every wave owns the rows of its k-tiles ----
#pragma unroll
  for (int j = 0; j < kWT; ++j) {
    const int kt = slot_kt(j);
    if (kt < nkt && slot_dw(j)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dW0_part[((int64_t)blockIdx.x * Kp + kt * 32 + unit_of(r, h)) * 32 + c] = accw[j][r];
    }
  }
#ifdef RM_BWD_STAMP
  if (lane == 0 && blockIdx.x < 256) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long *o = rm_bwd_stamp_buf + 8 * (blockIdx.x * 8 + wave);
    o[0] = bt1 - bt0; o[1] = bt2 - bt1; o[2] = RM_BT() - bt2; o[3] = bt_stage; o[4] = bt_kt; o[5] = bt0;
  }
#endif
}
#ifdef RM_BWD_STAMP
}  // namespace
extern "C" int rm_debug_bwd_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_bwd_stamp_buf), sizeof(unsigned long long) * n);
}
namespace {
#endif

// dW0[k][u] = sum_slabs part[slab][k][u]   (k < K, u < H0).  A block owns 64 consecutive
// slab elements; its 4 waves take every 4th slab (coalesced 256-byte reads), partial sums
// meet in LDS in a fixed order -> deterministic.
// The finishing reductions run 16 waves per block: a thread's chain of dependent-latency loads is
// nslab / 16 long (with 4 waves the 512-partial small-gradient reduction alone took 9 us).
constexpr int kFinG = 16;
__device__ __forceinline__ float fin_tree(float (*sm)[64], int o) {
  float v[kFinG];
#pragma unroll
  for (int q = 0; q < kFinG; ++q) v[q] = sm[q][o];
#pragma unroll
  for (int st = 1; st < kFinG; st *= 2)
#pragma unroll
    for (int q = 0; q < kFinG; q += 2 * st) v[q] += v[q + st];
  return v[0];
}
__device__ __forceinline__ void mlp_dw0_reduce_body(float (*sm)[64], int blk,
                                                    const float *__restrict__ part, int nslab, int K,
                                                    int Kp, int H0, float *__restrict__ dW0) {
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blk * 64 + o;  // element of the [Kp][32] slab
  const int64_t slab = (int64_t)Kp * 32;
  float acc = 0.f;
#pragma unroll 8
  for (int i = grp; i < nslab; i += kFinG) acc += part[(int64_t)i * slab + e];
  sm[grp][o] = acc;
  __syncthreads();
  if (grp == 0) {
    const float v = fin_tree(sm, o);
    const int k = e >> 5, u = e & 31;
    if (k < K && u < H0) dW0[k * H0 + u] = v;
  }
}

// Every remaining (tiny) gradient of the MLP in ONE pass over h_l / dh_l / g:
//   dW_l = h_{l-1}^T dh_l (l >= 1), db_l = colsum(dh_l), d w_out = h_last^T g, d w0 = sum g.
// Per-block partial layout: [2 x 1024 dW_l | 3 x 32 db_l | 32 d w_out | 1 sum g (+31 pad) | 32 g^T xd].
constexpr int kSgDense = kRmSgDense;  // offset of the g^T xd slots (mlp_internal.h)
constexpr int kSgStride = kRmSgStride;
// Stage 1 on the matrix pipe (the first version was a VALU kernel behind LDS staging, 20 us): a wave walks
// 32-example chunks, loads h_l / dh_l straight in MFMA operand layout (lane = unit, the two lane
// halves = the two examples of a k-step: coalesced 128-byte rows, no LDS staging),
//   dW_l += h_{l-1}^T dh_l on the matrix pipe, db_l / d w_out / sum g as per-lane column sums.
// The 4 waves of a block meet in LDS and write ONE partial per block.  20 -> ~8 us at B = 65536.
template <int NL>
__global__ __launch_bounds__(256) void mlp_small_grads_mfma(
    const float *__restrict__ h0, const float *__restrict__ h1, const float *__restrict__ h2,
    const float *__restrict__ d0, const float *__restrict__ d1, const float *__restrict__ d2,
    const float *__restrict__ g, const float *__restrict__ xd, int Dn, int64_t B,
    float *__restrict__ part) {
  __shared__ float red[4][kSgStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
  for (int t = lane; t < kSgStride; t += 64) red[wave][t] = 0.f;
  constexpr int NW = NL > 1 ? NL - 1 : 1;
  f32x16 acc[NW];
#pragma unroll
  for (int l = 0; l < NW; ++l)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[l][r] = 0.f;
  float db[NL], dwo = 0.f, sg = 0.f, dxd = 0.f;
#pragma unroll
  for (int l = 0; l < NL; ++l) db[l] = 0.f;
  // xd != NULL: also g^T xd (the gradient of the linear term's dense weights, layers.py:330-347:
  // the same g and the same dense columns are already streaming through this kernel)
  // (unconditional load from an always-valid address: a branch around it would serialise the
  // unrolled loads of a chunk)
  const float *xq = xd != nullptr ? xd + (c < Dn ? c : 0) : g;
  const int64_t xstride = xd != nullptr ? Dn : 1;
  const float *hp[3] = {h0, h1, h2};
  const float *dp[3] = {d0, d1, d2};
  const int64_t nchunks = (B + 31) / 32;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nchunks; chunk += (int64_t)gridDim.x * 4) {
    const int64_t r0 = chunk * 32;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int64_t row = r0 + 2 * s + h;
      const bool ok = row < B;
      const int64_t rc = ok ? row : B - 1;
      float hv[NL], dv[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        hv[l] = hp[l][rc * 32 + c];
        const float d = dp[l][rc * 32 + c];
        dv[l] = ok ? d : 0.f;  // rows past the batch contribute nothing
      }
      const float gv = ok ? g[rc] : 0.f;
#pragma unroll
      for (int l = 1; l < NL; ++l)
        acc[l - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(hv[l - 1], dv[l], acc[l - 1], 0, 0, 0);
#pragma unroll
      for (int l = 0; l < NL; ++l) db[l] += dv[l];
      dwo += gv * hv[NL - 1];
      sg += gv;
      dxd += gv * xq[rc * xstride];
    }
  }
  // the two lane halves hold disjoint examples of the same column
#pragma unroll
  for (int l = 0; l < NL; ++l) db[l] += __shfl_xor(db[l], 32, 64);
  dwo += __shfl_xor(dwo, 32, 64);
  sg += __shfl_xor(sg, 32, 64);
  dxd += __shfl_xor(dxd, 32, 64);
#pragma unroll
  for (int l = 1; l < NL; ++l)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(l - 1) * 1024 + unit_of(r, h) * 32 + c] = acc[l - 1][r];
  if (h == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) red[wave][2 * 1024 + l * 32 + c] = db[l];
    red[wave][2 * 1024 + 32 * NL + c] = dwo;
    if (c == 0) red[wave][2 * 1024 + 32 * NL + 32] = sg;
    if (c < Dn) red[wave][kSgDense + c] = dxd;
  }
  __syncthreads();
  float *o = part + (int64_t)blockIdx.x * kSgStride;
  for (int t = tid; t < kSgStride; t += 256) o[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

struct SgOut {
  float *dW[kMaxNL];  // dW[l] for l >= 1 ([H_{l-1}, H_l])
  float *db[kMaxNL];
  float *dw_out, *dw0;
  float *dxd;         // g^T xd [Dn] or NULL
  float *gsum;        // a second destination for sum g (the linear term's w0 gradient) or NULL
  int Dn;
  int H[kMaxNL];
};
__device__ __forceinline__ void mlp_small_grads_stage2_body(float (*sm)[64], int blk,
                                                            const float *__restrict__ part, int nblk,
                                                            int NL, const SgOut &o) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int src = blk * 64 + lane;  // element of the per-block partial
  float acc = 0.f;
  if (src < kSgStride) {
#pragma unroll 8
    for (int i = grp; i < nblk; i += kFinG) acc += part[(int64_t)i * kSgStride + src];
  }
  sm[grp][lane] = acc;
  __syncthreads();
  if (grp != 0 || src >= kSgStride) return;
  acc = fin_tree(sm, lane);
  if (src >= kSgDense) {
    if (o.dxd && src - kSgDense < o.Dn) o.dxd[src - kSgDense] = acc;
  } else if (src < 2 * 1024) {
    const int l = 1 + src / 1024, pq = src % 1024, p = pq >> 5, q = pq & 31;
    if (l < NL && p < o.H[l - 1] && q < o.H[l]) o.dW[l][p * o.H[l] + q] = acc;
  } else {
    const int v = src - 2 * 1024;
    if (v < 32 * NL) {
      const int l = v >> 5, q = v & 31;
      if (q < o.H[l] && o.db[l]) o.db[l][q] = acc;
    } else if (v < 32 * NL + 32) {
      const int q = v - 32 * NL;
      if (q < o.H[NL - 1] && o.dw_out) o.dw_out[q] = acc;
    } else if (v == 32 * NL + 32) {
      if (o.dw0) o.dw0[0] = acc;
      if (o.gsum) o.gsum[0] = acc;
    }
  }
}

// The two finishing reductions of the backward in ONE launch (each was a ~5 us launch of its own on
// a 250 us step): blocks [0, n_dw0) finish dW0 from mlp_bwd's slabs, the rest finish the small
// gradients from mlp_small_grads_mfma's per-block partials.
__global__ __launch_bounds__(64 * kFinG) void mlp_finish_kernel(const float *__restrict__ part, int nslab, int K,
                                                         int Kp, int H0, float *__restrict__ dW0, int n_dw0,
                                                         const float *__restrict__ part2, int nblk2, int NL,
                                                         SgOut o, int n_sg,
                                                         const float *__restrict__ loss_partial,
                                                         int64_t n_loss, float invB,
                                                         float *__restrict__ loss) {
  __shared__ float sm[kFinG][64];
  if ((int)blockIdx.x < n_dw0) {
    mlp_dw0_reduce_body(sm, blockIdx.x, part, nslab, K, Kp, H0, dW0);
  } else if ((int)blockIdx.x < n_dw0 + n_sg) {
    mlp_small_grads_stage2_body(sm, blockIdx.x - n_dw0, part2, nblk2, NL, o);
  } else {
    // the fused head's loss: mean over B of the per-tile sums, fixed order -> deterministic
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n_loss; i += 64 * kFinG) acc += loss_partial[i];
    acc = rm_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = fin_tree(sm, 0) * invB;
  }
}

size_t mlp_fwd_smem(int K, int NL, bool tail) {
  const int Kp = ((K + 63) / 64) * 64;
  return (size_t)(32 * (Kp + 4) + (NL - 1) * 1024 * (tail ? 2 : 1) + (NL + 1) * 32 +
                  RM_MLP_FWD_WAVES * 32 * kLDX) * sizeof(float);
}
size_t mlp_bwd_smem(int K, int Ds) {  // Ds: columns of the per-wave g*S tiles (0 = none)
  const int Kp = ((K + 63) / 64) * 64;
  return (size_t)(Kp * 36 + 8 * 32 * kLDT + 8 * 32 * 33 + 8 * 32 * Ds) * sizeof(float);
}

int mlp_check(const char *fn, int FD, int Dn, int NL, const int *H) {
  RM_REQUIRE(NL >= 1 && NL <= kMaxNL, "%s: %d hidden layers unsupported (1..3)", fn, NL);
  for (int l = 0; l < NL; ++l)
    RM_REQUIRE(H[l] >= 1 && H[l] <= 32, "%s: hidden width %d unsupported (<= 32)", fn, H[l]);
  RM_REQUIRE(FD >= 0 && Dn >= 0 && FD % 4 == 0 && FD + Dn >= 1 && FD + Dn <= 448,
             "%s: input width %d+%d unsupported (FD %% 4 == 0, FD+Dn <= 448)", fn, FD, Dn);
  return RM_OK;
}

}  // namespace

int rm_internal_mlp_finish(const float *dw0_part, int nslab, int K, int Kp, int H0, float *dW0, const float *sg_part,
                           int nblk2, int NL, const int *H, float *const *dW, float *const *db, float *d_w_out,
                           float *d_w0_out, float *d_xd_wsum, float *d_g_sum, int Dn, const float *loss_partial,
                           int64_t n_loss, int64_t B, float *loss, hipStream_t st) {
  SgOut o;
  for (int l = 0; l < kMaxNL; ++l) {
    o.dW[l] = (l >= 1 && l < NL && dW) ? dW[l] : nullptr;
    o.db[l] = (l < NL && db) ? db[l] : nullptr;
    o.H[l] = l < NL ? H[l] : 0;
  }
  o.dw_out = d_w_out;
  o.dw0 = d_w0_out;
  o.dxd = d_xd_wsum;
  o.gsum = d_g_sum;
  o.Dn = Dn;
  const int n_dw0 = Kp * 32 / 64, n_sg = (kSgStride + 63) / 64;
  const bool with_loss = loss != nullptr && loss_partial != nullptr;
  hipLaunchKernelGGL(mlp_finish_kernel, dim3(n_dw0 + n_sg + (with_loss ? 1 : 0)), dim3(64 * kFinG), 0, st, dw0_part,
                     nslab, K, Kp, H0, dW0, n_dw0, sg_part, nblk2, NL, o, n_sg, with_loss ? loss_partial : nullptr,
                     n_loss, 1.0f / (float)B, with_loss ? loss : nullptr);
  RM_CHECK_LAUNCH("rm_internal_mlp_finish");
  return RM_OK;
}

extern "C" int rm_mlp_supported(int FD, int Dn, int NL, const int *H) {
  if (NL < 1 || NL > kMaxNL || FD % 4 != 0 || FD + Dn < 1 || FD + Dn > 448) return 0;
  for (int l = 0; l < NL; ++l)
    if (H[l] < 1 || H[l] > 32) return 0;
  return 1;
}

static int mlp_tail_check(const char *fn, const rm_mlp_tail *t, int NL) {
  RM_REQUIRE(t->dlogit && t->loss_partial, "%s: tail needs dlogit and loss_partial", fn);
  RM_REQUIRE((t->y != nullptr) != (t->y_f != nullptr), "%s: tail needs exactly one of y / y_f", fn);
  RM_REQUIRE(t->task == 0 || t->task == 1, "%s: tail task must be 0 or 1", fn);
  for (int l = 0; l < NL; ++l) RM_REQUIRE(t->dh[l], "%s: tail needs dh[%d]", fn, l);
  return RM_OK;
}

extern "C" int rm_mlp_fwd(const float *xe, const float *xd, int FD, int Dn, int NL, const int *H,
                          const float *const *W, const float *const *bias, const float *w_out,
                          const float *w0_out, int act, int64_t B, float *const *h_out,
                          float *logit, const rm_mlp_tail *tail, rm_stream_t stream) {
  int rc = mlp_check("rm_mlp_fwd", FD, Dn, NL, H);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE((FD == 0 || (xe && rm_aligned16(xe))) && (Dn == 0 || xd) && W && bias && w_out && logit &&
                 h_out, "rm_mlp_fwd: NULL or unaligned argument");
  for (int l = 0; l < NL; ++l) RM_REQUIRE(W[l] && bias[l], "rm_mlp_fwd: NULL weight / bias of layer %d", l);
  if (tail && (rc = mlp_tail_check("rm_mlp_fwd", tail, NL)) != RM_OK) return rc;
  MlpW w;
  for (int l = 0; l < kMaxNL; ++l) {
    w.W[l] = l < NL ? W[l] : nullptr;
    w.b[l] = l < NL ? bias[l] : nullptr;
    w.H[l] = l < NL ? H[l] : 0;
  }
  const size_t smem = mlp_fwd_smem(FD + Dn, NL, tail != nullptr);
  const int64_t ntiles = (B + 31) / 32;
  dim3 grid((unsigned)rm_grid_cap((ntiles + RM_MLP_FWD_WAVES - 1) / RM_MLP_FWD_WAVES, 256));
  hipStream_t st = (hipStream_t)stream;
  float *h0 = h_out[0], *h1 = NL > 1 ? h_out[1] : nullptr, *h2 = NL > 2 ? h_out[2] : nullptr;
  const rm_mlp_tail tl = tail ? *tail : rm_mlp_tail{};
#define RM_MLP_FWD(NL_, TAIL_)                                                                       \
  {                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_fwd_kernel<NL_, TAIL_>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                \
    hipLaunchKernelGGL((mlp_fwd_kernel<NL_, TAIL_>), grid, dim3(64 * RM_MLP_FWD_WAVES), smem, st, xe, \
                       xd, FD, Dn, w, w_out, w0_out, act, B, h0, h1, h2, logit, tl);                  \
  }
  if (tail) {
    if (NL == 1) RM_MLP_FWD(1, true) else if (NL == 2) RM_MLP_FWD(2, true) else RM_MLP_FWD(3, true)
  } else {
    if (NL == 1) RM_MLP_FWD(1, false) else if (NL == 2) RM_MLP_FWD(2, false) else RM_MLP_FWD(3, false)
  }
#undef RM_MLP_FWD
  RM_CHECK_LAUNCH("rm_mlp_fwd");
  return RM_OK;
}

extern "C" int rm_embed_mlp_fwd_supported(int F, int D, int64_t table_ld, int Dn, int NL, const int *H) {
  // rows of 32 floats (the engines' table: one 128-byte line per row) or of D + 4 = 20 (the rows a row-sharded
  // table exchanges, recman_amd/dist.py: idx = positions in the received buffer, field_off = 0)
  if (D != 16 || (table_ld != 32 && table_ld != 20) || F < 1 || Dn < 0 || Dn > 16 || 16 * F + Dn > 448) return 0;
  return rm_mlp_supported(16 * F, Dn, NL, H);
}

extern "C" int rm_embed_mlp_fwd(const int64_t *idx, const float *table, int64_t table_ld,
                                const int64_t *field_off, int want_bias, int want_lin,
                                const float *lin_w_dense, const float *lin_w0, const float *xd, int Dn,
                                int64_t B, int F, int D, float *E, float *fm_sum, float *fm_logit,
                                float *lin_logit, int flags, int NL, const int *H, const float *const *W,
                                const float *const *bias, const float *w_out, const float *w0_out, int act,
                                float *const *h_out, float *logit, const rm_mlp_tail *tail,
                                rm_stream_t stream) {
  RM_REQUIRE(H && rm_embed_mlp_fwd_supported(F, D, table_ld, Dn, NL, H),
             "rm_embed_mlp_fwd: needs D = 16, table_ld = 32 or 20, Dn <= 16, 16 F + Dn <= 448 and hidden widths <= 32");
  int rc = mlp_check("rm_embed_mlp_fwd", 16 * F, Dn, NL, H);
  if (rc != RM_OK) return rc;
  RM_REQUIRE(B >= 0, "rm_embed_mlp_fwd: B < 0");
  if (B == 0) return RM_OK;
  RM_REQUIRE(idx && table && field_off && E && rm_aligned16(table) && rm_aligned16(E) && (Dn == 0 || xd) && W &&
                 bias && w_out && logit && h_out,
             "rm_embed_mlp_fwd: NULL or unaligned argument");
  RM_REQUIRE(fm_sum == nullptr || rm_aligned16(fm_sum), "rm_embed_mlp_fwd: fm_sum must be 16-byte aligned");
  RM_REQUIRE(lin_w_dense == nullptr || Dn > 0, "rm_embed_mlp_fwd: lin_w_dense without dense inputs");
  for (int l = 0; l < NL; ++l) RM_REQUIRE(W[l] && bias[l], "rm_embed_mlp_fwd: NULL weight / bias of layer %d", l);
  if (tail) {
    if ((rc = mlp_tail_check("rm_embed_mlp_fwd", tail, NL)) != RM_OK) return rc;
    // the head's other branches are the ones this kernel produces (or absent)
    RM_REQUIRE((tail->logit_a == nullptr || tail->logit_a == lin_logit || tail->logit_a == fm_logit) &&
                   (tail->logit_b == nullptr || tail->logit_b == lin_logit || tail->logit_b == fm_logit),
               "rm_embed_mlp_fwd: tail.logit_a / logit_b must be this call's lin_logit / fm_logit buffers");
  }
  MlpW w;
  for (int l = 0; l < kMaxNL; ++l) {
    w.W[l] = l < NL ? W[l] : nullptr;
    w.b[l] = l < NL ? bias[l] : nullptr;
    w.H[l] = l < NL ? H[l] : 0;
  }
  const EmbFront ef{idx, table, (int)table_ld, field_off, lin_w_dense, lin_w0, xd, F, want_bias, want_lin, E, fm_sum, fm_logit, lin_logit};
  const size_t smem = mlp_fwd_smem(16 * F + Dn, NL, tail != nullptr);
  const int64_t ntiles = (B + 31) / 32;
  dim3 grid((unsigned)rm_grid_cap((ntiles + RM_MLP_FWD_WAVES - 1) / RM_MLP_FWD_WAVES, 256));
  hipStream_t st = (hipStream_t)stream;
  float *h0 = h_out[0], *h1 = NL > 1 ? h_out[1] : nullptr, *h2 = NL > 2 ? h_out[2] : nullptr;
  const rm_mlp_tail tl = tail ? *tail : rm_mlp_tail{};
  const bool nt = (flags & 1) != 0;  // RM_EMBED_STREAM_ROWS
#define RM_EMF(NL_, TAIL_, NT_)                                                                          \
  {                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(embed_mlp_fwd_kernel<NL_, TAIL_, NT_>),     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                    \
    hipLaunchKernelGGL((embed_mlp_fwd_kernel<NL_, TAIL_, NT_>), grid, dim3(64 * RM_MLP_FWD_WAVES), smem, \
                       st, ef, xd, Dn, w, w_out, w0_out, act, B, h0, h1, h2, logit, tl);                 \
  }
#define RM_EMF_NL(TAIL_, NT_)                                                                       \
  if (NL == 1) RM_EMF(1, TAIL_, NT_) else if (NL == 2) RM_EMF(2, TAIL_, NT_) else RM_EMF(3, TAIL_, NT_)
  if (tail) {
    if (nt) { RM_EMF_NL(true, true) } else { RM_EMF_NL(true, false) }
  } else {
    if (nt) { RM_EMF_NL(false, true) } else { RM_EMF_NL(false, false) }
  }
#undef RM_EMF_NL
#undef RM_EMF
  RM_CHECK_LAUNCH("rm_embed_mlp_fwd");
  return RM_OK;
}

extern "C" int64_t rm_mlp_bwd_workspace(int FD, int Dn) {
  const int Kp = ((FD + Dn + 63) / 64) * 64;
  return (int64_t)512 * Kp * 32 + 512 * kSgStride;
}

extern "C" int rm_mlp_bwd(const float *xe, const float *xd, int FD, int Dn, int NL, const int *H,
                          const float *const *W, const float *w_out, int act, int64_t B,
                          const float *g, const float *const *h, const float *fm_sum, int D,
                          float *d_rows, float *const *dh, float *const *dW, float *const *db,
                          float *d_w_out, float *d_w0_out, float *d_xd_wsum, float *d_g_sum,
                          float *workspace, const rm_mlp_tail *tail, int flags, rm_stream_t stream) {
  int rc = mlp_check("rm_mlp_bwd", FD, Dn, NL, H);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  if (tail && (rc = mlp_tail_check("rm_mlp_bwd", tail, NL)) != RM_OK) return rc;
  RM_REQUIRE((FD == 0 || (xe && rm_aligned16(xe))) && (Dn == 0 || xd) && W && w_out && g && h && dh &&
                 dW && workspace && (FD == 0 || (d_rows && rm_aligned16(d_rows))),
             "rm_mlp_bwd: NULL or unaligned argument");
  RM_REQUIRE(!fm_sum || (D > 0 && D % 4 == 0 && rm_aligned16(fm_sum)), "rm_mlp_bwd: bad fm_sum / D");
  RM_REQUIRE(!d_xd_wsum || (Dn >= 1 && Dn <= 32), "rm_mlp_bwd: d_xd_wsum needs 1 <= Dn <= 32");
  MlpW w;
  for (int l = 0; l < kMaxNL; ++l) {
    w.W[l] = l < NL ? W[l] : nullptr;
    w.b[l] = nullptr;
    w.H[l] = l < NL ? H[l] : 0;
  }
  const int K = FD + Dn, Kp = ((K + 63) / 64) * 64;
  // the per-wave g*S tiles go to LDS when they fit next to everything else (D = 16: 16 KB more)
  const int s_lds = (fm_sum != nullptr && mlp_bwd_smem(K, D) <= 160 * 1024) ? 1 : 0;
  const size_t smem = mlp_bwd_smem(K, s_lds ? D : 0);
  const int64_t ntiles = (B + 31) / 32;
  const int nblk = rm_grid_cap(ntiles, 256);  // one 8-wave block per CU, one 32-example tile at a time
  hipStream_t st = (hipStream_t)stream;
  float *part = workspace;
  float *part2 = workspace + (int64_t)512 * Kp * 32;
  // 1. the dh chain, one wave per tile: dh_l for every layer (already there after a fused forward)
  if (!tail) {
    const dim3 cgrid((unsigned)((ntiles + 3) / 4));
    const float *h1p = NL > 1 ? h[1] : nullptr, *h2p = NL > 2 ? h[2] : nullptr;
    float *d1p = NL > 1 ? dh[1] : nullptr, *d2p = NL > 2 ? dh[2] : nullptr;
    if (NL == 1)
      hipLaunchKernelGGL((mlp_dh_chain_kernel<1>), cgrid, dim3(256), 0, st, w, w_out, act, B, g, h[0], h1p,
                         h2p, dh[0], d1p, d2p);
    else if (NL == 2)
      hipLaunchKernelGGL((mlp_dh_chain_kernel<2>), cgrid, dim3(256), 0, st, w, w_out, act, B, g, h[0], h1p,
                         h2p, dh[0], d1p, d2p);
    else
      hipLaunchKernelGGL((mlp_dh_chain_kernel<3>), cgrid, dim3(256), 0, st, w, w_out, act, B, g, h[0], h1p,
                         h2p, dh[0], d1p, d2p);
  }
  {  // 2. the small gradients
    SgOut o;
    for (int l = 0; l < kMaxNL; ++l) {
      o.dW[l] = (l >= 1 && l < NL) ? dW[l] : nullptr;
      o.db[l] = (l < NL && db) ? db[l] : nullptr;
      o.H[l] = l < NL ? H[l] : 0;
    }
    o.dw_out = d_w_out;
    o.dw0 = d_w0_out;
    o.dxd = d_xd_wsum;
    o.gsum = d_g_sum;
    o.Dn = Dn;
    const float *xdp = d_xd_wsum ? xd : nullptr;
    const float *h1p = NL > 1 ? h[1] : nullptr, *h2p = NL > 2 ? h[2] : nullptr;
    const float *d1p = NL > 1 ? dh[1] : nullptr, *d2p = NL > 2 ? dh[2] : nullptr;
    const int sblk = (int)rm_grid_cap((ntiles + 3) / 4, 512);  // 4-wave blocks, one 32-example chunk per wave up to B = 65536; one partial per block
    if (NL == 1)
      hipLaunchKernelGGL((mlp_small_grads_mfma<1>), dim3(sblk), dim3(256), 0, st, h[0], h1p, h2p,
                         dh[0], d1p, d2p, g, xdp, Dn, B, part2);
    else if (NL == 2)
      hipLaunchKernelGGL((mlp_small_grads_mfma<2>), dim3(sblk), dim3(256), 0, st, h[0], h1p, h2p,
                         dh[0], d1p, d2p, g, xdp, Dn, B, part2);
    else
      hipLaunchKernelGGL((mlp_small_grads_mfma<3>), dim3(sblk), dim3(256), 0, st, h[0], h1p, h2p,
                         dh[0], d1p, d2p, g, xdp, Dn, B, part2);
    // 3. dX (+ FM term) -> d_rows and the dW0 slabs, from x and dh0.  (After the small gradients:
    // those read h_l / dh_l, which the forward's epilogue wrote a moment ago - still in L2 / the
    // Infinity Cache now, gone from them after this kernel's 220 MB.)
    if (flags & RM_MLP_STREAM_DROWS) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_bwd_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(mlp_bwd_kernel<true>, dim3(nblk), dim3(512), smem, st, xe, xd, FD, Dn, W[0], H[0], B, g,
                         (const float *)dh[0], fm_sum, D, d_rows, part, s_lds);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_bwd_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      hipLaunchKernelGGL(mlp_bwd_kernel<false>, dim3(nblk), dim3(512), smem, st, xe, xd, FD, Dn, W[0], H[0], B, g,
                         (const float *)dh[0], fm_sum, D, d_rows, part, s_lds);
    }
    const int n_dw0 = Kp * 32 / 64, n_sg = (kSgStride + 63) / 64;
    const bool with_loss = tail && tail->loss;
    hipLaunchKernelGGL(mlp_finish_kernel, dim3(n_dw0 + n_sg + (with_loss ? 1 : 0)), dim3(64 * kFinG), 0, st, part,
                       nblk, K, Kp, H[0], dW[0], n_dw0, part2, sblk, NL, o, n_sg,
                       with_loss ? tail->loss_partial : nullptr, ntiles, 1.0f / (float)B,
                       with_loss ? tail->loss : nullptr);
  }
  RM_CHECK_LAUNCH("rm_mlp_bwd");
  return RM_OK;
}
