// CrossNet (DCN v1, vector form): all L layers fused, forward and backward.
// The class is ABSENT from the reference (recman/tf/core/DCN.py:7 has the import
// commented out; used at DCN.py:134-137); arithmetic per arXiv 1708.05123 eq. (3):
//     x_{l+1} = x0 (x_l . w_l) + b_l + x_l,   logit = x_L . w_out.
//
// Closed form (exact algebra): with Bp_l = sum_{j<l} b_j (parameters only)
//     x_l = c_l x0 + Bp_l,   c_0 = 1,   c_{l+1} = c_l + s_l,
//     s_l = x_l . w_l = c_l p_l + beta_l,   p_l = x0 . w_l,   beta_l = Bp_l . w_l,
//     logit = c_L p_out + beta_out,         p_out = x0 . w_out, beta_out = Bp_L . w_out
// so the FORWARD is L+1 dot products of the x0 row against fixed vectors plus a scalar recurrence:
// no x_l is ever formed.  The BACKWARD, with t_l = delta_{l+1} . x0:
//     t_l = g p_out + sum_{j>l} t_j p_j   (scalar recurrence on the saved p row),
//     d x0 = g c_L w_out + sum_j (t_j c_j) w_j
// needs the p row only: it never re-reads x0.  Both kernels are pure streams, HBM-bound by design:
//   fwd  reads x0 [B,d] once, writes logit [B] and p [B, p_ld]                (1,752 B/example at d=429, L=6)
//   bwd  reads the other branch's dx [B,FD] and p, writes dx0 [B,FD] + coef   (3,424 B/example)
// (round 1's kernels kept x0, x_l / delta_l and an accumulator in registers - 84..112 VGPRs of vector
// state, the per-layer vectors in LDS, 4 dependent LDS-shuffle reductions per layer - and sat 74-82 %
// of their wave-cycles in s_waitcnt at 28 % / 38 % of HBM: profiles/r02_cross_counters.md.)
//
// Mapping: one 64-lane wave per example row; lane `lane` owns the float4 slices q = lane + 64 s
// (s < NS, NS = 1 or 2: FD <= 512) of the embedding part and dense column `lane`, so a wave-instruction
// reads 1 KiB contiguous.  The L+1 weight vectors live in REGISTERS for the whole kernel ((L+1) NS
// float4 + L+1 floats per lane; staged once per block through LDS), no LDS in the loop.  Row sums: 4 DPP steps inside each 16-lane row + the gfx950 permlane swaps across rows -
// no LDS traffic either.  A wave walks U consecutive rows per iteration and loads the next
// iteration's rows before it computes the current ones (the kernels live on loads in flight).
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxL = 8;
// rows per wave and iteration (U) and waves per SIMD the register budget is cut for (WPS): measured
// variants in profiles/r02_cross_counters.md
#ifndef RM_CROSS_FWD_U
#define RM_CROSS_FWD_U 2
#endif
#ifndef RM_CROSS_FWD_WPS
#define RM_CROSS_FWD_WPS 4
#endif
#ifndef RM_CROSS_BWD_U
#define RM_CROSS_BWD_U 2
#endif
#ifndef RM_CROSS_BWD_WPS
#define RM_CROSS_BWD_WPS 4
#endif

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over the 64 lanes, every lane gets the total; fixed order -> deterministic
__device__ __forceinline__ float wave_allsum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the 8-lane half
  v += dpp_mov<0x140>(v);  // row_mirror: the other half of the 16-lane row
  // rows, then halves: v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of the
  // first register with the even rows (lower half) of the second, so with two copies of v the sum of the
  // two results is the xor-16 / xor-32 butterfly.  Inline asm: through __builtin_amdgcn_permlane16_swap
  // hipcc (ROCm 7.2) added the FIRST result to itself here (v_add v, v10, v10 behind the swap) - the sum
  // doubled instead of crossing rows; tests/test_gpu_cross.py pins it.  "s_nop 1" = the two wait states
  // the swap needs behind a VALU write of its operands (cdna_hip_programming.md T21).
  {
    float a = v, c = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));
    v = a + c;
  }
  {
    float a = v, c = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
    v = a + c;
  }
  return v;
}

// a value every lane holds identically -> one scalar register (frees a VGPR; VALU reads it as an SGPR operand)
__device__ __forceinline__ float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Per-lane parameter registers.  x0 = [xe | xd] is covered by NS float4 "embedding" slices per lane
// (slice q = lane + 64 s of xe; FD <= 256 NS) plus, for the Dn dense columns, ONE float per lane
// (column `lane`; ND = 1 needs Dn <= 64; ND = 2: any Dn, a slow dynamic loop; ND = 0: no dense part).
// Lanes whose slice / column lies past the end carry ZERO WEIGHTS and load from a clamped address:
// nothing is ever selected on loaded data, so no load has to be waited for before it is used
// (a select right behind a prefetch load put s_waitcnt vmcnt(0) into the prefetch itself).
template <int NS, int L>
struct CrossParams {
  float4 W[L + 1][NS];  // w_0 .. w_{L-1}, w_out: embedding slices
  float Wd[L + 1];      // the same vectors' dense column `lane`
  float beta[L + 1];    // Bp_l . w_l (l < L), Bp_L . w_out
};

// sm: [(2L+1)][P] floats, P = 256 NS + 64: rows w_0..w_{L-1}, w_out, b_0..b_{L-1}; each row = the
// embedding part zero-padded to 256 NS, then 64 dense slots (zero past Dn).  Coalesced loads, every
// thread issues all its loads before its first LDS store.
template <int NS>
__device__ __forceinline__ void stage_params(float *sm, const float *__restrict__ w,
                                             const float *__restrict__ b,
                                             const float *__restrict__ w_out, int L, int FD, int Dn) {
  constexpr int P = 256 * NS + 64;
  constexpr int PER = 8;
  const int d = FD + Dn;
  const int total = (2 * L + 1) * P;
  for (int base = 0; base < total; base += kBlock * PER) {
    float v[PER];
    bool ok[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = base + q * kBlock + threadIdx.x;
      const int r = i / P, e = i - r * P;
      const int src = e < 256 * NS ? e : FD + e - 256 * NS;
      ok[q] = i < total && (e < 256 * NS ? e < FD : e - 256 * NS < Dn);
      // one unconditional load from a clamped address per element (a branch per source array made
      // every element its own divergent region); zeroed by the select below
      const float *row = r < L ? w + (int64_t)r * d : (r == L ? w_out : b + (int64_t)(r - L - 1) * d);
      v[q] = (i < total ? row : w)[ok[q] ? src : 0];
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = base + q * kBlock + threadIdx.x;
      if (i < total) sm[i] = ok[q] ? v[q] : 0.f;
    }
  }
}

template <int NS, int L, int ND>
__device__ __forceinline__ void load_params(const float *sm, const float *__restrict__ w,
                                            const float *__restrict__ b, const float *__restrict__ w_out,
                                            int FD, int Dn, int lane, CrossParams<NS, L> &P_) {
  constexpr int P = 256 * NS + 64;
#pragma unroll
  for (int j = 0; j <= L; ++j) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      P_.W[j][s] = *reinterpret_cast<const float4 *>(sm + j * P + 4 * (lane + 64 * s));
    P_.Wd[j] = ND == 1 ? sm[j * P + 256 * NS + lane] : 0.f;
  }
  float4 Bp[NS];
  float Bd = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) Bp[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int d = FD + Dn;
#pragma unroll
  for (int j = 0; j <= L; ++j) {
    float part = Bd * P_.Wd[j];
#pragma unroll
    for (int s = 0; s < NS; ++s) part += dot4(Bp[s], P_.W[j][s]);
    if (ND == 2) {  // Dn > 64: the dense columns straight from memory (prologue only)
      for (int k = lane; k < Dn; k += 64) {
        float bp = 0.f;
        for (int i = 0; i < j && i < L; ++i) bp += b[(int64_t)i * d + FD + k];
        part += bp * (j < L ? w[(int64_t)j * d + FD + k] : w_out[FD + k]);
      }
    }
    P_.beta[j] = uniform(wave_allsum(part));
    if (j < L) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float4 bl = *reinterpret_cast<const float4 *>(sm + (L + 1 + j) * P + 4 * (lane + 64 * s));
        Bp[s].x += bl.x; Bp[s].y += bl.y; Bp[s].z += bl.z; Bp[s].w += bl.w;
      }
      if (ND == 1) Bd += sm[(L + 1 + j) * P + 256 * NS + lane];
    }
  }
}

// Row accesses go through a per-row BUFFER descriptor (base = the row, num_records = its bytes): lanes
// whose slice lies past the row read 0 / are dropped by the hardware range check, so every load and
// store of the loop is ONE unconditional instruction.  That matters beyond the saved compares: hipcc's
// s_waitcnt pass counts only memory operations that execute on every path, so a lane-predicated store
// (an `if` = a skippable block) made each "wait for the oldest load" a wait for nearly everything.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t row_rsrc(const void *p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(rsrc_t r, int off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float buf_load1(rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store4(rsrc_t r, int off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
__device__ __forceinline__ void buf_store1(rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, off, 0, 0);
}

template <int NS, int L, int ND, int U>
__global__ __launch_bounds__(kBlock, RM_CROSS_FWD_WPS) void cross_fwd_kernel(
    const float *__restrict__ xe, const float *__restrict__ xd, int FD, int Dn,
    const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ w_out,
    int64_t B, float *__restrict__ logit, float *__restrict__ p_out, int p_ld) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63;
  stage_params<NS>(sm, w, b, w_out, L, FD, Dn);
  __syncthreads();
  CrossParams<NS, L> P_;
  load_params<NS, L, ND>(sm, w, b, w_out, FD, Dn, lane, P_);
  const int d = FD + Dn;

  // (the wave number as a SCALAR: row numbers and the row descriptors stay on the scalar unit)
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * U;
  int64_t b0 = wave * U;
  if (b0 >= B) return;
  // two row buffers used in turn (a copy "cur = nxt" at the end of an iteration makes the wave wait
  // for its prefetch right there)
  float4 bufA[U][NS], bufB[U][NS];
  float dnsA[U], dnsB[U];

  auto fetch = [&](int64_t base, float4 (&x)[U][NS], float (&xdv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u < B ? base + u : B - 1;  // past the end: row B-1 again (same values)
      const rsrc_t re = row_rsrc(xe + r * FD, FD * 4);
#pragma unroll
      for (int s = 0; s < NS; ++s) x[u][s] = buf_load4(re, 16 * (lane + 64 * s));
      xdv[u] = ND == 1 ? buf_load1(row_rsrc(xd + r * Dn, Dn * 4), 4 * lane) : 0.f;
    }
  };
  auto rows = [&](int64_t base, const float4 (&x)[U][NS], const float (&xdv)[U]) {
    float p[U][L + 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j <= L; ++j) {
        float part = xdv[u] * P_.Wd[j];
#pragma unroll
        for (int s = 0; s < NS; ++s) part += dot4(x[u][s], P_.W[j][s]);
        p[u][j] = part;
      }
      if (ND == 2) {
        const int64_t r = base + u < B ? base + u : B - 1;
        for (int k = lane; k < Dn; k += 64) {
          const float xv = xd[r * Dn + k];
#pragma unroll
          for (int j = 0; j <= L; ++j) p[u][j] += xv * (j < L ? w[(int64_t)j * d + FD + k] : w_out[FD + k]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j <= L; ++j) p[u][j] = wave_allsum(p[u][j]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u < B ? base + u : B - 1;  // a clamped row stores row B-1's values again
      float c = 1.f;
#pragma unroll
      for (int l = 0; l < L; ++l) c += fmaf(c, p[u][l], P_.beta[l]);  // c_{l+1} = c_l + s_l
      const float out = fmaf(c, p[u][L], P_.beta[L]);
      buf_store1(row_rsrc(logit + r, 4), 4 * lane, out);  // lane 0 is in range
      if (p_out != nullptr) {
        float sel = p[u][0];
#pragma unroll
        for (int j = 1; j <= L; ++j) sel = lane == j ? p[u][j] : sel;
        buf_store1(row_rsrc(p_out + r * p_ld, 4 * (L + 1)), 4 * lane, sel);  // lanes 0..L
      }
    }
  };
  fetch(b0, bufA, dnsA);
  while (true) {
    // the next iteration's rows go out before this one's arithmetic; the fence keeps them there
    // (hipcc otherwise sinks the loads below the arithmetic: nothing in flight while the wave computes)
    fetch(b0 + stride, bufB, dnsB);
    __builtin_amdgcn_sched_barrier(0);
    rows(b0, bufA, dnsA);
    b0 += stride;
    if (b0 >= B) break;
    fetch(b0 + stride, bufA, dnsA);
    __builtin_amdgcn_sched_barrier(0);
    rows(b0, bufB, dnsB);
    b0 += stride;
    if (b0 >= B) break;
  }
}

// NDX = ND + 4 * HAVE_IN (HAVE_IN: add the other branch's dx_in_e)
template <int NS, int L, int NDX, int U>
__global__ __launch_bounds__(kBlock, RM_CROSS_BWD_WPS) void cross_bwd_kernel(
    int FD, int Dn, const float *__restrict__ w, const float *__restrict__ b,
    const float *__restrict__ w_out, int64_t B, const float *__restrict__ g,
    const float *__restrict__ p_in, int p_ld, const float *__restrict__ dx_in_e,
    float *__restrict__ d_xe, float *__restrict__ coef) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63;
  constexpr int NC = 2 * L + 2;
  constexpr int ND = NDX & 3;
  constexpr bool HAVE_IN = NDX >= 4;
  stage_params<NS>(sm, w, b, w_out, L, FD, Dn);
  __syncthreads();
  CrossParams<NS, L> P_;
  load_params<NS, L, ND>(sm, w, b, w_out, FD, Dn, lane, P_);  // (Wd unused here; beta needs the dense part)

  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * U;
  // per row: the other branch's dx slices and ONE more vector load: lane j <= L fetches p_j, the lanes
  // above it g (broadcast by v_readlane when used).  As uniform scalar loads the compiler turned them
  // into s_load right at their use - a scalar-cache round trip per row in the middle of the arithmetic -
  // and a prefetched pair of rows would hold 32 SGPRs.
  int64_t b0 = wave * U;
  if (b0 >= B) return;
  const float *pg_base = lane <= L ? p_in : g;          // loop-invariant per-lane selects:
  const int64_t pg_mul = lane <= L ? (int64_t)p_ld : 1;  // address = base + (r * mul + add) floats
  const int64_t pg_add = lane <= L ? lane : 0;
  float4 bufA[U][NS], bufB[U][NS], bufC[U][NS];
  float pgA[U], pgB[U], pgC[U];

  auto fetch = [&](int64_t base, float4 (&in)[U][NS], float (&pv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u < B ? base + u : B - 1;
      if (HAVE_IN) {
        const rsrc_t ri = row_rsrc(dx_in_e + r * FD, FD * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) in[u][s] = buf_load4(ri, 16 * (lane + 64 * s));
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) in[u][s] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      pv[u] = pg_base[r * pg_mul + pg_add];
    }
  };
  auto rows = [&](int64_t base, const float4 (&in)[U][NS], const float (&pv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u < B ? base + u : B - 1;  // a clamped row stores row B-1's values again
      float cp[L + 1];
#pragma unroll
      for (int j = 0; j <= L; ++j)
        cp[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[u]), j));
      const float gb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[u]), L + 1));
      float c[L + 1], t[L], a[L + 1];
      c[0] = 1.f;
#pragma unroll
      for (int l = 0; l < L; ++l) c[l + 1] = c[l] + fmaf(c[l], cp[l], P_.beta[l]);
#pragma unroll
      for (int l = L - 1; l >= 0; --l) {
        float tl = gb * cp[L];
#pragma unroll
        for (int j = L - 1; j > l; --j) tl = fmaf(t[j], cp[j], tl);
        t[l] = tl;
        a[l] = uniform(tl * c[l]);
      }
      // (a_j as scalars: in VGPRs hipcc packs the updates below into v_pk_fma_f32 whose 64-bit operand
      // pairs a_j with the NEIGHBOURING register - a prefetch destination - and so waits for vmcnt(0))
      a[L] = uniform(gb * c[L]);
      // coef row: [t_l c_l (l < L) | g c_L | t_l (l < L) | g], lanes 0..NC-1
      float sel = a[0];
#pragma unroll
      for (int j = 1; j <= L; ++j) sel = lane == j ? a[j] : sel;
#pragma unroll
      for (int j = 0; j < L; ++j) sel = lane == L + 1 + j ? t[j] : sel;
      sel = lane == NC - 1 ? gb : sel;
      buf_store1(row_rsrc(coef + r * NC, 4 * NC), 4 * lane, sel);
      const rsrc_t ro = row_rsrc(d_xe + r * FD, FD * 4);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float4 o = in[u][s];
#pragma unroll
        for (int j = 0; j <= L; ++j) {
          o.x = fmaf(a[j], P_.W[j][s].x, o.x); o.y = fmaf(a[j], P_.W[j][s].y, o.y);
          o.z = fmaf(a[j], P_.W[j][s].z, o.z); o.w = fmaf(a[j], P_.W[j][s].w, o.w);
        }
        buf_store4(ro, 16 * (lane + 64 * s), o);
      }
    }
  };
  // THREE row buffers in rotation: the dx0 stores read their data from the buffer registers, and a load
  // into registers an outstanding store still reads must wait for that store (vmcnt counts loads and
  // stores together) - with two buffers every prefetch started behind the previous rows' stores
  fetch(b0, bufA, pgA);
  while (true) {
    fetch(b0 + stride, bufB, pgB);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads in front of the arithmetic (see the forward)
    rows(b0, bufA, pgA);
    b0 += stride;
    if (b0 >= B) break;
    fetch(b0 + stride, bufC, pgC);
    __builtin_amdgcn_sched_barrier(0);
    rows(b0, bufB, pgB);
    b0 += stride;
    if (b0 >= B) break;
    fetch(b0 + stride, bufA, pgA);
    __builtin_amdgcn_sched_barrier(0);
    rows(b0, bufC, pgC);
    b0 += stride;
    if (b0 >= B) break;
  }
}

// d_w[l] = P[:,l] + T_l*Bp_l ; d_w_out = P[:,L] + G*Bp_L ; d_b[l] = G*w_out + sum_{j>l} T_j w_j
__global__ void cross_param_grads_kernel(const float *__restrict__ Pm, const float *__restrict__ cs,
                                         const float *__restrict__ w, const float *__restrict__ b,
                                         const float *__restrict__ w_out, int L, int d,
                                         float *__restrict__ d_w, float *__restrict__ d_b,
                                         float *__restrict__ d_w_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d) return;
  const float G = cs[L];
  float bp = 0.f;  // Bp_l = sum_{j<l} b_j
  for (int l = 0; l < L; ++l) {
    d_w[l * d + e] = Pm[e * (L + 1) + l] + cs[l] * bp;
    bp += b[l * d + e];
  }
  d_w_out[e] = Pm[e * (L + 1) + L] + G * bp;
  float tail = G * w_out[e];
  for (int l = L - 1; l >= 0; --l) {
    d_b[l * d + e] = tail;
    tail += cs[l] * w[l * d + e];
  }
}

}  // namespace

#define RM_CROSS_L(NS_, ND_, U_, KERNEL, ...)                                                                         \
  switch (L) {                                                                                                     \
    case 1: hipLaunchKernelGGL((KERNEL<NS_, 1, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 2: hipLaunchKernelGGL((KERNEL<NS_, 2, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 3: hipLaunchKernelGGL((KERNEL<NS_, 3, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 4: hipLaunchKernelGGL((KERNEL<NS_, 4, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 5: hipLaunchKernelGGL((KERNEL<NS_, 5, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 6: hipLaunchKernelGGL((KERNEL<NS_, 6, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 7: hipLaunchKernelGGL((KERNEL<NS_, 7, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    case 8: hipLaunchKernelGGL((KERNEL<NS_, 8, ND_, U_>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;       \
    default: break;                                                                                                \
  }
#define RM_CROSS_ND(NS_, U_, ADD_, KERNEL, ...)                                   \
  if (Dn == 0) {                                                            \
    RM_CROSS_L(NS_, 0 + ADD_, U_, KERNEL, __VA_ARGS__)                             \
  } else if (Dn <= 64) {                                                    \
    RM_CROSS_L(NS_, 1 + ADD_, U_, KERNEL, __VA_ARGS__)                             \
  } else {                                                                  \
    RM_CROSS_L(NS_, 2 + ADD_, U_, KERNEL, __VA_ARGS__)                             \
  }
#define RM_CROSS_DISPATCH(U_, ADD_, KERNEL, ...)                                 \
  const int ns_ = FD <= 256 ? 1 : 2;                                        \
  const size_t smem = (size_t)(2 * L + 1) * (256 * ns_ + 64) * sizeof(float); \
  if (ns_ == 1) {                                                           \
    RM_CROSS_ND(1, U_, ADD_, KERNEL, __VA_ARGS__)                                 \
  } else {                                                                  \
    RM_CROSS_ND(2, U_, ADD_, KERNEL, __VA_ARGS__)                                 \
  }

static int cross_check(const char *fn, int FD, int Dn, const float *w, const float *b,
                       const float *w_out, int L, int64_t B, int p_ld) {
  RM_REQUIRE(B >= 0 && FD >= 0 && Dn >= 0 && FD + Dn > 0, "%s: bad sizes", fn);
  RM_REQUIRE(L >= 1 && L <= kMaxL, "%s: L=%d unsupported (1..%d)", fn, L, kMaxL);
  RM_REQUIRE(FD % 4 == 0, "%s: FD=%d must be a multiple of 4", fn, FD);
  RM_REQUIRE(FD <= 512, "%s: FD=%d unsupported (<= 512)", fn, FD);
  RM_REQUIRE(w && b && w_out, "%s: NULL parameter", fn);
  RM_REQUIRE(p_ld >= L + 1, "%s: p_ld=%d must be >= L + 1", fn, p_ld);
  return RM_OK;
}

// wps waves per SIMD on every CU; each walks U rows per iteration
static dim3 cross_grid(int64_t B, int U, int wps) {
  return dim3(rm_grid_cap((B + (kBlock / 64) * U - 1) / ((kBlock / 64) * U), 256 * wps));
}

extern "C" int rm_cross_fwd(const float *xe, const float *xd, int FD, int Dn, const float *w,
                            const float *b, const float *w_out, int L, int64_t B, float *logit,
                            float *p_out, int p_ld, rm_stream_t stream) {
  int rc = cross_check("rm_cross_fwd", FD, Dn, w, b, w_out, L, B, p_out ? p_ld : L + 1);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE(FD > 0 && xe && rm_aligned16(xe) && (Dn == 0 || xd), "rm_cross_fwd: NULL or unaligned input (FD > 0 required)");
  RM_REQUIRE(logit, "rm_cross_fwd: logit must not be NULL");
  dim3 grid = cross_grid(B, RM_CROSS_FWD_U, RM_CROSS_FWD_WPS);
  hipStream_t st = (hipStream_t)stream;
  RM_CROSS_DISPATCH(RM_CROSS_FWD_U, 0, cross_fwd_kernel, xe, xd, FD, Dn, w, b, w_out, B, logit, p_out, p_ld)
  RM_CHECK_LAUNCH("rm_cross_fwd");
  return RM_OK;
}

extern "C" int rm_cross_bwd(int FD, int Dn, const float *w, const float *b, const float *w_out, int L,
                            int64_t B, const float *g, const float *p, int p_ld, const float *dx_in_e,
                            float *d_xe, float *coef, rm_stream_t stream) {
  int rc = cross_check("rm_cross_bwd", FD, Dn, w, b, w_out, L, B, p_ld);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE(g && p && coef, "rm_cross_bwd: g, p and coef must not be NULL");
  RM_REQUIRE(FD > 0 && d_xe && rm_aligned16(d_xe), "rm_cross_bwd: d_xe NULL or unaligned (FD > 0 required)");
  RM_REQUIRE(!dx_in_e || rm_aligned16(dx_in_e), "rm_cross_bwd: dx_in_e unaligned");
  dim3 grid = cross_grid(B, RM_CROSS_BWD_U, RM_CROSS_BWD_WPS);
  hipStream_t st = (hipStream_t)stream;
  if (dx_in_e != nullptr) {  // ND + 4: the kernel adds the other branch's dx
    RM_CROSS_DISPATCH(RM_CROSS_BWD_U, 4, cross_bwd_kernel, FD, Dn, w, b, w_out, B, g, p, p_ld, dx_in_e, d_xe, coef)
  } else {
    RM_CROSS_DISPATCH(RM_CROSS_BWD_U, 0, cross_bwd_kernel, FD, Dn, w, b, w_out, B, g, p, p_ld, dx_in_e, d_xe, coef)
  }
  RM_CHECK_LAUNCH("rm_cross_bwd");
  return RM_OK;
}

extern "C" int rm_cross_param_grads(const float *P, const float *colsum, const float *w,
                                    const float *b, const float *w_out, int L, int d, float *d_w,
                                    float *d_b, float *d_w_out, rm_stream_t stream) {
  RM_REQUIRE(L >= 1 && L <= kMaxL && d > 0, "rm_cross_param_grads: bad sizes");
  RM_REQUIRE(P && colsum && w && b && w_out && d_w && d_b && d_w_out,
             "rm_cross_param_grads: NULL argument");
  hipLaunchKernelGGL(cross_param_grads_kernel, dim3((d + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, P, colsum, w, b, w_out, L, d, d_w, d_b, d_w_out);
  RM_CHECK_LAUNCH("rm_cross_param_grads");
  return RM_OK;
}
